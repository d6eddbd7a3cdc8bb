#!/bin/bash
# Regenerates the per-round profile artefacts under gpurun_out/<tag>/ on a GPU box:  tools/profile_round.sh r02
# (copy what should be judged into profiles/ afterwards).  rocprofv3 is given python3 directly (no env/bash hop).
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps && rocprofv3 --kernel-trace --stats -d /tmp/ps -o s --output-format csv -- \
    python3 $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-operators > $OUT/bench_prof.log 2>&1
cp /tmp/ps/s_kernel_stats.csv $OUT/kernel_stats.csv
python3 $ROOT/tools/step_timeline.py /tmp/ps --full > $OUT/step_timeline.txt
python3 $ROOT/tools/kernel_by_shape.py /tmp/ps $OUT/kernel_by_shape.json > $OUT/kernel_by_shape.txt
tail -80 $OUT/step_timeline.txt
