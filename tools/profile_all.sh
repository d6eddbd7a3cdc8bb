#!/bin/bash
# Everything profiles/<tag>_* is made of, in one GPU call:   bash tools/profile_all.sh r04
# (tools/profile_round.sh for configs[1], kernel stats of configs[4] and of the decode paths, the two bench lines, the
# free-running kernel's phase stamps).  Outputs under gpurun_out/<tag>/; copy what should be judged into profiles/.
set -e
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
bash $ROOT/tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
echo "[profile_all] round profile done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats -d /tmp/p5 -o s --output-format csv -- python3 $ROOT/bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-operators --no-extras > $OUT/bench_cfg5_prof.log 2>&1
cp /tmp/p5/s_kernel_stats.csv $OUT/bench_cfg5_kernel_stats.csv
rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats -d /tmp/pd -o s --output-format csv -- python3 $ROOT/tools/bench_decode.py > $OUT/decode_prof.log 2>&1
cp /tmp/pd/s_kernel_stats.csv $OUT/decode_kernel_stats.csv
echo "[profile_all] cfg5 + decode kernel stats done"
cd $ROOT
python3 tools/exp_free_phases.py > $OUT/exp_free_phases.txt 2>&1
python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
echo "[profile_all] bench cfg2 done"
python3 bench.py --config cfg5 --steps 10 --warmup 5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
echo "[profile_all] bench cfg5 done"
