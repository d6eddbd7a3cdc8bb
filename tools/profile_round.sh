#!/bin/bash
# Regenerates the per-round profile artefacts on a GPU box:   bash tools/profile_round.sh r02 [quick]
#   gpurun_out/<tag>/kernel_stats.csv      rocprofv3 --kernel-trace --stats of bench.py (training steps only: no CPU baseline,
#                                          no isolated operator timings, so kernel counts are per step x steps)
#   gpurun_out/<tag>/step_timeline.txt     one optimiser step as a timeline + per-kernel totals
#   gpurun_out/<tag>/kernel_by_shape.json  chain kernels by (name, grid, workgroup)
#   gpurun_out/<tag>/pmc.json              HBM traffic per launch (FETCH_SIZE x2 + WRITE_SIZE, separate passes) for the dominant
#                                          kernels and the decoder step
#   gpurun_out/<tag>/pmc_gemm.txt          matrix-pipe counters of the bf16x6 GEMM kernel
# Copy what should be judged into profiles/ (named <tag>_*).  rocprofv3 is given python3 directly (no env/bash hop), PMC passes
# use --kernel-trace only.
set -e
TAG=${1:-r02}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps && rocprofv3 --kernel-trace --stats -d /tmp/ps -o s --output-format csv -- \
    python3 $ROOT/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-operators --no-extras > $OUT/bench_prof.log 2>&1
cp /tmp/ps/s_kernel_stats.csv $OUT/kernel_stats.csv
python3 $ROOT/tools/step_timeline.py /tmp/ps --full > $OUT/step_timeline.txt
python3 $ROOT/tools/kernel_by_shape.py /tmp/ps $OUT/kernel_by_shape.json > $OUT/kernel_by_shape.txt
tail -60 $OUT/step_timeline.txt
if [ "$QUICK" = "quick" ]; then exit 0; fi
rm -rf /tmp/pf /tmp/pw /tmp/pg
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f --output-format csv -- python3 $ROOT/tools/prof_decoder_fwd.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w --output-format csv -- python3 $ROOT/tools/prof_decoder_fwd.py > $OUT/pmc_write.log 2>&1
python3 $ROOT/tools/pmc_summary.py /tmp/pf /tmp/pw $OUT/pmc.json > $OUT/pmc_summary.log 2>&1 || tail -5 $OUT/pmc_summary.log
# whole-step HBM traffic, configs[1] and configs[4]
for CFG in cfg2 cfg5; do
  rm -rf /tmp/sf /tmp/sw
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/sf -o f --output-format csv -- python3 $ROOT/tools/prof_step.py $CFG > $OUT/pmc_step_fetch_$CFG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/sw -o w --output-format csv -- python3 $ROOT/tools/prof_step.py $CFG > $OUT/pmc_step_write_$CFG.log 2>&1
  python3 $ROOT/tools/pmc_step_summary.py /tmp/sf /tmp/sw $OUT/pmc_step_$CFG.json > $OUT/pmc_step_$CFG.txt 2>&1 || tail -5 $OUT/pmc_step_$CFG.txt
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pg -o g --output-format csv -- python3 $ROOT/tools/pmc_gemm.py > $OUT/pmc_gemm.log 2>&1 || true
python3 $ROOT/tools/pmc_agg.py /tmp/pg > $OUT/pmc_gemm.txt 2>&1 || true
cat $OUT/pmc.json; cat $OUT/pmc_gemm.txt
