set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/cfg5
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-operators 2>/dev/null | tail -1
rm -rf /tmp/p5 && rocprofv3 --kernel-trace --stats -d /tmp/p5 -o s --output-format csv -- python3 $ROOT/bench.py --config cfg5 --steps 6 --warmup 3 --no-cpu-baseline --no-operators --no-extras > $ROOT/gpurun_out/cfg5/prof.log 2>&1
cp /tmp/p5/s_kernel_stats.csv $ROOT/gpurun_out/cfg5/kernel_stats.csv
head -30 $ROOT/gpurun_out/cfg5/kernel_stats.csv | cut -c1-160
