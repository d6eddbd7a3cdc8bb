#!/bin/bash
# Upper bound on what pre-splitting the bf16x6 products' weight operand once per optimiser step could buy:
# tools/_ab/cheat_b.so is the library with -DVAG_CHEAT_B=1 (gemm_shared.h sp_store CHEAT: the B operand's three planes are one
# bf16 pack instead of the 11-instruction split -- wrong numbers, the instruction count of a pre-split operand, none of its extra
# ingest bytes); tools/_ab/head.so is the product library. Same box, same process order; per-shape product times, then the whole step.
# Build the two libraries first (tools/_ab is scratch, not tracked):
#   make -C vag-nmt_amd/csrc && mkdir -p tools/_ab && cp vag-nmt_amd/lib/libvagnmt.so tools/_ab/head.so
#   cd vag-nmt_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVAG_CHEAT_B=1 -c gemm.hip -o /tmp/gemm_cheat.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/cheat_b.so $(ls build/*.o | grep -v gemm.o) /tmp/gemm_cheat.o -ldl
#   (-DVAG_CHEAT_B=2: both operands' splits removed, cheat_ab.so)
set -e
out=${1:-gpurun_out/halfsplit}; mkdir -p $out
for v in head cheat_b; do
  VAG_LIB=$PWD/tools/_ab/$v.so timeout -k 10 200 python3 tools/exp_gemm_shapes.py > $out/shapes_$v.txt 2>$out/shapes_$v.err < /dev/null
  VAG_LIB=$PWD/tools/_ab/$v.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_$v.json 2>$out/bench_$v.err < /dev/null
done
for v in head cheat_b; do
  VAG_LIB=$PWD/tools/_ab/$v.so timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $out/bench2_$v.json 2>$out/bench2_$v.err < /dev/null
done
paste -d'|' $out/shapes_head.txt $out/shapes_cheat_b.txt > $out/shapes_side.txt
