#!/bin/bash
# round 3, GPU call 1: tile-table GEMM policies vs the one-tile kernels; decode nt-loads A/B
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c1; mkdir -p $out
export TMPDIR=/tmp
B=tools/bin/gemm_tab_bench
timeout 120 $B in_layers u8=8 u7=7 u6=6 m86=8,6 m8866=8,8,6,6 m8765=8,7,6,5 m86lpt=8,6//1 m86n=8,6//2 u7n=7//2 > $out/tab_in_layers.txt 2>&1
timeout 120 $B proj_out u8=8 u7=7 m86=8,6 m8866=8,8,6,6 m8765=8,7,6,5 > $out/tab_proj_out.txt 2>&1
timeout 180 $B qkv u8c6=8/6 u7c6=7/6 u7c8=7/8 u7c12=7/12 u7c24=7/24 m86c8=8,6/8 m8866c8=8,8,6,6/8 m8765c8=8,7,6,5/8 u8c8=8/8 > $out/tab_qkv.txt 2>&1
timeout 180 $B conv3 u8=8 u7=7 u6=6 c36=6,6,6,6,6,6,6,6,6,6,6,7 m86=8,6 m8866=8,8,6,6 m8765=8,7,6,5 m8765lpt=8,7,6,5//1 > $out/tab_conv3.txt 2>&1
timeout 120 $B integ u8=8 u7=7 u6=6 u5=5 m86=8,6 > $out/tab_integ.txt 2>&1
timeout 120 $B single u8=8 u4=4 u3=3 u2=2 > $out/tab_single.txt 2>&1
T=tools/bin/gemm_tab_bench_trace
timeout 120 $T in_layers u8=8 u7=7 m8866=8,8,6,6 m8765=8,7,6,5 > $out/trace_in_layers.txt 2>&1
timeout 120 $T conv3 u8=8 c36=6,6,6,6,6,6,6,6,6,6,6,7 m8765=8,7,6,5 > $out/trace_conv3.txt 2>&1
timeout 120 $T qkv u8c6=8/6 u7c8=7/8 > $out/trace_qkv.txt 2>&1
TTS_DEC_NT=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_nt0.json 2> $out/bench_nt0.err
TTS_DEC_NT=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_nt1.json 2> $out/bench_nt1.err
tail -n +1 $out/tab_*.txt | head -150
