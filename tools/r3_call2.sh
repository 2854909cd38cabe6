#!/bin/bash
# round 3, GPU call 2: batched-load epilogues vs the round-2 kernels; parity subset; quick bench
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c2; mkdir -p $out
export TMPDIR=/tmp
B=tools/bin/gemm_tab_bench
timeout 120 $B in_layers arith=0 u8=8 u7=7 > $out/tab_in_layers.txt 2>&1
timeout 120 $B proj_out arith=0 u8=8 > $out/tab_proj_out.txt 2>&1
timeout 180 $B qkv arith=0 u8c8=8/8 u8c6=8/6 u7c8=7/8 > $out/tab_qkv.txt 2>&1
timeout 180 $B conv3 arith=0 u8=8 u7=7 m8765=8,7,6,5 > $out/tab_conv3.txt 2>&1
timeout 120 $B integ arith=0 arith4=4 > $out/tab_integ.txt 2>&1
timeout 120 $B single arith=0 arith2=2 arith4=4 arith8=8 > $out/tab_single.txt 2>&1
T=tools/bin/gemm_tab_bench_trace
timeout 120 $T in_layers arith=0 u7=7 > $out/trace_in_layers.txt 2>&1
timeout 120 $T proj_out arith=0 > $out/trace_proj_out.txt 2>&1
timeout 120 $T conv3 arith=0 > $out/trace_conv3.txt 2>&1
timeout 120 $T qkv arith=0 > $out/trace_qkv.txt 2>&1
timeout 900 python -m pytest tests/test_diffusion_gpu.py tests/test_vocoder_gpu.py tests/test_ar_gpu.py -m gpu -x -q > $out/tests_subset.log 2>&1; echo "tests rc=$?" >> $out/tests_subset.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/bench.json 2> $out/bench.err
tail -n +1 $out/tab_*.txt | head -120; tail -3 $out/tests_subset.log; head -c 1500 $out/bench.json
