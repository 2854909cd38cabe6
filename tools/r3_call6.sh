#!/bin/bash
# round 3, GPU call 6: per-phase timeline of the 256-column kernel
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c6; mkdir -p $out
export TMPDIR=/tmp
timeout 120 tools/bin/gemm_tab_bench_trace in_layers arith=0 arith8=8 > $out/trace_in_layers.txt 2>&1
timeout 120 tools/bin/gemm_tab_bench in_layers arith=0 arith8=8 > $out/tab_in_layers.txt 2>&1
timeout 120 tools/bin/gemm_tab_bench qkv arith=0 arith8=8 > $out/tab_qkv.txt 2>&1
cat $out/tab_in_layers.txt $out/tab_qkv.txt; grep -A6 "256-column" $out/trace_in_layers.txt | cut -c1-400
