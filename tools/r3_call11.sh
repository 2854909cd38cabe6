#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c11; mkdir -p $out
export TMPDIR=/tmp
for v in 1 0; do for sh in in_layers qkv; do echo "#### stagger=$v" >> $out/tab.txt; timeout 120 tools/bin/gemm_tab_bench_stag$v "$sh" arith=0 big=0 >> $out/tab.txt 2>&1; done; done
for v in 0; do timeout 120 tools/bin/gemm_tab_bench_stag${v}_trace in_layers big=0 > $out/trace_stag$v.txt 2>&1; done
grep -v "single" $out/tab.txt | grep -A4 "####\|==" | head -60; grep -A5 "256-column" $out/trace_stag0.txt | cut -c1-330
