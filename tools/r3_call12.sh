#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c12; mkdir -p $out
export TMPDIR=/tmp
for v in s1p0 s0p0 s0p1 s1p1; do for sh in in_layers proj_out qkv conv3; do echo "#### $v" >> $out/tab.txt; timeout 120 tools/bin/gemm_tab_bench_$v "$sh" arith=0 big=0 2>&1 | grep -v "single" >> $out/tab.txt; done; done
grep -E "####|==|arith|big" $out/tab.txt | grep -v "single\|N3072 K1024 M1792" 
