#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c7; mkdir -p $out
export TMPDIR=/tmp
for sh in in_layers proj_out qkv "integ k1" "single k1" "single qkv"; do timeout 120 tools/bin/gemm_tab_bench "$sh" arith=0 arithpipe=0 >> $out/tab.txt 2>&1; done
cat $out/tab.txt
