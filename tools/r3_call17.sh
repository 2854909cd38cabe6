#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c17; mkdir -p $out; rm -f $out/*
for rep in 1 2; do for a in "" _ff; do echo "### build$a" >> $out/tab.txt; for sh in in_layers proj_out qkv; do timeout 120 tools/bin/gemm_tab_bench$a "$sh" arith=0 2>&1 | grep -v "single\|M1792\|pad_" | grep "==\|arith" >> $out/tab.txt; done; done; done
cat $out/tab.txt
