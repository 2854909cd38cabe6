#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c17; mkdir -p $out; rm -f $out/*
for sh in in_layers proj_out qkv; do timeout 120 tools/bin/gemm_tab_bench "$sh" arith=0 ctl=8/8/6 shareA=8/8/3 share22=8/8/4 none=8/8/5 2>&1 | grep -v "single\|M1792" >> $out/tab.txt; done
cat $out/tab.txt
