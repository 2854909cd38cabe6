#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c17; mkdir -p $out; rm -f $out/*
timeout 300 tools/bin/gemm_tab_bench "pad_" arith=0 wide=0 2>&1 >> $out/tab.txt
cat $out/tab.txt
