#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c17; mkdir -p $out; rm -f $out/*
TTS_COLD=1 timeout 200 tools/bin/gemm_tab_bench "single" arith=0 nopre=0 2>&1 >> $out/tab.txt
TTS_COLD=1 timeout 200 tools/bin/gemm_tab_bench "in_layers" arith=0 nopre=0 2>&1 | grep -v "single\|M1792" >> $out/tab.txt
cat $out/tab.txt
