#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c20; mkdir -p $out; rm -f $out/*
timeout 1200 python -m pytest tests/test_clvp.py tests/test_distributed_gpu.py -x -q -s -m gpu > $out/tests.txt 2>&1
tail -25 $out/tests.txt
