#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c21; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "rccl_one_rank or two_ranks" > $out/tests.txt 2>&1
tail -8 $out/tests.txt
