#!/bin/bash
# developer wrapper for one gpurun call: runs the given pytest selection, keeps the log under gpurun_out/once/
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/once; mkdir -p $out; rm -f $out/*
timeout 1500 python -m pytest "$@" > $out/tests.txt 2>&1
tail -25 $out/tests.txt
