#!/bin/bash
# round 4, GPU call 10: decode step without copy nodes (tokens read from pinned host memory, lists written to it) — AR tests, A/B of device_topk, full-size AR tests
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c10; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_ar_gpu.py -x -q > $out/tests_ar.log 2>&1; echo "tests_ar rc=$? [$(( $(date +%s) - t0 )) s]"; tail -4 $out/tests_ar.log
timeout 600 python tools/ar_option_ab.py device_topk 16 1 > $out/topk_ab.txt 2>&1; echo "ab rc=$? [$(( $(date +%s) - t0 )) s]"; tail -12 $out/topk_ab.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "ar_ or AR or config1 or batch16" > $out/tests_fullsize_ar.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"; tail -8 $out/tests_fullsize_ar.log
