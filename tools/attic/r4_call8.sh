#!/bin/bash
# round 4, GPU call 8: cross-kernel prefetch in the decode step (option dec_prefetch) — AR parity tests, on/off A/B at B = 16 and B = 1
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c8; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_ar_gpu.py -x -q > $out/tests_ar.log 2>&1; echo "tests_ar rc=$? [$(( $(date +%s) - t0 )) s]"; tail -4 $out/tests_ar.log
timeout 600 python tools/ar_option_ab.py dec_prefetch 16 1 > $out/prefetch_ab.txt 2>&1; echo "ab rc=$? [$(( $(date +%s) - t0 )) s]"; tail -12 $out/prefetch_ab.txt
