#!/bin/bash
# round 4, GPU call 11: latent pass with the thread-per-row attention kernel — AR tests + AR stage timing
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c11; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_ar_gpu.py -x -q -s > $out/tests_ar.log 2>&1; echo "tests_ar rc=$? [$(( $(date +%s) - t0 )) s]"; grep -i "latent\|passed\|failed\|error" $out/tests_ar.log | tail -8
TTS_TIMING=1 timeout 600 python tools/ar_option_ab.py device_topk 16 > $out/ab.txt 2>&1; echo "ab rc=$? [$(( $(date +%s) - t0 )) s]"; grep "^B=" $out/ab.txt | tail -4; grep "tts timing" $out/ab.txt | tail -3
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "test_ar_full_depth or teacher_forced" > $out/tests_fullsize_ar.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"; tail -5 $out/tests_fullsize_ar.log
