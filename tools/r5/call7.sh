#!/bin/bash
# round 5, GPU call 7: consumer-side ceiling of a GEMM work-group (fragment reads + MFMAs only)
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c7; mkdir -p $out
timeout 120 tools/bin/mfma_lds_bound > $out/mfma_lds_bound.txt 2>&1; echo "rc=$?"; cat $out/mfma_lds_bound.txt
