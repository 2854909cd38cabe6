#!/bin/bash
# round 5, GPU call 4: engine-side ablation inside the reference-precision AttentionBlock (option attn_f32_drop)
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c4; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python tools/r5/attn_modes.py small mid full20 > $out/attn_modes.log 2>&1; echo "attn_modes rc=$? [$(( $(date +%s) - t0 )) s]"; cat $out/attn_modes.log
