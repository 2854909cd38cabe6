#!/bin/bash
# round 5, GPU call 5: the latent conditioner in reference precision (option lc_attn_f32) — loops against the oracle incl. the benchmark-length problem, cost
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c5; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python tools/r5/attn_modes.py small mid full20 full870 cost > $out/attn_modes.log 2>&1; echo "attn_modes rc=$? [$(( $(date +%s) - t0 )) s]"; cat $out/attn_modes.log
