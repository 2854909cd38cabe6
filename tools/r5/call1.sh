#!/bin/bash
# round 5, GPU call 1: attention-mode ladder against the oracle + cost, then a short bench line
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c1; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python tools/r5/attn_modes.py small mid full20 full870 cost > $out/attn_modes.log 2>&1; echo "attn_modes rc=$? [$(( $(date +%s) - t0 )) s]"; cat $out/attn_modes.log
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 600 $out/bench_n1.json; echo
