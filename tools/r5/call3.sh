#!/bin/bash
# round 5, GPU call 3: forward dumps per arithmetic mode (CPU-side comparison with the torch emulations), dual-B proj_out kernel A/B, ragged tests
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c3; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python tools/r5/forward_dump.py $out/fwd > $out/forward_dump.log 2>&1; echo "forward_dump rc=$? [$(( $(date +%s) - t0 )) s]"; tail -2 $out/forward_dump.log
timeout 600 python tools/r5/attn_modes.py mid full20 cost > $out/attn_modes.log 2>&1; echo "attn_modes rc=$? [$(( $(date +%s) - t0 )) s]"; cat $out/attn_modes.log
timeout 900 python -m pytest tests/test_ragged_gpu.py -x -q -s > $out/ragged.log 2>&1; echo "ragged rc=$? [$(( $(date +%s) - t0 )) s]"; tail -12 $out/ragged.log
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab > $out/bench.json 2> $out/bench.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 260 $out/bench.json; echo
