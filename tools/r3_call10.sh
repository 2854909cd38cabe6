#!/bin/bash
# one-GPU measurements behind the predicted strong-scaling curve of configs[3] (64 candidates sharded 64/N per GPU): a shard of 64/N candidates on one GPU
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c10; mkdir -p $out
export TMPDIR=/tmp
for c in 8 16 32 64; do timeout 300 python bench.py --config 4 --candidates $c --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ab > $out/shard_$c.json 2> $out/shard_$c.err; done
python - <<'PY'
import json
for c in (8,16,32,64):
    d=json.load(open("gpurun_out/r3c10/shard_%d.json"%c)); print(c, d["value"], d["ms_per_step"], d["stage_ms_per_step"])
PY
