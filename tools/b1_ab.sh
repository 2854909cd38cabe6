#!/bin/bash
# developer A/B at one utterance (bench.py --candidates 1): $1 = environment assignment that selects the variant, e.g. TTS_ATT_RING3=1
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/b1ab; mkdir -p $out; rm -f $out/*
for rep in 1 2; do
  timeout 600 python bench.py --candidates 1 --steps 6 --warmup 2 --no-cpu-baseline --no-ab > $out/base.$rep.json 2>/dev/null
  env "$1" timeout 600 python bench.py --candidates 1 --steps 6 --warmup 2 --no-cpu-baseline --no-ab > $out/variant.$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b1ab/*.json')):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['stage_ms_per_step'])
PY
