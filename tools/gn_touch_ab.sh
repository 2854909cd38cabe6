#!/bin/bash
# developer A/B: weight touch in the GroupNorm kernel (TTS_GN_NOTOUCH=1 disables it), single utterance and the benchmark batch
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/gntouch; mkdir -p $out; rm -f $out/*
for rep in 1 2; do
for t in 0 1; do
  if [ $t = 1 ]; then export TTS_GN_NOTOUCH=1; else unset TTS_GN_NOTOUCH; fi
  timeout 600 python bench.py --candidates 1 --steps 6 --warmup 2 --no-cpu-baseline --no-ab > $out/b1_notouch$t.$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/b16_notouch$t.$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/gntouch/*.json')):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    print(f.split('/')[-1], d['value'], d['stage_ms_per_step'])
PY
