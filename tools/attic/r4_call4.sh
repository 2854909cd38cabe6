#!/bin/bash
# round 4, GPU call 4: attention kernel variants (setprio around the MFMA clusters, row sums on the VALU), single-utterance bench line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c4; mkdir -p $out
export TMPDIR=/tmp
for v in base setprio lsum_valu lsum_valusetprio base setprio lsum_valu lsum_valusetprio; do echo "== $v" >> $out/attn_variants.txt; timeout 60 tools/bin/attn_v_$v >> $out/attn_variants.txt 2>&1; done
grep -E "==|product|differ" $out/attn_variants.txt
timeout 600 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err; echo "bench b1 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c4/bench_b1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stage_ms_per_step') if k in d}); print(d.get('roofline_decode'))
PY
