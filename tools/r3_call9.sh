#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c9; mkdir -p $out
export TMPDIR=/tmp
for sh in conv3 in_layers qkv "integ k3" "single k3"; do timeout 120 tools/bin/gemm_tab_bench "$sh" arith=0 >> $out/tab.txt 2>&1; done
cat $out/tab.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c9/bench.json"))
print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for k in d["roofline"]["kernels"]: print(k)
PY
