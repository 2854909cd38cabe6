#!/bin/bash
# round 3, GPU call 5: the 256-column 8-phase kernel vs the 128-column kernels vs round 2
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c5; mkdir -p $out
export TMPDIR=/tmp
B=tools/bin/gemm_tab_bench
for sh in in_layers proj_out qkv conv3 integ; do timeout 200 $B $sh arith=0 arith8=8 > $out/tab_$sh.txt 2>&1; done
tail -n +1 $out/tab_*.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $out/bench.json 2> $out/bench.err
head -c 300 $out/bench.json; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3c5/bench.json"))
    print(d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
    for k in d["roofline"]["kernels"]: print(k)
except Exception as e: print("bench parse failed", e)
PY
timeout 600 python -m pytest tests/test_diffusion_gpu.py tests/test_vocoder_gpu.py tests/test_properties_gpu.py -m gpu -x -q > $out/tests_subset.log 2>&1; tail -3 $out/tests_subset.log
