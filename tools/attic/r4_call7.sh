#!/bin/bash
# round 4, GPU call 7: device top-k prefilter (tts_ar_step_sample) — AR tests, then the driver's bench command with the on/off A/B
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c7; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_ar_gpu.py -x -q -s > $out/tests_ar.log 2>&1; echo "tests_ar rc=$? [$(( $(date +%s) - t0 )) s]"; tail -12 $out/tests_ar.log
TTS_TIMING=1 timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4c7/bench_n1.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "stage_ms_per_step")})
print("topk", r.get("ar_device_topk_option"))
print("decode", {k: r["roofline_decode"][k] for k in ("achieved", "frac", "avg_step_us")})
print("f32 rerun", r.get("ar_f32_default_rerun"))
PY
grep "tts timing" $out/bench_n1.err | tail -12
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "AR or ar" > $out/tests_fullsize_ar.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"; tail -5 $out/tests_fullsize_ar.log
