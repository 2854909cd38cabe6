#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c19; mkdir -p $out; rm -f $out/*
timeout 900 python -m pytest tests/test_fp8_weights.py tests/test_ar_gpu.py::test_fp16_decode_weights_option tests/test_properties_gpu.py::test_stream_cus_partition_does_not_change_results -x -q -m gpu -s > $out/tests.txt 2>&1
tail -15 $out/tests.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c19/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_step'])
print(d['ar_weights_f16_option'])
print(d['ar_weights_fp8_option'])
print(d['roofline_decode'])
PY
