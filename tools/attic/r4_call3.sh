#!/bin/bash
# round 4, GPU call 3: decode step with the h4 residual-stream layout (A/B in tools/dec_bench), AR parity, a short bench line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c3; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 120 tools/bin/dec_bench > $out/dec_bench_plain.txt 2>&1; echo "dec_bench rc=$?"; grep -E "layer chain|us/launch" $out/dec_bench_plain.txt
timeout 120 tools/bin/dec_bench_trace > $out/dec_bench_trace.txt 2>&1; echo "dec_bench_trace rc=$?"; grep -A11 "layer chain" $out/dec_bench_trace.txt
timeout 900 python -m pytest tests/test_ar_gpu.py tests/test_fp8_weights.py tests/test_errors_gpu.py -m gpu -x -q -s > $out/tests_ar.log 2>&1; echo "tests_ar rc=$? [$(( $(date +%s) - t0 )) s]"; tail -3 $out/tests_ar.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_short.json 2> $out/bench_short.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c3/bench_short.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','stage_ms_per_step') if k in d})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','us_per_launch') if k in d['roofline']})
print('decode', d.get('roofline_decode'))
print('f32 rerun', d.get('ar_f32_default_rerun')); print('ref prec', d.get('reference_precision_option'))
print('kernels', d['roofline'].get('kernels'))
PY
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "test_full_size_ar_192 or test_ar_full_depth or test_config2_batch16" > $out/tests_fullsize.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"
grep -E "\[reference precision|\[throughput|passed|failed|AR|configs" $out/tests_fullsize.log | tail -20
