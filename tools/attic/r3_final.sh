#!/bin/bash
# round 3 final evidence: full GPU suite, smoke, the driver's bench command (with cpu_baseline), the single-utterance line
# (set TTS_SKIP_LONG_TESTS=1 / R3_NO_BENCH=1 for a shorter re-check after a small change)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3final; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
if [ -z "$R3_NO_BENCH" ]; then
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err
timeout 300 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err
fi
grep -E "passed|failed|rc=" $out/tests.log | tail -3; tail -3 $out/smoke.log; head -c 400 $out/bench_n1.json 2>/dev/null; echo; head -c 400 $out/bench_b1.json 2>/dev/null
