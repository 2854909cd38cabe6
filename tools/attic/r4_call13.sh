#!/bin/bash
# round 4, GPU call 13: after the time-MLP changes — diffusion + distributed GPU tests, smoke, the driver's bench command, the single-utterance line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c13; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_diffusion_gpu.py tests/test_distributed_gpu.py tests/test_e2e_gpu.py -q -s > $out/tests.log 2>&1; echo "tests rc=$? [$(( $(date +%s) - t0 )) s]" | tee -a $out/tests.log
grep -E "passed|failed" $out/tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log; tail -1 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 420 $out/bench_n1.json; echo
timeout 300 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err; echo "bench b1 rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 300 $out/bench_b1.json; echo
