#!/bin/bash
# round 4, GPU call 6: rocprofv3 passes (profile_r4.sh), the driver's bench command, the single-utterance line
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c6; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 600 $out/bench_n1.json; echo
timeout 300 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err; echo "bench b1 rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 300 $out/bench_b1.json; echo
timeout 1200 bash tools/profile_r4.sh > $out/profile.log 2>&1; echo "profile rc=$? [$(( $(date +%s) - t0 )) s]"; tail -45 $out/profile.log
cp gpurun_out/prof_r4/r4_* $out/ 2>/dev/null; cp gpurun_out/prof_r4/bench_under_rocprof.json $out/ 2>/dev/null
