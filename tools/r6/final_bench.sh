#!/bin/bash
# round 6, final evidence (benches + profiles): the driver's bench command, the other workloads, the rocprofv3 passes of tools/profile_r6.sh
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); out=gpurun_out/r6final; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
el() { echo "[$(( $(date +%s) - t0 )) s]"; }
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? $(el)"; head -c 400 $out/bench_n1.json; echo
timeout 300 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_n1_single_candidate.json 2> $out/b1.err; echo "bench b1 rc=$? $(el)"; head -c 260 $out/bench_n1_single_candidate.json; echo
timeout 300 python bench.py --candidates 1 --latency-mode --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_n1_single_candidate_latency_mode.json 2> $out/b1l.err; echo "bench b1 latency rc=$? $(el)"; head -c 260 $out/bench_n1_single_candidate_latency_mode.json; echo
timeout 300 python bench.py --config 4 --candidates 8 --steps 5 --warmup 2 --no-cpu-baseline --no-ab > $out/bench_n1_config4_shard8.json 2> $out/c4s.err; echo "bench config4 shard of 8 rc=$? $(el)"; head -c 260 $out/bench_n1_config4_shard8.json; echo
timeout 400 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_n1_config4_64cand.json 2> $out/c4.err; echo "bench config4 rc=$? $(el)"; head -c 260 $out/bench_n1_config4_64cand.json; echo
timeout 600 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_n1_config5_8prompts_200steps.json 2> $out/c5.err; echo "bench config5 rc=$? $(el)"; head -c 260 $out/bench_n1_config5_8prompts_200steps.json; echo
timeout 1500 bash tools/profile_r6.sh > $out/profile.log 2>&1; echo "profile rc=$? $(el)"; tail -30 $out/profile.log | cut -c1-200
cp gpurun_out/prof_r6/r6_* gpurun_out/prof_r6/bench_under_rocprof.json $out/ 2>/dev/null; ls $out
