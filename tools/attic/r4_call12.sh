#!/bin/bash
# round 4, GPU call 12 (final evidence): the whole GPU suite + smoke, the driver's bench command, the single-utterance line, rocprofv3 kernel stats of the bench command
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); out=gpurun_out/r4c12; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 2000 python -m pytest tests -m gpu -q -s > $out/tests.log 2>&1; echo "tests rc=$? [$(( $(date +%s) - t0 )) s]" | tee -a $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
grep -E "passed|failed" $out/tests.log | tail -3; tail -2 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 500 $out/bench_n1.json; echo
timeout 300 python bench.py --candidates 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err; echo "bench b1 rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 300 $out/bench_b1.json; echo
O=$R/$out/prof; mkdir -p $O
( cd /tmp && TTS_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python $R/bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench.log ); echo "rocprof rc=$? [$(( $(date +%s) - t0 )) s]"
BS=$(find $O/bench -name "*kernel_stats.csv" | head -1)
python tools/summarize_profiles.py stats "$BS" $out/r4_bench_kernel_stats_final.csv "TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1   (final round-4 code; the bench command itself: 1 warm-up + 2 timed passes; graphs off so that every kernel is traced)"
cp $O/bench_under_rocprof.json $out/bench_under_rocprof.json
rm -rf $O
head -16 $out/r4_bench_kernel_stats_final.csv | cut -c1-150
