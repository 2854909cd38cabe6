#!/bin/bash
# round 5: the driver's bench command with the hashed launch sampling + rocprofv3 kernel stats of the same command on the same box
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); out=gpurun_out/r5bench; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 300 $out/bench_n1.json; echo
O=$R/$out/prof; mkdir -p $O
( cd /tmp && TTS_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python $R/bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench.log ); echo "rocprof rc=$? [$(( $(date +%s) - t0 )) s]"
BS=$(find $O/bench -name "*kernel_stats.csv" | head -1)
python tools/summarize_profiles.py stats "$BS" $out/r5_bench_kernel_stats.csv "TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1   (final round-5 code; the bench command itself: 1 warm-up + 2 timed passes; graphs off so that every kernel is traced; same box as r5_bench_n1.json)"
cp $O/bench_under_rocprof.json $out/bench_under_rocprof.json; rm -rf $O
head -12 $out/r5_bench_kernel_stats.csv | cut -c1-150
