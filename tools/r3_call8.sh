#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c8; mkdir -p $out
export TMPDIR=/tmp
timeout 900 bash tools/profile_r3.sh > $out/profile.log 2>&1
unset TTS_NO_GRAPH
timeout 300 python bench.py --config 4 --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_config4.json 2> $out/bench_config4.err
timeout 400 python bench.py --config 5 --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_config5.json 2> $out/bench_config5.err
timeout 300 python bench.py --candidates 1 --steps 5 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err
tail -30 $out/profile.log; for f in config4 config5 b1; do head -c 250 $out/bench_$f.json; echo; done
