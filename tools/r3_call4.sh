#!/bin/bash
# round 3, GPU call 4: MALL residency of the decode slabs; full GPU suite with the round-3 tests; single-candidate and 16-candidate bench lines
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c4; mkdir -p $out
export TMPDIR=/tmp
timeout 120 tools/bin/dec_mall_bench > $out/dec_mall.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -s > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ab > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --candidates 1 --steps 5 --warmup 1 --no-cpu-baseline --no-ab > $out/bench_b1.json 2> $out/bench_b1.err
cat $out/dec_mall.txt; grep -E "passed|failed|rc=" $out/tests.log | tail -3; head -c 600 $out/bench.json; echo; head -c 600 $out/bench_b1.json
