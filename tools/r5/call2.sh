#!/bin/bash
# round 5, GPU call 2: fp16-subnormal probe, attention modes with the P scale, packed-f32 build A/B, the new bench keys, the ragged tests
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5c2; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
tools/bin/mfma_denorm_probe > $out/mfma_denorm_probe.txt 2>&1; cat $out/mfma_denorm_probe.txt
timeout 600 python tools/r5/attn_modes.py small mid full20 cost > $out/attn_modes.log 2>&1; echo "attn_modes rc=$? [$(( $(date +%s) - t0 )) s]"; cat $out/attn_modes.log
timeout 900 python -m pytest tests/test_ragged_gpu.py -x -q -s > $out/ragged.log 2>&1; echo "ragged rc=$? [$(( $(date +%s) - t0 )) s]"; tail -15 $out/ragged.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab > $out/bench_nopk_$rep.json 2> $out/bench_nopk_$rep.err; echo "bench no-packed rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 260 $out/bench_nopk_$rep.json; echo
  TTS_LIB_PATH=$PWD/tools/bin/libtortoise_mi355x_pk.so timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ab > $out/bench_pk_$rep.json 2> $out/bench_pk_$rep.err; echo "bench packed rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 260 $out/bench_pk_$rep.json; echo
done
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_full_line.json 2> $out/bench_full_line.err; echo "bench full line rc=$? [$(( $(date +%s) - t0 )) s]"; tail -c 3000 $out/bench_full_line.json; tail -5 $out/bench_full_line.err
