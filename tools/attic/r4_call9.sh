#!/bin/bash
# round 4, GPU call 9: which cross-kernel prefetch pairs pay (tools/dec_bench: chain of 30 layers in a hipGraph, per-pair masks, slab fractions, workgroup counts)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c9; mkdir -p $out
timeout 200 tools/bin/dec_bench 30 1 > $out/dec_bench_plain.txt 2>&1; echo "dec_bench rc=$?"; grep -E "layer chain" $out/dec_bench_plain.txt | sed 's/layer chain in a hipGraph (30 layers x 5 launches, B = 16, 165 keys, plain build, residual stream in the h4 layout (round 4), //'
timeout 200 tools/bin/dec_bench_trace 30 1 > $out/dec_bench_trace.txt 2>&1; echo "dec_bench_trace rc=$?"; grep -B1 -A11 "prefetch mask 8, 8/8 of a slab, 64\|prefetch mask 27, 8/8 of a slab, 64" $out/dec_bench_trace.txt | cut -c1-260
