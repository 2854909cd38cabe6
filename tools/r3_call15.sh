#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r3c15; mkdir -p $out; rm -f $out/overlap.txt
for n in 1 2 4; do timeout 300 python -u tools/overlap_probe.py a:stream_cus=$n b:stream_cus=-$n >> $out/overlap.txt 2>&1; done
cat $out/overlap.txt
