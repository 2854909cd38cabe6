#!/bin/bash
# round 4, GPU call 2: attention kernel A/B (four tile ranges vs the round-3 body), decode chain after the prologue fixes, the 256-column GEMM kernel at
# M = 28032 vs 28672, AR + diffusion parity, a short bench line.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c2; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 60 tools/bin/attn_bench > $out/attn_bench.txt 2>&1; echo "attn_bench rc=$?"; cat $out/attn_bench.txt | head -6
timeout 120 tools/bin/dec_bench_trace > $out/dec_bench_trace.txt 2>&1; echo "dec_bench rc=$?"; tail -12 $out/dec_bench_trace.txt
for f in "in_layers" "pad_in_layers" "proj_out" "qkv" "conv3"; do timeout 100 tools/bin/gemm_tab_bench "$f" arith=0 big=0 >> $out/gemm_tab.txt 2>&1; done; echo "gemm_tab rc=$? [$(( $(date +%s) - t0 )) s]"; grep -E "^==|arith|big" $out/gemm_tab.txt
timeout 900 python -m pytest tests/test_ar_gpu.py tests/test_diffusion_gpu.py -m gpu -x -q -s > $out/tests_a.log 2>&1; echo "tests_a rc=$? [$(( $(date +%s) - t0 )) s]"; tail -3 $out/tests_a.log
grep -E "\[reference precision" $out/tests_a.log | tail -30
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ab > $out/bench_short.json 2> $out/bench_short.err; echo "bench rc=$? [$(( $(date +%s) - t0 )) s]"; head -c 1500 $out/bench_short.json
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "test_diffusion_forward_full_depth or test_sampling_loop_80_steps or test_full_size_80_steps_at_bench_length or test_full_size_ar_192" > $out/tests_fullsize.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"
grep -E "\[reference precision|\[throughput|passed|failed|AR" $out/tests_fullsize.log | tail -30
