#!/bin/bash
# rocprofv3 passes behind profiles/r6_*.{csv,json} (run on the GPU box). Counters are collected in their own runs (--pmc with --kernel-trace only).
# TTS_NO_GRAPH=1: graph replays are launched kernel by kernel so every kernel is traced.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TTS_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python $R/bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/tools/diff_prof.py 2 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/tools/diff_prof.py 2 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python $R/tools/diff_prof.py 2 > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dec -o d -- python $R/tools/ar_decode_only.py 16 > $O/dec.log 2>&1
cd $R
BS=$(find $O/bench -name "*kernel_stats.csv" | head -1); FC=$(find $O/fetch -name "*counter_collection.csv" | head -1)
WC=$(find $O/write -name "*counter_collection.csv" | head -1); MC=$(find $O/mfma -name "*counter_collection.csv" | head -1); DC=$(find $O/dec -name "*counter_collection.csv" | head -1)
python tools/summarize_profiles.py stats "$BS" $O/r6_bench_kernel_stats.csv "TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-ab --steps 2 --warmup 1   (the bench command itself: 1 warm-up + 2 timed passes; graphs off so that every kernel is traced)"
python tools/summarize_profiles.py pmc "$FC" "$WC" $O/r6_pmc_hbm_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on TTS_NO_GRAPH=1 python tools/diff_prof.py 2 (B=16, T=870 diffusion batch + vocoder, round-6 kernels). hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE in bytes: MI355X_MICROARCH.md (HBM) - on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (global_load and LDS-DMA alike); WRITE_SIZE taken at face value (calibrated on the GroupNorm kernel in round 1: 57 MB of fp16 written per launch). Infinity-Cache hits are counted, not excluded."
python tools/summarize_profiles.py mfma "$MC" $O/r6_pmc_mfma_util.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES on TTS_NO_GRAPH=1 python tools/diff_prof.py 2 (round-6 kernels). mfma_util = matrix-pipe busy cycles / (cycles per SIMD x 1024 SIMDs), cycles per SIMD = GRBM_GUI_ACTIVE / 8. Counter collection serialises the launches and the chip clocks higher than in the un-profiled pipeline: read the utilisation ratio, not the durations."
python tools/summarize_profiles.py decode "$DC" $O/r6_pmc_decode_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE (own pass), TTS_NO_GRAPH=1 python tools/ar_decode_only.py 16 (B = 16 candidates, 30 layers, decode steps 0-15: context 69-84 keys), round-6 kernels (h4 residual-stream layout). FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes. WRITE_SIZE is not included (the step writes ~2 MB of activations, K/V rows and logits)."
find $O -name "*.csv" ! -name "r6_*" -size +2M -delete
ls -la $O | head -20; head -12 $O/r6_bench_kernel_stats.csv; head -c 400 $O/bench_under_rocprof.json; echo; cat $O/r6_pmc_decode_traffic.json | head -30
