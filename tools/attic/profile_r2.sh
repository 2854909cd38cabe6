#!/bin/bash
# rocprofv3 passes behind profiles/r2_*.{csv,json} (run on the GPU box through tools/gpu_call.sh cmd:...). Counters are collected in
# their own runs (--pmc with --kernel-trace only). TTS_NO_GRAPH=1: graph replays are launched kernel by kernel so every kernel is traced.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TTS_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -o e2e -- python $R/tools/e2e_once.py 80 > $O/e2e.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python $R/tools/diff_prof.py 2 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python $R/tools/diff_prof.py 2 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -o m -- python $R/tools/diff_prof.py 2 > $O/mfma.log 2>&1
cd $R
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
ES=$(find $O/e2e -name "*kernel_stats.csv" | head -1); FC=$(find $O/fetch -name "*counter_collection.csv" | head -1)
WC=$(find $O/write -name "*counter_collection.csv" | head -1); MC=$(find $O/mfma -name "*counter_collection.csv" | head -1)
python tools/summarize_profiles.py stats "$ES" $O/r2_e2e_kernel_stats.csv "TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python tools/e2e_once.py 80   (2 end-to-end passes: 16 candidates, 192 decode steps, 80 diffusion steps, vocoder)"
python tools/summarize_profiles.py pmc "$FC" "$WC" $O/r2_pmc_hbm_traffic.json "PLACEHOLDER"
python tools/summarize_profiles.py mfma "$MC" $O/r2_pmc_mfma_util.json "PLACEHOLDER"
# raw CSVs are large: keep only the summaries
find $O -name "*.csv" ! -name "r2_*" -size +2M -delete
ls -la $O
