#!/bin/bash
# round 4, GPU call 1: reference-precision attention (attn_f32) parity at small / mid / full depth, GEMM cube calibration, decode layer-chain
# latency breakdown, SQ counters of the attention kernel. Everything lands under gpurun_out/r4c1/.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c1; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_diffusion_gpu.py -m gpu -x -q -s > $out/tests_diffusion.log 2>&1; rc=$?; echo "tests_diffusion rc=$rc [$(( $(date +%s) - t0 )) s]"
grep -E "\[reference precision|\[throughput" $out/tests_diffusion.log | tail -40
timeout 200 tools/bin/gemm_cube_bench zero > $out/gemm_cube.txt 2>&1; echo "gemm_cube rc=$? [$(( $(date +%s) - t0 )) s]"
timeout 120 tools/bin/dec_bench_trace > $out/dec_bench_trace.txt 2>&1; echo "dec_bench rc=$? [$(( $(date +%s) - t0 )) s]"
timeout 60 tools/bin/attn_bench > $out/attn_bench.txt 2>&1; echo "attn_bench rc=$?"
(cd /tmp && rocprofv3 -L > $OLDPWD/$out/rocprof_counters.txt 2>&1)
R=$(pwd)
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$out/attn_pmc_a -o a -- $R/tools/bin/attn_bench > $R/$out/attn_pmc_a.log 2>&1; echo "pmc a rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/$out/attn_pmc_b -o b -- $R/tools/bin/attn_bench > $R/$out/attn_pmc_b.log 2>&1; echo "pmc b rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS SQ_VALU_TRANS_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/$out/attn_pmc_c -o c -- $R/tools/bin/attn_bench > $R/$out/attn_pmc_c.log 2>&1; echo "pmc c rc=$?"
cd $R
find $out -name "*.csv" -size +3M -delete
echo "[$(( $(date +%s) - t0 )) s] full-depth tests"
if [ $rc = 0 ]; then
  timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "test_diffusion_forward_full_depth or test_sampling_loop_80_steps or test_full_size_80_steps_at_bench_length" > $out/tests_fullsize.log 2>&1; echo "tests_fullsize rc=$? [$(( $(date +%s) - t0 )) s]"
  grep -E "\[reference precision|\[throughput|passed|failed" $out/tests_fullsize.log | tail -40
fi
