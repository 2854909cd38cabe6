#!/bin/bash
# developer probe: the kernel microbenches' own bit-equality checks while another process keeps the GPU busy
cd "$(dirname "$0")/.." || exit 1
python - <<'PY' &
import sys, os
sys.path.insert(0, os.getcwd())
import tortoise_cpp_amd_loader as l, numpy as np
pkg = l.load()
from tortoise_cpp_amd import synth_weights as sw
src = "/tmp/tts_synth/small"
if not os.path.exists(src + "/.done"):
    sw.write_all(src, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321); open(src + "/.done", "w").write("ok")
e = pkg.Engine(0); e.load(src); rs = np.random.RandomState(0)
print("load running", flush=True)
while True:
    e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)
PY
LOADPID=$!
sleep 14
for i in 1 2 3; do timeout 60 tools/bin/attn_bench 2>&1 | grep -i "differ" ; done
for i in 1 2; do timeout 120 tools/bin/gemm_small_diag 2>&1 | grep -E "1792|3584" | grep -v "bit-identical" | head -12; done
kill $LOADPID
