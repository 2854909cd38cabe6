"""Developer probe: which shapes of the diffusion stage lose bit-reproducibility while another PROCESS keeps the GPU busy (argv[1] = 'noload': without it)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402
pkg = tortoise_cpp_amd_loader.load()
from tortoise_cpp_amd import synth_weights as sw  # noqa: E402
src = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "small")
if not os.path.exists(os.path.join(src, ".done")):
    sw.write_all(src, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)
    open(os.path.join(src, ".done"), "w").write("ok")
load = None
if not (len(sys.argv) > 1 and sys.argv[1] == "noload"):
    load = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tortoise_cpp_amd_loader as l, numpy as np; pkg = l.load(); e = pkg.Engine(0); e.load(%r); rs = np.random.RandomState(0)\nwhile True:\n    e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)" % (ROOT, src)])
import time
if load:
    time.sleep(12)  # until the other process has loaded its models and is launching kernels
rs = np.random.RandomState(1)
reps = 16
try:
    for mode in (0, 1):
        e = pkg.Engine(0); e.load(diffusion=src + "/ggml-diffusion-model.bin"); e.set_option("attn_f32", mode)
        for L in (24, 43, 57, 100, 200):
            lat = rs.randn(L, 1024).astype(np.float32)
            T = e.frames(L)
            x = rs.randn(100, T).astype(np.float32)
            ref = e.diffusion_forward(lat, x, 500, False)
            bad = sum(not np.array_equal(e.diffusion_forward(lat, x, 500, False), ref) for _ in range(reps))
            print("attn_f32=%d forward L=%3d T=%3d: %2d of %d differ" % (mode, L, T, bad, reps), flush=True)
        for B, L in ((1, 24), (2, 24), (4, 24), (4, 57), (2, 100)):
            lats = [rs.randn(L, 1024).astype(np.float32) for _ in range(B)]
            e.seed(1); ref = e.diffusion(lats, n_steps=3, noise_mode=pkg.NOISE_DEVICE)
            bad = 0
            for _ in range(reps):
                e.seed(1); m = e.diffusion(lats, n_steps=3, noise_mode=pkg.NOISE_DEVICE)
                bad += not all(np.array_equal(a, b) for a, b in zip(m, ref))
            print("attn_f32=%d 3-step loop B=%d L=%3d: %2d of %d differ" % (mode, B, L, bad, reps), flush=True)
        e.close()
finally:
    if load:
        load.kill()
