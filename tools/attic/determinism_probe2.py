"""Developer probe: bit-reproducibility of ONE diffusion forward / one sampling loop while another process keeps the GPU busy, per option set."""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
load = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tortoise_cpp_amd_loader as l, numpy as np; pkg = l.load(); e = pkg.Engine(0); e.load(%r); rs = np.random.RandomState(0)\nwhile True:\n    e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)" % (ROOT, src)])
rs = np.random.RandomState(1)
try:
    for L in (24, 57):
        lat = rs.randn(L, 1024).astype(np.float32)
        for name, opts in (("default", {}), ("attn_f32=1", {"attn_f32": 1})):
            e = pkg.Engine(0); e.load(diffusion=src + "/ggml-diffusion-model.bin")
            for k, v in opts.items():
                e.set_option(k, v)
            T = e.frames(L)
            x = rs.randn(100, T).astype(np.float32)
            ref = [e.diffusion_forward(lat, x, 500, cf) for cf in (False, True)]
            bad = [0, 0]
            worst = 0.0
            for rep in range(reps):
                for i, cf in enumerate((False, True)):
                    o = e.diffusion_forward(lat, x, 500, cf)
                    if not np.array_equal(o, ref[i]):
                        bad[i] += 1
                        worst = max(worst, float(np.abs(o - ref[i]).max()))
            print("L=%d T=%d %s: forwards that differ from the first of %d: conditioned %d, unconditioned %d (largest difference %.2e)" % (L, T, name, reps, bad[0], bad[1], worst), flush=True)
            e.close()
finally:
    load.kill()
