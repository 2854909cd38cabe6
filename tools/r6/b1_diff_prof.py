"""Developer tool (round 6): per-kernel-family device time (the engine's own HIP-event timers) of the SINGLE-UTTERANCE diffusion stage (B = 1: 2 sequences x 870 frames =
1 792 packed rows), default path and option latency_mode.   python tools/r6/b1_diff_prof.py [steps] [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
import bench
d = "/tmp/tts_bench_models"
bench.ensure_models(d, False, True)
eng = pkg.Engine(0)
eng.load(diffusion=d + "/ggml-diffusion-model.bin")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, L = int(sys.argv[2]) if len(sys.argv) > 2 else 1, 200
lats = [np.random.RandomState(c).randn(L, 1024).astype(np.float32) for c in range(B)]
FAMS = ["diff_gemm", "diff_gemm_k3r", "diff_gemm_k3", "diff_gemm_qkv", "diff_gemm_k1", "diff_gemm_k1r", "diff_gemm_misc", "diff_attn", "diff_gn_fused", "diff_gn_apply", "diff_gn_stats", "diff_update"]
for lat, hoist in ((0, 0), (0, 100000), (1, 100000)):
    eng.set_option("latency_mode", lat)
    eng.set_option("hoist_integrator", hoist)
    eng.seed(0)
    eng.diffusion(lats, n_steps=2, noise_mode=pkg.NOISE_DEVICE)
    eng.prof_reset(False)
    t0 = time.time(); eng.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE); t1 = time.time()
    print("latency_mode=%d hoist_integrator=%d B=%d: 80 steps, graph replay: %.1f ms whole stage call (%.3f ms/step)" % (lat, hoist, B, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / 80))
    eng.set_option("prof_eager_every", 1)
    eng.prof_reset(True)
    t0 = time.time(); eng.diffusion(lats, n_steps=steps, noise_mode=pkg.NOISE_DEVICE); t1 = time.time()
    tot = 0.0
    for f in FAMS:
        ms, n, w = eng.prof_get(f)
        if n == 0:
            continue
        if f != "diff_gemm":
            tot += ms
        print("   %-15s %8.3f ms/step %6.1f launches/step %7.1f us/launch  %s" % (f, ms / steps, n / steps, 1e3 * ms / max(n, 1),
              ("%.0f TF/s" % (w / (ms * 1e-3) / 1e12)) if w > 0 and ms > 0 else ""))
    print("   sum of the bracketed families: %.3f ms/step (eager, every launch between two events)" % (tot / steps))
    eng.set_option("prof_eager_every", 8)
    eng.prof_reset(False)
