"""Developer tool: per-kernel-family device time of the diffusion + vocoder stages at the bench size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
import bench
d = "/tmp/tts_bench_models"
bench.ensure_models(d, False, True)
eng = pkg.Engine(0)
eng.load(diffusion=d + "/ggml-diffusion-model.bin", vocoder=d + "/ggml-vocoder-model.bin")
B, L, steps = 16, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 8
lats = [np.random.RandomState(c).randn(L, 1024).astype(np.float32) for c in range(B)]
eng.seed(0)
eng.diffusion(lats, n_steps=2, noise_mode=pkg.NOISE_DEVICE)
for prof in (False, True):
    eng.prof_reset(prof)
    t0 = time.time(); mels = eng.diffusion(lats, n_steps=steps, noise_mode=pkg.NOISE_DEVICE); t1 = time.time()
    print("diffusion %d steps prof=%s: %.1f ms/step" % (steps, prof, 1e3 * (t1 - t0) / steps))
    if prof:
        for f in ["diff_gemm", "diff_gemm_k3r", "diff_gemm_qkv", "diff_gemm_k1", "diff_gemm_k1r", "diff_attn", "diff_gn_fused", "diff_gn_stats", "diff_update"]:
            ms, n, w = eng.prof_get(f)
            print("   %-14s %8.2f ms/step %6d launches/step %7.1f us/launch  %s" % (f, ms / steps, n // steps, 1e3 * ms / max(n, 1),
                  ("%.0f TF/s" % (w / (ms * 1e-3) / 1e12)) if w > 0 and ms > 0 else ""))
eng.prof_reset(True)
t0 = time.time(); au = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE); t1 = time.time()
print("vocoder: %.1f ms" % (1e3 * (t1 - t0)))
for f in ["voc_conv", "voc_convt", "voc_kernel_gemm", "voc_lvc"]:
    ms, n, w = eng.prof_get(f)
    print("   %-16s %8.2f ms %5d launches" % (f, ms, n))
