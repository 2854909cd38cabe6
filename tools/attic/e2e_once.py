"""One end-to-end pass at the bench configuration without event timing (for rocprofv3 runs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
import bench
d = "/tmp/tts_bench_models"
bench.ensure_models(d, False, True)
eng = pkg.Engine(0)
eng.load(d)
toks = bench.synthetic_prompt()
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
n_diff = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for it in range(2):
    eng.seed(it)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, 16, 192, mask_stop=True)
    mels = eng.diffusion(lats, n_steps=n_diff, noise_mode=pkg.NOISE_DEVICE)
    audio = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
print("ok", sum(len(a) for a in audio))
