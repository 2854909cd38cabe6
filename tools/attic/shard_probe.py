"""Developer probe: does a context holding candidates [c0, c0 + b) of a batch reproduce the batch BIT FOR BIT (codes, latents, mel)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tortoise_cpp_amd_loader  # noqa: E402
pkg = tortoise_cpp_amd_loader.load()
from tortoise_cpp_amd import synth_weights as sw  # noqa: E402
d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "small")
if not os.path.exists(os.path.join(d, ".done")):
    sw.write_all(d, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)
    open(os.path.join(d, ".done"), "w").write("ok")
toks = np.array([255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0], np.int32)
voice = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "mol.bin"), np.float32)[:1024]
B, S = 4, 16
for topk in (1, 0):
    full = pkg.Engine(0); full.load(d); full.set_option("device_topk", topk); full.seed(3)
    codes, rows, lats, _ = full.autoregressive(toks, voice, B, S, mask_stop=True, retire=True)
    mels = full.diffusion(lats, n_steps=4, noise_mode=pkg.NOISE_DEVICE)
    audio = full.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
    full.close()
    for r in range(2):
        e = pkg.Engine(0); e.load(d); e.set_option("device_topk", topk)
        e.set_option("rng_shard_offset", 2 * r); e.set_option("rng_shard_total", B); e.seed(3)
        c2, r2, l2, _ = e.autoregressive(toks, voice, 2, S, mask_stop=True, retire=True)
        m2 = e.diffusion(l2, n_steps=4, noise_mode=pkg.NOISE_DEVICE)
        a2 = e.vocoder(m2, noise_mode=pkg.NOISE_DEVICE)
        for k in range(2):
            g = 2 * r + k
            print("device_topk=%d rank %d cand %d: codes equal %s, rows %d/%d, latents max abs diff %.3e (max %.2f), mel max abs diff %.3e, audio max abs diff %.3e (max %.2f)"
                  % (topk, r, g, bool((c2[k] == codes[g]).all()), r2[k], rows[g], float(np.abs(l2[k] - lats[g]).max()), float(np.abs(lats[g]).max()),
                     float(np.abs(m2[k] - mels[g]).max()), float(np.abs(a2[k] - audio[g]).max()), float(np.abs(audio[g]).max())), flush=True)
        e.close()
