"""Developer probe: is one context's pipeline bit-reproducible while ANOTHER process keeps the GPU busy? Per stage, per option set."""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
load = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tortoise_cpp_amd_loader as l, numpy as np; pkg = l.load(); e = pkg.Engine(0); e.load(%r); rs = np.random.RandomState(0)\nwhile True:\n    e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)" % (ROOT, src)])
toks = np.array([255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0], np.int32)
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)[:1024]
try:
    for name, opts in (("default", {}), ("diff_graph=0", {"diff_graph": 0}), ("device_topk=0", {"device_topk": 0})):
        e = pkg.Engine(0); e.load(src)
        for k, v in opts.items():
            e.set_option(k, v)
        ref = None
        bad = {"codes": 0, "latents": 0, "mel": 0, "mel_fixed_lat": 0, "audio_fixed_mel": 0}
        for rep in range(reps):
            e.seed(3)
            codes, rows, lats, _ = e.autoregressive(toks, voice, 2, 16, mask_stop=True, retire=True)
            mels = e.diffusion(lats, n_steps=4, noise_mode=pkg.NOISE_DEVICE)
            if ref is None:
                ref = (codes, lats, mels, e.vocoder(mels, noise_mode=pkg.NOISE_DEVICE))
                continue
            e.seed(3); e.autoregressive(toks, voice, 2, 16, mask_stop=True, retire=True)  # same RNG position as the reference run had
            m_fixed = e.diffusion(ref[1], n_steps=4, noise_mode=pkg.NOISE_DEVICE)
            a_fixed = e.vocoder(ref[2], noise_mode=pkg.NOISE_DEVICE)
            bad["codes"] += not (codes == ref[0]).all()
            bad["latents"] += not all(np.array_equal(a, b) for a, b in zip(lats, ref[1]))
            bad["mel"] += not all(np.array_equal(a, b) for a, b in zip(mels, ref[2]))
            bad["mel_fixed_lat"] += not all(np.array_equal(a, b) for a, b in zip(m_fixed, ref[2]))
            bad["audio_fixed_mel"] += not all(np.array_equal(a, b) for a, b in zip(a_fixed, ref[3]))
        print(name, "runs that differ from the first, of", reps - 1, ":", bad, flush=True)
        e.close()
finally:
    load.kill()
