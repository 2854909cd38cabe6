"""Developer tool (round 6): the prompt pass and the latent pass of ONE candidate (the two non-decode pieces of a single utterance's AR stage), for rocprofv3 --kernel-trace --stats.
python tools/r6/ar_small_prof.py [B] [repeats]"""
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
eng.load(ar=d + "/ggml-model.bin")
toks = bench.synthetic_prompt()
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
codes = np.random.RandomState(0).randint(0, 8192, (B, 502)).astype(np.int32)
tp, tl = [], []
for r in range(R + 1):
    eng.ar_begin(toks, voice, B, 192)
    t0 = time.time(); eng.ar_prefill(); t1 = time.time()
    lat = eng.ar_latents(codes, 201); t2 = time.time()
    if r:
        tp.append(1e3 * (t1 - t0)); tl.append(1e3 * (t2 - t1))
print("B=%d: prompt pass %.2f ms, latent pass (201 mel positions) %.2f ms (best of %d; mean %.2f / %.2f)" % (B, min(tp), min(tl), R, sum(tp) / R, sum(tl) / R))
