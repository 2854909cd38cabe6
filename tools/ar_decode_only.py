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
eng.load(ar=d + "/ggml-model.bin")
toks = bench.synthetic_prompt()
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
B, S = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng.ar_begin(toks, voice, B, S)
eng.ar_prefill()
prev = np.full(B, 100, np.int32)
t0 = time.time()
for i in range(S):
    eng.ar_step(prev, i)
print("decode %.2f ms/step" % (1e3 * (time.time() - t0) / S))
