"""Developer tool: per-kernel-family device time of the AR stage (decode loop vs latent pass)."""
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
B, S = 16, 192
fams = ["ar_gemv", "ar_epilogue", "ar_attention", "ar_layernorm"]
def report(tag, wall):
    print(tag, "wall %.1f ms" % (wall * 1e3))
    for f in fams:
        ms, n, w = eng.prof_get(f)
        extra = "  %.2f TB/s" % (w / (ms * 1e-3) / 1e12) if f == "ar_gemv" and ms > 0 else ""
        print("   %-14s %8.2f ms %7d launches %6.2f us/launch%s" % (f, ms, n, 1e3 * ms / max(n, 1), extra))
for prof in (False, True):
    eng.ar_begin(toks, voice, B, S)
    eng.prof_reset(prof)
    t0 = time.time(); lg = eng.ar_prefill(); t1 = time.time()
    if prof: report("prefill", t1 - t0)
    eng.prof_reset(prof)
    t0 = time.time()
    prev = np.full(B, 100, np.int32)
    for i in range(S):
        lg = eng.ar_step(prev, i)
    t1 = time.time()
    print("decode %d steps prof=%s: %.2f ms/step" % (S, prof, 1e3 * (t1 - t0) / S))
    if prof: report("decode", t1 - t0)
    eng.prof_reset(False)
    t0 = time.time(); s = eng.sample(lg, prev.reshape(B, 1)); t1 = time.time()
    print("host sampler B=16: %.2f ms" % (1e3 * (t1 - t0)))
    codes = np.random.RandomState(0).randint(0, 8192, (B, 502)).astype(np.int32)
    eng.prof_reset(prof)
    t0 = time.time(); lat = eng.ar_latents(codes, 201); t1 = time.time()
    print("latents n_mel=201 prof=%s: %.1f ms" % (prof, 1e3 * (t1 - t0)))
    if prof: report("latents", t1 - t0)
