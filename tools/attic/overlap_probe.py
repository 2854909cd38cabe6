#!/usr/bin/env python3
"""Developer probe: does the AR stage of the NEXT batch overlap with the diffusion stage of the CURRENT one when they run on two
contexts (two HIP streams) of one process?  Prints each stage's time alone and while the other stage is running."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader
import bench

pkg = tortoise_cpp_amd_loader.load()
md = "/tmp/tts_bench_models"
bench.ensure_models(md, False, True)
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
toks = bench.synthetic_prompt(0)
ea = pkg.Engine(0); eb = pkg.Engine(0)
for k, v in [x.split("=") for x in sys.argv[1:]]:
    (ea if k.startswith("a:") else eb).set_option(k[2:], float(v))
ea.load(ar=md + "/ggml-model.bin")
eb.load(diffusion=md + "/ggml-diffusion-model.bin", vocoder=md + "/ggml-vocoder-model.bin")
print("options:", sys.argv[1:])


def ar():
    t = time.time(); r = ea.autoregressive(toks, voice, 16, 192, mask_stop=True); return time.time() - t, r


def diff(lats):
    t = time.time(); m = eb.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE); t1 = time.time(); eb.vocoder(m, noise_mode=pkg.NOISE_DEVICE); return t1 - t, time.time() - t1


ea.seed(1); eb.seed(1)
_, r = ar(); lats = r[2]
diff(lats)
print("alone: AR %s ms" % [round(1e3 * ar()[0], 1) for _ in range(3)])
print("alone: diffusion, vocoder %s ms" % [tuple(round(1e3 * x, 1) for x in diff(lats)) for _ in range(2)])
res = {"ar": [], "diff": []}
stop = threading.Event()


def ta():
    while not stop.is_set():
        res["ar"].append(round(1e3 * ar()[0], 1))


th = threading.Thread(target=ta); t0 = time.time(); th.start()
for _ in range(4):
    res["diff"].append(tuple(round(1e3 * x, 1) for x in diff(lats)))
stop.set(); th.join(); dt = time.time() - t0
print("together: AR %s ms | diffusion, vocoder %s ms | %d AR passes + 4 diffusion passes in %.1f ms" % (res["ar"], res["diff"], len(res["ar"]), 1e3 * dt))
