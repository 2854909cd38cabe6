#!/usr/bin/env python3
"""Developer experiment: the decode step is a chain of 151 short launches, each with a cold-start memory round trip. Do two independent half-batches
(8 + 8 candidates on two contexts = two streams) interleave their chains well enough to beat one batch of 16, although every weight is then streamed
twice per step? Prints the AR stage time of each variant."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402

pkg = tortoise_cpp_amd_loader.load()
import bench  # noqa: E402

MODELS = "/tmp/tts_bench_models"
bench.ensure_models(MODELS, False, True)
engs = [pkg.Engine(0) for _ in range(2)]
for e in engs:
    e.load(ar=os.path.join(MODELS, "ggml-model.bin"))
toks = bench.synthetic_prompt()
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)


def run(e, B, out, i):
    e.seed(7 + i)
    out[i] = e.autoregressive(toks, voice, B, 192, mask_stop=True, want_latents=False)


def timed(bs):
    best = 1e9
    for _ in range(3):
        out = [None] * len(bs)
        th = [threading.Thread(target=run, args=(engs[i], b, out, i)) for i, b in enumerate(bs)]
        t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.time() - t0)
    return best * 1e3


timed([8, 8])
print("one context, 16 candidates, 192 decode steps:   %.1f ms" % timed([16]))
print("two contexts, 8 + 8 concurrently:               %.1f ms" % timed([8, 8]))
print("one context, 8 candidates:                      %.1f ms" % timed([8]))
