#!/usr/bin/env python3
"""Developer experiment: does running the diffusion stage as two concurrent half-batches (two contexts = two HIP streams on one GPU) hide the
GEMM tail rounds and epilogue bursts that profiles/r2_gemm_tile_phases.txt shows? Times 16 candidates x 80 steps (T = 870) as one batch on
one context, then as 8 + 8 on two contexts driven from two host threads. Prints one line per variant."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402

pkg = tortoise_cpp_amd_loader.load()
sys.path.insert(0, ROOT)
import bench  # noqa: E402

MODELS = "/tmp/tts_bench_models"
bench.ensure_models(MODELS, False, True)
engs = [pkg.Engine(0) for _ in range(2)]
for e in engs:
    e.load(diffusion=os.path.join(MODELS, "ggml-diffusion-model.bin"))
    e.seed(5)
rs = np.random.RandomState(3)
lats = [rs.randn(200, 1024).astype(np.float32) for _ in range(16)]


def run(e, ls, out, i):
    out[i] = e.diffusion(ls, n_steps=80, noise_mode=pkg.NOISE_DEVICE)


def timed(parts):
    best = 1e9
    for _ in range(3):
        out = [None] * len(parts)
        th = [threading.Thread(target=run, args=(engs[i], p, out, i)) for i, p in enumerate(parts)]
        t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        best = min(best, time.time() - t0)
    return best * 1e3


timed([lats[:2]])
timed([lats[:2], lats[2:4]])
print("one context, 16 candidates:        %.1f ms" % timed([lats]))
print("two contexts, 8 + 8 concurrently:  %.1f ms" % timed([lats[:8], lats[8:]]))
print("one context, 8 candidates:         %.1f ms" % timed([lats[:8]]))
