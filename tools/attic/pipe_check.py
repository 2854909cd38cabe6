"""Experiment: two engines (two HIP streams) on one GPU, AR of one half-batch overlapped with diffusion of the other."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
import bench
d = "/tmp/tts_bench_models"
bench.ensure_models(d, False, True)
toks = bench.synthetic_prompt()
voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
S = 192
def full(eng, B, seed):
    eng.seed(seed)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True)
    mels = eng.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE)
    return eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
e1 = pkg.Engine(0); e1.load(d)
full(e1, 16, 0)
t0 = time.time(); full(e1, 16, 1); t1 = time.time()
print("single engine B=16: %.3f s" % (t1 - t0))
e2 = pkg.Engine(0); e2.load(d)
full(e2, 8, 0); full(e1, 8, 0)
# two threads, each a full pipeline over 8 candidates, started together (stages interleave naturally)
def run(e, out, seed): out.append(full(e, 8, seed))
for trial in range(2):
    o1, o2 = [], []
    th1 = threading.Thread(target=run, args=(e1, o1, 10)); th2 = threading.Thread(target=run, args=(e2, o2, 11))
    t0 = time.time(); th1.start(); th2.start(); th1.join(); th2.join(); t1 = time.time()
    print("two engines x B=8 concurrently: %.3f s" % (t1 - t0))
# staggered: engine 2 starts its AR when engine 1 enters diffusion
def ar_then(e, B, seed, ev_ar_done, res):
    e.seed(seed)
    codes, rows, lats, steps = e.autoregressive(toks, voice, B, S, mask_stop=True)
    ev_ar_done.set()
    mels = e.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE)
    res.append(e.vocoder(mels, noise_mode=pkg.NOISE_DEVICE))
for trial in range(2):
    r1, r2 = [], []
    ev1, ev2 = threading.Event(), threading.Event()
    t0 = time.time()
    th1 = threading.Thread(target=ar_then, args=(e1, 8, 20, ev1, r1)); th1.start()
    ev1.wait()
    th2 = threading.Thread(target=ar_then, args=(e2, 8, 21, ev2, r2)); th2.start()
    th1.join(); th2.join(); t1 = time.time()
    print("staggered 8+8: %.3f s" % (t1 - t0))
