"""Developer probe (build the debug library first: tools/build_debug_lib.sh): per-launch buffer hashes (-DTTS_DEBUG_CHECKSUM) of ONE diffusion forward, repeated while another process keeps the GPU
busy; prints the first launch whose output differs from the first repetition's."""
import os, subprocess, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402
pkg = tortoise_cpp_amd_loader.load()
pkg.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libtortoise_mi355x_dbg.so")
from tortoise_cpp_amd import synth_weights as sw  # noqa: E402
src = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "small")
if not os.path.exists(os.path.join(src, ".done")):
    sw.write_all(src, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)
    open(os.path.join(src, ".done"), "w").write("ok")
load = None
if os.environ.get("AGGRESSOR") == "probe":  # the stand-alone matrix-vector probe (3 copies) instead of an engine process
    load = subprocess.Popen("for i in 1 2 3; do %s 60 agg$i 48 & done; wait" % os.path.join(ROOT, "tools", "bin", "mp_corruption_probe"), shell=True, stdout=subprocess.DEVNULL, preexec_fn=os.setsid)
    time.sleep(3)
elif os.environ.get("AGGRESSOR") == "ar":  # an engine process that only runs the AR stage (decode-step graphs, no LDS-DMA kernels)
    load = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tortoise_cpp_amd_loader as l, numpy as np; pkg = l.load(); e = pkg.Engine(0); e.load(ar=%r + '/ggml-model.bin'); v = np.zeros(1024, np.float32); t = np.array([255, 5, 6, 7, 0], np.int32)\nwhile True:\n    e.seed(1); e.autoregressive(t, v, 16, 60, mask_stop=True)" % (ROOT, src)], stderr=subprocess.DEVNULL)
    time.sleep(12)
elif os.environ.get("NOLOAD") is None:
    load = subprocess.Popen([sys.executable, "-c", "import sys, os, ctypes; sys.path.insert(0, %r)\nif os.environ.get('LOADPAD'):\n    hip = ctypes.CDLL('libamdhip64.so'); pp = ctypes.c_void_p(); print('pad', hip.hipMalloc(ctypes.byref(pp), ctypes.c_size_t(int(os.environ['LOADPAD']) << 20)), hex(pp.value or 0), flush=True)\nimport tortoise_cpp_amd_loader as l, numpy as np; pkg = l.load(); e = pkg.Engine(0); e.load(%r); rs = np.random.RandomState(0)\nwhile True:\n    e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)" % (ROOT, src)],
                            stderr=subprocess.DEVNULL)  # LOADPAD=<MB>: the other process first allocates a dummy buffer, so that its buffers get other virtual addresses
    time.sleep(12)
if os.environ.get("VICTIMPAD"):  # this process first takes VICTIMPAD small device allocations, so that its small buffers get other virtual addresses than the load's
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    keep = []
    for i in range(int(os.environ["VICTIMPAD"])):
        pp = ctypes.c_void_p()
        hip.hipMalloc(ctypes.byref(pp), ctypes.c_size_t(int(os.environ.get("VICTIMPAD_BYTES", 0)) or 4096 * (1 + i % 7)))
        keep.append(pp)
    print("victim pad:", len(keep), "allocations, first", hex(keep[0].value or 0), "last", hex(keep[-1].value or 0))
rs = np.random.RandomState(1)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 43
reps = 30
log = tempfile.mktemp()
try:
    e = pkg.Engine(0); e.load(diffusion=src + "/ggml-diffusion-model.bin")
    lat = rs.randn(L, 1024).astype(np.float32)
    x = rs.randn(100, e.frames(L)).astype(np.float32)
    e.diffusion_forward(lat, x, 500, False)  # warm-up (allocations)
    traces = []
    saved = os.dup(2)
    for rep in range(reps):
        fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        os.dup2(fd, 2)
        e.diffusion_forward(lat, x, 500, False)
        os.dup2(saved, 2); os.close(fd)
        traces.append([l.strip() for l in open(log) if l.startswith("[cks]")])
        for l in open(log):
            if l.startswith("[dif]") or l.startswith("[retry]") or l.startswith("[kernarg]"):
                print("rep", rep, l.strip())
    ref = traces[0]
    print("launch outputs hashed per forward:", len(ref))
    from collections import Counter
    first = Counter()
    for t in traces[1:]:
        d = [i for i, (a, b) in enumerate(zip(ref, t)) if a != b]
        if d:
            first[" ".join(ref[d[0]].split()[1:-1]) + " (#%d)" % d[0]] += 1
            print("  differs from rep 0 at", len(d), "of", len(ref), "hashes; first:", ref[d[0]][:40], "->", t[d[0]][-17:])
        else:
            print("  identical")
    print("first divergent launch, counted over the repetitions:", dict(first))
    e.close()
finally:
    if load:
        if os.environ.get("AGGRESSOR") == "probe":
            os.killpg(os.getpgid(load.pid), 9)  # the shell AND the probes it started (our own process group, by its id)
        else:
            load.kill()
