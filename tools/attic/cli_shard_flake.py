"""Developer probe: `tortoise --candidates 4` against `--devices 2 --device-map 0,0` (two worker processes sharing one GPU), repeated, per option set."""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402
tortoise_cpp_amd_loader.load()
from tortoise_cpp_amd import synth_weights as sw  # noqa: E402
src = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "small")
if not os.path.exists(os.path.join(src, ".done")):
    sw.write_all(src, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)
    open(os.path.join(src, ".done"), "w").write("ok")
tmp = tempfile.mkdtemp()
d = os.path.join(tmp, "models"); os.mkdir(d)
for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
    os.symlink(os.path.join(src, f), os.path.join(d, f))
shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), os.path.join(d, "tokenizer.json"))
exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
base = [exe, "--models", d, "--message", "this is a test message.", "--voice", os.path.join(ROOT, "models", "mol.bin"), "--seed", "3", "--codes", "16", "--steps", "4",
        "--candidates", "4"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
optsets = ([], ["--option", "device_topk=0"]) if len(sys.argv) < 3 else ([],)
load = None
if len(sys.argv) > 2:  # a third process keeps the GPU busy meanwhile (the test suite's own engine does)
    load = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import tortoise_cpp_amd_loader as l, numpy as np, time; pkg = l.load(); e = pkg.Engine(0); e.load(%r); rs = np.random.RandomState(0)\nwhile True:\n    try:\n        e.diffusion([rs.randn(30, 1024).astype(np.float32) for _ in range(4)], n_steps=6, noise_mode=pkg.NOISE_DEVICE)\n    except Exception as ex:\n        print('load:', ex, flush=True)" % (ROOT, src)])
for opts in optsets:
    def run(tag, extra):
        out = os.path.join(tmp, tag + ".wav")
        r = subprocess.run(base + opts + ["--output", out] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        files = [out] + [os.path.join(tmp, "%s.wav.%d.wav" % (tag, c)) for c in range(1, 4)]
        return [np.frombuffer(open(f, "rb").read()[44:], np.float32).copy() for f in files]
    one = run("one", [])
    one_b = run("one", [])
    print(opts, "single process twice: max diff", [float(np.abs(a - b).max()) for a, b in zip(one, one_b)], flush=True)
    for rep in range(reps):
        two = run("two", ["--devices", "2", "--device-map", "0,0"])
        line = []
        for c in range(4):
            dd = np.abs(one[c] - two[c])
            nz = np.nonzero(dd)[0]
            line.append("c%d: %.1e%s" % (c, dd.max(), (" [%d..%d of %d, %d samples]" % (nz[0], nz[-1], len(dd), len(nz))) if len(nz) else ""))
        print(opts, "rep", rep, " | ".join(line), flush=True)
if load:
    load.kill()
shutil.rmtree(tmp)
