"""Developer tool: timeline of the single-utterance diffusion stage (1 candidate, T = 870, N steps, launched eagerly) under
    cd /tmp && TTS_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d OUT -o b1 -- python tools/b1_timeline.py run 20
then   python tools/b1_timeline.py summarize OUT/.../b1_kernel_trace.csv profiles/r2_b1_diffusion_timeline.txt
-> per kernel: launches per step, mean duration, and the mean idle gap that FOLLOWS it on the device (next start - this end)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import tortoise_cpp_amd_loader
    pkg = tortoise_cpp_amd_loader.load()
    import bench
    d = "/tmp/tts_bench_models"
    bench.ensure_models(d, False, True)
    eng = pkg.Engine(0)
    eng.load(diffusion=d + "/ggml-diffusion-model.bin")
    steps = int(sys.argv[2])
    lat = [np.random.RandomState(0).randn(200, 1024).astype(np.float32)]
    eng.seed(0)
    eng.diffusion(lat, n_steps=steps, noise_mode=pkg.NOISE_DEVICE)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    # the sampling loop = everything from the first ddpm update's step onwards; keep it simple: the last 60 % of the launches
    ev = ev[len(ev) * 4 // 10:]
    n_upd = sum(1 for e in ev if "ddpm_update" in e[2]) or 1
    agg = {}
    for i, (s, e, name) in enumerate(ev[:-1]):
        short = name.split("(")[0].replace("void tts::", "").replace("tts::", "")[:70]
        a = agg.setdefault(short, [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        a[2] += max(0, ev[i + 1][0] - e)
    span = ev[-1][1] - ev[0][0]
    busy = sum(a[1] for a in agg.values())
    out = ["# single-utterance diffusion (1 candidate, T = 870), eager launches, rocprofv3 --kernel-trace; %d sampling steps in the window" % n_upd,
           "# window %.2f ms: kernels busy %.2f ms (%.0f %%), idle between kernels %.2f ms; per step %.3f ms" %
           (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, span / 1e6 / n_upd),
           "%-72s %9s %9s %11s %11s" % ("kernel", "per step", "mean us", "gap after", "us per step")]
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        out.append("%-72s %9.1f %9.2f %11.2f %11.1f" % (k, a[0] / n_upd, a[1] / a[0] / 1e3, a[2] / a[0] / 1e3, (a[1] + a[2]) / n_upd / 1e3))
    open(sys.argv[3], "w").write("\n".join(out) + "\n")
    print("\n".join(out))
