"""CPU: the engine's dumped forwards (tools/r5/forward_dump.py) against the oracle, torch-f32 and the torch emulations of the fp16 roundings."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle as O  # noqa: E402
import torch_ref as TR  # noqa: E402
from regen_parity_floor import ensure_models  # noqa: E402

torch.set_num_threads(8)
d = sys.argv[1]
for kind, L in (("mid", 12), ("full", 20), ("full", 43)):
    path = ensure_models(kind)
    od = O.Diffusion(O.Model(path))
    T = od.T_of(L)
    lat = np.random.RandomState(100 + L).randn(L, 1024).astype(np.float32)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    ce = od.code_embedding(lat, T)
    nets = {"t64": TR.TorchDiffusion(path, O.buckets, dtype=torch.float64), "t32": TR.TorchDiffusion(path, O.buckets),
            "e32 qk,v,p,o": TR.TorchDiffusion(path, O.buckets, f16_attention="qk,v,p,o"), "e32 all": TR.TorchDiffusion(path, O.buckets, f16_attention=True),
            "e64 qk,v,p,o": TR.TorchDiffusion(path, O.buckets, dtype=torch.float64, f16_attention="qk,v,p,o")}
    for cf in (0, 1):
        for ts in (3999, 557):
            te = O.timestep_embedding(ts)
            ref = {k: n.forward(None if cf else ce, x_t, te) for k, n in nets.items()}
            ref["oracle"] = od.forward(None if cf else ce, x_t, ts)
            sc = np.abs(ref["t64"]).max()
            line = "%s L=%d cf=%d t=%d |" % (kind, L, cf, ts)
            for mode in ("default", "allfp16", "f32"):
                y = np.load(os.path.join(d, "%s_L%d_%s_cf%d_t%d.npy" % (kind, L, mode, cf, ts)))
                line += " %s:" % mode + " ".join("%s %.2e" % (k.split()[0] if k != "e32 qk,v,p,o" and k != "e64 qk,v,p,o" else k.replace(" qk,v,p,o", "-w"), np.abs(y - r).mean() / sc * 100) for k, r in ref.items()) + " |"
            print(line, flush=True)
            print("      refs vs t64: " + " ".join("%s %.2e" % (k, np.abs(r - ref["t64"]).mean() / sc * 100) for k, r in ref.items() if k != "t64"), flush=True)
