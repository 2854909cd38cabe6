"""GPU: one network forward per arithmetic mode, dumped for CPU-side comparison with the torch emulations (tools/r5/forward_compare.py).
Inputs are seeded, the CPU side regenerates them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tortoise_cpp_amd_loader  # noqa: E402

pkg = tortoise_cpp_amd_loader.load()
sys.path.insert(0, os.path.join(ROOT, "tools", "r5"))
from attn_modes import models  # noqa: E402

CASES = (("mid", 12), ("full", 20), ("full", 43))
MODES = {"default": {"attn_f32": 0, "attn_proj_f16": 0}, "allfp16": {"attn_f32": 0, "attn_proj_f16": 1}, "f32": {"attn_f32": 1, "attn_proj_f16": 0}}


def inputs(L, T):
    lat = np.random.RandomState(100 + L).randn(L, 1024).astype(np.float32)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    return lat, x_t


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    eng = pkg.Engine(0)
    for kind, L in CASES:
        eng.load(diffusion=models(kind) + "/ggml-diffusion-model.bin")
        T = eng.frames(L)
        lat, x_t = inputs(L, T)
        for mode, opts in MODES.items():
            for k, v in opts.items():
                eng.set_option(k, v)
            for cf in (0, 1):
                for ts in (3999, 557):
                    y = eng.diffusion_forward(lat, x_t, ts, bool(cf))
                    np.save(os.path.join(out, "%s_L%d_%s_cf%d_t%d.npy" % (kind, L, mode, cf, ts)), y)
    eng.close()
    print("dumped", len(os.listdir(out)), "files")
