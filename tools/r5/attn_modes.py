"""GPU: the three arithmetic modes of the diffusion AttentionBlock against the oracle on the loop problems of the parity tests, and what each costs.
  default          fp16 q/k/v/P/attention output, proj_out weight as a split pair (round 5)
  attn_proj_f16=1  the all-fp16 block of rounds 1-4
  attn_f32=1       reference precision (every product on split pairs)
usage: python tools/r5/attn_modes.py [small mid full20 full870 cost]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tortoise_cpp_amd_loader  # noqa: E402

pkg = tortoise_cpp_amd_loader.load()
import oracle as O  # noqa: E402
import conftest as CT  # noqa: E402

MODES = (("default (split W, f32 conditioner)", {"attn_f32": 0, "attn_proj_f16": 0, "lc_attn_f32": 1}),
         ("split W, fp16 conditioner", {"attn_f32": 0, "attn_proj_f16": 0, "lc_attn_f32": 0}),
         ("all-fp16 blocks, f32 conditioner", {"attn_f32": 0, "attn_proj_f16": 1, "lc_attn_f32": 1}),
         ("all-fp16 (rounds 1-4)", {"attn_f32": 0, "attn_proj_f16": 1, "lc_attn_f32": 0}),
         ("attn_f32", {"attn_f32": 1, "attn_proj_f16": 0, "lc_attn_f32": 1}))


def models(kind):
    if kind == "small":
        return CT._synth_dir(pkg, "small", ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)
    if kind == "mid":
        return CT._synth_dir(pkg, "mid", ar_layers=6, diff_main=3, diff_tail=1, diff_integ=1, diff_lc=2, seed=777)
    d = os.environ.get("TTS_BENCH_MODELS", "/tmp/tts_bench_models")
    if not os.path.exists(d + "/.done"):
        from tortoise_cpp_amd import synth_weights as sw
        sw.write_all(d, seed=1234)
        open(d + "/.done", "w").write("ok")
    return d


def problem(kind, L, lat_seed, noise_seed, steps=80, ablation_style=False):
    d = models(kind)
    od = O.Diffusion(O.Model(d + "/ggml-diffusion-model.bin"))
    T = od.T_of(L)
    if ablation_style:  # tools/regen_parity_floor.py --ablate: one RandomState for latents then noise
        rs = np.random.RandomState(lat_seed)
        lat = rs.randn(L, 1024).astype(np.float32)
        noise = rs.randn(steps + 1, 100 * T).astype(np.float32)
    else:
        lat = np.random.RandomState(lat_seed).randn(L, 1024).astype(np.float32)
        noise = np.random.RandomState(noise_seed).randn(steps + 1, 100 * T).astype(np.float32)
    t0 = time.time()
    want = od.sample(lat, n_steps=steps, noise=noise)
    print("  oracle %s L=%d T=%d: %.0f s" % (kind, L, T, time.time() - t0), flush=True)
    return d, lat, noise, want, steps


def main():
    what = sys.argv[1:] or ["small", "mid", "full20", "cost"]
    eng = pkg.Engine(0)
    probs = {"small": ("small", 12, 12, 5), "mid": ("mid", 12, 12, 5), "full20": ("full", 20, 9, None), "full870": ("full", 200, 31, 6)}
    for w in what:
        if w not in probs:
            continue
        kind, L, ls, ns = probs[w]
        d, lat, noise, want, steps = problem(kind, L, ls, ns, ablation_style=ns is None)
        eng.load(diffusion=d + "/ggml-diffusion-model.bin")
        for name, opts in MODES:
            for k, v in opts.items():
                eng.set_option(k, v)
            mel = eng.diffusion([lat], n_steps=steps, noise=[noise])[0]
            e = np.abs(mel - want)
            print("%-8s %-34s max %.3e mean %.3e" % (w, name, e.max(), e.mean()), flush=True)
    if "cost" in what:
        d = models("full")
        eng.load(diffusion=d + "/ggml-diffusion-model.bin")
        rs = np.random.RandomState(1)
        lat, xt = rs.randn(200, 1024).astype(np.float32), rs.randn(100, 870).astype(np.float32)
        ys = []
        eng.set_option("attn_f32", 0)
        for db in (1, 0):
            eng.set_option("proj_dual_b", db)
            ys.append(eng.diffusion_forward(lat, xt, 557, False))
        print("dual-B proj_out kernel vs two K segments, one full-depth forward at T=870: max rel diff %.2e" % (np.abs(ys[0] - ys[1]).max() / np.abs(ys[1]).max()), flush=True)
        for B in (16, 1):
            lats = [rs.randn(200, 1024).astype(np.float32) for _ in range(B)]
            for rep in range(2):
                for name, opts in MODES:
                    for k, v in opts.items():
                        eng.set_option(k, v)
                    eng.seed(1)
                    eng.diffusion(lats, n_steps=4, noise_mode=pkg.NOISE_DEVICE)
                    t0 = time.time()
                    eng.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE)
                    print("cost B=%d %-34s diffusion stage %.1f ms" % (B, name, 1e3 * (time.time() - t0)), flush=True)
    for k, v in (("attn_f32", 0), ("attn_proj_f16", 0), ("lc_attn_f32", 1)):
        eng.set_option(k, v)
    eng.close()


if __name__ == "__main__":
    main()
