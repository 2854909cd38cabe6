"""CPU: how far apart may two CORRECT evaluations of the 80-step sampling loop be?  

north_star asks for mel "within 1e-3 relative" of the reference; the reference's own test gates at max abs 0.01 (main.cpp:6223).
The GPU engine lands 1.4e-3 .. 3.6e-3 from the oracle over 80 steps. This file measures, on the reduced-depth synthetic weights and
with the SAME explicit noise for every evaluation, what that number has to be compared with:

  floor_f32     torch-f32 vs torch-f64 of the reference's graph (same fp16 rounding points): what ANY two f32 summation orders cost;
  oracle        oracle (f32, hand-written loops) vs the f64 evaluation: must sit on that floor (it is the checker);
  engine_math   the ENGINE's arithmetic evaluated exactly (f64): AttentionBlock with fp16 q/k/v/P/attention output/proj_out weight
                (north-star: "MFMA ... for the dense fp16 GEMMs in attention") vs the reference's F32 AttentionBlock — the part of the
                GPU-vs-oracle distance that is a design decision, not evaluation order.

Measured (tests/golden/parity_floor.json: small / mid weights at L = 9 .. 30, full depth at L = 20 / 32, T = 87 / 139): floor_f32 7-8e-4 at reduced
depth and 1.1e-3 at full depth, the oracle 7-9e-4, engine_math 1.0-1.4e-3 at reduced depth and 1.9-2.1e-3 at full depth, `pair` (an f32 emulation of
the engine's arithmetic vs the oracle, i.e. what a correct GPU implementation should show against the oracle) 1.0-1.4e-3 / 2.0-2.5e-3.
So north-star's 1e-3 is the distance between two CORRECT f32 evaluations of the reference's own graph over this loop: no f32 implementation can
promise to stay inside it against another one, and the engine's fp16 attention (a north-star design decision) costs about one more floor.
Round 5: the five fp16 roundings of that block were ablated one at a time (tests/torch_ref.py switches, tools/regen_parity_floor.py --ablate, table
under "ablation" in the record). Only the two that are the SAME perturbation at every step survive the loop — the proj_out WEIGHT and anything inside the latent conditioner
(evaluated once per utterance) — so the engine's default now keeps q, k, v, P and the attention output as fp16 MFMA operands, multiplies proj_out on a split-precision weight
and runs the conditioner in reference precision: emulated 1.14-1.16 x the f32-vs-f32 mean at full depth, measured on the GPU 1.09-1.21 x. The GPU gates in
tests/test_diffusion_gpu.py / tests/test_fullsize_gpu.py are conftest.loop_gate / loop_gate_mean for BOTH modes (default and option attn_f32): max(1e-3, 1.5 x oracle_vs_t32) =
1.4e-3 / 1.6e-3 / 2.0e-3 on the maximum and — round 6 — 1.30 x the torch-vs-oracle mean of the very problem a test runs on the mean (tools/regen_parity_floor.py regenerates
every column, `--problems` the torch-vs-oracle distances on the GPU tests' own inputs, `--seeds` the spread of the floor over seeds). The 2 x pair gates of rounds 1-4
(2.8e-3 .. 5.0e-3) are gone.
"""
import json
import os

import numpy as np
import pytest
import torch

import torch_ref as TR
from conftest import GOLDEN

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
FLOOR_JSON = os.path.join(GOLDEN, "parity_floor.json")


def run_loops(oracle, path, L, seed=5, steps=80):
    od = oracle.Diffusion(oracle.Model(path))
    nets = {
        "t32": TR.TorchDiffusion(path, oracle.buckets),
        "t64": TR.TorchDiffusion(path, oracle.buckets, dtype=torch.float64),
        "e64": TR.TorchDiffusion(path, oracle.buckets, dtype=torch.float64, f16_attention=True),
    }
    T = od.T_of(L)
    N = 100 * T
    rs = np.random.RandomState(seed)
    lat = rs.randn(L, 1024).astype(np.float32)
    noise = rs.randn(steps + 1, N).astype(np.float32)
    tm = oracle.default_timestep_map(steps)
    ce = od.code_embedding(lat, T)

    def loop(net):  # the driver of main.cpp:5723-6033 around a torch network; the update is the oracle's (pinned bit-exact, test_host_parity)
        x = noise[0].copy()
        for idx in range(steps):
            t = steps - 1 - idx
            te = oracle.timestep_embedding(int(tm[t]))
            xc = x.reshape(100, T)
            x = oracle.diffusion_update(tm, t, net.forward(ce, xc, te), net.forward(None, xc, te), x, noise[idx + 1], T)
        return x.reshape(100, T)

    res = {k: loop(n) for k, n in nets.items()}
    res["orc"] = od.sample(lat, steps, noise=noise.reshape(-1))

    def d(a, b):
        return float(np.abs(res[a] - res[b]).max())

    return {"T": int(T), "floor_f32": d("t32", "t64"), "oracle": d("orc", "t64"), "oracle_vs_t32": d("orc", "t32"), "engine_math": d("e64", "t64")}


def test_loop_level_parity_floor(small_models, oracle):
    r = run_loops(oracle, small_models + "/ggml-diffusion-model.bin", L=8)
    print("80-step loop, small weights, T = %(T)d: torch-f32 vs f64 %(floor_f32).2e | oracle vs f64 %(oracle).2e | oracle vs torch-f32 "
          "%(oracle_vs_t32).2e | engine arithmetic (fp16 attention, evaluated in f64) vs f64 %(engine_math).2e" % r)
    rec = json.load(open(FLOOR_JSON))
    # the floor exists and is where the committed record says it is (a factor 2.5 either way: it is a max over 100 x T chaotic values)
    assert rec["small"]["floor_f32"] / 2.5 < r["floor_f32"] < rec["small"]["floor_f32"] * 2.5
    assert 2e-4 < r["floor_f32"] < 2.5e-3
    # the oracle is an f32 evaluation like any other: on the floor, not above it
    assert r["oracle"] < 2.0 * r["floor_f32"] + 1e-4
    # the engine's fp16 attention arithmetic costs about one more floor, not an order of magnitude
    assert r["engine_math"] < 3.0 * r["floor_f32"] + 1e-4
    # two correct f32 evaluations (the oracle, torch-f32) sit about one floor apart: this is what gates the engine's reference-precision mode
    assert rec["small"]["oracle_vs_t32"] / 2.5 < r["oracle_vs_t32"] < rec["small"]["oracle_vs_t32"] * 2.5
    if os.environ.get("TTS_REGEN_FLOOR"):
        rec["small_L8"] = r
        json.dump(rec, open(FLOOR_JSON, "w"), indent=1)


def test_floor_record_is_consistent():
    """The committed floors the GPU gates are derived from (tools/regen_parity_floor.py regenerates every column). Round 6:
      gate_f32       = max(1e-3, 1.5 x the largest recorded distance between the oracle and a torch-f32 evaluation — over the class's samples, the exact problems of the
                       GPU tests and the seed distribution);
      mean gate      = conftest.MEAN_RATIO_MAX (1.30) x the torch-f32-vs-oracle MEAN of the very problem a test runs (`problems`), and for problems without a floor of
                       their own 1.30 x max(largest recorded problem mean, mu + 3 sigma of the class's seed distribution) (`gate_f32_mean`);
      seed_distribution = the same comparison under >= 5 (latents, noise) seeds at one problem size: the floor's relative sigma is 1-3 %, so a recorded single-sample floor
                       is a fair denominator for the ratio gate, and the 1.30 leaves the measured engine / floor ratios (1.04 .. 1.24) their sigma of air.
    Both arithmetic modes of the engine (and option latency_mode) are held to them. No floor is so far below north-star's 1e-3 that 1e-3 would be a promise one f32
    implementation could keep against another."""
    from conftest import MEAN_RATIO_MAX
    rec = json.load(open(FLOOR_JSON))
    for key in ("small", "mid", "full"):
        f = rec[key]
        for fld in ("floor_f32", "oracle", "engine_math", "pair"):
            assert f[fld] == max(x[fld] for x in f["samples"]), (key, fld)
            assert 4e-4 < f[fld] < 5e-3, (key, fld, f[fld])
        assert "gate" not in f  # the self-referential 2 x pair gate of rounds 1-4 is gone
        dist = f["seed_distribution"]
        assert len(dist["rows"]) >= 5 and all(r["T"] == dist["rows"][0]["T"] for r in dist["rows"])
        means = np.array([r["mean"] for r in dist["rows"]])
        assert dist["mean_mu"] == pytest.approx(means.mean()) and dist["mean_sigma"] == pytest.approx(means.std(ddof=1))
        assert dist["mean_sigma"] / dist["mean_mu"] < 0.05, (key, dist["mean_sigma"] / dist["mean_mu"])  # measured 0.9 .. 2.5 %
        both = [x["oracle_vs_t32"] for x in f["samples"]] + [p["oracle_vs_t32"] for p in f["problems"].values()] + [r["max"] for r in dist["rows"]]
        assert f["oracle_vs_t32"] == max(both) and all(2e-4 < v < 2e-3 for v in both), (key, both)
        assert f["gate_f32"] == pytest.approx(max(1e-3, 1.5 * f["oracle_vs_t32"]), rel=1e-2)
        top = max([p["oracle_vs_t32_mean"] for p in f["problems"].values()] + [dist["mean_mu_plus_3sigma"]])
        assert f["gate_f32_mean"] == pytest.approx(MEAN_RATIO_MAX * top, rel=1e-3)
        assert f["gate_f32"] < 2.1e-3 < 0.01  # far tighter than the reference's own gate (main.cpp:6223) at every depth
    t = rec["trained"]  # the trained-statistics weights (tests/test_trained_stats_gpu.py): the floor of its one loop problem
    assert "test_trained_stats_loop_80_steps" in t["problems"] and t["gate_f32"] == pytest.approx(max(1e-3, 1.5 * t["oracle_vs_t32"]), rel=1e-2)
    # every loop problem of the GPU tests that can be rebuilt without the engine is on record (tools/regen_parity_floor.py --problems)
    assert {"test_sampling_loop_80_steps[small]", "test_sampling_loop_matches_oracle[cand 0]", "test_sampling_loop_matches_oracle[cand 1]",
            "test_sampling_loop_200_steps_config5"} <= set(rec["small"]["problems"])
    assert "test_sampling_loop_80_steps[mid]" in rec["mid"]["problems"]
    assert {"test_full_size_80_steps_at_bench_length", "test_config5_shape_200_steps"} <= set(rec["full"]["problems"])


def test_ablation_record_says_what_the_default_mode_relies_on():
    """The round-5 ablation table (full depth, L = 20 and 32; reduced depth): of the five fp16 roundings of the rounds 1-4 AttentionBlock only the proj_out WEIGHT — and the
    same roundings inside the once-per-utterance latent conditioner (+lc) — move the 80-step mean; q, k, v, P and the attention output together stay within 1.25 x the
    f32-vs-f32 mean. That is the arithmetic the engine's default mode keeps in fp16."""
    rec = json.load(open(FLOOR_JSON))
    rows = [(k, key, r) for k in ("mid", "full") for key, r in rec[k].get("ablation", {}).items()]
    assert len(rows) >= 3 and sum(1 for k, _, _ in rows if k == "full") >= 2
    for kind, key, r in rows:
        none = r["none"]["mean"]
        assert r["w"]["mean"] > 1.4 * none, (kind, key)                    # the weight rounding alone: 1.5 .. 2.1 x
        assert r["qk,v,p,o"]["mean"] < 1.25 * none, (kind, key)            # everything else together: 1.13 .. 1.16 x
        for one in ("qk", "v", "p", "o"):
            assert r[one]["mean"] < 1.15 * none, (kind, key, one)
        assert r["qk,v,p,o,w"]["mean"] > 1.4 * none
        if "qk,v,p,o+lc" in r:                                             # an fp16 conditioner undoes it: 1.7 .. 1.9 x
            assert r["qk,v,p,o+lc"]["mean"] > 1.4 * none, (kind, key)


def test_ablation_switches_are_independent(small_models, oracle):
    """tests/torch_ref.py: f16_attention accepts any subset of the five roundings; the empty set IS the reference's F32 block (same numbers as the plain branch), the full set
    is what f16_attention=True always meant, and single switches change the output by the size of one fp16 rounding."""
    path = small_models + "/ggml-diffusion-model.bin"
    od = oracle.Diffusion(oracle.Model(path))
    L = 8
    T = od.T_of(L)
    lat = np.random.RandomState(1).randn(L, 1024).astype(np.float32)
    x_t = np.random.RandomState(2).randn(100, T).astype(np.float32)
    ce, te = od.code_embedding(lat, T), oracle.timestep_embedding(557)
    ref = TR.TorchDiffusion(path, oracle.buckets).forward(ce, x_t, te)
    assert np.array_equal(TR.TorchDiffusion(path, oracle.buckets, f16_attention="").forward(ce, x_t, te), ref)
    full = TR.TorchDiffusion(path, oracle.buckets, f16_attention=True).forward(ce, x_t, te)
    assert np.array_equal(TR.TorchDiffusion(path, oracle.buckets, f16_attention="qk,v,p,o,w").forward(ce, x_t, te), full)
    sc = np.abs(ref).max()
    for one in ("qk", "v", "p", "o", "w"):
        y = TR.TorchDiffusion(path, oracle.buckets, f16_attention=one).forward(ce, x_t, te)
        d = np.abs(y - ref).max() / sc
        assert 0 < d < 5e-3, (one, d)
    with pytest.raises(AssertionError):
        TR.TorchDiffusion(path, oracle.buckets, f16_attention="qk,z")
