"""Option latency_mode (round 6): small diffusion batches — one utterance, at most 2 048 packed rows — take the GroupNorm statistics
from the epilogue of the GEMM that produced the tensor (fixed-point sums, gemm_f16.h: GEMM_OUT_*_STATS) and normalise with gn_apply_kernel instead of the reducing
GroupNorm kernels (one 512-thread workgroup per (sequence, group): 64 workgroups, 8.7 us per launch, 43 launches per sampling step). The arithmetic differs from the
batch path only in how the variance is formed (exact sums, E[x^2] - E[x]^2 in f64, against two-pass f32), so it is held to the SAME oracle gates as the default
mode (main.cpp:3191-3499 are the GroupNorm call sites of the reference's graph), must stay within rounding noise of the batch path, and must be reproducible
run to run (integer accumulation: no dependence on the order workgroups finish in)."""
import numpy as np
import pytest

from conftest import ATTN_MODES, check_loop

pytestmark = pytest.mark.gpu


def _latents(L, seed):
    return np.random.RandomState(seed).randn(L, 1024).astype(np.float32)


@pytest.fixture()
def lat_engine(engine):
    yield engine
    engine.set_option("latency_mode", 0)
    engine.set_option("attn_f32", 0)


@pytest.mark.parametrize("models,L,timestep", [("small", 12, 3999), ("mid", 43, 2025), ("small", 1, 0), ("small", 100, 1000), ("small", 200, 500), ("small", 250, 77)])
@pytest.mark.parametrize("cond_free", [False, True])
def test_forward_latency_mode(lat_engine, oracle, small_models, mid_models, models, L, timestep, cond_free):
    """One network evaluation: latency mode vs the oracle (the default mode's gate) and vs the batch path (rounding noise). L = 200 is the benchmark's utterance
    (T = 870: 1 792 packed rows with both branches in diffusion(); one branch here), L = 250 the 1024-thread GroupNorm shape of the batch path."""
    engine = lat_engine
    d = small_models if models == "small" else mid_models
    engine.load(diffusion=d + "/ggml-diffusion-model.bin")
    od = oracle.Diffusion(oracle.Model(d + "/ggml-diffusion-model.bin"))
    lat = _latents(L, L)
    T = engine.frames(L)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    want = od.forward(None if cond_free else od.code_embedding(lat, T), x_t, timestep)
    scale = np.abs(want).max()
    for mode, what in ATTN_MODES:
        engine.set_option("attn_f32", mode)
        engine.set_option("latency_mode", 0)
        base = engine.diffusion_forward(lat, x_t, timestep, cond_free)
        engine.set_option("latency_mode", 1)
        got = engine.diffusion_forward(lat, x_t, timestep, cond_free)
        again = engine.diffusion_forward(lat, x_t, timestep, cond_free)
        e, d_batch = float(np.abs(got - want).max() / scale), float(np.abs(got - base).max() / scale)
        print("latency forward %s L=%d t=%d cond_free=%s [%s]: vs oracle %.2e, vs batch path %.2e" % (models, L, timestep, cond_free, what, e, d_batch))
        assert e < (6e-4 if mode else 1e-3), (mode, e)
        # the two paths round the same fp16 operands from statistics that differ in the last bits: operand elements flip by one fp16 ulp and a single forward is
        # chaotic at that level (two correct f32 evaluations keep 2-5e-4 from each other: tests/test_oracle_vs_torch.py)
        assert d_batch < 6e-4, (mode, d_batch)
        assert (got == again).all(), "latency mode must be reproducible run to run"


@pytest.mark.parametrize("kind", ["small", "mid"])
def test_sampling_loop_80_steps_latency_mode(lat_engine, oracle, small_models, mid_models, kind, oracle_sample):
    """The 80-step loop of tests/test_fullsize_gpu.py::test_sampling_loop_80_steps (same latents, same explicit noise) in latency mode: both arithmetic modes inside the
    ONE pair of gates of the default path."""
    engine = lat_engine
    d = small_models if kind == "small" else mid_models
    engine.load(diffusion=d + "/ggml-diffusion-model.bin")
    lat = _latents(12, 12)
    noise = np.random.RandomState(5).randn(81, 100 * engine.frames(12)).astype(np.float32)
    want = oracle_sample(d, lat, noise, 80)
    for mode, what in ATTN_MODES:
        engine.set_option("attn_f32", mode)
        engine.set_option("latency_mode", 0)
        base = engine.diffusion([lat], n_steps=80, noise=[noise])[0]
        engine.set_option("latency_mode", 1)
        mel = engine.diffusion([lat], n_steps=80, noise=[noise])[0]
        again = engine.diffusion([lat], n_steps=80, noise=[noise])[0]
        print("80-step loop, latency mode, %s [%s]: %s; batch path on the same problem: max %.2e mean %.2e" %
              (kind, what, check_loop(np.abs(mel - want), kind, mode, problem="test_sampling_loop_80_steps[%s]" % kind), np.abs(base - want).max(), np.abs(base - want).mean()))
        assert (mel == again).all(), "latency mode must be reproducible run to run"


def test_ragged_pair_latency_mode(lat_engine, oracle, small_models, oracle_sample):
    """Two candidates of different length in one latency-mode batch (4 sequences; chunks of 8 rows never straddle two sequences) against the oracle, and each
    candidate against itself run alone in latency mode: the statistics are exact sums of per-chunk partials, so a candidate's result does not depend on its
    neighbours here either."""
    engine = lat_engine
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    import oracle_jobs
    lats, noise = oracle_jobs.small_pair_inputs(engine.frames)
    engine.set_option("latency_mode", 1)
    mels = engine.diffusion(lats, n_steps=80, noise=noise)
    for c, l in enumerate(lats):
        want = oracle_sample(small_models, l, noise[c], 80, bg="small_pair%d" % c)
        print("latency mode, ragged pair cand %d: %s" % (c, check_loop(np.abs(mels[c] - want), "small", 0, "cand %d" % c, problem="test_sampling_loop_matches_oracle[cand %d]" % c)))
        alone = engine.diffusion([l], n_steps=80, noise=[noise[c]])[0]
        assert (alone == mels[c]).all(), "candidate %d differs between the pair and the single run" % c


@pytest.mark.parametrize("lens,steps", [((12,), 80), ((20, 9), 80), ((16, 16), 37), ((9,), 200)])
def test_hoisted_integrator_is_bit_identical(lat_engine, small_models, mid_models, pkg, lens, steps):
    """Small batches evaluate the conditioning_timestep_integrator layers — which see only (code embedding, timestep), never x_t (main.cpp:3322-3499) — for ALL sampling
    steps before the loop, many timesteps per batch with per-sequence scale / shift (diffusion.hip: precompute_integrator; option hoist_integrator, default 1). Same
    arithmetic per sequence: the mel equals the one computed with the layers inside every step, bit for bit — one utterance, a ragged pair, an equal-length pair (the
    shared unconditioned sequence), 200 steps; explicit noise and device noise. (Under option latency_mode the hoisted layers run on the batch path's GroupNorm
    kernels while the in-step ones take their statistics from the GEMM epilogues: there the two settings agree to the loop's chaos level, like latency mode and the
    default do, and only reproducibility is asserted.)"""
    engine = lat_engine
    for d in (small_models, mid_models):
        engine.load(diffusion=d + "/ggml-diffusion-model.bin")
        lats = [_latents(L, 40 + i) for i, L in enumerate(lens)]
        rs = np.random.RandomState(11)
        noise = [rs.randn(steps + 1, 100 * engine.frames(L)).astype(np.float32) for L in lens]
        for lat_mode in (0, 1):
            engine.set_option("latency_mode", lat_mode)
            out = {}
            for hoist in (0, 1):
                engine.set_option("hoist_integrator", hoist)
                engine.seed(5)
                out[hoist] = (engine.diffusion(lats, n_steps=steps, noise=noise), engine.diffusion(lats, n_steps=steps, noise_mode=pkg.NOISE_DEVICE))
            engine.set_option("hoist_integrator", 1)
            for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
                assert np.isfinite(a).all() and np.isfinite(b).all()
                if lat_mode == 0:
                    assert (a == b).all(), (lens, steps, float(np.abs(a - b).max()))
                else:
                    assert np.abs(a - b).max() < 5e-3, (lens, steps, float(np.abs(a - b).max()))


@pytest.mark.parametrize("models,L", [("small", 1), ("small", 12), ("mid", 43), ("small", 100), ("small", 200), ("small", 250)])
def test_attention_64_query_workgroups_are_bit_identical(lat_engine, small_models, mid_models, models, L):
    """Option attn_q64 (the diffusion attention kernel with 64 instead of 128 queries per workgroup: twice as many, half as long workgroups for grids that leave the CUs
    with at most one; measured without gain and off by default, profiles/r6_small_batch.txt). Every query's arithmetic is the same in the same key order (main.cpp:3232-3275): the forward
    with the option forced on equals the forward with it off bit for bit, conditioned and conditioning-free, for lengths on and off the 64 / 128 boundaries."""
    engine = lat_engine
    d = small_models if models == "small" else mid_models
    engine.load(diffusion=d + "/ggml-diffusion-model.bin")
    lat = _latents(L, L)
    x_t = np.random.RandomState(7).randn(100, engine.frames(L)).astype(np.float32)
    try:
        for cond_free in (False, True):
            out = {}
            for q64 in (0, 1):
                engine.set_option("attn_q64", q64)
                out[q64] = engine.diffusion_forward(lat, x_t, 1234, cond_free)
            assert np.isfinite(out[0]).all() and (out[0] == out[1]).all(), (L, cond_free, float(np.abs(out[0] - out[1]).max()))
    finally:
        engine.set_option("attn_q64", 0)
