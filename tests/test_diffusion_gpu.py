"""GPU parity of the diffusion stage against the oracle, through the C ABI."""
import numpy as np
import pytest

from conftest import ATTN_MODES, check_loop, loop_gate

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _latents(L, seed):
    return np.random.RandomState(seed).randn(L, 1024).astype(np.float32)


# L=100 (T=435): attention tiles that are far from the diagonal and not the tail tile (constant-bias path);
# L=250 (T=1088): the 1024-thread GroupNorm path (T > 896) and several query blocks per sequence;
# L=500 (T=2176): the largest sequence the AR stage can produce (500 latent rows)
@pytest.mark.parametrize("models,L,timestep", [("small", 12, 3999), ("small", 43, 51), ("mid", 43, 2025), ("small", 1, 0),
                                               ("small", 100, 1000), ("small", 250, 500), ("small", 500, 100)])
@pytest.mark.parametrize("cond_free", [False, True])
def test_forward_matches_oracle(engine, oracle, small_models, mid_models, models, L, timestep, cond_free):
    """One diffusion_graph evaluation (eps | variance logits), conditioned and conditioning-free."""
    d = small_models if models == "small" else mid_models
    engine.load(diffusion=d + "/ggml-diffusion-model.bin")
    od = oracle.Diffusion(oracle.Model(d + "/ggml-diffusion-model.bin"))
    lat = _latents(L, L)
    T = engine.frames(L)
    assert T == od.T_of(L)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    ce = None if cond_free else od.code_embedding(lat, T)
    want = od.forward(ce, x_t, timestep)
    try:
        for mode, what in ATTN_MODES:
            engine.set_option("attn_f32", mode)
            got = engine.diffusion_forward(lat, x_t, timestep, cond_free)
            assert got.shape == want.shape == (200, T)
            # north star: 1e-3 relative with fp16 MFMA operands in the attention core (f32 accumulate); the reference-precision mode sits on the
            # single-forward chaos floor of the fp16-rounded convolutions (two f32 evaluations: 2-5e-4, tests/test_oracle_vs_torch.py)
            e = rel_err(got, want)
            print("forward rel err %s L=%d t=%d cond_free=%s [%s]: %.2e" % (models, L, timestep, cond_free, what, e))
            assert e < (6e-4 if mode else 1e-3), (mode, e)
    finally:
        engine.set_option("attn_f32", 0)


@pytest.mark.parametrize("gn_eps,lut", [(1e-5, 0), (1e-6, 1)])
def test_numerics_switches(engine, oracle, small_models, gn_eps, lut):
    """The two [ggml-unverified] switches (GroupNorm epsilon, fp16 SiLU table) against the oracle's implementation of the
    same switches, conditioned forward. The flash attention keeps the hardware exp in LUT mode (documented): within the
    same 1e-3 gate."""
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    engine.set_option("gn_eps", gn_eps)
    engine.set_option("ggml_lut", lut)
    oracle.set_flags(gn_eps=gn_eps, lut=lut)
    try:
        od = oracle.Diffusion(oracle.Model(small_models + "/ggml-diffusion-model.bin"))
        L = 30
        lat = _latents(L, 5)
        T = engine.frames(L)
        x_t = np.random.RandomState(9).randn(100, T).astype(np.float32)
        got = engine.diffusion_forward(lat, x_t, 1500, False)
        want = od.forward(od.code_embedding(lat, T), x_t, 1500)
        e = rel_err(got, want)
        print("switches gn_eps=%g lut=%d: rel err %.2e" % (gn_eps, lut, e))
        assert e < 1e-3, e
    finally:
        oracle.set_flags()
        engine.set_option("gn_eps", 1e-6)  # the engine fixture is shared by the whole session
        engine.set_option("ggml_lut", 0)


def test_sampling_loop_matches_oracle(engine, oracle, small_models, oracle_sample):
    """diffusion(): the full 80-step schedule, 2 candidates of different length in one batch (ragged layout), explicit noise —
    gates: conftest.loop_gate / loop_gate_mean (the distance two correct f32 evaluations keep, both arithmetic modes; the reference's own gate is max abs
    0.01, main.cpp:6223). (Coarse schedules are a worse test, not a faster one: with 4-6
    respaced steps the first update multiplies the eps error by up to 153 before the +-1 clamp and single bins land 5e-2 apart on two
    correct implementations; over 80 steps the same two implementations agree to ~2e-3.)"""
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    import oracle_jobs
    n_steps = 80
    lats, noise = oracle_jobs.small_pair_inputs(engine.frames)
    wants = [oracle_sample(small_models, l, noise[c], n_steps, bg="small_pair%d" % c) for c, l in enumerate(lats)]
    try:
        for mode, what in ATTN_MODES:
            engine.set_option("attn_f32", mode)
            mels = engine.diffusion(lats, n_steps=n_steps, noise=noise)
            for c, want in enumerate(wants):
                assert mels[c].shape == want.shape
                err = np.abs(mels[c] - want)
                print("80-step sampling loop cand %d [%s]: %s" % (c, what, check_loop(err, "small", mode, "cand %d" % c, problem="test_sampling_loop_matches_oracle[cand %d]" % c)))
    finally:
        engine.set_option("attn_f32", 0)


def test_sampling_loop_200_steps_config5(engine, oracle, small_models, oracle_sample):
    """configs[4] runs 200 diffusion steps (timestep_map = round(i * 3999 / 199), the generalisation the reference hard-codes away for 80): the
    device loop over that schedule against the oracle's, same explicit noise (201 vectors), gate conftest.loop_gate."""
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    import oracle_jobs
    lat, noise = oracle_jobs.small_200_inputs(engine.frames)
    T = engine.frames(9)
    want = oracle_sample(small_models, lat, noise, 200, bg="small_200")
    try:
        for mode, what in ATTN_MODES:
            engine.set_option("attn_f32", mode)
            mel = engine.diffusion([lat], n_steps=200, noise=[noise])[0]
            err = np.abs(mel - want)
            assert mel.shape == want.shape == (100, T) and np.isfinite(mel).all() and np.abs(mel).max() <= 1.0 + 1e-6
            print("200-step sampling loop (T=%d) [%s]: %s" % (T, what, check_loop(err, "small", mode, problem="test_sampling_loop_200_steps_config5")))
    finally:
        engine.set_option("attn_f32", 0)


def test_what_the_default_arithmetic_relies_on(engine, oracle, mid_models, oracle_sample):
    """The round-5 finding, on the engine itself (mid depth, T = 52, 80 steps, same explicit noise): the two roundings that are the SAME perturbation at every step — the
    proj_out weight (option attn_proj_f16) and anything inside the once-per-utterance latent conditioner (option lc_attn_f32) — each move the mean distance from the oracle,
    the default (neither) sits with the reference-precision mode. CPU emulation of the same ladder: tests/golden/parity_floor.json "ablation"."""
    engine.load(diffusion=mid_models + "/ggml-diffusion-model.bin")
    L = 12
    lat = np.random.RandomState(L).randn(L, 1024).astype(np.float32)
    noise = np.random.RandomState(5).randn(81, 100 * engine.frames(L)).astype(np.float32)
    want = oracle_sample(mid_models, lat, noise, 80)  # the problem of tests/test_fullsize_gpu.py::test_sampling_loop_80_steps[mid]
    modes = {"default": (0, 0, 1), "fp16 conditioner": (0, 0, 0), "fp16 proj_out weight": (0, 1, 1), "all fp16 (rounds 1-4)": (0, 1, 0), "attn_f32": (1, 0, 1)}
    mean = {}
    try:
        for name, (f32, pw16, lc) in modes.items():
            for k, v in (("attn_f32", f32), ("attn_proj_f16", pw16), ("lc_attn_f32", lc)):
                engine.set_option(k, v)
            mean[name] = float(np.abs(engine.diffusion([lat], n_steps=80, noise=[noise])[0] - want).mean())
    finally:
        for k, v in (("attn_f32", 0), ("attn_proj_f16", 0), ("lc_attn_f32", 1)):
            engine.set_option(k, v)
    print("mean abs distance from the oracle, mid depth, 80 steps: " + ", ".join("%s %.2e" % kv for kv in mean.items()))
    assert mean["default"] < 1.2 * mean["attn_f32"]                       # measured 6.1e-5 vs 5.6e-5
    assert mean["fp16 conditioner"] > 1.25 * mean["default"]             # 9.1e-5
    assert mean["fp16 proj_out weight"] > 1.2 * mean["default"]          # 8.2e-5
    assert mean["all fp16 (rounds 1-4)"] > 1.5 * mean["default"]         # 1.08e-4


def test_reference_noise_stream(engine, oracle, small_models):
    """noise_mode REFERENCE consumes the ctx RNG exactly like the reference: x_T then one vector per step."""
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    od = oracle.Diffusion(oracle.Model(small_models + "/ggml-diffusion-model.bin"))
    lat = _latents(10, 4)
    engine.seed(1234)
    mel = engine.diffusion([lat], n_steps=80)[0]
    rng = oracle.Rng(1234)
    want = od.sample(lat, n_steps=80, rng=rng)
    assert np.abs(mel - want).max() <= loop_gate("small")
    assert engine.rng_uniform() == rng.uniform()


def test_reference_noise_pipeline_changes_nothing(engine, small_models):
    """Round 6: with ONE candidate in the reference's draw order the host draws block k + 1 of the noise while the device runs the earlier steps (option noise_pipeline,
    default 1; diffusion.hip: diff_sample). Same draws in the same order (main.cpp:5638, 6020-6021): the mel and the RNG state after the call equal the ones of the
    draw-everything-first path bit for bit, for 80 and 200 steps; two candidates (candidate-major draw order) take the old path under either setting."""
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    try:
        for lats, steps in (([_latents(10, 4)], 80), ([_latents(23, 6)], 200), ([_latents(9, 1), _latents(14, 2)], 80)):
            out = {}
            for pipe in (0, 1):
                engine.set_option("noise_pipeline", pipe)
                engine.seed(4321)
                mels = engine.diffusion(lats, n_steps=steps)
                out[pipe] = (mels, engine.rng_uniform())
            assert out[0][1] == out[1][1], "RNG state after the call differs"
            for a, b in zip(out[0][0], out[1][0]):
                assert np.isfinite(a).all() and (a == b).all()
    finally:
        engine.set_option("noise_pipeline", 1)


def test_device_noise_is_deterministic_and_normal(engine, small_models, pkg):
    engine.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    lat = _latents(30, 5)
    engine.seed(7)
    a = engine.diffusion([lat, lat], n_steps=3, noise_mode=pkg.NOISE_DEVICE)
    engine.seed(7)
    b = engine.diffusion([lat, lat], n_steps=3, noise_mode=pkg.NOISE_DEVICE)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    assert not (a[0] == a[1]).all()  # independent per-candidate streams
    assert np.isfinite(a[0]).all()


def test_time_mlp_guard_is_silent_in_a_single_process(engine):
    """The timestep MLP is evaluated twice per call and repeated on disagreement (diffusion.hip: precompute_time; DESIGN.md section 6: a tripwire kept from the
    round-4 hunt for the packed-FMA fault that a second engine process on the same GPU exposed). Every call of this module ran alone: not one repetition."""
    assert engine.time_mlp_retries() == 0
