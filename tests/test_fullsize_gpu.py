"""GPU parity at the size the benchmark runs: full-depth synthetic weights — 30 GPT-2 layers, 4 latent-
conditioner + 3 integrator + 10 main + 3 tail diffusion blocks, UnivNet — the very files bench.py times (conftest.full_models),
against the oracle through the C ABI. Error compounds with depth (30 fp16-QKV rounding points, 13 AttentionBlocks with fp16 P.V),
so the reduced-depth tests in test_ar_gpu.py / test_diffusion_gpu.py do not cover this.

Gates: AR logits 1e-4 relative (f32 on both sides), latents and mel/audio of ONE evaluation 1e-3 relative (north star), the 80-/200-step
sampling loop conftest.loop_gate / loop_gate_mean — since round 5 ONE pair of gates for the default arithmetic and for option attn_f32: the distance two correct
f32 evaluations of the reference's graph keep from each other on the same problems (1.9e-3 max, 7.3e-5 mean at full depth; the reference's own gate: 0.01, main.cpp:6223).
configs[1] = test_config1_end_to_end, configs[2] = test_config2_batch16 (+ the two full-shape tests), configs[3] = test_config3_shape_64_candidates,
configs[4] = test_config5_shape_200_steps."""
import os

import numpy as np
import pytest

from conftest import ATTN_MODES, DEFAULT_TOKENS, check_loop, loop_gate, loop_gate_mean

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def bench_prompt():
    return np.array([255] + [3 + (7 * j) % 250 for j in range(64)] + [0], np.int32)  # the synthetic 64-token prompt


@pytest.fixture(scope="module")
def full_engine(pkg, full_models):
    eng = pkg.Engine(0)
    eng.load(full_models)
    yield eng
    eng.close()


def test_ar_full_depth(full_engine, oracle, full_models, voice):
    """30 layers: prompt pass (P = 68, the bench prompt), 8 decode steps at B = 16 (one full candidate tile, as in bench.py), a 40-row
    latent pass."""
    eng = full_engine
    ar = oracle.AR(oracle.Model(full_models + "/ggml-model.bin"))
    assert eng.ar_layers == ar.n_layers == 30
    toks, B = bench_prompt(), 16
    eng.ar_begin(toks, voice, B, 16)
    ar.start(toks, voice, B, len(toks) + 2 + 17)
    errs = [rel_err(eng.ar_prefill(), ar.prefill())]
    rs = np.random.RandomState(1)
    for i in range(8):
        prev = rs.randint(0, 8192, B).astype(np.int32)
        errs.append(rel_err(eng.ar_step(prev, i), ar.step(prev, i)))
    print("full-depth AR logits rel err: prefill %.1e, steps %s" % (errs[0], " ".join("%.1e" % e for e in errs[1:])))
    assert max(errs) < 1e-4, errs
    codes = rs.randint(0, 8192, (2, 502)).astype(np.int32)
    codes[:, 0] = 8192
    lg, lo = eng.ar_latents(codes, 40), ar.latents(codes, 40)
    e = rel_err(lg, lo)
    print("full-depth latents (40 rows) rel err %.1e" % e)
    assert lg.shape == lo.shape == (2, 40, 1024) and e < 1e-3


@pytest.mark.parametrize("L", [43, 200])  # T = 187 (the reference fixture's size) and T = 870 (the benchmark's)
def test_diffusion_forward_full_depth(full_engine, oracle, full_models, L):
    eng = full_engine
    od = oracle.Diffusion(oracle.Model(full_models + "/ggml-diffusion-model.bin"))
    lat = np.random.RandomState(L).randn(L, 1024).astype(np.float32)
    T = eng.frames(L)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    ce = od.code_embedding(lat, T)
    try:
        for cond_free, timestep in ((False, 3999), (True, 3999), (False, 557)):
            want = od.forward(None if cond_free else ce, x_t, timestep)
            for mode, what in ATTN_MODES:
                eng.set_option("attn_f32", mode)
                got = eng.diffusion_forward(lat, x_t, timestep, cond_free)
                e = rel_err(got, want)
                print("full-depth diffusion forward T=%d t=%d cond_free=%s [%s]: rel err %.2e" % (T, timestep, cond_free, what, e))
                # default mode: north star's 1e-3; reference precision: the single-forward floor two f32 evaluations of this graph keep
                # (tests/test_oracle_vs_torch.py: torch-f32 and the oracle are each 2-5e-4 from an f64 evaluation at full depth)
                assert got.shape == want.shape == (200, T) and e < (6e-4 if mode else 1e-3), (mode, e)
    finally:
        eng.set_option("attn_f32", 0)


@pytest.mark.parametrize("T", [187, 870])  # the reference fixture's length and the benchmark's
def test_vocoder_full(full_engine, oracle, full_models, T):
    eng = full_engine
    ov = oracle.Vocoder(oracle.Model(full_models + "/ggml-vocoder-model.bin"))
    rs = np.random.RandomState(3)
    mel = np.clip(rs.randn(100, T) * 0.5, -1, 1).astype(np.float32)
    nz = rs.randn(64, T + 10).astype(np.float32)
    au, ao = eng.vocoder([mel], noise=[nz])[0], ov.run(mel, noise=nz)
    e = rel_err(au, ao)
    print("vocoder T=%d rel err %.2e" % (T, e))
    assert au.shape == ao.shape and e < 1e-3


@pytest.mark.parametrize("models,L", [("small", 12), ("mid", 12)])
def test_sampling_loop_80_steps(engine, oracle, small_models, mid_models, models, L, oracle_sample):
    """tts_diffusion over the full 80-step schedule against the oracle's diffusion() with the same explicit noise; gate conftest.loop_gate
    (the reference's own: max abs 0.01 on the mel, main.cpp:6223); mean reported."""
    d = small_models if models == "small" else mid_models
    engine.load(diffusion=d + "/ggml-diffusion-model.bin")
    lat = np.random.RandomState(L).randn(L, 1024).astype(np.float32)
    T = engine.frames(L)
    noise = np.random.RandomState(5).randn(81, 100 * T).astype(np.float32)
    want = oracle_sample(d, lat, noise, 80)
    try:
        for mode, what in ATTN_MODES:
            engine.set_option("attn_f32", mode)
            mel = engine.diffusion([lat], n_steps=80, noise=[noise])[0]
            err = np.abs(mel - want)
            assert np.abs(want).max() <= 1.5
            print("80-step loop (%s weights, T=%d) [%s]: %s" % (models, T, what, check_loop(err, models, mode, problem="test_sampling_loop_80_steps[%s]" % models)))
    finally:
        engine.set_option("attn_f32", 0)


def test_config1_end_to_end(full_engine, oracle, full_models, voice, oracle_bg):
    """configs[1]: the default message, mol.bin voice, --seed 0, ONE candidate, 80 diffusion steps, full-size weights, every stage
    against the oracle with the RNG stream in lock-step (AR uniforms -> x_T -> 80 noise vectors -> vocoder noise, the reference's
    order). The stop token is masked (random weights do not produce one): 40 codes -> L = 48, T = 208."""
    eng = full_engine
    toks, S, seed = DEFAULT_TOKENS, 40, 0
    eng.seed(seed)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, 1, S, mask_stop=True)
    if oracle_bg:  # the oracle's whole chain was computed beside the earlier tests of the session (tests/oracle_jobs.py: config1)
        o = oracle_bg["config1"].result(timeout=1800)
    else:
        import oracle_jobs
        o = oracle_jobs.config1(full_models, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "mol.bin"), [int(t) for t in toks], S, seed)
    rc, codes_o, steps_o = o["rc"], o["codes_o"], o["steps_o"]
    assert rc == 0 and steps == steps_o == S
    ids_identical = bool((codes == codes_o).all())
    if not ids_identical:
        # A split must be explained by sub-tolerance logits at the first divergent step (the reference's fp16 rounding of QKV turns 1e-7
        # round-off into occasional 5e-4 jumps that can flip a multinomial draw) — and then the later stages are STILL checked, teacher-forced:
        # the engine continues on the oracle's codes (both sides have consumed 2 uniforms per step, so the RNG stream is in lock-step).
        ar = oracle.AR(oracle.Model(full_models + "/ggml-model.bin"))
        j = int(np.argwhere(codes[0] != codes_o[0])[0][0])
        ar.start(toks, voice, 1, len(toks) + 2 + S + 1)
        eng.ar_begin(toks, voice, 1, S)
        lo, lg = ar.prefill(), eng.ar_prefill()
        for i in range(j - 1):
            lo, lg = ar.step(codes_o[:, 1 + i], i), eng.ar_step(codes_o[:, 1 + i], i)
        assert rel_err(lg, lo) < 1e-4
        print("configs[1]: AR trajectories split at step %d with logits within %.1e; later stages teacher-forced on the oracle's codes" % (j - 1, rel_err(lg, lo)))
        rows = np.array([oracle.trimmed_rows(codes_o[0])], np.int32)
        lats = [eng.ar_latents(codes_o, int(rows[0]) + 1)[0, :int(rows[0])]]
        del ar
    L = int(rows[0])
    assert L == o["L"]
    lat_o = o["lat_o"]
    e_lat = rel_err(lats[0], lat_o)
    # diffusion: reference noise order from the shared stream; the oracle runs on ITS latents (end-to-end comparison). Both arithmetic
    # modes of the AttentionBlock start from the same RNG state; the reference-precision run is the one carried on to the vocoder.
    state = os.path.join(os.environ.get("TMPDIR", "/tmp"), "tts_cfg1_rng_state.txt")
    eng.rng_save_state(state)
    mel_fast = eng.diffusion([lats[0]], n_steps=80)[0]
    eng.rng_load_state(state)
    try:
        eng.set_option("attn_f32", 1)
        mel = eng.diffusion([lats[0]], n_steps=80)[0]
    finally:
        eng.set_option("attn_f32", 0)
    mel_o = o["mel_o"]
    dm, dm_fast = np.abs(mel - mel_o), np.abs(mel_fast - mel_o)
    au = eng.vocoder([mel])[0]
    au_o = o["au_o"]
    da = np.abs(au - au_o)
    assert eng.rng_uniform() == o["u_final"]  # every stage consumed the stream exactly like the reference
    # vocoder gate proper: same mel, same explicit noise on both sides (the end-to-end audio difference above also carries the
    # mel difference through a network that amplifies it; it is reported, the reference gates each stage on its own fixture)
    nz = np.random.RandomState(4).randn(64, mel_o.shape[1] + 10).astype(np.float32)
    e_voc = rel_err(eng.vocoder([mel_o], noise=[nz])[0], o["voc_on_mel_o"])
    print("configs[1] end to end: ids %s (%d codes), latents rel %.1e, mel max abs %.2e mean %.2e in reference precision / %.2e mean %.2e in the "
          "default mode (gates %.2e / %.2e for both), vocoder on the oracle's mel rel %.2e; end-to-end audio max abs %.2e of range %.2f"
          % ("identical" if ids_identical else "teacher-forced", S, e_lat, dm.max(), dm.mean(), dm_fast.max(), dm_fast.mean(),
             loop_gate("full"), loop_gate_mean("full"), e_voc, da.max(), np.abs(au_o).max()))
    assert e_lat < 1e-3
    check_loop(dm, "full", 1, "configs[1]"); check_loop(dm_fast, "full", 0, "configs[1]")  # (the reference's gate on target_mel: 0.01, main.cpp:6223)
    assert e_voc < 1e-3


def test_config2_batch16(full_engine, oracle, full_models, voice, pkg):
    """configs[2] at the benchmark's own shapes: 16 candidates of the 64-token prompt, batched through all three stages exactly as
    bench.py does (device noise), with full-size weights. Checked against the oracle where the oracle can follow: AR ids of every
    candidate (teacher-forced through all 24 decode steps at B = 16), the latents of candidates 0 and 15, the batched 80-step sampling
    loop (32 sequences: every candidate's conditioned and unconditioned copy) for the last candidate with the reference's 0.01
    gate, and the vocoder on the batch's own mel for candidates 0 and 15."""
    eng = full_engine
    toks, B, S = bench_prompt(), 16, 24
    eng.seed(77)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True)
    assert steps == S and (rows == rows[0]).all()
    ar = oracle.AR(oracle.Model(full_models + "/ggml-model.bin"))
    # teacher-forced: the oracle replays the device's ids; logits of the last step within 1e-4 for all 16 candidates
    ar.start(toks, voice, B, len(toks) + 2 + S + 1)
    eng.ar_begin(toks, voice, B, S)
    lo, lg = ar.prefill(), eng.ar_prefill()
    worst = rel_err(lg, lo)
    for i in range(S - 1):
        lo, lg = ar.step(codes[:, 1 + i], i), eng.ar_step(codes[:, 1 + i], i)
        worst = max(worst, rel_err(lg, lo))
    print("configs[2] AR B=16: worst logits rel err over %d steps %.1e" % (S, worst))
    assert worst < 1e-4
    L = int(rows[0])
    lat_o = ar.latents(codes[[0, 15]], L + 1)
    for k, c in enumerate((0, 15)):
        assert rel_err(lats[c], lat_o[k, :L]) < 1e-3
    del ar
    # diffusion + vocoder as one batch of 16 (32 sequences with the unconditioned copies), device noise as bench.py uses it
    mels = eng.diffusion(lats, n_steps=4, noise_mode=pkg.NOISE_DEVICE)
    assert all(np.isfinite(m).all() and np.abs(m).max() <= 1.5 for m in mels)
    # the same batch over the full 80-step schedule with explicit noise, against the oracle for the first and the last candidate:
    # conftest.loop_gate (the reference's own gate: max abs 0.01, main.cpp:6223)
    T = eng.frames(L)
    rs = np.random.RandomState(9)
    noise = [rs.randn(81, 100 * T).astype(np.float32) for _ in range(B)]
    od = oracle.Diffusion(oracle.Model(full_models + "/ggml-diffusion-model.bin"))
    c = 15  # the last candidate of the batch (one full-depth oracle loop costs ~50 s; position in the batch cannot matter: see the invariance test below)
    want = od.sample(lats[c], n_steps=80, noise=noise[c])
    del od
    try:
        for mode, what in reversed(ATTN_MODES):  # the default-mode batch last: its mels go on to the vocoder, as in bench.py
            eng.set_option("attn_f32", mode)
            mels = eng.diffusion(lats, n_steps=80, noise=noise)
            err = np.abs(mels[c] - want)
            print("configs[2] batched 80-step sampling loop cand %d (T=%d) [%s]: %s" % (c, T, what, check_loop(err, "full", mode)))
    finally:
        eng.set_option("attn_f32", 0)
    nz = [rs.randn(64, T + 10).astype(np.float32) for _ in range(B)]
    aus = eng.vocoder(mels, noise=nz)
    ov = oracle.Vocoder(oracle.Model(full_models + "/ggml-vocoder-model.bin"))
    for c in (0, 15):
        assert rel_err(aus[c], ov.run(mels[c], noise=nz[c])) < 1e-3


def test_config2_full_shape_batch_invariance(full_engine, pkg):
    """configs[2] at its REAL shapes — 16 candidates of L = 200 latent rows (T = 870), 80 diffusion steps, full-size weights, device noise, the
    vocoder on all 16 — against the single-candidate path on the same inputs: candidate c of the batch (32 sequences in one row space, shared
    unconditioned integrator, 128-row GEMM tiles) must equal candidate c run alone (2 sequences, 64-row tiles) with the same noise stream
    (stream = global candidate id, option rng_shard_offset). A size-independent property: no oracle run is needed at this size, and the
    single-candidate path is the one the oracle checks at T = 870 (test_diffusion_forward_full_depth)."""
    eng = full_engine
    B, L = 16, 200
    rs = np.random.RandomState(21)
    lats = [rs.randn(L, 1024).astype(np.float32) for _ in range(B)]
    eng.seed(99)
    mels = eng.diffusion(lats, n_steps=80, noise_mode=pkg.NOISE_DEVICE)
    audio = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
    assert all(m.shape == (100, 870) and np.isfinite(m).all() and np.abs(m).max() <= 1.0 + 1e-6 for m in mels)
    try:
        for c in (0, 7, 15):
            eng.set_option("rng_shard_offset", c)
            eng.set_option("rng_shard_total", B)
            eng.seed(99)
            m1 = eng.diffusion([lats[c]], n_steps=80, noise_mode=pkg.NOISE_DEVICE)[0]
            a1 = eng.vocoder([m1], noise_mode=pkg.NOISE_DEVICE)[0]
            dm, da = float(np.abs(m1 - mels[c]).max()), float(np.abs(a1 - audio[c]).max() / np.abs(audio[c]).max())
            print("configs[2] full shape, candidate %d: batch-of-16 vs alone: mel max abs diff %.1e, audio rel diff %.1e" % (c, dm, da))
            assert dm <= 1e-5 and da <= 1e-5
    finally:
        eng.set_option("rng_shard_offset", 0)
        eng.set_option("rng_shard_total", 0)


@pytest.mark.skipif(bool(os.environ.get("TTS_SKIP_LONG_TESTS")), reason="TTS_SKIP_LONG_TESTS set (about 2.5 minutes of oracle time on the host)")
def test_full_size_80_steps_at_bench_length(full_engine, oracle, full_models, oracle_bg):
    """The benchmark's own diffusion problem for one candidate — full-size weights, L = 200 latent rows, T = 870 mel frames, all 80 steps — against
    the oracle with the same explicit noise (160 full-size oracle forwards, about 2.5 minutes of host time; runs by default since round 3)."""
    import oracle_jobs
    L = 200
    T = full_engine.frames(L)
    lat, noise = oracle_jobs.bench_length_inputs(full_engine.frames)
    if oracle_bg:  # computed beside the earlier tests of the session (tests/oracle_jobs.py)
        want = oracle_bg["bench_length"].result(timeout=1800)
    else:
        want = oracle.Diffusion(oracle.Model(full_models + "/ggml-diffusion-model.bin")).sample(lat, n_steps=80, noise=noise)
    try:
        for mode, what in ATTN_MODES:
            full_engine.set_option("attn_f32", mode)
            mel = full_engine.diffusion([lat], n_steps=80, noise=[noise])[0]
            err = np.abs(mel - want)
            assert T == 870 and np.abs(want).max() <= 1.5
            print("full-size 80-step loop at T=%d [%s]: %s" % (T, what, check_loop(err, "full", mode, problem="test_full_size_80_steps_at_bench_length")))
    finally:
        full_engine.set_option("attn_f32", 0)


def test_config5_shape_200_steps(full_engine, oracle, full_models, pkg, oracle_bg):
    """configs[4] on one GPU at its real shapes: 2 distinct prompts x 16 candidates of L = 200 latent rows (T = 870), 200 diffusion steps
    (timestep_map = round(i * 3999 / 199)), full-size weights, device noise. Size-independent property at that size: a candidate of either
    prompt's batch equals the same candidate run alone (noise stream = global candidate id). Against the oracle where the oracle can follow
    in suite time: the 200-step loop at full depth for one short candidate (400 full-depth oracle forwards at T = 39), explicit noise."""
    eng = full_engine
    B, L, steps = 16, 200, 200
    rs = np.random.RandomState(41)
    for prompt in range(2):
        lats = [rs.randn(L, 1024).astype(np.float32) for _ in range(B)]
        eng.seed(500 + prompt)
        mels = eng.diffusion(lats, n_steps=steps, noise_mode=pkg.NOISE_DEVICE)
        assert all(m.shape == (100, 870) and np.isfinite(m).all() and np.abs(m).max() <= 1.0 + 1e-6 for m in mels)
        try:
            c = (3, 12)[prompt]
            eng.set_option("rng_shard_offset", c)
            eng.set_option("rng_shard_total", B)
            eng.seed(500 + prompt)
            m1 = eng.diffusion([lats[c]], n_steps=steps, noise_mode=pkg.NOISE_DEVICE)[0]
            dm = float(np.abs(m1 - mels[c]).max())
            print("configs[4] shape, prompt %d candidate %d, 200 steps: batch-of-16 vs alone: mel max abs diff %.1e" % (prompt, c, dm))
            assert dm <= 1e-5
        finally:
            eng.set_option("rng_shard_offset", 0)
            eng.set_option("rng_shard_total", 0)
    import oracle_jobs
    T = eng.frames(9)
    lat, noise = oracle_jobs.config5_inputs(eng.frames, steps)
    assert np.array_equal(lat, rs.randn(9, 1024).astype(np.float32))  # the same stream as the two batches above
    if oracle_bg:
        want = oracle_bg["config5"].result(timeout=1800)
    else:
        want = oracle.Diffusion(oracle.Model(full_models + "/ggml-diffusion-model.bin")).sample(lat, n_steps=steps, noise=noise)
    try:
        for mode, what in ATTN_MODES:
            eng.set_option("attn_f32", mode)
            mel = eng.diffusion([lat], n_steps=steps, noise=[noise])[0]
            err = np.abs(mel - want)
            print("configs[4] schedule at full depth, 200 steps, T=%d [%s]: %s" % (T, what, check_loop(err, "full", mode, problem="test_config5_shape_200_steps")))
    finally:
        eng.set_option("attn_f32", 0)


def test_config3_shape_64_candidates(full_engine, pkg, voice):
    """configs[3] on one GPU at its real shape: ONE batch of 64 candidates (four candidate tiles of 16 in every decode launch, 128 sequences of T = 870 in the
    diffusion row space, full-size weights) against the shards the 8-GPU form deals out (8 candidates per rank, options rng_shard_offset / rng_shard_total): the
    shard's AR codes are the batch's bit for bit, and a candidate's 80-step diffusion (device noise keyed by the global candidate id) and vocoder output equal the batch's."""
    eng = full_engine
    toks, B, S = bench_prompt(), 64, 24
    eng.seed(7)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True)
    assert steps == S and codes.shape == (B, 502) and len(set(int(r) for r in rows)) == 1
    rs = np.random.RandomState(64)
    dl = [rs.randn(200, 1024).astype(np.float32) for _ in range(B)]
    eng.seed(7)
    mels = eng.diffusion(dl, n_steps=80, noise_mode=pkg.NOISE_DEVICE)
    audio = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
    assert all(m.shape == (100, 870) and np.isfinite(m).all() for m in mels)
    try:
        for rank in (0, 5, 7):  # three of the eight ranks of the 8-GPU form
            eng.set_option("rng_shard_offset", rank * 8)
            eng.set_option("rng_shard_total", B)
            eng.seed(7)
            c8, r8, l8, _ = eng.autoregressive(toks, voice, 8, S, mask_stop=True)
            assert (c8 == codes[rank * 8:(rank + 1) * 8]).all(), rank
            for k in (0, 7):
                assert np.abs(l8[k] - lats[rank * 8 + k]).max() <= 1e-5 * np.abs(lats[rank * 8 + k]).max()
        c = 45
        eng.set_option("rng_shard_offset", c)
        eng.seed(7)
        m1 = eng.diffusion([dl[c]], n_steps=80, noise_mode=pkg.NOISE_DEVICE)[0]
        a1 = eng.vocoder([m1], noise_mode=pkg.NOISE_DEVICE)[0]
        dm, da = float(np.abs(m1 - mels[c]).max()), float(np.abs(a1 - audio[c]).max() / np.abs(audio[c]).max())
        print("configs[3] shape: candidate %d of the batch of 64 vs alone: mel max abs diff %.1e, audio rel diff %.1e" % (c, dm, da))
        assert dm <= 1e-5 and da <= 1e-5
    finally:
        eng.set_option("rng_shard_offset", 0)
        eng.set_option("rng_shard_total", 0)


def test_full_size_ar_192_steps_teacher_forced(full_engine, oracle, full_models, voice):
    """The benchmark's own AR problem: 30 layers, the 64-token prompt, 16 candidates, all 192 decode steps (context 68 .. 260), the oracle replaying the
    device's sampled ids; logits of every step within 1e-4, then the latent pass over the full L = 200 rows for candidates 0 and 15."""
    eng = full_engine
    toks, B, S = bench_prompt(), 16, 192
    eng.seed(5)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True)
    assert steps == S and int(rows[0]) == 200
    ar = oracle.AR(oracle.Model(full_models + "/ggml-model.bin"))
    ar.start(toks, voice, B, len(toks) + 2 + S + 1)
    eng.ar_begin(toks, voice, B, S)
    worst = rel_err(eng.ar_prefill(), ar.prefill())
    for i in range(S - 1):
        worst = max(worst, rel_err(eng.ar_step(codes[:, 1 + i], i), ar.step(codes[:, 1 + i], i)))
    lat_o = ar.latents(codes[[0, 15]], 201)
    e_lat = max(rel_err(lats[c], lat_o[k, :200]) for k, c in enumerate((0, 15)))
    print("full-size AR, 16 candidates x 192 steps teacher-forced: worst logits rel err %.1e; latents (200 rows) rel err %.1e" % (worst, e_lat))
    assert worst < 1e-4 and e_lat < 1e-3


def test_full_size_ar_device_topk_on_off(full_engine, voice):
    """The benchmark's AR workload (16 candidates x 192 masked steps, full-depth weights) with the sampler's top-k selected on the device (default)
    and with the reference's full-logits hand-over: same codes, same latents, same RNG position, and hardly a list that needed its full row."""
    eng = full_engine
    toks = np.array([255] + [int(x) for x in (np.arange(62) * 7 + 3) % 250] + [0], np.int32)
    out = []
    try:
        for on in (1, 0):
            eng.set_option("device_topk", on)
            eng.seed(1234)
            codes, rows, lats, steps = eng.autoregressive(toks, voice, 16, 192, mask_stop=True)
            out.append((codes, rows, lats, steps, eng.rng_uniform(), eng.topk_fallbacks()))
    finally:
        eng.set_option("device_topk", 1)
    a, b = out
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[3] == b[3] == 192 and a[4] == b[4]
    assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    print("full-size AR, 16 x 192 steps: device top-k == full logits; %d of %d candidate-steps fetched their full row" % (a[5], 16 * 191))
    assert a[5] <= 16 * 191 // 100
