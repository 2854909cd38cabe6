"""GPU parity of the autoregressive stage against the oracle (CPU restatement), through the C ABI."""
import time

import numpy as np
import pytest

from conftest import DEFAULT_TOKENS

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("B", [1, 4, 16, 19])  # 16 = one full candidate tile of the decode kernels, 19 = two tiles, ragged
def test_prefill_and_steps_logits(engine, oracle, small_models, voice, B):
    engine.load(ar=small_models + "/ggml-model.bin")
    m = oracle.Model(small_models + "/ggml-model.bin")
    ar = oracle.AR(m)
    assert engine.ar_layers == ar.n_layers == 2
    toks = DEFAULT_TOKENS
    engine.ar_begin(toks, voice, B, 8)
    ar.start(toks, voice, B, len(toks) + 2 + 9)
    lg, lo = engine.ar_prefill(), ar.prefill()
    assert rel_err(lg, lo) < 1e-4  # f32 both sides: only summation order differs
    rs = np.random.RandomState(B)
    for i in range(6 if B <= 4 else 3):
        prev = rs.randint(0, 8192, B).astype(np.int32)
        lg, lo = engine.ar_step(prev, i), ar.step(prev, i)
        assert rel_err(lg, lo) < 1e-4, i


@pytest.mark.parametrize("models", ["small", "full"])
def test_loader_paths_give_the_same_bits(engine, small_models, full_models, voice, models):
    """tts_load_ar builds its device layouts three ways: on one host thread (load_threads = 1: rounds 1-5), on several host threads (load_device_pack = 0), and — the
    default since round 6 — with kernels from the uploaded file tensors (ar.hip: pk_*_kernel). The layouts are index permutations plus one rounding per element, so the
    three loads must give bit-identical logits in the prompt pass, the decode steps (16 candidates: every decode slab layout and the head) and the latent pass (the
    split-precision copies). Full size: every layer shape the benchmark runs."""
    d = small_models if models == "small" else full_models
    toks, B = DEFAULT_TOKENS, 16
    rs = np.random.RandomState(5)
    prevs = [rs.randint(0, 8192, B).astype(np.int32) for _ in range(3)]
    codes = rs.randint(0, 8192, (B, 12)).astype(np.int32)
    out = {}
    try:
        for name, opts in (("serial host", {"load_threads": 1}), ("host threads", {"load_threads": 0, "load_device_pack": 0}), ("device", {"load_threads": 0, "load_device_pack": 1})):
            for k, v in opts.items():
                engine.set_option(k, v)
            t0 = time.time()
            engine.load(ar=d + "/ggml-model.bin")
            print("tts_load_ar, %s weights, %s: %.2f s" % (models, name, time.time() - t0))
            engine.ar_begin(toks, voice, B, 8)
            got = [engine.ar_prefill().copy()]
            for i, prev in enumerate(prevs):
                got.append(engine.ar_step(prev, i).copy())
            engine.seed(9)
            c, rows, lats, steps = engine.autoregressive(toks, voice, 2, 12, mask_stop=True)
            got += [np.asarray(c).copy()] + [np.asarray(l).copy() for l in lats]
            out[name] = got
    finally:
        engine.set_option("load_threads", 0)
        engine.set_option("load_device_pack", 1)
    for name in ("host threads", "device"):
        assert len(out[name]) == len(out["serial host"])
        for a_, b_ in zip(out["serial host"], out[name]):
            assert a_.shape == b_.shape and (a_.view(np.uint32) == b_.view(np.uint32)).all(), name


def test_long_context_crosses_attention_chunk(engine, oracle, small_models, voice):
    """A 280-id prompt: the decode attention walks its keys in chunks of 288, so these steps cross from one chunk to two
    (283 -> 294 keys); the prompt pass runs 18 position tiles; the latent pass takes 281 prompt K/V rows from the cache."""
    engine.load(ar=small_models + "/ggml-model.bin")
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks = np.random.RandomState(11).randint(1, 250, 280).astype(np.int32)
    B = 3
    engine.ar_begin(toks, voice, B, 16)
    ar.start(toks, voice, B, len(toks) + 2 + 17)
    assert rel_err(engine.ar_prefill(), ar.prefill()) < 1e-4
    rs = np.random.RandomState(12)
    for i in range(12):
        prev = rs.randint(0, 8192, B).astype(np.int32)
        assert rel_err(engine.ar_step(prev, i), ar.step(prev, i)) < 1e-4, i
    codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    lg, lo = engine.ar_latents(codes, 20), ar.latents(codes, 20)
    assert lg.shape == lo.shape == (B, 20, 1024)
    assert rel_err(lg, lo) < 1e-4


def test_ggml_lut_mode(engine, oracle, small_models, voice):
    """Option ggml_lut = 1 (GELU through fp16 on both sides, decode softmax exp through fp16) against the
    oracle's emulation of the same tables. A value that straddles an fp16 rounding boundary can round differently on the
    two sides (5e-4 relative on that element), hence the wider gate than the default mode's 1e-4."""
    engine.load(ar=small_models + "/ggml-model.bin")
    engine.set_option("ggml_lut", 1)
    oracle.set_flags(lut=1)
    try:
        ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
        toks, B = DEFAULT_TOKENS, 4
        engine.ar_begin(toks, voice, B, 8)
        ar.start(toks, voice, B, len(toks) + 2 + 9)
        errs = [rel_err(engine.ar_prefill(), ar.prefill())]
        rs = np.random.RandomState(21)
        for i in range(5):
            prev = rs.randint(0, 8192, B).astype(np.int32)
            errs.append(rel_err(engine.ar_step(prev, i), ar.step(prev, i)))
        codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
        codes[:, 0] = 8192
        errs.append(rel_err(engine.ar_latents(codes, 24), ar.latents(codes, 24)))
        print("ggml_lut AR rel errs:", ["%.1e" % e for e in errs])
        assert max(errs) < 1e-3, errs
        # the switch does something: default-mode logits differ from LUT-mode logits
        oracle.set_flags(lut=0)
        ar2 = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
        ar2.start(toks, voice, B, len(toks) + 2 + 9)
        engine.ar_begin(toks, voice, B, 8)
        assert rel_err(engine.ar_prefill(), ar2.prefill()) > 1e-6
    finally:
        oracle.set_flags()
        engine.set_option("ggml_lut", 0)  # the engine fixture is shared by the whole session


def test_latents(engine, oracle, small_models, voice):
    engine.load(ar=small_models + "/ggml-model.bin")
    m = oracle.Model(small_models + "/ggml-model.bin")
    ar = oracle.AR(m)
    toks = DEFAULT_TOKENS
    B = 2
    rs = np.random.RandomState(5)
    codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    engine.ar_begin(toks, voice, B, 8)
    ar.start(toks, voice, B, 32)
    n_mel = 40
    lg, lo = engine.ar_latents(codes, n_mel), ar.latents(codes, n_mel)
    assert lg.shape == lo.shape == (B, 40, 1024)
    assert rel_err(lg, lo) < 1e-4


def test_sampled_ids_teacher_forced(engine, oracle, small_models, voice):
    """Feed the oracle's trajectory to the device step by step: logits agree to f32 round-off at every
    step and the product sampler (same RNG stream) picks the oracle's id. A handful of flips is the
    physical limit: the reference rounds QKV to fp16 (main.cpp:2789-2790), which turns 1e-7 summation-
    order noise into occasional 5e-4 jumps, so a uniform draw can land on the other side of a CDF edge."""
    engine.load(ar=small_models + "/ggml-model.bin")
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks, B, S, seed = DEFAULT_TOKENS, 4, 40, 245645656
    rng = oracle.Rng(seed)
    engine.seed(seed)
    ar.start(toks, voice, B, len(toks) + 2 + S + 1)
    engine.ar_begin(toks, voice, B, S)
    lo, lg = ar.prefill(), engine.ar_prefill()
    ids = np.tile(np.array([1] * (len(toks) + 1) + [8192], np.int32), (B, 1))
    mism, worst = 0, 0.0
    for i in range(S):
        worst = max(worst, rel_err(lg, lo))
        so = oracle.sample(lo, ids, rng)
        sg = engine.sample(lg, ids)
        mism += int((so != sg).sum())
        ids = so.reshape(B, 1)
        lo, lg = ar.step(so, i), engine.ar_step(so, i)
    assert worst < 1e-4, worst
    assert mism <= 2, "%d of %d sampled ids differ" % (mism, B * S)
    assert engine.rng_uniform() == rng.uniform()  # RNG streams in lock-step (2 uniforms / candidate / step)


@pytest.mark.parametrize("B,seed", [(1, 0), (4, 245645656)])
def test_autoregressive_driver(engine, oracle, small_models, voice, B, seed):
    """Whole autoregressive() driver at a fixed seed against the oracle's driver: identical token ids
    (any divergence must be explained by sub-tolerance logits at the first divergent step), identical
    padding/trim bookkeeping, latents within 1e-3."""
    engine.load(ar=small_models + "/ggml-model.bin")
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks, S = DEFAULT_TOKENS, 40
    engine.seed(seed)
    codes_g, rows_g, lats_g, steps_g = engine.autoregressive(toks, voice, B, S, mask_stop=True)
    rng = oracle.Rng(seed)
    rc, codes_o, steps_o, raw = ar.generate(toks, voice, B, rng, S, mask_stop=True)
    assert rc == 0 and steps_g == steps_o == S
    assert (codes_g[:, 0] == 8192).all() and (codes_g[:, 501] == 8193).all()
    if not (codes_g == codes_o).all():
        c, j = np.argwhere(codes_g != codes_o)[0]
        step = j - 1
        # replay the oracle's prefix on the device: the logits that produced the divergent sample agree
        ar.start(toks, voice, B, len(toks) + 2 + S + 1)
        engine.ar_begin(toks, voice, B, S)
        lo, lg = ar.prefill(), engine.ar_prefill()
        for i in range(step):
            lo, lg = ar.step(codes_o[:, 1 + i], i), engine.ar_step(codes_o[:, 1 + i], i)
        assert rel_err(lg, lo) < 1e-4
        print("trajectories split at candidate %d step %d with logits within %.1e (fp16-QKV rounding flip); "
              "bookkeeping and latents are checked on the device's own codes" % (c, step, rel_err(lg, lo)))
        # up to the split everything is identical, and every other candidate's stream is untouched by it
        assert (codes_g[:, :j] == codes_o[:, :j]).all()
        codes_o = codes_g  # from here on: the oracle is driven with the device's trajectory
    for c in range(B):
        assert rows_g[c] == oracle.trimmed_rows(codes_o[c])
    lat_o = ar.latents(codes_o, min(502, int(rows_g.max()) + 1))
    for c in range(B):
        assert rel_err(lats_g[c], lat_o[c, :rows_g[c]]) < 1e-3
    assert engine.rng_uniform() == rng.uniform()


def test_fp16_decode_weights_option(pkg, oracle, small_models, voice):
    """Option ar_weights = 1 (throughput mode): the decode step streams fp16 copies of the weights. Logits stay within 5e-3 of the
    oracle's f32 evaluation (teacher-forced), differ from the f32 mode's, and the prompt pass / latent pass remain f32-exact."""
    eng = pkg.Engine(0)
    eng.set_option("ar_weights", 1)
    eng.load(ar=small_models + "/ggml-model.bin")
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks, B = DEFAULT_TOKENS, 5
    eng.ar_begin(toks, voice, B, 8)
    ar.start(toks, voice, B, len(toks) + 2 + 9)
    assert rel_err(eng.ar_prefill(), ar.prefill()) < 1e-4  # the prompt pass keeps the f32 weights
    rs = np.random.RandomState(2)
    errs = []
    for i in range(6):
        prev = rs.randint(0, 8192, B).astype(np.int32)
        errs.append(rel_err(eng.ar_step(prev, i), ar.step(prev, i)))
    print("fp16 decode weights: logits rel err per step", ["%.1e" % e for e in errs])
    assert max(errs) < 5e-3 and max(errs) > 1e-5
    codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    assert rel_err(eng.ar_latents(codes, 12), ar.latents(codes, 12)) < 1e-4
    eng.close()
    # toggling the option on an engine loaded without the fp16 slabs is an error, not a silent f32 run
    e2 = pkg.Engine(0)
    e2.load(ar=small_models + "/ggml-model.bin")
    e2.set_option("ar_weights", 1)
    e2.ar_begin(toks, voice, 2, 4)
    e2.ar_prefill()
    with pytest.raises(pkg.TtsError):
        e2.ar_step(np.array([1, 2], np.int32), 0)
    e2.close()


def test_maximum_sizes(engine, oracle, small_models, voice):
    """The reference's limits at once: 404 text ids (all text positions), 500 sampled codes (what apply_padding accepts, main.cpp:4517),
    decode context 406 + 500 positions, the latent pass over all 502 mel positions (907 rows per candidate)."""
    engine.load(ar=small_models + "/ggml-model.bin")
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks = np.random.RandomState(3).randint(1, 250, 404).astype(np.int32)
    toks[0], toks[-1] = 255, 0
    B, S = 2, 500
    engine.ar_begin(toks, voice, B, S)
    ar.start(toks, voice, B, len(toks) + 2 + S + 1)
    errs = [rel_err(engine.ar_prefill(), ar.prefill())]
    rs = np.random.RandomState(4)
    for i in range(S - 1):
        prev = rs.randint(0, 8192, B).astype(np.int32)
        lo = ar.step(prev, i)
        lg = engine.ar_step(prev, i)
        if i in (0, 1, 250, 497, 498):
            errs.append(rel_err(lg, lo))
    print("max-size AR logits rel errs:", ["%.1e" % e for e in errs])
    assert max(errs) < 1e-4
    codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    lg, lo = engine.ar_latents(codes, 502), ar.latents(codes, 502)
    assert lg.shape == lo.shape == (B, 500, 1024)
    assert rel_err(lg, lo) < 1e-4


def _prompt_ids(toks, B):
    return np.tile(np.array([1] * (len(toks) + 1) + [8192], np.int32), (B, 1))


def _sampled_run(engine, toks, voice, B, S, seed, mask, fused):
    """The decode loop spelled with the stepwise ABI. fused(i) -> True: tts_ar_step_sample (device top-k), False: tts_ar_step + tts_sample."""
    engine.seed(seed)
    engine.ar_begin(toks, voice, B, S)
    lg = engine.ar_prefill()
    if mask:
        lg[:, 8193] = -1e30
    s = engine.sample(lg, _prompt_ids(toks, B))
    out, fb = [s], 0
    for i in range(S - 1):
        if fused(i):
            s = engine.ar_step_sample(s, i, mask_stop=mask)
            fb += engine.topk_fallbacks()
        else:
            lg = engine.ar_step(s, i)
            if mask:
                lg[:, 8193] = -1e30
            s = engine.sample(lg, s.reshape(B, 1))
        out.append(s)
    return np.stack(out), engine.rng_uniform(), fb


@pytest.mark.parametrize("B,mask", [(16, True), (16, False), (3, False), (1, True)])
def test_device_topk_step_is_step_then_sample(engine, small_models, voice, B, mask):
    """tts_ar_step_sample (the sampler's top-k selected by sample_prefilter_kernel, 64..128 logits per candidate cross PCIe) returns exactly what
    tts_ar_step + tts_sample return from the full rows: same ids, same RNG position; also when the two step graphs alternate."""
    engine.load(ar=small_models + "/ggml-model.bin")
    toks, S, seed = DEFAULT_TOKENS, 28, 4242 + B
    want, u_want, _ = _sampled_run(engine, toks, voice, B, S, seed, mask, lambda i: False)
    got, u_got, fb = _sampled_run(engine, toks, voice, B, S, seed, mask, lambda i: True)
    assert (got == want).all() and u_got == u_want
    assert fb == 0  # continuous logits: no list ever needed its full row
    mixed, u_mixed, _ = _sampled_run(engine, toks, voice, B, S, seed, mask, lambda i: i % 3 != 1)
    assert (mixed == want).all() and u_mixed == u_want
    if mask:
        assert (got != 8193).all()


@pytest.mark.parametrize("kind", ["all_equal", "coarse", "sparse_ties"])
def test_device_topk_falls_back_to_the_full_row_on_ties(pkg, engine, small_models, voice, tmp_path, kind):
    """A head whose logits tie: the prefilter finds no threshold keeping 64..128 of them (all_equal, coarse) or the survivors tie (sparse_ties), the
    host fetches the rows it cannot decide and still returns the ids of the two-call path."""
    from tortoise_cpp_amd import synth_weights as SW
    t = SW.read_ggml(small_models + "/ggml-model.bin")
    rs = np.random.RandomState(3)
    t["inference_model.lm_head.1.weight"][:] = 0
    bias = {"all_equal": np.zeros(8194), "coarse": np.round(rs.randn(8194) * 2), "sparse_ties": np.round(rs.randn(8194) * 300) / 16}[kind]
    t["inference_model.lm_head.1.bias"][:] = bias.astype(np.float32)
    path = str(tmp_path / "ar_ties.bin")
    w = SW.GgmlWriter(path)
    for name, arr in t.items():
        w.add(name, arr)
    w.close()
    eng = engine
    eng.load(ar=path)
    toks, B, S = DEFAULT_TOKENS, 4, 5
    want, u_want, _ = _sampled_run(eng, toks, voice, B, S, 11, False, lambda i: False)
    got, u_got, fb = _sampled_run(eng, toks, voice, B, S, 11, False, lambda i: True)
    assert (got == want).all() and u_got == u_want
    if kind != "sparse_ties":
        assert fb == B * (S - 1)  # every list was refused
    print("%s: %d of %d candidate-steps sampled from their full row" % (kind, fb, B * (S - 1)))


@pytest.mark.parametrize("flags", [dict(mask_stop=True), dict(retire=True), dict()])
def test_autoregressive_driver_device_topk_on_off(pkg, engine, small_models, voice, flags):
    """Option device_topk only moves the top-k selection onto the device: codes, rows, latents and the RNG position of tts_autoregressive do not change."""
    engine.load(ar=small_models + "/ggml-model.bin")
    toks, B, S = DEFAULT_TOKENS, 16, 48
    res = []
    for on in (1, 0):
        engine.set_option("device_topk", on)
        engine.seed(77)
        try:
            codes, rows, lats, steps = engine.autoregressive(toks, voice, B, S, **flags)
        except pkg.TtsError as e:  # strict mode may legitimately run out of steps: then both modes must
            res.append(("err", str(e)))
            continue
        res.append((codes, rows, lats, steps, engine.rng_uniform(), engine.topk_fallbacks()))
    engine.set_option("device_topk", 1)
    failed = [isinstance(r[0], str) for r in res]
    if any(failed):
        assert all(failed), res
        return
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and res[0][3] == res[1][3] and res[0][4] == res[1][4]
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)
    assert res[0][5] == 0
