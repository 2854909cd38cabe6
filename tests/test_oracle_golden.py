"""CPU tests: the oracle (hand restatement) against the committed golden vectors, which were generated
from the real reference code by tests/golden/make_golden.py, and against oracle/_ref directly when present."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, MODELS


def test_f16_round_matches_numpy(oracle):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.randn(20000).astype(np.float32) * s for s in (1e-9, 1e-5, 1e-3, 1, 100, 7e4)])
    x = np.concatenate([x, np.array([0, -0.0, 65504, 65519.9, 65520, 6e-8, 2.98e-8, 2.9802322e-8, 1e-45], np.float32)])
    got = np.array([oracle.lib().orc_f16_round(float(v)) for v in x], np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert (got == want).all()


def test_rng_golden(oracle):
    g = json.load(open(os.path.join(GOLDEN, "host_golden.json")))
    r = oracle.Rng(245645656)
    assert [r.uniform() for _ in range(8)] == [np.float32(v) for v in g["uniform_seed_245645656"]]
    r.load_state(os.path.join(GOLDEN, "reference_assets", "test_autoregressive_seed.bin"))
    assert [r.uniform() for _ in range(8)] == [np.float32(v) for v in g["uniform_seed_245645656"]]
    r.load_state(os.path.join(GOLDEN, "reference_assets", "test_diffusion_seed.bin"))
    assert (r.normal(64) == np.array(g["normal_diffusion_seed"], np.float32)).all()


def test_tokenizer_and_buckets_golden(oracle):
    g = json.load(open(os.path.join(GOLDEN, "host_golden.json")))
    tk = oracle.Tokenizer(os.path.join(MODELS, "tokenizer.json"))
    for msg, ids in g["tokenizer"].items():
        assert list(tk.encode(msg)) == ids
    assert list(tk.encode("this is a test message.")) == [255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0]
    for n, tab in g["buckets"].items():
        assert (oracle.buckets(int(n)) == np.array(tab)).all()


def test_schedule_and_update_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "schedule_golden.npz"))
    tm = oracle.default_timestep_map(80)
    assert (tm == g["timestep_map"]).all()
    assert tm[1] == 51 and tm[79] == 3999  # literal table main.cpp:5641-5648
    s = oracle.schedule(tm)
    for k in oracle.SCHED_KEYS:
        assert (s[k] == g[k]).all(), k
    # SURVEY §8c spot values
    assert abs(s["betas"][1] - 2.920470051e-03) < 1e-12 and abs(s["acp"][79] - 4.246653922e-05) < 1e-14
    for t in (0, 51, 2025, 3999):
        assert (oracle.timestep_embedding(t) == g["temb_%d" % t]).all()
    for t in (79, 40, 1, 0):
        got = oracle.diffusion_update(tm, t, g["upd%d_oc" % t], g["upd%d_ou" % t], g["upd%d_x" % t], g["upd%d_nz" % t], 23)
        assert (got == g["upd%d_out" % t]).all(), t


def test_padding_and_trim(oracle):
    codes = np.array([5, 6, 83, 83, 7], np.int32)
    p = oracle.apply_padding(codes)
    assert p[0] == 8192 and p[501] == 8193 and list(p[1:6]) == [5, 6, 83, 83, 7]
    assert list(p[498:501]) == [45, 45, 248] and (p[6:498] == 83).all()
    assert oracle.trimmed_rows(p) == 5 + 8  # rows kept until MORE than 8 consecutive 83s
    R = oracle.ref()
    if R is not None:
        rs = np.random.RandomState(3)
        for _ in range(20):
            n = rs.randint(1, 400)
            c = rs.choice([83, 83, 17, 4000, 8193, 8139], n).astype(np.int32)
            want = np.empty(502, np.int32)
            R.ref_apply_padding(c, n, want)
            got = oracle.apply_padding(c)
            assert (got == want).all()
            lat = rs.randn(1, 500, 1024).astype(np.float32)
            out = np.empty(500 * 1024, np.float32)
            rows = np.empty(1, np.int32)
            R.ref_trim_latents(lat.reshape(-1), want, 1, out, rows)
            assert rows[0] == oracle.trimmed_rows(got)


def test_oracle_stages_run_and_are_consistent(oracle, small_models, voice):
    """Decode-with-cache, the cache-less latent pass and prefill are three views of the same transformer."""
    from conftest import DEFAULT_TOKENS
    ar = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks, B = DEFAULT_TOKENS, 2
    ar.start(toks, voice, B, 40)
    lg0 = ar.prefill()
    assert np.isfinite(lg0).all() and (lg0[0] == lg0[1]).all()
    codes = np.random.RandomState(0).randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    lat_a = ar.latents(codes, 12)
    lat_b = ar.latents(codes, 30)  # causal: a longer pass leaves the prefix unchanged
    assert np.abs(lat_a - lat_b[:, :12]).max() < 1e-4
    md = oracle.Diffusion(oracle.Model(small_models + "/ggml-diffusion-model.bin"))
    T = md.T_of(9)
    assert T == 9 * 4 * 24000 // 22050
    x = np.random.RandomState(1).randn(100, T).astype(np.float32)
    out = md.forward(None, x, 51)
    assert out.shape == (200, T) and np.isfinite(out).all()
