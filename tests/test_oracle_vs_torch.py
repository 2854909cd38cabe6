"""CPU: the oracle (oracle/orc_*.cpp, hand-written loops) against PyTorch's own ops, so that the op semantics and the block
wiring of the three network stages are torch-pinned. Op level first — conv1d as fp16 im2col, GroupNorm,
LayerNorm, tanh-GELU, SiLU, softmax — then every block and the whole graphs assembled from torch ops in tests/torch_ref.py
on synthetic weights in the reference's file format. What remains "[ggml-unverified]" after this file: the ggml fork's
constants only (GroupNorm eps, fp16 activation tables), both switches on either side.

Tolerances: single ops 1e-5 of the output range (f32 summation order). Whole diffusion / vocoder graphs: the reference rounds
every convolution operand to fp16, and an operand that sits on a rounding boundary rounds the other way under a different f32
summation order; through ~20 convolutions this puts ANY two f32 evaluations 2-6e-4 of the output range apart (measured below:
torch-f32 against the same graph in f64 is as far as the oracle is). The gate is therefore triangulated: the oracle must be
within 1e-3 of the f64 evaluation (north-star tolerance) and no further from it than 3x what torch's own f32 evaluation is."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import torch_ref as TR
from conftest import DEFAULT_TOKENS

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))


@pytest.fixture(scope="module")
def ops(oracle):
    L = oracle.lib()
    ci, cf = C.c_int, C.c_float
    L.orc_op_conv1d_f16.argtypes = [_f32p, ci, ci, _f32p, ci, ci, _f32p, ci, ci, _f32p]
    L.orc_op_groupnorm.argtypes = [_f32p, ci, ci, ci, cf, _f32p, _f32p, _f32p]
    L.orc_op_layernorm.argtypes = [_f32p, ci, ci, cf, _f32p, _f32p]
    L.orc_op_unary.argtypes = [ci, _f32p, C.c_int64]
    L.orc_op_softmax.argtypes = [_f32p, ci]
    L.orc_op_gemm_kn.argtypes = [ci, ci, ci, _f32p, _f32p, _f32p, _f32p]
    return L


# ---- single ops ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,dil,pad", [(1, 1, 0), (3, 1, 1), (5, 1, 2), (7, 1, 0), (3, 3, 3), (3, 9, 9), (3, 27, 27)])
def test_conv1d_f16_vs_torch(ops, K, dil, pad):
    """ggml conv1d = fp16 im2col x fp16 weights, f32 accumulate == F.conv1d on fp16-rounded operands."""
    rs = np.random.RandomState(K * 100 + dil)
    T, Cin, Cout = 95, 37, 29
    x = rs.randn(T, Cin).astype(np.float32)
    w = (rs.randn(Cout, Cin, K) * 0.2).astype(np.float32)  # file layout ne = [K, Cin, Cout]
    b = rs.randn(Cout).astype(np.float32)
    Tout = T + 2 * pad - dil * (K - 1)
    y = np.empty((Tout, Cout), np.float32)
    ops.orc_op_conv1d_f16(x.reshape(-1), T, Cin, w.reshape(-1), K, Cout, b, pad, dil, y.reshape(-1))
    want = TR.conv1d_f16(torch.from_numpy(x.T.copy()), torch.from_numpy(w), torch.from_numpy(b), padding=pad, dilation=dil).numpy().T
    assert y.shape == want.shape and rel(y, want) < 1e-5
    # and the rounding points matter: without them the result differs by ~fp16 epsilon
    plain = F.conv1d(torch.from_numpy(x.T.copy())[None], torch.from_numpy(w), torch.from_numpy(b), padding=pad, dilation=dil)[0].numpy().T
    assert rel(y, plain) > 1e-5


@pytest.mark.parametrize("eps", [1e-6, 1e-5])
def test_groupnorm_vs_torch(ops, eps):
    rs = np.random.RandomState(3)
    T, Cn = 77, 128
    x = (rs.randn(T, Cn) * 2 + 0.7).astype(np.float32)
    g, b = rs.randn(Cn).astype(np.float32), rs.randn(Cn).astype(np.float32)
    y = np.empty_like(x)
    ops.orc_op_groupnorm(x.reshape(-1), T, Cn, 32, eps, g, b, y.reshape(-1))
    want = F.group_norm(torch.from_numpy(x.T.copy())[None], 32, torch.from_numpy(g), torch.from_numpy(b), eps)[0].numpy().T
    assert rel(y, want) < 1e-5


def test_layernorm_vs_torch(ops):
    rs = np.random.RandomState(4)
    x = (rs.randn(9, 1024) * 3 - 1).astype(np.float32)
    g, b = rs.randn(1024).astype(np.float32), rs.randn(1024).astype(np.float32)
    y = x.copy()
    ops.orc_op_layernorm(y.reshape(-1), 9, 1024, 1e-5, g, b)
    want = F.layer_norm(torch.from_numpy(x), (1024,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    assert rel(y, want) < 1e-5


def test_activations_and_softmax_vs_torch(ops):
    x = np.linspace(-12, 12, 4001).astype(np.float32)
    t = torch.from_numpy(x)
    for code, want in ((0, F.gelu(t, approximate="tanh")), (1, F.silu(t))):
        y = x.copy()
        ops.orc_op_unary(code, y, y.size)
        assert np.abs(y - want.numpy()).max() < 2e-6
    s = (np.random.RandomState(5).randn(301) * 4).astype(np.float32)
    want = torch.softmax(torch.from_numpy(s), 0).numpy()
    ops.orc_op_softmax(s, s.size)
    assert np.abs(s - want).max() < 1e-6


def test_gemm_vs_torch(ops):
    rs = np.random.RandomState(6)
    M, N, K = 19, 70, 333
    a, bt, bias = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32), rs.randn(N).astype(np.float32)
    c = np.empty((M, N), np.float32)
    ops.orc_op_gemm_kn(M, N, K, a.reshape(-1), bt.reshape(-1), bias, c.reshape(-1))
    assert rel(c, a @ bt + bias) < 1e-5


def test_upscale_index_vs_interpolate():
    """ggml's nearest upscale (src = (int)(dst / ((float)T / L)), f32 division: what the oracle and the engine implement) against
    F.interpolate(mode='nearest') (= exact floor(dst * L / T)) for every latent length the AR stage can produce. They are the same
    map except where dst * L / T is an exact integer and the f32 quotient lands just below it: there ggml picks the previous source
    row. Characterised here so the difference is a documented property of the reference's op, not an accident: 21 of the 500
    lengths, at most 3 of their T positions, always exactly one row earlier."""
    bad = []
    for L in range(1, 501):
        T = L * 4 * 24000 // 22050
        idx = TR.TorchDiffusion.upscale_index(L, T)
        ref = F.interpolate(torch.arange(L, dtype=torch.float32)[None, None], size=T, mode="nearest")[0, 0].numpy().astype(np.int64)
        assert (ref == (np.arange(T) * L) // T).all()
        d = idx != ref
        if d.any():
            assert ((ref - idx)[d] == 1).all() and ((np.arange(T) * L) % T == 0)[d].all()
            bad.append((L, int(d.sum())))
    print("ggml-vs-torch nearest index differences (L, positions):", bad)
    assert len(bad) == 21 and max(n for _, n in bad) == 3


# ---- blocks and whole graphs -------------------------------------------------------------------------------------------
def triangulate(name, oracle_out, t32, t64):
    eo, et, eot = rel(oracle_out, t64), rel(t32, t64), rel(oracle_out, t32)
    print("%s: oracle vs f64 %.2e | torch-f32 vs f64 %.2e | oracle vs torch-f32 %.2e" % (name, eo, et, eot))
    assert eo < 1e-3 and eo < 3 * et + 1e-5, (name, eo, et)


@pytest.fixture(scope="module")
def tdiff(small_models, oracle):
    p = small_models + "/ggml-diffusion-model.bin"
    return TR.TorchDiffusion(p, oracle.buckets), oracle.Diffusion(oracle.Model(p)), TR.TorchDiffusion(p, oracle.buckets, dtype=torch.float64)


@pytest.mark.parametrize("L", [5, 23])
def test_diffusion_code_embedding_vs_torch(tdiff, L):
    """latent conditioner: conv k3, AttentionBlocks (GroupNorm, fp16 qkv conv, T5 bias, softmax, proj_out), code_norm, scale/shift,
    nearest upsample."""
    td, od, td64 = tdiff
    lat = np.random.RandomState(L).randn(L, 1024).astype(np.float32)
    T = od.T_of(L)
    e = rel(od.code_embedding(lat, T), td.code_embedding(lat, T))
    print("code embedding L=%d: %.2e" % (L, e))
    assert e < 2e-4


@pytest.mark.parametrize("L,timestep,cond", [(12, 3999, True), (12, 3999, False), (30, 51, True)])
def test_diffusion_forward_vs_torch(tdiff, oracle, L, timestep, cond):
    """The whole diffusion_graph: time MLP, integrator ResBlock + AttentionBlock, inp_block, concat + integrating conv, main and
    tail layers, out head."""
    td, od, td64 = tdiff
    T = od.T_of(L)
    lat = np.random.RandomState(L).randn(L, 1024).astype(np.float32)
    x_t = np.random.RandomState(7).randn(100, T).astype(np.float32)
    ce = od.code_embedding(lat, T) if cond else None
    got = od.forward(ce, x_t, timestep)
    te = oracle.timestep_embedding(timestep)
    assert got.shape == (200, T)
    triangulate("diffusion forward L=%d t=%d cond=%s" % (L, timestep, cond), got, td.forward(ce, x_t, te), td64.forward(ce, x_t, te))


def test_diffusion_forward_mid_depth_vs_torch(mid_models, oracle):
    p = mid_models + "/ggml-diffusion-model.bin"
    td, td64 = TR.TorchDiffusion(p, oracle.buckets), TR.TorchDiffusion(p, oracle.buckets, dtype=torch.float64)
    od = oracle.Diffusion(oracle.Model(p))
    L, T = 16, od.T_of(16)
    lat = np.random.RandomState(1).randn(L, 1024).astype(np.float32)
    x_t = np.random.RandomState(2).randn(100, T).astype(np.float32)
    ce = od.code_embedding(lat, T)
    te = oracle.timestep_embedding(2025)
    triangulate("diffusion forward (3 main blocks)", od.forward(ce, x_t, 2025), td.forward(ce, x_t, te), td64.forward(ce, x_t, te))


@pytest.mark.parametrize("T", [1, 9, 33])
def test_vocoder_vs_torch(small_models, oracle, T):
    """UnivNet: reflect pad + conv_pre, ConvTranspose1d (stride 8/8/4, crop stride/2), kernel predictor, dilated convs, the
    location-variable convolution through unfold (main.cpp:4337-4456), gated residual, conv_post."""
    tv = TR.TorchVocoder(small_models + "/ggml-vocoder-model.bin")
    tv64 = TR.TorchVocoder(small_models + "/ggml-vocoder-model.bin", dtype=torch.float64)
    ov = oracle.Vocoder(oracle.Model(small_models + "/ggml-vocoder-model.bin"))
    rs = np.random.RandomState(T)
    mel = np.clip(rs.randn(100, T) * 0.5, -1, 1).astype(np.float32)
    nz = rs.randn(64, T + 10).astype(np.float32)
    got = ov.run(mel, noise=nz)
    md = mel.copy().reshape(-1)
    oracle.lib().orc_denormalize_mel(md, md.size)
    assert got.shape == ((T + 10) * 256 - 6,)
    triangulate("vocoder T=%d" % T, got, tv.forward(md.reshape(100, T), nz), tv64.forward(md.reshape(100, T), nz))


def test_ar_vs_torch(small_models, oracle, voice):
    """GPT-2 stack: prefill logits, three decode steps through the oracle's KV cache against a cache-free torch evaluation at the
    reference's mel positions (0, then i + 2: main.cpp:5244), and the latent pass."""
    ta = TR.TorchAR(small_models + "/ggml-model.bin")
    oa = oracle.AR(oracle.Model(small_models + "/ggml-model.bin"))
    toks = DEFAULT_TOKENS
    oa.start(toks, voice, 1, len(toks) + 2 + 8)
    lg = oa.prefill()[0]
    e = rel(lg, ta.logits_after(toks, voice, [8192], [0]))
    print("AR prefill logits: %.2e" % e)
    assert e < 2e-4
    mel, pos = [8192], [0]
    for i, tok in enumerate([17, 4000, 8191]):
        lg = oa.step(np.array([tok], np.int32), i)[0]
        mel.append(tok)
        pos.append(i + 2)
        e = rel(lg, ta.logits_after(toks, voice, mel, pos))
        print("AR step %d logits: %.2e" % (i, e))
        assert e < 2e-4
    codes = np.full((1, 502), 83, np.int32)
    codes[0, 0] = 8192
    codes[0, 1:12] = np.random.RandomState(0).randint(0, 8192, 11)
    n_mel = 14
    e = rel(oa.latents(codes, n_mel)[0], ta.latents(toks, voice, codes[0], n_mel))
    print("AR latents: %.2e" % e)
    assert e < 2e-4
