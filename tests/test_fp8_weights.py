"""Option ar_weights = 2 ("fp8 weights for decode"): the decode step streams OCP fp8 e4m3 copies of the weight
matrices with one power-of-two scale per output column. No counterpart in the reference; what is checked:
  * CPU: the library's quantiser is OCP e4m3 with round-to-nearest-even and saturation (every code, every tie, the subnormals) — against
    torch.float8_e4m3fn;
  * GPU: on a model file whose matrices are EXACTLY representable (e4m3 value x power-of-two column scale) the fp8 mode reproduces the f32
    mode (packing orders, in-register conversion, scale application are exact: only the activation split is left, ~1e-6);
  * GPU: on ordinary weights the logits stay within the quantisation's reach of the oracle, the prompt pass and the latent pass stay
    f32-exact, and a mode the state was not loaded for is an error."""
import os

import numpy as np
import pytest
import torch

from conftest import DEFAULT_TOKENS


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_fp8_quantiser_is_ocp_e4m3(pkg):
    codes = np.arange(256, dtype=np.uint8)
    vals = torch.from_numpy(codes).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    fin = np.isfinite(vals)
    assert fin.sum() == 254 and vals[fin].max() == 448.0  # no infinities, two NaN codes
    enc = pkg.host_fp8_e4m3(vals[fin])
    same = (enc == codes[fin]) | ((vals[fin] == 0) & ((enc & 0x7F) == 0))
    assert same.all()
    pos = np.sort(vals[fin & (vals >= 0)])
    mids = ((pos[1:] + pos[:-1]) / 2).astype(np.float32)  # ties: to the even mantissa
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.randn(20000).astype(np.float32) * s for s in (1e-3, 1e-2, 0.1, 1, 10, 100, 300)] +
                       [mids, -mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                        np.array([448, 449, 464, 480, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 0.0], np.float32)])
    want = torch.from_numpy(x).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = pkg.host_fp8_e4m3(x)
    bad = (got != want) & ~(((got & 0x7F) == 0) & ((want & 0x7F) == 0))  # +0 / -0 are the same value
    assert not bad.any(), [(float(x[i]), int(got[i]), int(want[i])) for i in np.nonzero(bad)[0][:8]]
    assert pkg.host_fp8_e4m3(np.array([np.nan], np.float32))[0] & 0x7F == 0x7F


def _representable(w, axis):
    """w with every entry replaced by (e4m3 value) x (power-of-two scale of its output column); axis = the output dimension."""
    amax = np.abs(w).max(axis=1 - axis, keepdims=True)
    sc = 2.0 ** np.ceil(np.log2(np.maximum(amax, 1e-30) / 448.0))
    q = torch.from_numpy((w / sc).astype(np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy()
    return (q * sc).astype(np.float32)


@pytest.mark.gpu
def test_fp8_mode_is_exact_on_representable_weights(pkg, small_models, voice, tmp_path):
    from tortoise_cpp_amd import synth_weights as sw
    t = sw.read_ggml(small_models + "/ggml-model.bin")
    path = str(tmp_path / "ggml-model-fp8-exact.bin")
    w = sw.GgmlWriter(path)
    for name, a in t.items():
        if name.endswith(("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")):
            a = _representable(a, axis=1)      # HF Conv1D: [in][out]
        elif name == "inference_model.lm_head.1.weight":
            a = _representable(a, axis=0)      # nn.Linear: [out][in]
        elif name.endswith(("ln_1.weight", "ln_2.weight")) or name == "inference_model.lm_head.0.weight":
            a = np.ones_like(a)                # the LayerNorm gain is folded into the matrix it feeds: keep the matrix representable
        w.add(name, a)
    w.close()
    outs = []
    for mode in (0, 2):
        e = pkg.Engine(0)
        e.set_option("ar_weights", mode)
        e.load(ar=path)
        B = 5
        e.ar_begin(DEFAULT_TOKENS, voice, B, 8)
        lg = [e.ar_prefill()]
        rs = np.random.RandomState(2)
        for i in range(6):
            lg.append(e.ar_step(rs.randint(0, 8192, B).astype(np.int32), i))
        outs.append(np.stack(lg))
        e.close()
    assert (outs[0][0] == outs[1][0]).all()  # the prompt pass does not use the fp8 slabs
    err = rel_err(outs[1][1:], outs[0][1:])
    print("fp8 mode vs f32 mode on exactly representable weights: logits rel err %.1e" % err)
    assert err < 1e-5


@pytest.mark.gpu
def test_fp8_decode_weights_option(pkg, oracle, small_models, voice):
    eng = pkg.Engine(0)
    eng.set_option("ar_weights", 2)
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
    print("fp8 decode weights: logits rel err per step", ["%.1e" % e for e in errs])
    assert max(errs) < 0.12 and max(errs) > 1e-3  # e4m3: 2^-4 relative per weight, averaged over K = 1024 .. 4096 products
    codes = rs.randint(0, 8192, (B, 502)).astype(np.int32)
    codes[:, 0] = 8192
    assert rel_err(eng.ar_latents(codes, 12), ar.latents(codes, 12)) < 1e-4
    # another reduced-precision mode than the one the slabs were packed for is an error, not a silent run on the wrong slabs
    eng.set_option("ar_weights", 1)
    eng.ar_begin(toks, voice, 2, 4)
    eng.ar_prefill()
    with pytest.raises(pkg.TtsError, match="before tts_load_ar"):
        eng.ar_step(np.array([1, 2], np.int32), 0)
    with pytest.raises(pkg.TtsError):
        eng.set_option("ar_weights", 3)
    eng.close()
