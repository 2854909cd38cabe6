"""Error behaviour of the C ABI on a device context: malformed weight files (the reference's loaders reject unknown
names and wrong shapes, main.cpp:834-870), calls out of order, arguments out of range. Every failure is a negative
status + tts_last_error text; nothing throws across the ABI, and the context stays usable afterwards."""
import os
import struct

import numpy as np
import pytest

from conftest import DEFAULT_TOKENS

pytestmark = pytest.mark.gpu


def _rewrite(src, dst, edit):
    """Copy a weight file record by record, letting `edit(name, ne, data)` return a replacement (name, ne, data) or None
    to drop the record."""
    with open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read(4))
        while True:
            hdr = f.read(12)
            if len(hdr) < 12:
                break
            n_dims, ln, tt = struct.unpack("<iii", hdr)
            ne = list(struct.unpack("<%di" % n_dims, f.read(4 * n_dims)))
            name = f.read(ln).decode()
            data = f.read(4 * int(np.prod(ne)))
            r = edit(name, ne, data)
            if r is None:
                continue
            name, ne, data = r
            nb = name.encode()
            g.write(struct.pack("<iii", len(ne), len(nb), tt) + struct.pack("<%di" % len(ne), *ne) + nb + data)


def test_malformed_vocoder_files(pkg, small_models, tmp_path):
    good = os.path.join(small_models, "ggml-vocoder-model.bin")
    eng = pkg.Engine(0)
    try:
        with pytest.raises(pkg.TtsError, match="failed to open"):
            eng.load(vocoder=str(tmp_path / "missing.bin"))
        p = str(tmp_path / "magic.bin")
        open(p, "wb").write(b"\x00\x01\x02\x03" + open(good, "rb").read()[4:4096])
        with pytest.raises(pkg.TtsError, match="bad magic"):
            eng.load(vocoder=p)
        p = str(tmp_path / "trunc.bin")
        open(p, "wb").write(open(good, "rb").read()[:100000])
        with pytest.raises(pkg.TtsError, match="truncated"):
            eng.load(vocoder=p)
        p = str(tmp_path / "unknown.bin")
        _rewrite(good, p, lambda n, ne, d: ("conv_pre.weird", ne, d) if n == "conv_pre.bias" else (n, ne, d))
        with pytest.raises(pkg.TtsError, match="conv_pre"):
            eng.load(vocoder=p)
        p = str(tmp_path / "missing_tensor.bin")
        _rewrite(good, p, lambda n, ne, d: None if n == "conv_post.1.bias" else (n, ne, d))
        with pytest.raises(pkg.TtsError, match="conv_post.1.bias"):
            eng.load(vocoder=p)
        p = str(tmp_path / "shape.bin")
        _rewrite(good, p, lambda n, ne, d: (n, [ne[0] // 2], d[:len(d) // 2]) if n == "conv_pre.bias" else (n, ne, d))
        with pytest.raises(pkg.TtsError, match="conv_pre.bias"):
            eng.load(vocoder=p)
        p = str(tmp_path / "huge.bin")  # a shape far larger than the file must not become an allocation
        _rewrite(good, p, lambda n, ne, d: (n, [2 ** 30, 2 ** 30], d) if n == "conv_pre.bias" else (n, ne, d))
        with pytest.raises(pkg.TtsError, match="truncated"):
            eng.load(vocoder=p)
        # the context is still usable
        with pytest.raises(pkg.TtsError, match="not loaded"):
            eng.vocoder([np.zeros((100, 4), np.float32)])
        eng.load(vocoder=good)
        assert eng.vocoder([np.zeros((100, 4), np.float32)], noise=[np.zeros((64, 14), np.float32)])[0].shape == (14 * 256 - 6,)
    finally:
        eng.close()


def test_call_order_and_argument_limits(pkg, small_models, voice):
    eng = pkg.Engine(0)
    try:
        with pytest.raises(pkg.TtsError, match="unknown option"):
            eng.set_option("no_such_option", 1)
        with pytest.raises(pkg.TtsError, match="not loaded"):
            eng.ar_begin(DEFAULT_TOKENS, voice, 1, 8)
        with pytest.raises(pkg.TtsError, match="not loaded"):
            eng.diffusion([np.zeros((4, 1024), np.float32)], n_steps=2, noise_mode=pkg.NOISE_DEVICE)
        eng.load(small_models)
        with pytest.raises(pkg.TtsError):
            eng.B = 1
            eng.ar_prefill()  # before tts_ar_begin
        with pytest.raises(pkg.TtsError, match="404"):
            eng.ar_begin(np.zeros(405, np.int32), voice, 1, 8)
        with pytest.raises(pkg.TtsError, match="out of range"):
            eng.ar_begin(np.array([255, 256, 0], np.int32), voice, 1, 8)
        with pytest.raises(pkg.TtsError, match="exceeds"):
            eng.ar_begin(DEFAULT_TOKENS, voice, 1, 700)
        eng.ar_begin(DEFAULT_TOKENS, voice, 2, 4)
        eng.ar_prefill()
        with pytest.raises(pkg.TtsError, match="out of range"):
            eng.ar_step(np.array([5, 8194], np.int32), 0)
        with pytest.raises(pkg.TtsError, match="beyond the KV cache"):
            eng.ar_step(np.array([5, 6], np.int32), 5)
        assert np.isfinite(eng.ar_step(np.array([5, 6], np.int32), 0)).all()  # still usable after the refusals
        bad = np.full((1, 502), 9000, np.int32)
        with pytest.raises(pkg.TtsError, match="out of range"):
            eng.ar_latents(bad, 8)
        with pytest.raises(pkg.TtsError, match="out of range"):
            eng.diffusion([np.zeros((501, 1024), np.float32)], n_steps=2, noise_mode=pkg.NOISE_DEVICE)
        with pytest.raises(pkg.TtsError, match="bad argument"):
            eng.diffusion([np.zeros((4, 1024), np.float32)], n_steps=1, noise_mode=pkg.NOISE_DEVICE)
        with pytest.raises(pkg.TtsError, match="exceeds the 500"):
            eng.autoregressive(DEFAULT_TOKENS, voice, 1, 501)
    finally:
        eng.close()


def test_ar_weight_beyond_the_split_precision_range_is_rejected(pkg, small_models, tmp_path):
    """The AR stage holds 64 W as an fp16 hi | lo pair (ar.hip: W16_SCALE): |W| >= 937 would become an fp16 infinity inside the MFMA operands. Such a file
    fails at load with the tensor's name (round 6; weights of a trained GPT-2 are three orders of magnitude below), one at |W| = 100 loads."""
    good = os.path.join(small_models, "ggml-model.bin")
    eng = pkg.Engine(0)

    def scaled(target, peak):
        def edit(n, ne, d):
            if n == target:
                a = np.frombuffer(d, np.float32).copy()
                a[7] = peak
                return n, ne, a.tobytes()
            return n, ne, d
        return edit
    try:
        name = "inference_model.transformer.h.1.mlp.c_proj.weight"
        p = str(tmp_path / "w100.bin")
        _rewrite(good, p, scaled(name, 100.0))
        eng.load(ar=p)
        p = str(tmp_path / "w2000.bin")
        _rewrite(good, p, scaled(name, 2000.0))
        with pytest.raises(pkg.TtsError, match="h.1.mlp.c_proj.weight.*split-precision"):
            eng.load(ar=p)
        eng.load(ar=good)  # the context stays usable
    finally:
        eng.close()
