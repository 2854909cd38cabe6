"""Voice-conditioning encoder: 80-band mel of the reference clips -> the 1024-float latent a --voice file holds.
The reference only reads the finished latent (main.cpp:5179-5184; README.md:54-72 is an offline PyTorch recipe), so, as for CLVP, the chain is
  torch restatement of upstream tortoise-tts (tests/torch_ref.py: TorchVoiceEncoder, f64)  ==  numpy oracle (oracle.VoiceEncoder)  [CPU]
  numpy oracle  ~  HIP engine (tts_load_voice_encoder / tts_voice_latent) on synthetic weights                                   [GPU]
— parity UNPINNED against upstream weights. The audio front-end (STFT, mel filterbank, normalisation) is the caller's."""
import os

import numpy as np
import pytest
import torch

import torch_ref as TR


@pytest.fixture(scope="session")
def venc_models(pkg):
    d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "venc")
    os.makedirs(d, exist_ok=True)
    from tortoise_cpp_amd import synth_weights as sw
    out = {}
    for name, blocks in (("small", 2), ("full", 6)):
        p = os.path.join(d, "ggml-conditioning-model-%s.bin" % name)
        if not os.path.exists(p + ".done"):
            sw.write_voice_encoder(p, blocks=blocks, seed=50 + blocks)
            open(p + ".done", "w").write("ok")
        out[name] = p
    return out


def _mels(seed, lens):
    rs = np.random.RandomState(seed)
    return [(rs.randn(80, n) * 1.5 - 2.0).astype(np.float32) for n in lens]  # log-mel-like range


def test_voice_encoder_oracle_vs_torch(oracle, venc_models):
    mels = _mels(0, (50, 33))
    o = oracle.VoiceEncoder(oracle.Model(venc_models["small"])).latent(mels)
    t64 = TR.TorchVoiceEncoder(venc_models["small"], torch.float64).latent(mels)
    t32 = TR.TorchVoiceEncoder(venc_models["small"]).latent(mels)
    scale = np.abs(t64).max()
    assert np.abs(o - t64).max() < 2e-5 * scale and np.abs(t32 - t64).max() < 2e-5 * scale
    assert scale > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("which,lens", [("small", (50, 33, 7)), ("full", (517, 301))])
def test_voice_encoder_engine_vs_oracle(pkg, oracle, venc_models, which, lens):
    """fp16-operand GEMMs (f32 accumulate) against the f32 oracle: 1e-3 of the latent's range (north star; measured 3-4e-4); deterministic; a clip's contribution does not
    depend on the other clips (mean of single-clip latents == the multi-clip latent)."""
    mels = _mels(4, lens)
    e = pkg.Engine(0)
    e.load_voice_encoder(venc_models[which])
    got = e.voice_latent(mels)
    want = oracle.VoiceEncoder(oracle.Model(venc_models[which])).latent(mels)
    err = float(np.abs(got - want).max() / np.abs(want).max())
    print("voice encoder %s: max err %.1e of range %.2f" % (which, err, np.abs(want).max()))
    assert np.isfinite(got).all() and err < 1e-3
    assert (got == e.voice_latent(mels)).all()
    singles = np.mean([e.voice_latent([m]).astype(np.float64) for m in mels], axis=0)
    assert np.abs(singles - got).max() < 1e-5 * np.abs(want).max()
    e.close()


@pytest.mark.gpu
def test_voice_encoder_errors(pkg, venc_models, small_models):
    e = pkg.Engine(0)
    with pytest.raises(pkg.TtsError, match="tts_load_voice_encoder not called"):
        e.voice_latent(_mels(0, (5,)))
    with pytest.raises(pkg.TtsError, match="not a conditioning-encoder file"):
        e.load_voice_encoder(small_models + "/ggml-model.bin")
    e.load_voice_encoder(venc_models["small"])
    assert e.voice_latent(_mels(0, (1,))).shape == (1024,)  # a one-frame clip is legal
    e.close()


# ---- the other voice latent: diffusion conditioning (100-band mel -> the 2048 floats the reference keeps as a weight) -----------------
@pytest.fixture(scope="session")
def dcond_models(pkg):
    d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "dcond")
    os.makedirs(d, exist_ok=True)
    from tortoise_cpp_amd import synth_weights as sw
    out = {}
    for name, blocks in (("small", 2), ("full", 5)):
        p = os.path.join(d, "ggml-diffusion-conditioning-model-%s.bin" % name)
        if not os.path.exists(p + ".done"):
            sw.write_diffusion_conditioning_encoder(p, blocks=blocks, seed=70 + blocks)
            open(p + ".done", "w").write("ok")
        out[name] = p
    return out


def _mels100(seed, lens):
    rs = np.random.RandomState(seed)
    # the scale upstream feeds this encoder: UN-normalised log(clamp(mel, 1e-5)) in [-11.51, 2.31] (tts_host_mel_diffusion100(..., normalize = 0))
    return [np.clip(rs.randn(100, n) * 2.5 - 4.5, -11.512925, 2.3143387).astype(np.float32) for n in lens]


def test_diffusion_conditioning_oracle_vs_torch(oracle, dcond_models):
    mels = _mels100(0, (301, 120, 7))  # odd / even lengths through both stride-2 convolutions, a clip shorter than the receptive field
    o = oracle.DiffusionConditioning(oracle.Model(dcond_models["small"])).latent(mels)
    t64 = TR.TorchDiffusionConditioning(dcond_models["small"], torch.float64).latent(mels)
    t32 = TR.TorchDiffusionConditioning(dcond_models["small"]).latent(mels)
    scale = np.abs(t64).max()
    assert np.abs(o - t64).max() < 2e-5 * scale and np.abs(t32 - t64).max() < 2e-5 * scale and scale > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("which,lens", [("small", (301, 120, 7, 1)), ("full", (938, 517))])
def test_diffusion_conditioning_engine_vs_oracle(pkg, oracle, dcond_models, which, lens):
    mels = _mels100(5, lens)
    e = pkg.Engine(0)
    e.load_diffusion_conditioning_encoder(dcond_models[which])
    got = e.diffusion_conditioning_latent(mels)
    want = oracle.DiffusionConditioning(oracle.Model(dcond_models[which])).latent(mels)
    err = float(np.abs(got - want).max() / np.abs(want).max())
    print("diffusion conditioning %s: max err %.1e of range %.2f" % (which, err, np.abs(want).max()))
    assert np.isfinite(got).all() and err < 1e-3
    assert (got == e.diffusion_conditioning_latent(mels)).all()
    e.close()


@pytest.mark.gpu
def test_set_diffusion_conditioning_latent(pkg, small_models, tmp_path):
    """tts_set_diffusion_conditioning_latent == a weight file with that latent baked in (the reference's only way, main.cpp:1557-1560)."""
    from tortoise_cpp_amd import synth_weights as sw
    t = sw.read_ggml(small_models + "/ggml-diffusion-model.bin")
    rs = np.random.RandomState(8)
    lat2 = (t["diffusion_conditioning_latent"].reshape(-1) + rs.randn(2048).astype(np.float32) * 0.1).astype(np.float32)
    path = str(tmp_path / "ggml-diffusion-model-other-voice.bin")
    w = sw.GgmlWriter(path)
    for k, v in t.items():
        w.add(k, lat2.reshape(v.shape) if k == "diffusion_conditioning_latent" else v)
    w.close()
    latents = rs.randn(9, 1024).astype(np.float32)
    x_t = rs.randn(100, pkg.Engine.frames(9)).astype(np.float32)
    e = pkg.Engine(0)
    with pytest.raises(pkg.TtsError, match="tts_load_diffusion not called"):
        e.set_diffusion_conditioning_latent(lat2)
    e.load(diffusion=small_models + "/ggml-diffusion-model.bin")
    base = e.diffusion_forward(latents, x_t, 500, False)
    e.set_diffusion_conditioning_latent(lat2)
    swapped = e.diffusion_forward(latents, x_t, 500, False)
    e2 = pkg.Engine(0)
    e2.load(diffusion=path)
    baked = e2.diffusion_forward(latents, x_t, 500, False)
    assert (swapped == baked).all() and not (swapped == base).all()
    e.close(); e2.close()


@pytest.mark.gpu
def test_make_voice_tool_and_cli_voice_flags(pkg, venc_models, dcond_models, small_models, tmp_path):
    """Audio clips -> tools/make_voice.py (host mel front-end + both encoders) -> `tortoise --voice V.bin --diffusion-latent V.diffusion.bin`:
    the files have the right sizes, equal the library calls on the same mel, and the CLI run with them differs from the stock voice's."""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    rs = np.random.RandomState(12)
    clips = []
    for i, (rate, secs) in enumerate(((24000, 1.3), (16000, 0.9))):
        t = np.arange(int(rate * secs)) / rate
        x = (0.3 * np.sin(2 * np.pi * (180 + 40 * i) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rs.randn(len(t))).astype(np.float32)
        p = str(tmp_path / ("clip%d.wav" % i))
        assert pkg.write_wav(p, x, rate) == 0
        clips.append((p, x, rate))
    out = str(tmp_path / "voice")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_voice.py"), "--clips"] + [c[0] for c in clips] +
                       ["--conditioning-model", venc_models["small"], "--diffusion-conditioning-model", dcond_models["small"], "--out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    v, dl = np.fromfile(out + ".bin", np.float32), np.fromfile(out + ".diffusion.bin", np.float32)
    assert v.shape == (1024,) and dl.shape == (2048,) and np.isfinite(v).all() and np.isfinite(dl).all()
    e = pkg.Engine(0)
    e.load_voice_encoder(venc_models["small"])
    from scipy.signal import resample_poly
    first = e.voice_latent([pkg.host_mel_voice80(resample_poly(clips[0][1], 147, 160).astype(np.float32))])  # 24 kHz -> 22.05 kHz
    assert first.shape == (1024,) and not np.allclose(first, v)  # two clips average to something else than the first alone
    e.close()
    d = tmp_path / "models"
    d.mkdir()
    for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
        os.symlink(os.path.join(small_models, f), d / f)
    shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), d / "tokenizer.json")
    base = [exe, "--models", str(d), "--message", "this is a test message.", "--seed", "0", "--codes", "16", "--steps", "4"]
    outs = {}
    for tag, extra in (("stock", ["--voice", os.path.join(ROOT, "models", "mol.bin")]),
                       ("mine", ["--voice", out + ".bin", "--diffusion-latent", out + ".diffusion.bin"]),
                       ("mine_ar_only", ["--voice", out + ".bin"])):
        w = tmp_path / (tag + ".wav")
        r = subprocess.run(base + extra + ["--output", str(w)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[tag] = w.read_bytes()
    assert outs["mine"] != outs["stock"] and outs["mine"] != outs["mine_ar_only"]
    r = subprocess.run(base + ["--voice", out + ".bin", "--diffusion-latent", out + ".bin", "--output", str(tmp_path / "bad.wav")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "2048 floats" in r.stderr  # a 1024-float file is not a diffusion latent
