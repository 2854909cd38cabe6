"""tools/convert_weights.py (the PyTorch -> reference-container export the reference's README promises but does not ship).
Round trip without the trained checkpoints: synthetic weights in the reference format are turned back into PyTorch-style state dicts
the way tortoise-tts stores them — k = 1 convolutions as 3-D Conv1d weights, UnivNet convolutions as weight_norm (weight_g, weight_v)
pairs under 'model_g', the GPT-2 stack under its training-time names (gpt.h.N, final_norm, mel_head) plus unrelated keys — saved with
torch.save, converted by the CLI, and the result must be the original container: same names, same shapes, same values (the fused
weight_norm tensors to f32 round-off), accepted by the reference-format reader of the oracle."""
import os
import subprocess
import sys

import numpy as np
import torch

from conftest import ROOT


def _to_state_dicts(src, tmp):
    import tortoise_cpp_amd_loader
    tortoise_cpp_amd_loader.load()
    from tortoise_cpp_amd import synth_weights as sw
    ar = sw.read_ggml(os.path.join(src, "ggml-model.bin"))
    sd = {}
    for k, v in ar.items():  # training-time naming
        k2 = (k.replace("inference_model.transformer.h.", "gpt.h.").replace("inference_model.transformer.ln_f.", "gpt.ln_f.")
               .replace("inference_model.lm_head.0.", "final_norm.").replace("inference_model.lm_head.1.", "mel_head."))
        sd[k2] = torch.from_numpy(v)
    sd["text_head.weight"] = torch.zeros(256, 1024)  # keys the reference never reads
    sd["conditioning_encoder.init.weight"] = torch.zeros(4, 4)
    torch.save(sd, os.path.join(tmp, "autoregressive.pth"))
    df = sw.read_ggml(os.path.join(src, "ggml-diffusion-model.bin"))
    sd = {}
    for k, v in df.items():
        if k == "diffusion_conditioning_latent":
            torch.save(torch.from_numpy(v.reshape(1, 2048)), os.path.join(tmp, "voice_diff.pth"))
            continue
        t = torch.from_numpy(v)
        if k.endswith((".qkv.weight", ".proj_out.weight", ".in_layers.2.weight")) or k == "integrating_conv.weight":
            t = t[:, :, None]  # nn.Conv1d(kernel_size=1)
        if k == "unconditioned_embedding":
            t = t.reshape(1, 1024, 1)
        sd[k] = t
    sd["contextual_embedder.init.weight"] = torch.zeros(3, 3)
    torch.save(sd, os.path.join(tmp, "diffusion_decoder.pth"))
    vc = sw.read_ggml(os.path.join(src, "ggml-vocoder-model.bin"))
    sd = {}
    rs = np.random.RandomState(0)
    for k, v in vc.items():
        t = torch.from_numpy(v)
        if k == "conv_post.1.weight":
            t = t.reshape(1, 32, 7)
        if k.endswith(".weight") and t.ndim == 3:  # weight_norm(dim=0): weight = g * v / ||v||
            scale = torch.from_numpy(rs.uniform(0.5, 2.0, (t.shape[0], 1, 1)).astype(np.float32))
            vv = t * scale
            sd[k[:-len("weight")] + "weight_v"] = vv
            sd[k[:-len("weight")] + "weight_g"] = torch.sqrt((t.double() ** 2).sum(dim=(1, 2), keepdim=True)).float()
        else:
            sd[k] = t
    torch.save({"model_g": sd}, os.path.join(tmp, "vocoder.pth"))
    return ar, df, vc


def test_checkpoint_round_trip(small_models, tmp_path, oracle):
    tmp = str(tmp_path)
    ar, df, vc = _to_state_dicts(small_models, tmp)
    out = os.path.join(tmp, "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--ar", tmp + "/autoregressive.pth", "--diffusion",
                        tmp + "/diffusion_decoder.pth", "--diffusion-conditioning-latent", tmp + "/voice_diff.pth", "--vocoder",
                        tmp + "/vocoder.pth", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    from tortoise_cpp_amd import synth_weights as sw
    for name, want, tol in (("ggml-model.bin", ar, 0.0), ("ggml-diffusion-model.bin", df, 0.0), ("ggml-vocoder-model.bin", vc, 1e-6)):
        got = sw.read_ggml(os.path.join(out, name))
        assert sorted(got) == sorted(want), (name, set(got) ^ set(want))
        for k in want:
            assert got[k].shape == want[k].shape, (name, k, got[k].shape, want[k].shape)
            if tol == 0.0:
                assert (got[k] == want[k]).all(), (name, k)
            else:
                assert np.abs(got[k] - want[k]).max() <= tol * max(1.0, np.abs(want[k]).max()), (name, k)
        oracle.Model(os.path.join(out, name))  # the reference-format reader takes it
    # byte-identical containers for the two files that need no arithmetic
    for name in ("ggml-model.bin", "ggml-diffusion-model.bin"):
        a, b = open(os.path.join(out, name), "rb").read(), open(os.path.join(small_models, name), "rb").read()
        assert len(a) == len(b) and sorted(a[:64]) == sorted(b[:64])
    # the oracle runs the converted vocoder and matches the original weights' output (fused weight_norm within round-off)
    rs = np.random.RandomState(1)
    mel = np.clip(rs.randn(100, 7) * 0.5, -1, 1).astype(np.float32)
    nz = rs.randn(64, 17).astype(np.float32)
    a0 = oracle.Vocoder(oracle.Model(os.path.join(small_models, "ggml-vocoder-model.bin"))).run(mel, noise=nz)
    a1 = oracle.Vocoder(oracle.Model(os.path.join(out, "ggml-vocoder-model.bin"))).run(mel, noise=nz)
    assert np.abs(a0 - a1).max() <= 1e-3 * np.abs(a0).max()


def test_clvp_checkpoint_round_trip(tmp_path, oracle):
    """--clvp: a state dict shaped like upstream's clvp2.pth (0-dim temperature, rotary inv_freq buffers) -> the CLVP container; the
    oracle scores identically from the original and the converted file; a checkpoint with foreign keys is refused."""
    import tortoise_cpp_amd_loader
    tortoise_cpp_amd_loader.load()
    from tortoise_cpp_amd import synth_weights as sw
    src = str(tmp_path / "src.bin")
    sw.write_clvp(src, depth=2, seed=11)
    sd = {k: torch.from_numpy(v) for k, v in sw.read_ggml(src).items()}
    sd["temperature"] = sd["temperature"].reshape(())
    for enc in sw.CLVP_ENCODERS:
        sd[enc + ".transformer.attn_layers.rotary_pos_emb.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    torch.save(sd, str(tmp_path / "clvp2.pth"))
    out = str(tmp_path / "out")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--clvp", str(tmp_path / "clvp2.pth"), "--out", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "2 encoder layers" in r.stdout, r.stdout + r.stderr
    rs = np.random.RandomState(0)
    text, sp = rs.randint(0, 256, 12), [rs.randint(0, 8192, n) for n in (9, 21)]
    a = oracle.Clvp(oracle.Model(src)).score(text, sp)
    b = oracle.Clvp(oracle.Model(out + "/ggml-clvp-model.bin")).score(text, sp)
    assert (a == b).all()
    sd["text_pos_emb.weight"] = torch.zeros(4, 4)  # the non-xformers CLVP flavour
    torch.save(sd, str(tmp_path / "other.pth"))
    cmd[cmd.index("--clvp") + 1] = str(tmp_path / "other.pth")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "unexpected tensor" in r.stdout + r.stderr


def test_conditioning_encoder_from_the_ar_checkpoint(small_models, tmp_path, oracle):
    """--conditioning-encoder: the conditioning_encoder.* tensors that ride in upstream's autoregressive.pth (and that the reference's
    ggml-model.bin leaves out) become ggml-conditioning-model.bin; the oracle computes the same voice latent from the converted file."""
    import tortoise_cpp_amd_loader
    tortoise_cpp_amd_loader.load()
    from tortoise_cpp_amd import synth_weights as sw
    src = str(tmp_path / "venc.bin")
    sw.write_voice_encoder(src, blocks=2, seed=3)
    ar = sw.read_ggml(os.path.join(small_models, "ggml-model.bin"))
    sd = {(k.replace("inference_model.transformer.h.", "gpt.h.").replace("inference_model.transformer.ln_f.", "gpt.ln_f.")
            .replace("inference_model.lm_head.0.", "final_norm.").replace("inference_model.lm_head.1.", "mel_head.")): torch.from_numpy(v) for k, v in ar.items()}
    sd.update({k: torch.from_numpy(v) for k, v in sw.read_ggml(src).items()})
    torch.save(sd, str(tmp_path / "autoregressive.pth"))
    out = str(tmp_path / "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--ar", str(tmp_path / "autoregressive.pth"),
                        "--conditioning-encoder", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "2 attention blocks" in r.stdout, r.stdout + r.stderr
    got = sw.read_ggml(out + "/ggml-model.bin")
    assert sorted(got) == sorted(ar)  # the AR container is unchanged by the extra tensors
    mels = [np.random.RandomState(1).randn(80, 21).astype(np.float32)]
    a = oracle.VoiceEncoder(oracle.Model(src)).latent(mels)
    b = oracle.VoiceEncoder(oracle.Model(out + "/ggml-conditioning-model.bin")).latent(mels)
    assert (a == b).all()


def test_diffusion_conditioning_encoder_from_the_diffusion_checkpoint(tmp_path, oracle):
    import tortoise_cpp_amd_loader
    tortoise_cpp_amd_loader.load()
    from tortoise_cpp_amd import synth_weights as sw
    src = str(tmp_path / "dcond.bin")
    sw.write_diffusion_conditioning_encoder(src, blocks=1, seed=4)
    sd = {k: torch.from_numpy(v) for k, v in sw.read_ggml(src).items()}
    sd["layers.0.resblk.in_layers.2.weight"] = torch.zeros(4, 4, 1)  # the rest of the checkpoint is ignored
    torch.save(sd, str(tmp_path / "diffusion_decoder.pth"))
    out = str(tmp_path / "out")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--diffusion-conditioning-encoder",
                        str(tmp_path / "diffusion_decoder.pth"), "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "1 attention blocks" in r.stdout, r.stdout + r.stderr
    mels = [np.random.RandomState(1).randn(100, 37).astype(np.float32)]
    a = oracle.DiffusionConditioning(oracle.Model(src)).latent(mels)
    b = oracle.DiffusionConditioning(oracle.Model(out + "/ggml-diffusion-conditioning-model.bin")).latent(mels)
    assert (a == b).all()


def test_expected_tensor_lists_match_the_synthetic_writers(tmp_path):
    """tools/convert_weights.py --list-expected is the contract a real checkpoint is checked against; the synthetic writers
    (tortoise_cpp_amd/synth_weights.py) are what every parity test runs on. They were written independently (loader contracts vs generator):
    names and container shapes must agree tensor for tensor, for all six containers, so that neither drifts away from the loaders unnoticed."""
    import importlib.util
    import tortoise_cpp_amd_loader
    tortoise_cpp_amd_loader.load()
    from tortoise_cpp_amd import synth_weights as sw
    spec = importlib.util.spec_from_file_location("convert_weights", os.path.join(ROOT, "tools", "convert_weights.py"))
    cw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cw)
    d = str(tmp_path)
    sw.write_all(d, ar_layers=2, diff_main=2, diff_tail=1, diff_integ=1, diff_lc=2, seed=5)
    sw.write_clvp(d + "/clvp.bin", depth=2, seed=6)
    sw.write_voice_encoder(d + "/venc.bin", blocks=2, seed=7)
    sw.write_diffusion_conditioning_encoder(d + "/dcond.bin", blocks=2, seed=8)
    cases = (("ar", "ggml-model.bin", dict(layers=2)), ("diffusion", "ggml-diffusion-model.bin", dict(lc=2, integ=1, main=2, tail=1)),
             ("vocoder", "ggml-vocoder-model.bin", {}), ("clvp", "clvp.bin", dict(depth=2)), ("conditioning-encoder", "venc.bin", dict(blocks=2)),
             ("diffusion-conditioning-encoder", "dcond.bin", dict(blocks=2)))
    for kind, fn, arch in cases:
        have = {k: tuple(v.shape) for k, v in sw.read_ggml(os.path.join(d, fn)).items()}
        exp = {k: cw.container_shape(kind, k, shp) for k, shp in cw.expected_tensors(kind, **arch).items()}
        if kind == "diffusion":
            exp["diffusion_conditioning_latent"] = (1, 2048)  # from --diffusion-conditioning-latent, not from the checkpoint
        assert sorted(have) == sorted(exp), (kind, sorted(set(have) ^ set(exp))[:8])
        for k in exp:
            assert have[k] == exp[k], (kind, k, have[k], exp[k])
        assert not cw.check_against(kind, {k: np.zeros(s, np.float32) for k, s in have.items()}), kind
    # the upstream architecture: the counts a maintainer should see in --list-expected
    assert len(cw.expected_tensors("ar")) == 4 + 30 * 12 + 6 and len(cw.expected_tensors("clvp")) == 5 + 2 * (20 * 11 + 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--list-expected", "vocoder"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "res_stack.2.kernel_predictor.kernel_conv.weight" in r.stdout and "24576x64x3" in r.stdout


def test_converter_reports_a_name_diff_instead_of_a_keyerror(small_models, tmp_path):
    """a checkpoint whose tensor names differ from the expected ones (here: one renamed, one missing) is refused with the list of what is
    missing, before anything is written"""
    ar, df, vc = _to_state_dicts(small_models, str(tmp_path))
    sd = torch.load(str(tmp_path / "diffusion_decoder.pth"), weights_only=True)
    sd["code_norm_renamed.weight"] = sd.pop("code_norm.weight")
    del sd["layers.0.attn.qkv.bias"]
    torch.save(sd, str(tmp_path / "bad.pth"))
    out = str(tmp_path / "out_bad")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--diffusion", str(tmp_path / "bad.pth"),
                        "--diffusion-conditioning-latent", str(tmp_path / "voice_diff.pth"), "--out", out], capture_output=True, text=True, timeout=300)
    msg = r.stdout + r.stderr
    assert r.returncode != 0 and "code_norm.weight" in msg and "layers.0.attn.qkv.bias" in msg and "KeyError" not in msg
    assert not os.path.exists(os.path.join(out, "ggml-diffusion-model.bin"))
