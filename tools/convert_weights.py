#!/usr/bin/env python3
"""PyTorch checkpoints of tortoise-tts -> the reference's weight container.

The reference's README promises export scripts that are not in its repository (/root/reference/README.md:34, 77); its loaders
(main.cpp:682-792, 1244-1536, 1808-1923) define the contract: legacy-ggml container (magic 0x67676d6c, F32 records,
no alignment), tensor names = the PyTorch state_dict keys of the inference modules, PyTorch dim order reversed into ne[], k = 1
convolutions stored squeezed to 2-D. This tool produces exactly that from

  autoregressive.pth      UnifiedVoice state dict. Either the `inference_model.*` keys (GPT2InferenceModel, present after
                          post_init_gpt2_config) or the training-time names (gpt.h.N..., gpt.ln_f, final_norm, mel_head), which are mapped.
  diffusion_decoder.pth   DiffusionTts state dict (+ --diffusion-conditioning-latent: the [1, 2048] diffusion conditioning latent of the
                          voice, which the reference bakes into the file as `diffusion_conditioning_latent`).
  vocoder.pth             UnivNet generator state dict (optionally under 'model_g'); weight_norm parametrisations (weight_g / weight_v) are fused.

    python tools/convert_weights.py --ar autoregressive.pth --diffusion diffusion_decoder.pth --diffusion-conditioning-latent voice_diff.pth \\
                                    --vocoder vocoder.pth --out models/

The trained checkpoints are not available offline: tests/test_convert_weights.py round-trips synthetic weights through PyTorch-style state
dicts (3-D k=1 convs, weight_norm pairs, training-time AR names) and requires the converter to reproduce the original container.
"""
import argparse
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402

tortoise_cpp_amd_loader.load()
from tortoise_cpp_amd.synth_weights import GgmlWriter  # noqa: E402


def _np(t):
    return t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)


def fuse_weight_norm(sd):
    """weight = g * v / ||v|| (norm over every dim but 0: torch.nn.utils.weight_norm's default dim=0)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[:-len(".weight_g")]
            g, w = _np(v), _np(sd[base + ".weight_v"])
            norm = np.sqrt((w.astype(np.float64) ** 2).sum(axis=tuple(range(1, w.ndim)), keepdims=True))
            out[base + ".weight"] = (g.astype(np.float64) * w / norm).astype(np.float32)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = _np(v)
    return out


# ---- autoregressive (main.cpp:682-792) --------------------------------------------------------------------------------------------
AR_GLOBAL = ["text_embedding.weight", "text_pos_embedding.emb.weight", "mel_embedding.weight", "mel_pos_embedding.emb.weight"]
AR_TRAIN_TO_INFER = [(r"^gpt\.h\.", "inference_model.transformer.h."), (r"^gpt\.ln_f\.", "inference_model.transformer.ln_f."),
                     (r"^final_norm\.", "inference_model.lm_head.0."), (r"^mel_head\.", "inference_model.lm_head.1.")]
AR_LAYER = ["ln_1.weight", "ln_1.bias", "attn.c_attn.weight", "attn.c_attn.bias", "attn.c_proj.weight", "attn.c_proj.bias", "ln_2.weight",
            "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias"]


def convert_ar(sd, path):
    sd = {k: _np(v) for k, v in sd.items()}
    if not any(k.startswith("inference_model.") for k in sd):
        for k in list(sd):
            for pat, rep in AR_TRAIN_TO_INFER:
                if re.match(pat, k):
                    sd[re.sub(pat, rep, k)] = sd[k]
    w = GgmlWriter(path)
    for k in AR_GLOBAL:
        w.add(k, sd[k])
    n = 0
    while "inference_model.transformer.h.%d.ln_1.weight" % n in sd:
        for t in AR_LAYER:
            w.add("inference_model.transformer.h.%d.%s" % (n, t), sd["inference_model.transformer.h.%d.%s" % (n, t)])
        n += 1
    for k in ("inference_model.transformer.ln_f.weight", "inference_model.transformer.ln_f.bias", "inference_model.lm_head.0.weight",
              "inference_model.lm_head.0.bias", "inference_model.lm_head.1.weight", "inference_model.lm_head.1.bias"):
        w.add(k, sd[k])
    w.close()
    return n


# ---- diffusion (main.cpp:1244-1536) -----------------------------------------------------------------------------------------------
def _attn_names(p):
    return [p + s for s in (".norm.weight", ".norm.bias", ".qkv.weight", ".qkv.bias", ".proj_out.weight", ".proj_out.bias",
                            ".relative_pos_embeddings.relative_attention_bias.weight")]


def _res_names(p):
    return [p + s for s in (".in_layers.0.weight", ".in_layers.0.bias", ".in_layers.2.weight", ".in_layers.2.bias", ".emb_layers.1.weight",
                            ".emb_layers.1.bias", ".out_layers.0.weight", ".out_layers.0.bias", ".out_layers.3.weight", ".out_layers.3.bias")]


SQUEEZE_K1 = re.compile(r"(\.qkv\.weight|\.proj_out\.weight|\.in_layers\.2\.weight|^integrating_conv\.weight)$")


def convert_diffusion(sd, cond_latent, path):
    sd = {k: _np(v) for k, v in sd.items()}
    names = ["latent_conditioner.0.weight", "latent_conditioner.0.bias"]
    n_lc = 0
    while "latent_conditioner.%d.norm.weight" % (n_lc + 1) in sd:
        n_lc += 1
        names += _attn_names("latent_conditioner.%d" % n_lc)
    names += ["code_norm.weight", "code_norm.bias", "time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias"]
    n_integ = 0
    while "conditioning_timestep_integrator.%d.resblk.in_layers.0.weight" % n_integ in sd:
        names += _res_names("conditioning_timestep_integrator.%d.resblk" % n_integ) + _attn_names("conditioning_timestep_integrator.%d.attn" % n_integ)
        n_integ += 1
    names += ["inp_block.weight", "inp_block.bias", "integrating_conv.weight", "integrating_conv.bias"]
    n_main = 0
    while "layers.%d.resblk.in_layers.0.weight" % n_main in sd:
        names += _res_names("layers.%d.resblk" % n_main) + _attn_names("layers.%d.attn" % n_main)
        n_main += 1
    n_tail = 0
    while "layers.%d.in_layers.0.weight" % (n_main + n_tail) in sd:
        names += _res_names("layers.%d" % (n_main + n_tail))
        n_tail += 1
    names += ["out.0.weight", "out.0.bias", "out.2.weight", "out.2.bias"]
    w = GgmlWriter(path)
    w.add("diffusion_conditioning_latent", _np(cond_latent).reshape(1, 2048))
    for k in names:
        t = sd[k]
        if SQUEEZE_K1.search(k) and t.ndim == 3:  # k = 1 convolutions are stored 2-D (main.cpp:1260, 1308, 1379)
            assert t.shape[2] == 1, (k, t.shape)
            t = t[:, :, 0]
        w.add(k, t)
    w.add("unconditioned_embedding", sd["unconditioned_embedding"].reshape(-1))
    w.close()
    return n_lc, n_integ, n_main, n_tail


# ---- vocoder (main.cpp:1808-1923) -------------------------------------------------------------------------------------------------
def convert_vocoder(sd, path):
    if "model_g" in sd:
        sd = sd["model_g"]
    sd = fuse_weight_norm(sd)
    w = GgmlWriter(path)
    for k in ("conv_pre.weight", "conv_pre.bias"):
        w.add(k, sd[k])
    i = 0
    while "res_stack.%d.convt_pre.1.weight" % i in sd:
        p = "res_stack.%d." % i
        kp = p + "kernel_predictor."
        names = [kp + "input_conv.0.weight", kp + "input_conv.0.bias"]
        c = 0
        while kp + "residual_convs.%d.1.weight" % c in sd:
            for j in (1, 3):
                names += [kp + "residual_convs.%d.%d.weight" % (c, j), kp + "residual_convs.%d.%d.bias" % (c, j)]
            c += 1
        names += [kp + "kernel_conv.weight", kp + "kernel_conv.bias", kp + "bias_conv.weight", kp + "bias_conv.bias", p + "convt_pre.1.weight",
                  p + "convt_pre.1.bias"]
        c = 0
        while p + "conv_blocks.%d.1.weight" % c in sd:
            names += [p + "conv_blocks.%d.1.weight" % c, p + "conv_blocks.%d.1.bias" % c]
            c += 1
        for k in names:
            w.add(k, sd[k])
        i += 1
    post = sd["conv_post.1.weight"]
    w.add("conv_post.1.weight", post.reshape(post.shape[-2], post.shape[-1]))  # [1, 32, 7] -> [32, 7] (main.cpp:1919-1923)
    w.add("conv_post.1.bias", sd["conv_post.1.bias"])
    w.close()
    return i


def convert_conditioning_encoder(sd, path):
    """ggml-conditioning-model.bin for tts_load_voice_encoder: the `conditioning_encoder.*` tensors of upstream tortoise-tts' autoregressive.pth
    (the tensors convert_ar leaves out: the reference reads the finished voice latent from a file), under their state-dict names."""
    w = GgmlWriter(path)
    n = 0
    for k in sorted(sd):
        if k.startswith("conditioning_encoder."):
            if ".relative_pos_embeddings." in k:
                raise SystemExit("convert_conditioning_encoder: '%s' — an encoder with relative position embeddings is not the UnifiedVoice one" % k)
            w.add(k, _np(sd[k]))
            n += k.endswith(".norm.weight")
    w.close()
    if n == 0:
        raise SystemExit("convert_conditioning_encoder: no conditioning_encoder.* tensors in the checkpoint")
    return n


def convert_diffusion_conditioning_encoder(sd, path):
    """ggml-diffusion-conditioning-model.bin for tts_load_diffusion_conditioning_encoder: the `contextual_embedder.*` tensors of upstream
    tortoise-tts' diffusion_decoder.pth (left out of ggml-diffusion-model.bin, which carries one voice's finished latent instead)."""
    w = GgmlWriter(path)
    n = 0
    for k in sorted(sd):
        if k.startswith("contextual_embedder."):
            w.add(k, _np(sd[k]))
            n += k.endswith(".norm.weight")
    w.close()
    if n == 0:
        raise SystemExit("convert_diffusion_conditioning_encoder: no contextual_embedder.* tensors in the checkpoint")
    return n


def convert_clvp(sd, path):
    """ggml-clvp-model.bin for tts_load_clvp: upstream tortoise-tts CLVP (clvp2.pth, use_xformers=True), tensors under their state-dict names.
    Kept: embeddings, latent projections, temperature (0-dim -> [1]), every `*.attn_layers.layers.N.{0.g, 1.*}` parameter and the final norms;
    the rotary inv_freq buffers are recomputed by the loader (kept if present); anything else (e.g. positional embeddings of the non-xformers
    variant) is an error, so that a checkpoint of the other CLVP flavour is not silently mis-read."""
    w = GgmlWriter(path)
    depth = 0
    for k in sorted(sd):
        v = _np(sd[k])
        known = (k in ("text_emb.weight", "speech_emb.weight", "to_text_latent.weight", "to_speech_latent.weight", "temperature")
                 or ".attn_layers.layers." in k or k.endswith((".transformer.norm.weight", ".transformer.norm.bias", "rotary_pos_emb.inv_freq")))
        if not known:
            raise SystemExit("convert_clvp: unexpected tensor '%s' (not the use_xformers=True CLVP of tortoise-tts?)" % k)
        if k == "temperature":
            v = v.reshape(1)
        if k.startswith("text_transformer.") and k.endswith(".0.g"):
            depth += 1
        w.add(k, v)
    w.close()
    return depth // 2


# ---- what each checkpoint is expected to hold (--list-expected) ---------------------------------------------------------------------
# name -> PyTorch shape as tortoise-tts stores it, for the upstream architecture (or the one given). Written from the loaders' contracts
# (main.cpp:682-792, 1244-1536, 1808-1923; csrc/extras.hip for the three containers the reference does not have), NOT from a checkpoint:
# the trained files are not obtainable offline, so the first person who has them gets a name / shape diff instead of a silent mismatch
# (check_against below; tests/test_convert_weights.py requires this list and the synthetic writers to agree).
def expected_tensors(kind, **arch):
    D = 1024
    out = {}

    def attn(p, d, rel=True, conv=True):
        out[p + ".norm.weight"] = (d,); out[p + ".norm.bias"] = (d,)
        out[p + ".qkv.weight"] = (3 * d, d, 1) if conv else (3 * d, d); out[p + ".qkv.bias"] = (3 * d,)
        out[p + ".proj_out.weight"] = (d, d, 1) if conv else (d, d); out[p + ".proj_out.bias"] = (d,)
        if rel:
            out[p + ".relative_pos_embeddings.relative_attention_bias.weight"] = (32, 16)

    def res(p):
        out[p + ".in_layers.0.weight"] = (D,); out[p + ".in_layers.0.bias"] = (D,)
        out[p + ".in_layers.2.weight"] = (D, D, 1); out[p + ".in_layers.2.bias"] = (D,)
        out[p + ".emb_layers.1.weight"] = (2 * D, D); out[p + ".emb_layers.1.bias"] = (2 * D,)
        out[p + ".out_layers.0.weight"] = (D,); out[p + ".out_layers.0.bias"] = (D,)
        out[p + ".out_layers.3.weight"] = (D, D, 3); out[p + ".out_layers.3.bias"] = (D,)

    if kind == "ar":  # autoregressive.pth, inference names (the training-time names gpt.h.N / gpt.ln_f / final_norm / mel_head are mapped onto them)
        out.update({"text_embedding.weight": (256, D), "text_pos_embedding.emb.weight": (404, D), "mel_embedding.weight": (8194, D),
                    "mel_pos_embedding.emb.weight": (608, D)})
        for i in range(arch.get("layers", 30)):
            p = "inference_model.transformer.h.%d." % i
            out.update({p + "ln_1.weight": (D,), p + "ln_1.bias": (D,), p + "attn.c_attn.weight": (D, 3 * D), p + "attn.c_attn.bias": (3 * D,),
                        p + "attn.c_proj.weight": (D, D), p + "attn.c_proj.bias": (D,), p + "ln_2.weight": (D,), p + "ln_2.bias": (D,),
                        p + "mlp.c_fc.weight": (D, 4 * D), p + "mlp.c_fc.bias": (4 * D,), p + "mlp.c_proj.weight": (4 * D, D), p + "mlp.c_proj.bias": (D,)})
        out.update({"inference_model.transformer.ln_f.weight": (D,), "inference_model.transformer.ln_f.bias": (D,), "inference_model.lm_head.0.weight": (D,),
                    "inference_model.lm_head.0.bias": (D,), "inference_model.lm_head.1.weight": (8194, D), "inference_model.lm_head.1.bias": (8194,)})
    elif kind == "diffusion":  # diffusion_decoder.pth (+ the voice's latent from --diffusion-conditioning-latent)
        out["latent_conditioner.0.weight"] = (D, D, 3); out["latent_conditioner.0.bias"] = (D,)
        for i in range(1, 1 + arch.get("lc", 4)):
            attn("latent_conditioner.%d" % i, D)
        out.update({"code_norm.weight": (D,), "code_norm.bias": (D,), "time_embed.0.weight": (D, D), "time_embed.0.bias": (D,),
                    "time_embed.2.weight": (D, D), "time_embed.2.bias": (D,)})
        for i in range(arch.get("integ", 3)):
            res("conditioning_timestep_integrator.%d.resblk" % i); attn("conditioning_timestep_integrator.%d.attn" % i, D)
        out.update({"inp_block.weight": (D, 100, 3), "inp_block.bias": (D,), "integrating_conv.weight": (D, 2 * D, 1), "integrating_conv.bias": (D,)})
        n_main, n_tail = arch.get("main", 10), arch.get("tail", 3)
        for i in range(n_main):
            res("layers.%d.resblk" % i); attn("layers.%d.attn" % i, D)
        for i in range(n_main, n_main + n_tail):
            res("layers.%d" % i)
        out.update({"out.0.weight": (D,), "out.0.bias": (D,), "out.2.weight": (200, D, 3), "out.2.bias": (200,), "unconditioned_embedding": (1, D, 1)})
    elif kind == "vocoder":  # vocoder.pth ['model_g'], weight_norm pairs fused: shapes of the fused `weight`
        out["conv_pre.weight"] = (32, 64, 7); out["conv_pre.bias"] = (32,)
        for i, stride in enumerate((8, 8, 4)):
            p, kp = "res_stack.%d." % i, "res_stack.%d.kernel_predictor." % i
            out[kp + "input_conv.0.weight"] = (64, 100, 5); out[kp + "input_conv.0.bias"] = (64,)
            for c in range(3):
                for j in (1, 3):
                    out[kp + "residual_convs.%d.%d.weight" % (c, j)] = (64, 64, 3); out[kp + "residual_convs.%d.%d.bias" % (c, j)] = (64,)
            out[kp + "kernel_conv.weight"] = (24576, 64, 3); out[kp + "kernel_conv.bias"] = (24576,)
            out[kp + "bias_conv.weight"] = (256, 64, 3); out[kp + "bias_conv.bias"] = (256,)
            out[p + "convt_pre.1.weight"] = (32, 32, 2 * stride); out[p + "convt_pre.1.bias"] = (32,)
            for c in range(4):
                out[p + "conv_blocks.%d.1.weight" % c] = (32, 32, 3); out[p + "conv_blocks.%d.1.bias" % c] = (32,)
        out["conv_post.1.weight"] = (1, 32, 7); out["conv_post.1.bias"] = (1,)
    elif kind == "clvp":  # clvp2.pth (use_xformers=True)
        dim, heads, ffm = arch.get("dim", 768), arch.get("heads", 12), arch.get("ff_mult", 2)
        inner, ff = heads * 64, dim * ffm
        out.update({"text_emb.weight": (256, dim), "speech_emb.weight": (8192, dim), "to_text_latent.weight": (dim, dim), "to_speech_latent.weight": (dim, dim),
                    "temperature": ()})
        for enc in ("text_transformer", "speech_transformer"):
            for i in range(arch.get("depth", 20)):
                a, f = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i), "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i + 1)
                out[a + "0.g"] = (dim,)
                for nm in ("to_q", "to_k", "to_v"):
                    out[a + "1.%s.weight" % nm] = (inner, dim)
                out[a + "1.to_out.weight"] = (dim, inner); out[a + "1.to_out.bias"] = (dim,)
                out[f + "0.g"] = (dim,)
                out[f + "1.net.0.proj.weight"] = (2 * ff, dim); out[f + "1.net.0.proj.bias"] = (2 * ff,)
                out[f + "1.net.3.weight"] = (dim, ff); out[f + "1.net.3.bias"] = (dim,)
            out[enc + ".transformer.norm.weight"] = (dim,); out[enc + ".transformer.norm.bias"] = (dim,)
    elif kind == "conditioning-encoder":  # autoregressive.pth: conditioning_encoder.*
        out["conditioning_encoder.init.weight"] = (D, 80, 1); out["conditioning_encoder.init.bias"] = (D,)
        for i in range(arch.get("blocks", 6)):
            attn("conditioning_encoder.attn.%d" % i, D, rel=False)
    elif kind == "diffusion-conditioning-encoder":  # diffusion_decoder.pth: contextual_embedder.*
        out["contextual_embedder.0.weight"] = (1024, 100, 3); out["contextual_embedder.0.bias"] = (1024,)
        out["contextual_embedder.1.weight"] = (2048, 1024, 3); out["contextual_embedder.1.bias"] = (2048,)
        for i in range(arch.get("blocks", 5)):
            attn("contextual_embedder.%d" % (2 + i), 2048)
    else:
        raise ValueError(kind)
    return out


EXPECTED_KINDS = ("ar", "diffusion", "vocoder", "clvp", "conditioning-encoder", "diffusion-conditioning-encoder")


def container_shape(kind, name, shape):
    """shape of the tensor as the CONTAINER holds it (the reference's loaders: k = 1 convolutions of the diffusion model squeezed to 2-D, conv_post
    [1, 32, 7] -> [32, 7], unconditioned_embedding flattened, a 0-dim temperature as [1]); everything else as in the checkpoint"""
    if kind == "diffusion" and SQUEEZE_K1.search(name) and len(shape) == 3:
        return shape[:2]
    if kind == "diffusion" and name == "unconditioned_embedding":
        return (int(np.prod(shape)),)
    if kind == "vocoder" and name == "conv_post.1.weight":
        return shape[-2:]
    if kind == "clvp" and name == "temperature":
        return (1,)
    return tuple(shape)


def check_against(kind, sd, prefix=""):
    """name / shape diff of a state dict against expected_tensors(kind) (architecture counts taken from the state dict where they are discoverable);
    returns a list of human-readable lines, empty when everything expected is there with the expected shape"""
    have = {k: tuple(getattr(v, "shape", ())) for k, v in sd.items() if k.startswith(prefix)}

    def count(fmt, start=0):
        n = start
        while fmt % n in have:
            n += 1
        return n - start
    arch = {}
    if kind == "ar":
        arch["layers"] = count("inference_model.transformer.h.%d.ln_1.weight") or count("gpt.h.%d.ln_1.weight") or 30
    elif kind == "diffusion":
        arch = {"lc": count("latent_conditioner.%d.norm.weight", 1) or 4, "integ": count("conditioning_timestep_integrator.%d.resblk.in_layers.0.weight") or 3,
                "main": count("layers.%d.resblk.in_layers.0.weight") or 10}
        arch["tail"] = (count("layers.%d.in_layers.0.weight", arch["main"])) or 3
    elif kind == "clvp":
        n = 0
        while "text_transformer.transformer.attn_layers.layers.%d.0.g" % (2 * n) in have:
            n += 1
        arch["depth"] = n or 20
    elif kind == "conditioning-encoder":
        arch["blocks"] = count("conditioning_encoder.attn.%d.norm.weight") or 6
    elif kind == "diffusion-conditioning-encoder":
        arch["blocks"] = count("contextual_embedder.%d.norm.weight", 2) or 5
    lines = []
    for name, shape in expected_tensors(kind, **arch).items():
        if name not in have:
            lines.append("missing  %-80s expected %s" % (name, list(shape)))
        elif have[name] != tuple(shape) and int(np.prod(have[name] or (1,))) != int(np.prod(shape or (1,))):
            lines.append("shape    %-80s expected %s, checkpoint has %s" % (name, list(shape), list(have[name])))
    return lines


def main():
    import torch
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ar")
    ap.add_argument("--diffusion")
    ap.add_argument("--diffusion-conditioning-latent", help=".pth / .npy / raw f32 file holding the voice's [1, 2048] diffusion conditioning latent")
    ap.add_argument("--vocoder")
    ap.add_argument("--conditioning-encoder", action="store_true", help="with --ar: also write ggml-conditioning-model.bin (the checkpoint's "
                                                                         "conditioning_encoder.* tensors: mel -> voice latent, tts_load_voice_encoder; not in the reference)")
    ap.add_argument("--diffusion-conditioning-encoder", help="diffusion_decoder.pth -> ggml-diffusion-conditioning-model.bin (its contextual_embedder.* tensors: "
                                                              "100-band mel -> the diffusion conditioning latent, tts_load_diffusion_conditioning_encoder; not in the reference)")
    ap.add_argument("--clvp", help="clvp2.pth of upstream tortoise-tts -> ggml-clvp-model.bin (candidate re-ranking, tts_load_clvp; not in the reference)")
    ap.add_argument("--out")
    ap.add_argument("--list-expected", nargs="*", metavar="KIND", help="print every tensor name / shape the converter looks for per checkpoint kind (%s; "
                                                                        "default: all, upstream architecture) and exit" % ", ".join(EXPECTED_KINDS))
    a = ap.parse_args()
    if a.list_expected is not None:
        for kind in (a.list_expected or EXPECTED_KINDS):
            exp = expected_tensors(kind)
            print("# %s: %d tensors" % (kind, len(exp)))
            for name, shape in exp.items():
                print("%-28s %-88s %s" % (kind, name, "x".join(map(str, shape)) or "scalar"))
        return
    if not a.out:
        ap.error("--out is required")
    os.makedirs(a.out, exist_ok=True)
    load = lambda p: torch.load(p, map_location="cpu", weights_only=True)

    def checked(kind, sd, what):  # a name / shape diff before the conversion touches anything: no silent mismatch, no bare KeyError
        if kind == "ar" and not any(k.startswith("inference_model.") for k in sd):
            sd = dict(sd)
            for k in list(sd):
                for pat, rep in AR_TRAIN_TO_INFER:
                    if re.match(pat, k):
                        sd[re.sub(pat, rep, k)] = sd[k]
        if kind == "vocoder":
            sd = fuse_weight_norm(sd["model_g"] if "model_g" in sd else sd)
        diff = check_against(kind, sd)
        if diff:
            sys.exit("%s does not look like the expected %s checkpoint (python tools/convert_weights.py --list-expected %s prints the full list):\n  "
                     % (what, kind, kind) + "\n  ".join(diff[:40]) + ("\n  ... %d more" % (len(diff) - 40) if len(diff) > 40 else ""))

    if a.ar:
        checked("ar", load(a.ar), a.ar)
        print("ggml-model.bin: %d transformer layers" % convert_ar(load(a.ar), os.path.join(a.out, "ggml-model.bin")))
    if a.conditioning_encoder:
        if not a.ar:
            sys.exit("--conditioning-encoder needs --ar (the encoder's tensors live in autoregressive.pth)")
        checked("conditioning-encoder", load(a.ar), a.ar)
        print("ggml-conditioning-model.bin: %d attention blocks" % convert_conditioning_encoder(load(a.ar), os.path.join(a.out, "ggml-conditioning-model.bin")))
    if a.diffusion:
        if not a.diffusion_conditioning_latent:
            sys.exit("--diffusion needs --diffusion-conditioning-latent (the reference bakes the voice's diffusion latent into the weight file)")
        checked("diffusion", load(a.diffusion), a.diffusion)
        p = a.diffusion_conditioning_latent
        lat = np.load(p) if p.endswith(".npy") else load(p) if p.endswith((".pth", ".pt")) else np.fromfile(p, np.float32)
        print("ggml-diffusion-model.bin: blocks (latent conditioner, integrator, main, tail) = %s"
              % (convert_diffusion(load(a.diffusion), lat, os.path.join(a.out, "ggml-diffusion-model.bin")),))
    if a.diffusion_conditioning_encoder:
        checked("diffusion-conditioning-encoder", load(a.diffusion_conditioning_encoder), a.diffusion_conditioning_encoder)
        print("ggml-diffusion-conditioning-model.bin: %d attention blocks"
              % convert_diffusion_conditioning_encoder(load(a.diffusion_conditioning_encoder), os.path.join(a.out, "ggml-diffusion-conditioning-model.bin")))
    if a.clvp:
        checked("clvp", load(a.clvp), a.clvp)
        print("ggml-clvp-model.bin: %d encoder layers" % convert_clvp(load(a.clvp), os.path.join(a.out, "ggml-clvp-model.bin")))
    if a.vocoder:
        checked("vocoder", load(a.vocoder), a.vocoder)
        print("ggml-vocoder-model.bin: %d res stacks" % convert_vocoder(load(a.vocoder), os.path.join(a.out, "ggml-vocoder-model.bin")))


if __name__ == "__main__":
    main()
