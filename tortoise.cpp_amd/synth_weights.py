"""Writer for the reference's weight container + a seeded synthetic-weight generator.

File format (what the reference loaders read, /root/reference/main.cpp:811-888, 1545-1625,
1932-2012): little-endian; u32 magic 0x67676d6c; then records until EOF:
    i32 n_dims, i32 name_len, i32 ttype (0 = F32), i32 ne[n_dims] (ne[0] = innermost = last
    PyTorch dim), name bytes, raw data. No padding, no hparams block.

Tensor names/shapes are the reference's (main.cpp:682-792, 1244-1536, 1808-1923; SURVEY.md
Appendix A). The trained weights are not available offline, so tests and bench use these synthetic
ones (fixed seed). Layer counts may be reduced for fast tests: both this engine and the oracle
discover them from the file (the reference hard-codes 30 / 4+3+10+3 / 3x4).
"""
import struct
import numpy as np

MAGIC = 0x67676D6C


class GgmlWriter:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.f.write(struct.pack("<I", MAGIC))

    def add(self, name, arr):
        """arr: numpy float32 in PyTorch dim order (outermost first)."""
        arr = np.ascontiguousarray(arr, np.float32)
        ne = list(arr.shape[::-1])
        nb = name.encode()
        self.f.write(struct.pack("<iii", len(ne), len(nb), 0))
        self.f.write(struct.pack("<%di" % len(ne), *ne))
        self.f.write(nb)
        self.f.write(arr.tobytes())

    def close(self):
        self.f.close()


def read_ggml(path):
    """Returns {name: ndarray (PyTorch dim order)}. Used by tests."""
    out = {}
    with open(path, "rb") as f:
        (magic,) = struct.unpack("<I", f.read(4))
        assert magic == MAGIC
        while True:
            hdr = f.read(12)
            if len(hdr) < 12:
                break
            n_dims, ln, tt = struct.unpack("<iii", hdr)
            ne = struct.unpack("<%di" % n_dims, f.read(4 * n_dims))
            name = f.read(ln).decode()
            n = int(np.prod(ne))
            out[name] = np.frombuffer(f.read(4 * n), np.float32).reshape(ne[::-1]).copy()
    return out


class _Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def normal(self, shape, std):
        return self.rng.standard_normal(shape, dtype=np.float32) * np.float32(std)

    def lecun(self, shape, fan_in, gain=1.0):
        return self.normal(shape, gain / np.sqrt(fan_in))

    def gamma(self, n):
        return (1.0 + self.normal((n,), 0.05)).astype(np.float32)

    def beta(self, n):
        return self.normal((n,), 0.05)


class _TrainedGen(_Gen):
    """'Trained-statistics' synthetic weights (round 6): what the small-sigma generator above never shows the fp16 operand paths —
    normalisation gains of order 0.1 .. 12 (log-uniform) with biases of order 0.5, heavy-tailed conv / linear weights (1 value in 10^4 is an outlier of 30 sigma),
    relative-position biases of +-10, embeddings at unit scale. The branch gains are divided by the RMS of the preceding normalisation gain, so the residual
    stream keeps the magnitude a trained network keeps while every GEMM operand carries the large dynamic range."""
    GAMMA_RMS = float(np.sqrt((12.0 ** 2 - 0.1 ** 2) / (2.0 * np.log(120.0))))  # rms of a log-uniform variable on [0.1, 12] = 3.88

    def normal(self, shape, std):
        w = self.rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        n = int(np.prod(shape))
        if n >= 4096:
            k = max(1, n // 10000)
            idx = self.rng.choice(n, size=k, replace=False)
            w.reshape(-1)[idx] = np.float32(30.0 * std) * self.rng.choice(np.array([-1.0, 1.0], np.float32), size=k)
        return w

    def lecun(self, shape, fan_in, gain=1.0):
        return self.normal(shape, gain / (np.sqrt(fan_in) * self.GAMMA_RMS))

    def gamma(self, n):
        return np.exp(self.rng.uniform(np.log(0.1), np.log(12.0), n)).astype(np.float32)

    def beta(self, n):
        return (self.rng.standard_normal(n) * 0.5).astype(np.float32)


def _gen(seed, stats):
    return _TrainedGen(seed) if stats == "trained" else _Gen(seed)


def write_ar(path, n_layers=30, seed=1234):
    """ggml-model.bin: GPT-2 30x1024 (main.cpp:682-792)."""
    g = _Gen(seed)
    w = GgmlWriter(path)
    D = 1024
    w.add("text_embedding.weight", g.normal((256, D), 0.02))
    w.add("text_pos_embedding.emb.weight", g.normal((404, D), 0.02))
    w.add("mel_embedding.weight", g.normal((8194, D), 0.02))
    w.add("mel_pos_embedding.emb.weight", g.normal((608, D), 0.02))
    rs = 1.0 / np.sqrt(2.0 * n_layers)
    for i in range(n_layers):
        p = "inference_model.transformer.h.%d." % i
        w.add(p + "ln_1.weight", g.gamma(D)); w.add(p + "ln_1.bias", g.beta(D))
        # HF Conv1D: PyTorch [in][out]  (ggml ne = [out, in])
        w.add(p + "attn.c_attn.weight", g.normal((D, 3 * D), 0.02)); w.add(p + "attn.c_attn.bias", g.normal((3 * D,), 0.02))
        w.add(p + "attn.c_proj.weight", g.normal((D, D), 0.02 * rs)); w.add(p + "attn.c_proj.bias", g.normal((D,), 0.02))
        w.add(p + "ln_2.weight", g.gamma(D)); w.add(p + "ln_2.bias", g.beta(D))
        w.add(p + "mlp.c_fc.weight", g.normal((D, 4 * D), 0.02)); w.add(p + "mlp.c_fc.bias", g.normal((4 * D,), 0.02))
        w.add(p + "mlp.c_proj.weight", g.normal((4 * D, D), 0.02 * rs)); w.add(p + "mlp.c_proj.bias", g.normal((D,), 0.02))
    w.add("inference_model.transformer.ln_f.weight", g.gamma(D)); w.add("inference_model.transformer.ln_f.bias", g.beta(D))
    w.add("inference_model.lm_head.0.weight", g.gamma(D)); w.add("inference_model.lm_head.0.bias", g.beta(D))
    # nn.Linear: PyTorch [out][in]; a larger std gives a peaked (non-uniform) token distribution
    w.add("inference_model.lm_head.1.weight", g.normal((8194, D), 0.08)); w.add("inference_model.lm_head.1.bias", g.normal((8194,), 0.02))
    w.close()


def _add_attn(w, g, p, D=1024):
    w.add(p + ".norm.weight", g.gamma(D)); w.add(p + ".norm.bias", g.beta(D))
    w.add(p + ".qkv.weight", g.lecun((3 * D, D), D, 1.0)); w.add(p + ".qkv.bias", g.normal((3 * D,), 0.02))
    w.add(p + ".proj_out.weight", g.lecun((D, D), D, 0.5)); w.add(p + ".proj_out.bias", g.normal((D,), 0.02))
    if isinstance(g, _TrainedGen):  # bias = 8 x table (main.cpp:3232-3275): +-10
        w.add(p + ".relative_pos_embeddings.relative_attention_bias.weight", g.rng.uniform(-1.25, 1.25, (32, 16)).astype(np.float32))
    else:
        w.add(p + ".relative_pos_embeddings.relative_attention_bias.weight", g.normal((32, 16), 0.1))


def _add_res(w, g, p, D=1024):
    w.add(p + ".in_layers.0.weight", g.gamma(D)); w.add(p + ".in_layers.0.bias", g.beta(D))
    w.add(p + ".in_layers.2.weight", g.lecun((D, D), D, 1.0)); w.add(p + ".in_layers.2.bias", g.normal((D,), 0.02))
    w.add(p + ".emb_layers.1.weight", g.lecun((2 * D, D), D, 0.5)); w.add(p + ".emb_layers.1.bias", g.normal((2 * D,), 0.02))
    w.add(p + ".out_layers.0.weight", g.gamma(D)); w.add(p + ".out_layers.0.bias", g.beta(D))
    w.add(p + ".out_layers.3.weight", g.lecun((D, D, 3), 3 * D, 0.5)); w.add(p + ".out_layers.3.bias", g.normal((D,), 0.02))


def write_diffusion(path, n_main=10, n_tail=3, n_integ=3, n_lc=4, seed=1235, stats="small"):
    """ggml-diffusion-model.bin (main.cpp:1244-1536). stats = "trained": see _TrainedGen."""
    g = _gen(seed, stats)
    w = GgmlWriter(path)
    D = 1024
    w.add("diffusion_conditioning_latent", g.normal((1, 2 * D), 1.0 if stats == "trained" else 0.1))
    w.add("latent_conditioner.0.weight", g.lecun((D, D, 3), 3 * D, 1.0)); w.add("latent_conditioner.0.bias", g.normal((D,), 0.02))
    for i in range(1, 1 + n_lc):
        _add_attn(w, g, "latent_conditioner.%d" % i)
    w.add("code_norm.weight", g.gamma(D)); w.add("code_norm.bias", g.beta(D))
    w.add("time_embed.0.weight", g.lecun((D, D), D, 1.0)); w.add("time_embed.0.bias", g.normal((D,), 0.02))
    w.add("time_embed.2.weight", g.lecun((D, D), D, 1.0)); w.add("time_embed.2.bias", g.normal((D,), 0.02))
    for i in range(n_integ):
        _add_res(w, g, "conditioning_timestep_integrator.%d.resblk" % i)
        _add_attn(w, g, "conditioning_timestep_integrator.%d.attn" % i)
    w.add("inp_block.weight", g.lecun((D, 100, 3), 300, 1.0)); w.add("inp_block.bias", g.normal((D,), 0.02))
    w.add("integrating_conv.weight", g.lecun((D, 2 * D), 2 * D, 1.0)); w.add("integrating_conv.bias", g.normal((D,), 0.02))
    for i in range(n_main):
        _add_res(w, g, "layers.%d.resblk" % i)
        _add_attn(w, g, "layers.%d.attn" % i)
    for i in range(n_main, n_main + n_tail):
        _add_res(w, g, "layers.%d" % i)
    w.add("out.0.weight", g.gamma(D)); w.add("out.0.bias", g.beta(D))
    w.add("out.2.weight", g.lecun((200, D, 3), 3 * D, 0.5)); w.add("out.2.bias", g.normal((200,), 0.02))
    w.add("unconditioned_embedding", g.normal((1, D, 1), 1.0 if stats == "trained" else 0.5).reshape(D))
    w.close()


def write_vocoder(path, seed=1236):
    """ggml-vocoder-model.bin: UnivNet (main.cpp:1808-1923)."""
    g = _Gen(seed)
    w = GgmlWriter(path)
    w.add("conv_pre.weight", g.lecun((32, 64, 7), 448, 1.0)); w.add("conv_pre.bias", g.normal((32,), 0.02))
    strides = [8, 8, 4]
    for i in range(3):
        p = "res_stack.%d." % i
        kp = p + "kernel_predictor."
        w.add(kp + "input_conv.0.weight", g.lecun((64, 100, 5), 500, 0.3)); w.add(kp + "input_conv.0.bias", g.normal((64,), 0.02))
        for c in range(3):
            for j in (1, 3):
                w.add(kp + "residual_convs.%d.%d.weight" % (c, j), g.lecun((64, 64, 3), 192, 0.7))
                w.add(kp + "residual_convs.%d.%d.bias" % (c, j), g.normal((64,), 0.02))
        w.add(kp + "kernel_conv.weight", g.lecun((24576, 64, 3), 192, 0.1)); w.add(kp + "kernel_conv.bias", g.normal((24576,), 0.02))
        w.add(kp + "bias_conv.weight", g.lecun((256, 64, 3), 192, 0.3)); w.add(kp + "bias_conv.bias", g.normal((256,), 0.02))
        K = 2 * strides[i]
        # ConvTranspose1d: PyTorch [Cin][Cout][K]
        w.add(p + "convt_pre.1.weight", g.lecun((32, 32, K), 32 * 2, 1.0)); w.add(p + "convt_pre.1.bias", g.normal((32,), 0.02))
        for c in range(4):
            w.add(p + "conv_blocks.%d.1.weight" % c, g.lecun((32, 32, 3), 96, 1.0)); w.add(p + "conv_blocks.%d.1.bias" % c, g.normal((32,), 0.02))
    w.add("conv_post.1.weight", g.lecun((1, 32, 7), 224, 0.5).reshape(32, 7)); w.add("conv_post.1.bias", g.normal((1,), 0.02))
    w.close()


CLVP_ENCODERS = ("text_transformer", "speech_transformer")


def write_clvp(path, depth=20, dim=768, heads=12, ff_mult=2, seed=1237):
    """ggml-clvp-model.bin: the CLVP re-ranker of upstream tortoise-tts (tortoise/models/clvp.py with use_xformers=True: two x-transformers
    encoders of `depth` (attention, GEGLU feed-forward) pairs with RMSNorm pre-norm and rotary position embedding, masked mean, latent
    projection, cosine similarity x exp(temperature)), tensor names = the upstream state dict's. The reference has no CLVP (main.cpp:6575
    takes candidate 0): the container is this repository's, in the reference's file format."""
    g = _Gen(seed)
    w = GgmlWriter(path)
    inner, ff = heads * 64, dim * ff_mult
    w.add("text_emb.weight", g.normal((256, dim), 0.5))
    w.add("speech_emb.weight", g.normal((8192, dim), 0.5))
    w.add("to_text_latent.weight", g.lecun((dim, dim), dim))
    w.add("to_speech_latent.weight", g.lecun((dim, dim), dim))
    w.add("temperature", np.array([1.0], np.float32))
    rs = 1.0 / np.sqrt(2.0 * depth)
    for enc in CLVP_ENCODERS:
        for i in range(depth):
            a = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i)
            f = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i + 1)
            w.add(a + "0.g", g.gamma(dim))
            for nm in ("to_q", "to_k", "to_v"):
                w.add(a + "1.%s.weight" % nm, g.lecun((inner, dim), dim, 1.5))
            w.add(a + "1.to_out.weight", g.lecun((dim, inner), inner, rs)); w.add(a + "1.to_out.bias", g.normal((dim,), 0.02))
            w.add(f + "0.g", g.gamma(dim))
            w.add(f + "1.net.0.proj.weight", g.lecun((2 * ff, dim), dim)); w.add(f + "1.net.0.proj.bias", g.normal((2 * ff,), 0.02))
            w.add(f + "1.net.3.weight", g.lecun((dim, ff), ff, rs)); w.add(f + "1.net.3.bias", g.normal((dim,), 0.02))
        w.add(enc + ".transformer.norm.weight", g.gamma(dim)); w.add(enc + ".transformer.norm.bias", g.beta(dim))
    w.close()


def write_voice_encoder(path, blocks=6, seed=1238):
    """ggml-conditioning-model.bin: the conditioning encoder of upstream tortoise-tts' UnifiedVoice (tortoise/models/autoregressive.py:
    ConditioningEncoder(80, 1024, attn_blocks=6, num_attn_heads=16) = Conv1d(80, 1024, 1) + 6 AttentionBlocks (GroupNorm32, qkv Conv1d k=1,
    QKVAttentionLegacy, proj_out) -> position 0), which turns an 80-band mel of a reference clip into the 1024-float voice latent the reference
    reads from --voice (main.cpp:5179-5184; README.md:54-72 is the offline recipe). Tensor names = the upstream state dict's. SURVEY 8 f3."""
    g = _Gen(seed)
    w = GgmlWriter(path)
    D = 1024
    w.add("conditioning_encoder.init.weight", g.lecun((D, 80, 1), 80)); w.add("conditioning_encoder.init.bias", g.normal((D,), 0.02))
    for i in range(blocks):
        p = "conditioning_encoder.attn.%d." % i
        w.add(p + "norm.weight", g.gamma(D)); w.add(p + "norm.bias", g.beta(D))
        w.add(p + "qkv.weight", g.lecun((3 * D, D, 1), D, 1.5)); w.add(p + "qkv.bias", g.normal((3 * D,), 0.02))
        w.add(p + "proj_out.weight", g.lecun((D, D, 1), D, 0.5)); w.add(p + "proj_out.bias", g.normal((D,), 0.02))
    w.close()


def write_diffusion_conditioning_encoder(path, blocks=5, seed=1239):
    """ggml-diffusion-conditioning-model.bin: `contextual_embedder` of upstream tortoise-tts' DiffusionTts (tortoise/models/diffusion_decoder.py:
    Conv1d(100, 1024, 3, padding=1, stride=2), Conv1d(1024, 2048, 3, padding=1, stride=2), 5 x AttentionBlock(2048, 16 heads, relative position
    embeddings)); get_conditioning = mean over the frames of all clips -> the 2048-float `diffusion_conditioning_latent` the reference bakes
    into ggml-diffusion-model.bin (main.cpp:1557-1560 reads it as a weight). Tensor names = the upstream state dict's. SURVEY 8 f3."""
    g = _Gen(seed)
    w = GgmlWriter(path)
    w.add("contextual_embedder.0.weight", g.lecun((1024, 100, 3), 300)); w.add("contextual_embedder.0.bias", g.normal((1024,), 0.02))
    w.add("contextual_embedder.1.weight", g.lecun((2048, 1024, 3), 3072)); w.add("contextual_embedder.1.bias", g.normal((2048,), 0.02))
    D = 2048
    for i in range(blocks):
        p = "contextual_embedder.%d." % (2 + i)
        w.add(p + "norm.weight", g.gamma(D)); w.add(p + "norm.bias", g.beta(D))
        w.add(p + "qkv.weight", g.lecun((3 * D, D, 1), D, 1.5)); w.add(p + "qkv.bias", g.normal((3 * D,), 0.02))
        w.add(p + "proj_out.weight", g.lecun((D, D, 1), D, 0.5)); w.add(p + "proj_out.bias", g.normal((D,), 0.02))
        w.add(p + "relative_pos_embeddings.relative_attention_bias.weight", g.normal((32, 16), 0.1))
    w.close()


def write_all(out_dir, ar_layers=30, diff_main=10, diff_tail=3, diff_integ=3, diff_lc=4, seed=1234):
    import os
    os.makedirs(out_dir, exist_ok=True)
    write_ar(os.path.join(out_dir, "ggml-model.bin"), ar_layers, seed)
    write_diffusion(os.path.join(out_dir, "ggml-diffusion-model.bin"), diff_main, diff_tail, diff_integ, diff_lc, seed + 1)
    write_vocoder(os.path.join(out_dir, "ggml-vocoder-model.bin"), seed + 2)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--ar-layers", type=int, default=30)
    ap.add_argument("--diff-main", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    write_all(a.out_dir, a.ar_layers, a.diff_main, seed=a.seed)
