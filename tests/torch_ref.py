"""TEST INFRASTRUCTURE — an independent PyTorch (CPU, fp32) assembly of the three network stages from torch's own ops
(F.conv1d, F.group_norm, F.layer_norm, F.conv_transpose1d, F.pad(reflect), Tensor.unfold, F.interpolate, F.gelu(tanh),
F.silu, softmax), reading the reference-format weight files with synth_weights.read_ggml (PyTorch dim order).

Purpose ("pin what can be pinned offline"): the oracle (oracle/orc_*.cpp) restates the reference's ggml graphs
with hand-written loops; this module restates the same graphs with PyTorch semantics, which is what the ggml graphs were
transcribed from (tortoise-tts' DiffusionTts / UnivNet / GPT-2). tests/test_oracle_vs_torch.py compares the two on synthetic
weights: op semantics (conv1d-as-fp16-im2col, GroupNorm, conv_transpose_1d, reflect pad, unfold/LVC, nearest upscale,
tanh-GELU) and the wiring of every block are then torch-pinned; what stays unverified is only what neither side can know
offline (ggml-fork constants: GroupNorm eps, fp16 activation tables — both exposed as switches).

Reference line numbers are /root/reference/main.cpp. The fp16 rounding points are the reference's: conv1d rounds weights and
the im2col'd input to fp16 and accumulates in f32; the AR stack rounds QKV to fp16 (main.cpp:2789-2790).
"""
import zlib

import numpy as np
import torch
import torch.nn.functional as F

torch.set_grad_enabled(False)


def h16(x):
    """fp16 round trip (ggml F32->F16->F32); keeps the working dtype (f32, or f64 for the triangulation runs)."""
    return x.float().half().to(x.dtype)


def stochastic_h16(x, seed):
    """unbiased stochastic rounding to fp16: RNE of x + u * ulp(x), u uniform in (-0.5, 0.5), generator seeded per (tensor, variant)"""
    g = torch.Generator().manual_seed(int(seed) & 0x7fffffff)
    e = torch.floor(torch.log2(x.abs().float().clamp(min=2.0 ** -14)))
    ulp = torch.pow(2.0, e - 10.0)
    u = torch.rand(x.shape, generator=g) - 0.5
    return h16(x.float() + u * ulp).to(x.dtype)


def dither_h16(x, n, j, seed):
    """Round 6 (generalised): the j-th member of a PERIOD-n dither cycle of fp16 roundings of x. Every member is one of the two fp16 neighbours
    of x (floor / ceil on the fp16 grid); over one cycle exactly round(n * frac) members are the upper neighbour, spread evenly (Bresenham) from a per-element random
    phase, so the cycle MEAN is within ulp / (2 n) of x and every partial sum of the rounding errors stays within one ulp — no linear drift (RNE: the same error at
    every step) and no random walk (independent stochastic rounding). n = 2 is the antithetic pair; members are plain fp16 tensors (one GEMM operand each)."""
    xf = x.double()
    lo = xf.float().half()                                     # RNE, then step to the neighbour at or below x
    lo = torch.where(lo.double() > xf, torch.nextafter(lo, torch.tensor(-float("inf"), dtype=torch.half)), lo)
    hi = torch.nextafter(lo, torch.tensor(float("inf"), dtype=torch.half))
    lod, hid = lo.double(), hi.double()
    frac = ((xf - lod) / (hid - lod)).clamp(0.0, 1.0)          # position between the neighbours
    k = torch.round(frac * n)                                  # members of the cycle that take the upper neighbour
    g = torch.Generator().manual_seed(int(seed) & 0x7fffffff)
    phase = torch.rand(x.shape, generator=g, dtype=torch.float64)
    up = torch.floor((j % n + 1) * k / n + phase) - torch.floor((j % n) * k / n + phase)
    return torch.where(up > 0.5, hid, lod).to(x.dtype)


def antithetic_h16(x, j):
    """VERDICT r5 item 4 as written: W_a = fp16(W) on even steps, W_b = fp16(2 W - W_a) on odd ones"""
    a = h16(x)
    return a if j % 2 == 0 else h16(2.0 * x - a)


def conv1d_f16(x, w, b, padding=0, dilation=1):
    """x [Cin, T], w [Cout, Cin, K] -> [Cout, Tout]; fp16-rounded operands, f32 accumulate."""
    return F.conv1d(h16(x)[None], h16(w), b, padding=padding, dilation=dilation)[0]


def gn32(x, g, b, eps):
    """x [C, T] -> GroupNorm(32) over (C/32, T) per group (ggml_group_norm on [T,1,C])."""
    return F.group_norm(x[None], 32, g, b, eps)[0]


class Weights:
    def __init__(self, path, dtype=torch.float32):
        import tortoise_cpp_amd_loader
        tortoise_cpp_amd_loader.load()
        from tortoise_cpp_amd import synth_weights  # the container reader (format: main.cpp:811-888)
        self.dtype = dtype
        self.t = {k: torch.from_numpy(v).to(dtype) for k, v in synth_weights.read_ggml(path).items()}

    def tensor(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.dtype)

    def __getitem__(self, k):
        return self.t[k]

    def has(self, k):
        return k in self.t


# ------------------------------------------------------------------------------------------------------------------
# diffusion (diffusion_graph, main.cpp:3066-4044)
# ------------------------------------------------------------------------------------------------------------------
class TorchDiffusion:
    def __init__(self, path, buckets, gn_eps=1e-6, dtype=torch.float32, f16_attention=False):
        """f16_attention: emulate the ENGINE's AttentionBlock arithmetic instead of the reference's F32 one (DESIGN.md section 4,
        north-star "MFMA ... for the dense fp16 GEMMs in attention"): q, k, v, the unnormalised softmax numerators, the attention
        output and the proj_out weight are rounded to fp16, accumulation stays f32/f64. Used by tests/test_parity_floor.py to
        separate what the fp16 attention costs from what ANY f32 evaluation order costs.
        dtype float64: the same graph (same fp16 rounding points) evaluated in double — the yardstick for how far two f32
        evaluations with different summation orders may legitimately be apart. buckets(n) -> int array [n, n] of T5 relative-position buckets, index [query][key] (main.cpp:4722-4749; pinned against
        the reference's own code in tests/test_host_parity.py — passed in so this file stays free of oracle imports)."""
        self.w = Weights(path, dtype)
        self.buckets = buckets
        self.eps = gn_eps
        # round 5: the five fp16 roundings of the engine's throughput AttentionBlock as INDEPENDENT switches, so the 80-step distance from
        # the oracle can be attributed to each of them alone (tools/regen_parity_floor.py --ablate):
        #   "qk" q and k operands of QK^T, "v" the V operand of P.V, "p" the unnormalised softmax numerators (row sums then run over the ROUNDED numerators, as
        #   the engine's all-ones MFMA does), "o" the attention output handed to proj_out, "w" the proj_out weight.
        # f16_attention = True is all five; a set / string of names selects a subset ("" = the reference's F32 block evaluated through the same code path).
        if f16_attention is True:
            f16_attention = ("qk", "v", "p", "o", "w")
        elif isinstance(f16_attention, str):
            f16_attention = tuple(x for x in f16_attention.split(",") if x)
        self.f16_attention = frozenset(f16_attention or ())
        # "wd" (round 5 experiment): the proj_out weight as ONE fp16 operand again, but a DIFFERENT unbiased stochastic rounding of it at every sampling step
        # (self.w_variant, set by the loop driver): the rounding error stops being the same perturbation at all 80 steps
        # "wa" / "wb<n>" (round 6): the antithetic pair / a period-n Bresenham dither cycle of fp16 roundings of the proj_out weight, member = sampling step mod n
        self.w_dither = 0
        for name in self.f16_attention:
            if name.startswith("wb"):
                self.w_dither = int(name[2:])
        assert all(x in {"qk", "v", "p", "o", "w", "wd", "wa"} or x.startswith("wb") for x in self.f16_attention), self.f16_attention
        self.step_index = 0  # set by the loop driver for "wa" / "wb<n>"
        self.w_variant = 0
        w = self.w
        self.n_lc = 0
        while w.has("latent_conditioner.%d.norm.weight" % (self.n_lc + 1)):
            self.n_lc += 1
        self.n_integ = 0
        while w.has("conditioning_timestep_integrator.%d.resblk.in_layers.0.weight" % self.n_integ):
            self.n_integ += 1
        self.n_main = 0
        while w.has("layers.%d.resblk.in_layers.0.weight" % self.n_main):
            self.n_main += 1
        self.n_tail = 0
        while w.has("layers.%d.in_layers.0.weight" % (self.n_main + self.n_tail)):
            self.n_tail += 1

    def attention(self, x, p):
        """AttentionBlock (main.cpp:3184-3288): x [C, T]."""
        w = self.w
        C, T = x.shape
        h = gn32(x, w[p + ".norm.weight"], w[p + ".norm.bias"], self.eps)
        qkv = conv1d_f16(h, w[p + ".qkv.weight"].reshape(3 * C, C, 1), w[p + ".qkv.bias"])  # [3C, T]
        q, k, v = qkv.reshape(16, 192, T).split(64, dim=1)  # channel = head*192 + {q | k | v}
        rel = w[p + ".relative_pos_embeddings.relative_attention_bias.weight"]  # [32 buckets, 16 heads]
        bk = torch.from_numpy(np.asarray(self.buckets(T), np.int64))  # [query, key]
        bias = rel[bk].permute(2, 0, 1) * 8.0  # [head, query, key]
        if self.f16_attention:  # the engine: fp16 MFMA operands, f32 accumulate; row sums taken over the ROUNDED numerators
            r = self.f16_attention
            if "qk" in r:
                q, k = h16(q), h16(k)
            if "v" in r:
                v = h16(v)
            att = torch.einsum("hdi,hdj->hij", q, k) * (1.0 / 8.0) + bias
            e = torch.exp(att - att.max(dim=-1, keepdim=True).values)
            if "p" in r:
                e = h16(e)
            a = (torch.einsum("hij,hdj->hdi", e, v) / e.sum(dim=-1)[:, None, :]).reshape(C, T)
            if "o" in r:
                a = h16(a)
            pw = w[p + ".proj_out.weight"]
            if "wd" in r:
                pw = stochastic_h16(pw, (zlib.crc32(p.encode()) & 0xffff) * 1009 + self.w_variant)  # (a str hash is randomised per process)
            if "wa" in r:
                pw = antithetic_h16(pw, self.step_index)
            if self.w_dither:
                pw = dither_h16(pw, self.w_dither, self.step_index, zlib.crc32(p.encode()))
            o = F.conv1d(a[None], (h16(pw) if "w" in r else pw).reshape(C, C, 1), w[p + ".proj_out.bias"])[0]
            return x + o
        att = torch.einsum("hdi,hdj->hij", q, k) * (1.0 / 8.0) + bias
        att = torch.softmax(att, dim=-1)
        a = torch.einsum("hij,hdj->hdi", att, v).reshape(C, T)
        o = F.conv1d(a[None], w[p + ".proj_out.weight"].reshape(C, C, 1), w[p + ".proj_out.bias"])[0]  # F32 linear
        return x + o

    def resblock(self, x, p, emb):
        """ResBlock (main.cpp:3659-3782): x [C, T], emb [1024] (time embedding after the MLP, not yet activated)."""
        w = self.w
        C = x.shape[0]
        h = F.silu(gn32(x, w[p + ".in_layers.0.weight"], w[p + ".in_layers.0.bias"], self.eps))
        h = conv1d_f16(h, w[p + ".in_layers.2.weight"].reshape(C, C, 1), w[p + ".in_layers.2.bias"])
        ss = F.linear(F.silu(emb), w[p + ".emb_layers.1.weight"], w[p + ".emb_layers.1.bias"])  # [2048] = scale | shift
        scale, shift = ss[:C], ss[C:]
        h = gn32(h, w[p + ".out_layers.0.weight"], w[p + ".out_layers.0.bias"], self.eps)
        h = F.silu(h * (scale[:, None] + 1.0) + shift[:, None])
        h = conv1d_f16(h, w[p + ".out_layers.3.weight"], w[p + ".out_layers.3.bias"], padding=1)
        return x + h

    @staticmethod
    def upscale_index(L, T):
        """ggml_upscale_ext nearest: src = (int)(dst / ((float)T / L)) in f32 (main.cpp:3321 via SURVEY 3.7)."""
        sf = np.float32(T) / np.float32(L)
        idx = (np.arange(T, dtype=np.float32) / sf).astype(np.int64)
        return np.minimum(idx, L - 1)

    def code_embedding(self, latents, T):
        """latents [L, 1024] -> [T, 1024] (main.cpp:3156-3321)."""
        w = self.w
        x = w.tensor(latents).T.contiguous()  # [C, L]
        C, L = x.shape
        x = conv1d_f16(x, w["latent_conditioner.0.weight"], w["latent_conditioner.0.bias"], padding=1)
        for i in range(self.n_lc):
            x = self.attention(x, "latent_conditioner.%d" % (i + 1))
        x = gn32(x, w["code_norm.weight"], w["code_norm.bias"], self.eps)
        cl = w["diffusion_conditioning_latent"].reshape(-1)
        x = x * (cl[:C, None] + 1.0) + cl[C:, None]
        idx = torch.from_numpy(self.upscale_index(L, T))
        return x[:, idx].T.contiguous().float().numpy()

    def time_embedding(self, te):
        w = self.w
        e = F.linear(w.tensor(te), w["time_embed.0.weight"], w["time_embed.0.bias"])
        return F.linear(F.silu(e), w["time_embed.2.weight"], w["time_embed.2.bias"])

    def forward(self, code_emb, x_t, te):
        """code_emb [T,1024] or None; x_t [100,T]; te = sinusoidal timestep embedding [1024] (host math, pinned separately).
        Returns [200, T]."""
        w = self.w
        x_t = w.tensor(x_t)
        T = x_t.shape[1]
        emb = self.time_embedding(te)
        if code_emb is None:
            ce = w["unconditioned_embedding"].reshape(-1, 1).repeat(1, T)
        else:
            ce = w.tensor(code_emb).T.contiguous()
        for i in range(self.n_integ):
            p = "conditioning_timestep_integrator.%d" % i
            ce = self.resblock(ce, p + ".resblk", emb)
            ce = self.attention(ce, p + ".attn")
        xi = conv1d_f16(x_t, w["inp_block.weight"], w["inp_block.bias"], padding=1)
        x = conv1d_f16(torch.cat([xi, ce], 0), w["integrating_conv.weight"].reshape(1024, 2048, 1), w["integrating_conv.bias"])
        for i in range(self.n_main):
            x = self.resblock(x, "layers.%d.resblk" % i, emb)
            x = self.attention(x, "layers.%d.attn" % i)
        for i in range(self.n_tail):
            x = self.resblock(x, "layers.%d" % (self.n_main + i), emb)
        h = F.silu(gn32(x, w["out.0.weight"], w["out.0.bias"], self.eps))
        return conv1d_f16(h, w["out.2.weight"], w["out.2.bias"], padding=1).float().numpy()


# ------------------------------------------------------------------------------------------------------------------
# vocoder (vocoder_graph, main.cpp:4068-4483)
# ------------------------------------------------------------------------------------------------------------------
class TorchVocoder:
    def __init__(self, path, dtype=torch.float32):
        self.w = Weights(path, dtype)

    def forward(self, mel_denorm, noise):
        """mel_denorm [100, T] (already denormalised), noise [64, T+10] -> audio [(T+10)*256 - 6]."""
        w = self.w
        lk = lambda v: F.leaky_relu(v, 0.2)
        mel = w.tensor(mel_denorm)
        z = w.tensor(noise)
        T = mel.shape[1]
        Tm = T + 10
        pm = torch.cat([mel, w.tensor(np.full((100, 10), -11.5129, np.float32))], 1)  # 10 silent frames (main.cpp:6051-6054)
        x = conv1d_f16(F.pad(z[None], (3, 3), mode="reflect")[0], w["conv_pre.weight"], w["conv_pre.bias"])  # [32, Tm]
        for i, (s, hop) in enumerate(zip((8, 8, 4), (8, 64, 256))):
            rs = "res_stack.%d" % i
            # leaky -> ConvTranspose1d(32, 32, 2s, stride s, padding s/2): F32 (main.cpp:4145-4167)
            x = F.conv_transpose1d(lk(x)[None], w[rs + ".convt_pre.1.weight"], w[rs + ".convt_pre.1.bias"], stride=s, padding=s // 2)[0]
            kp = rs + ".kernel_predictor"
            c = lk(conv1d_f16(pm, w[kp + ".input_conv.0.weight"], w[kp + ".input_conv.0.bias"], padding=2))
            for r in range(3):
                rp = kp + ".residual_convs.%d" % r
                c1 = lk(conv1d_f16(c, w[rp + ".1.weight"], w[rp + ".1.bias"], padding=1))
                c2 = lk(conv1d_f16(c1, w[rp + ".3.weight"], w[rp + ".3.bias"], padding=1))
                c = c + c2
            kern = conv1d_f16(c, w[kp + ".kernel_conv.weight"], w[kp + ".kernel_conv.bias"], padding=1)  # [24576, Tm]
            kb = conv1d_f16(c, w[kp + ".bias_conv.weight"], w[kp + ".bias_conv.bias"], padding=1)  # [256, Tm]
            kern = kern.reshape(4, 32, 64, 3, Tm)  # [layer, in, out, tap, frame] (channel = ((c*32+i)*64+o)*3+k, main.cpp:4323, 4371)
            kb = kb.reshape(4, 64, Tm)
            for cidx, d in enumerate((1, 3, 9, 27)):
                cb = rs + ".conv_blocks.%d.1" % cidx
                y = lk(conv1d_f16(lk(x), w[cb + ".weight"], w[cb + ".bias"], padding=d, dilation=d))  # [32, len]
                # location-variable convolution through unfold (main.cpp:4337-4456): windows of hop + 2 samples per frame, 3 taps
                yp = F.pad(y, (1, 1))
                win = yp.unfold(1, hop + 2, hop)  # [32, Tm, hop + 2]
                win = win.unfold(2, 3, 1)  # [32, Tm, hop, 3]
                o = torch.einsum("ilsk,iokl->ols", win, kern[cidx]) + kb[cidx][:, :, None]  # [64, Tm, hop]
                o = o.reshape(64, Tm * hop)
                x = x + torch.sigmoid(o[:32]) * torch.tanh(o[32:])
        return conv1d_f16(lk(x), w["conv_post.1.weight"].reshape(1, 32, 7), w["conv_post.1.bias"])[0].float().numpy()


# ------------------------------------------------------------------------------------------------------------------
# autoregressive GPT-2 (main.cpp:2053-3040)
# ------------------------------------------------------------------------------------------------------------------
class TorchAR:
    def __init__(self, path, dtype=torch.float32):
        self.w = Weights(path, dtype)
        self.n_layers = 0
        while self.w.has("inference_model.transformer.h.%d.ln_1.weight" % self.n_layers):
            self.n_layers += 1

    def stack(self, x):
        """x [S, 1024]: full causal pass, QKV rounded to fp16 (main.cpp:2718-2983)."""
        w = self.w
        S = x.shape[0]
        mask = torch.full((S, S), float("-inf"), dtype=x.dtype).triu(1)
        for l in range(self.n_layers):
            p = "inference_model.transformer.h.%d" % l
            h = F.layer_norm(x, (1024,), w[p + ".ln_1.weight"], w[p + ".ln_1.bias"], 1e-5)
            qkv = h16(h @ w[p + ".attn.c_attn.weight"] + w[p + ".attn.c_attn.bias"])  # HF Conv1D: weight [in, out]
            q, k, v = [t.reshape(S, 16, 64).transpose(0, 1) for t in qkv.split(1024, dim=1)]
            att = torch.softmax(q @ k.transpose(1, 2) * (1.0 / 8.0) + mask, dim=-1)
            a = (att @ v).transpose(0, 1).reshape(S, 1024)
            x = x + (a @ w[p + ".attn.c_proj.weight"] + w[p + ".attn.c_proj.bias"])
            h = F.layer_norm(x, (1024,), w[p + ".ln_2.weight"], w[p + ".ln_2.bias"], 1e-5)
            f = F.gelu(h @ w[p + ".mlp.c_fc.weight"] + w[p + ".mlp.c_fc.bias"], approximate="tanh")
            x = x + (f @ w[p + ".mlp.c_proj.weight"] + w[p + ".mlp.c_proj.bias"])
        return x

    def final_norms(self, x):
        w = self.w
        x = F.layer_norm(x, (1024,), w["inference_model.transformer.ln_f.weight"], w["inference_model.transformer.ln_f.bias"], 1e-5)
        return F.layer_norm(x, (1024,), w["inference_model.lm_head.0.weight"], w["inference_model.lm_head.0.bias"], 1e-5)

    def inputs(self, tokens, voice, mel_ids, mel_pos):
        w = self.w
        tok = torch.from_numpy(np.asarray(tokens, np.int64))
        rows = [w.tensor(voice)[None],
                w["text_embedding.weight"][tok] + w["text_pos_embedding.emb.weight"][: len(tok)]]
        mel = torch.from_numpy(np.asarray(mel_ids, np.int64))
        pos = torch.from_numpy(np.asarray(mel_pos, np.int64))
        rows.append(w["mel_embedding.weight"][mel] + w["mel_pos_embedding.emb.weight"][pos])
        return torch.cat(rows, 0)

    def logits_after(self, tokens, voice, mel_ids, mel_pos):
        """Logits at the last position of [voice | text | mel_ids at mel positions mel_pos] — a cache-free evaluation of what the
        prefill (mel_ids = [8192], pos [0]) and decode step i (positions 0, 2, 3, ..., i + 2: the reference's quirk,
        main.cpp:5244) produce through the KV cache."""
        w = self.w
        x = self.final_norms(self.stack(self.inputs(tokens, voice, mel_ids, mel_pos))[-1:])
        return F.linear(x, w["inference_model.lm_head.1.weight"], w["inference_model.lm_head.1.bias"])[0].float().numpy()

    def latents(self, tokens, voice, codes, n_mel):
        """Latent pass (main.cpp:2053-2519): mel positions 0 .. n_mel-1; returns the n_mel mel rows after both LayerNorms."""
        x = self.inputs(tokens, voice, codes[:n_mel], np.arange(n_mel))
        return self.final_norms(self.stack(x))[1 + len(tokens):].float().numpy()



# ------------------------------------------------------------------------------------------------------------------------------
# CLVP re-ranker. No reference code exists for it (main.cpp:6575 takes candidate 0): this is the upstream
# tortoise-tts architecture (tortoise/models/clvp.py, use_xformers=True; vendored x-transformers Encoder with use_rmsnorm, ff_glu,
# rotary_pos_emb, ff_mult=2) written with torch ops, used to pin the numpy oracle (oracle.Clvp) — "parity unpinned" against upstream weights.
# ------------------------------------------------------------------------------------------------------------------------------
class TorchCLVP:
    def __init__(self, path, dtype=torch.float32):
        from tortoise_cpp_amd import synth_weights as sw
        self.w = {k: torch.from_numpy(v).to(dtype) for k, v in sw.read_ggml(path).items()}
        self.dtype = dtype
        self.dim = self.w["text_emb.weight"].shape[1]
        self.depth = 0
        while "text_transformer.transformer.attn_layers.layers.%d.0.g" % (2 * self.depth) in self.w:
            self.depth += 1
        self.heads = self.w["text_transformer.transformer.attn_layers.layers.0.1.to_q.weight"].shape[0] // 64

    def _rotary(self, n):
        inv = 1.0 / (10000.0 ** (torch.arange(0, 32, 2, dtype=self.dtype) / 32.0))
        fr = torch.einsum("i,j->ij", torch.arange(n, dtype=self.dtype), inv)
        return torch.cat((fr, fr), dim=-1)  # [n, 32]

    @staticmethod
    def _rot_half(x):
        x1, x2 = x[..., :16], x[..., 16:]
        return torch.cat((-x2, x1), dim=-1)

    def encode(self, enc, x):
        w, n, H = self.w, x.shape[0], self.heads
        fr = self._rotary(n)
        for i in range(self.depth):
            a = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i)
            f = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i + 1)
            def rms(t, g):
                nrm = torch.linalg.vector_norm(t, dim=-1, keepdim=True) * (self.dim ** -0.5)
                return t / nrm.clamp(min=1e-8) * g
            y = rms(x, w[a + "0.g"])
            q = (y @ w[a + "1.to_q.weight"].T).reshape(n, H, 64).transpose(0, 1)
            k = (y @ w[a + "1.to_k.weight"].T).reshape(n, H, 64).transpose(0, 1)
            v = (y @ w[a + "1.to_v.weight"].T).reshape(n, H, 64).transpose(0, 1)
            def rope(t):
                tl, tr = t[..., :32], t[..., 32:]
                return torch.cat((tl * fr.cos() + self._rot_half(tl) * fr.sin(), tr), dim=-1)
            q, k, v = rope(q), rope(k), rope(v)  # tortoise-tts models/xtransformers.py Attention.forward: apply_rotary_pos_emb on (ql, kl, vl)
            att = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
            o = (att @ v).transpose(0, 1).reshape(n, H * 64)
            x = x + o @ w[a + "1.to_out.weight"].T + w[a + "1.to_out.bias"]
            y = rms(x, w[f + "0.g"])
            u = y @ w[f + "1.net.0.proj.weight"].T + w[f + "1.net.0.proj.bias"]
            val, gate = u.chunk(2, dim=-1)
            x = x + (val * torch.nn.functional.gelu(gate)) @ w[f + "1.net.3.weight"].T + w[f + "1.net.3.bias"]
        return torch.nn.functional.layer_norm(x, (self.dim,), w[enc + ".transformer.norm.weight"], w[enc + ".transformer.norm.bias"], 1e-5)

    def latent(self, enc, emb, proj, tokens):
        x = self.w[emb][torch.as_tensor(np.asarray(tokens), dtype=torch.long)]
        z = self.encode(enc, x).mean(dim=0) @ self.w[proj].T
        return torch.nn.functional.normalize(z, p=2, dim=-1)

    def score(self, text, speech_list):
        """cosine similarity x exp(temperature) of the text with every speech-code sequence."""
        zt = self.latent("text_transformer", "text_emb.weight", "to_text_latent.weight", text)
        t = self.w["temperature"].reshape(()).exp()
        return np.array([float((zt * self.latent("speech_transformer", "speech_emb.weight", "to_speech_latent.weight", sp)).sum() * t)
                         for sp in speech_list])



# ------------------------------------------------------------------------------------------------------------------------------
# Voice-conditioning encoder: upstream tortoise-tts UnifiedVoice.get_conditioning (ConditioningEncoder + arch_util
# AttentionBlock / QKVAttentionLegacy) with torch ops; pins oracle.VoiceEncoder. No reference code exists (the reference reads the
# finished latent from --voice).
# ------------------------------------------------------------------------------------------------------------------------------
class TorchVoiceEncoder:
    def __init__(self, path, dtype=torch.float32):
        from tortoise_cpp_amd import synth_weights as sw
        self.w = {k: torch.from_numpy(v).to(dtype) for k, v in sw.read_ggml(path).items()}
        self.dtype = dtype
        self.blocks = 0
        while "conditioning_encoder.attn.%d.norm.weight" % self.blocks in self.w:
            self.blocks += 1

    def clip(self, mel):  # mel [80, T]
        w = self.w
        x = torch.as_tensor(np.asarray(mel)).to(self.dtype)[None]
        h = F.conv1d(x, w["conditioning_encoder.init.weight"], w["conditioning_encoder.init.bias"])
        for i in range(self.blocks):
            p = "conditioning_encoder.attn.%d." % i
            y = F.group_norm(h, 32, w[p + "norm.weight"], w[p + "norm.bias"], 1e-5)
            qkv = F.conv1d(y, w[p + "qkv.weight"], w[p + "qkv.bias"])
            bs, width, length = qkv.shape
            ch = width // (3 * 16)
            q, k, v = qkv.reshape(bs * 16, ch * 3, length).split(ch, dim=1)
            scale = 1 / (ch ** 0.25)
            wt = torch.softmax(torch.einsum("bct,bcs->bts", q * scale, k * scale), dim=-1)
            a = torch.einsum("bts,bcs->bct", wt, v).reshape(bs, -1, length)
            h = h + F.conv1d(a, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
        return h[0, :, 0]

    def latent(self, mels):
        return torch.stack([self.clip(m) for m in mels]).mean(dim=0).to(torch.float64).numpy()



def _t5_bucket(rel, num_buckets=32, max_distance=64):
    """x-transformers / T5 bidirectional bucket of rel = k_pos - q_pos (tortoise arch_util.RelativePositionBias._relative_position_bucket)."""
    import math
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


class TorchDiffusionConditioning:
    """Upstream DiffusionTts.get_conditioning (contextual_embedder) with torch ops: 100-band mel of the reference clips -> the 2048-float
    diffusion conditioning latent. Pins oracle.DiffusionConditioning; no reference code exists (the reference reads the latent as a weight)."""

    def __init__(self, path, dtype=torch.float32):
        from tortoise_cpp_amd import synth_weights as sw
        self.w = {k: torch.from_numpy(v).to(dtype) for k, v in sw.read_ggml(path).items()}
        self.dtype = dtype
        self.blocks = 0
        while "contextual_embedder.%d.norm.weight" % (2 + self.blocks) in self.w:
            self.blocks += 1

    def clip(self, mel):  # [100, T] -> [2048, T2]
        w = self.w
        h = torch.as_tensor(np.asarray(mel)).to(self.dtype)[None]
        h = F.conv1d(h, w["contextual_embedder.0.weight"], w["contextual_embedder.0.bias"], stride=2, padding=1)
        h = F.conv1d(h, w["contextual_embedder.1.weight"], w["contextual_embedder.1.bias"], stride=2, padding=1)
        n = h.shape[-1]
        pos = torch.arange(n)
        bucket = _t5_bucket(pos[None, :] - pos[:, None])  # [q, k]
        for i in range(self.blocks):
            p = "contextual_embedder.%d." % (2 + i)
            y = F.group_norm(h, 32, w[p + "norm.weight"], w[p + "norm.bias"], 1e-5)
            qkv = F.conv1d(y, w[p + "qkv.weight"], w[p + "qkv.bias"])
            bs, width, length = qkv.shape
            ch = width // (3 * 16)
            q, k, v = qkv.reshape(bs * 16, ch * 3, length).split(ch, dim=1)
            scale = 1 / (ch ** 0.25)
            wt = torch.einsum("bct,bcs->bts", q * scale, k * scale)
            bias = w[p + "relative_pos_embeddings.relative_attention_bias.weight"][bucket].permute(2, 0, 1)  # [h, q, k]
            wt = torch.softmax(wt + bias * (ch ** 0.5), dim=-1)
            a = torch.einsum("bts,bcs->bct", wt, v).reshape(bs, -1, length)
            h = h + F.conv1d(a, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
        return h[0]

    def latent(self, mels):
        return torch.cat([self.clip(m) for m in mels], dim=-1).mean(dim=-1).to(torch.float64).numpy()
