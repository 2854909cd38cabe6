"""TEST INFRASTRUCTURE — ctypes bindings for the oracle (oracle/liboracle.so, the CPU restatement of
the reference's hot path) and for oracle/_ref/libref.so (the reference's own ggml-free code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (tortoise.cpp_amd/) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

# Default team size for the TESTS only (the oracle's loops are small; a 256-thread OpenMP team on the GPU box's host is slower than 16).
# bench.py's cpu_baseline does not rely on this: it sets the team size explicitly (4 threads and all host cores) and reports both.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and _ref/libref.so when /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(so)
        for f in os.listdir(HERE) if f.endswith((".cpp", ".h"))
    ):
        subprocess.check_call(["make", "-C", HERE, "-j8", "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/main.cpp"):
        ref = os.path.join(HERE, "_ref", "libref.so")
        if force or not os.path.exists(ref) or os.path.getmtime(os.path.join(HERE, "ref_shim.cpp")) > os.path.getmtime(ref):
            subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        vp = C.c_void_p
        sig = {
            "orc_rng_new": (vp, [C.c_uint32]), "orc_rng_free": (None, [vp]),
            "orc_rng_seed": (None, [vp, C.c_uint32]), "orc_rng_load_state": (C.c_int, [vp, C.c_char_p]),
            "orc_rng_u32": (C.c_uint32, [vp]), "orc_rng_uniform": (C.c_float, [vp]),
            "orc_rng_normal_fill": (None, [vp, _f32p, C.c_int64]),
            "orc_sample": (None, [_f32p, _i32p, C.c_int, C.c_int, vp, _i32p, vp]),
            "orc_buckets": (None, [C.c_int, _i32p]),
            "orc_timestep_embedding": (None, [C.c_int, _f32p]),
            "orc_schedule": (None, [_i32p, C.c_int] + [_f64p] * 7),
            "orc_default_timestep_map": (None, [C.c_int, _i32p]),
            "orc_diffusion_update": (None, [_i32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int]),
            "orc_apply_padding": (None, [_i32p, C.c_int, _i32p]),
            "orc_trimmed_rows": (C.c_int, [_i32p]),
            "orc_tokenizer_new": (vp, [C.c_char_p]), "orc_tokenizer_free": (None, [vp]),
            "orc_tokenizer_vocab_size": (C.c_int, [vp]),
            "orc_tokenize": (C.c_int, [vp, C.c_char_p, _i32p, C.c_int]),
            "orc_set_flags": (None, [C.c_float, C.c_int]), "orc_f16_round": (C.c_float, [C.c_float]),
            "orc_model_load": (vp, [C.c_char_p]), "orc_model_free": (None, [vp]),
            "orc_model_get": (C.c_int, [vp, C.c_char_p, vp, C.c_int64]),
            "orc_ar_new": (vp, [vp]), "orc_ar_free": (None, [vp]), "orc_ar_layers": (C.c_int, [vp]),
            "orc_ar_start": (None, [vp, _i32p, C.c_int, _f32p, C.c_int, C.c_int]),
            "orc_ar_prefill": (None, [vp, _f32p]), "orc_ar_step": (None, [vp, _i32p, C.c_int, _f32p]),
            "orc_ar_latents": (None, [vp, _i32p, C.c_int, C.c_int, _f32p]),
            "orc_autoregressive": (C.c_int, [vp, _i32p, C.c_int, _f32p, C.c_int, vp, C.c_int, C.c_int, _i32p, _i32p, vp]),
            "orc_diff_new": (vp, [vp]), "orc_diff_free": (None, [vp]), "orc_diff_T": (C.c_int, [C.c_int]),
            "orc_diff_code_embedding": (None, [vp, _f32p, C.c_int, C.c_int, _f32p]),
            "orc_diff_forward": (None, [vp, vp, _f32p, C.c_int, C.c_int, _f32p]),
            "orc_diffusion": (None, [vp, _f32p, C.c_int, C.c_int, vp, vp, _f32p]),
            "orc_voc_new": (vp, [vp]), "orc_voc_free": (None, [vp]), "orc_voc_audio_len": (C.c_int, [C.c_int]),
            "orc_denormalize_mel": (None, [_f32p, C.c_int64]),
            "orc_voc_forward": (None, [vp, _f32p, C.c_int, _f32p, _f32p]),
            "orc_vocoder": (None, [vp, _f32p, C.c_int, vp, vp, _f32p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Rng:
    def __init__(self, seed=0):
        self.h = lib().orc_rng_new(seed)

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_rng_free(self.h)
            self.h = None

    def seed(self, s):
        lib().orc_rng_seed(self.h, s)

    def load_state(self, path):
        assert lib().orc_rng_load_state(self.h, path.encode()) == 0

    def u32(self):
        return lib().orc_rng_u32(self.h)

    def uniform(self):
        return lib().orc_rng_uniform(self.h)

    def normal(self, n):
        out = np.empty(n, np.float32)
        lib().orc_rng_normal_fill(self.h, out, n)
        return out


def sample(logits, ids, rng, want_probs=False):
    """logits [B,8194] f32; ids [B,k] int32 (penalised ids). Returns samples[B] (and probs)."""
    logits = np.ascontiguousarray(logits, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    B = logits.shape[0]
    out = np.empty(B, np.int32)
    probs = np.empty((B, 8194), np.float32) if want_probs else None
    lib().orc_sample(logits, ids.reshape(-1), ids.size, B, rng.h, out, _ptr(probs))
    return (out, probs) if want_probs else out


def buckets(n):
    out = np.empty((n, n), np.int32)
    lib().orc_buckets(n, out.reshape(-1))
    return out


def timestep_embedding(t):
    out = np.empty(1024, np.float32)
    lib().orc_timestep_embedding(t, out)
    return out


def default_timestep_map(steps=80):
    out = np.empty(steps, np.int32)
    lib().orc_default_timestep_map(steps, out)
    return out


SCHED_KEYS = ["betas", "acp", "post_logvar", "coef1", "coef2", "sqrt_recip", "sqrt_recipm1"]


def schedule(tm):
    tm = np.ascontiguousarray(tm, np.int32)
    arrs = [np.empty(len(tm), np.float64) for _ in range(7)]
    lib().orc_schedule(tm, len(tm), *arrs)
    return dict(zip(SCHED_KEYS, arrs))


def diffusion_update(tm, t, out_cond, out_uncond, x, noise, T):
    tm = np.ascontiguousarray(tm, np.int32)
    x = np.array(x, np.float32).reshape(-1).copy()
    lib().orc_diffusion_update(tm, len(tm), t, np.ascontiguousarray(out_cond, np.float32).reshape(-1),
                               np.ascontiguousarray(out_uncond, np.float32).reshape(-1), x,
                               np.ascontiguousarray(noise, np.float32).reshape(-1), T)
    return x


def apply_padding(codes):
    codes = np.ascontiguousarray(codes, np.int32)
    out = np.empty(502, np.int32)
    lib().orc_apply_padding(codes, len(codes), out)
    return out


def trimmed_rows(codes502):
    return lib().orc_trimmed_rows(np.ascontiguousarray(codes502, np.int32))


class Tokenizer:
    def __init__(self, json_path):
        self.h = lib().orc_tokenizer_new(json_path.encode())
        assert self.h, json_path

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_tokenizer_free(self.h)
            self.h = None

    def encode(self, msg):
        out = np.empty(4096, np.int32)
        n = lib().orc_tokenize(self.h, msg.encode("utf-8"), out, 4096)
        return out[:n].copy()


def set_flags(gn_eps=1e-6, lut=0):
    lib().orc_set_flags(gn_eps, lut)


class Model:
    def __init__(self, path):
        self.h = lib().orc_model_load(path.encode())
        if not self.h:
            raise IOError("oracle: cannot load " + path)

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_model_free(self.h)
            self.h = None

    def tensor(self, name):
        n = lib().orc_model_get(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, np.float32)
        lib().orc_model_get(self.h, name.encode(), _ptr(out), n)
        return out


class AR:
    """Oracle autoregressive stage (orc_ar.cpp)."""
    V = 8194

    def __init__(self, model):
        self.model = model
        self.h = lib().orc_ar_new(model.h)
        self.B = 0

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_ar_free(self.h)
            self.h = None

    @property
    def n_layers(self):
        return lib().orc_ar_layers(self.h)

    def start(self, tokens, voice, B, max_pos):
        self.B = B
        self._P, self._max_pos = len(tokens) + 2, max_pos
        assert max_pos >= self._P, "cache smaller than the prompt"
        lib().orc_ar_start(self.h, np.ascontiguousarray(tokens, np.int32), len(tokens),
                           np.ascontiguousarray(voice, np.float32), B, max_pos)

    def prefill(self):
        out = np.empty((self.B, self.V), np.float32)
        lib().orc_ar_prefill(self.h, out.reshape(-1))
        return out

    def step(self, toks, i):
        assert self._P + i < self._max_pos, "step %d beyond the cache given to start()" % i  # the C side does not check
        out = np.empty((self.B, self.V), np.float32)
        lib().orc_ar_step(self.h, np.ascontiguousarray(toks, np.int32), i, out.reshape(-1))
        return out

    def latents(self, codes502, n_mel=502):
        codes502 = np.ascontiguousarray(codes502, np.int32).reshape(-1, 502)
        nb = codes502.shape[0]
        n_out = min(500, n_mel)
        out = np.empty((nb, n_out, 1024), np.float32)
        lib().orc_ar_latents(self.h, codes502.reshape(-1), nb, n_mel, out.reshape(-1))
        return out

    def generate(self, tokens, voice, B, rng, max_steps, mask_stop=False):
        codes = np.empty((B, 502), np.int32)
        steps = np.zeros(1, np.int32)
        raw = np.full((B, max_steps), -1, np.int32)
        self.B = B
        rc = lib().orc_autoregressive(self.h, np.ascontiguousarray(tokens, np.int32), len(tokens),
                                      np.ascontiguousarray(voice, np.float32), B, rng.h, max_steps,
                                      1 if mask_stop else 0, codes.reshape(-1), steps, _ptr(raw))
        return rc, codes, int(steps[0]), raw


class Diffusion:
    def __init__(self, model):
        self.model = model
        self.h = lib().orc_diff_new(model.h)

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_diff_free(self.h)
            self.h = None

    @staticmethod
    def T_of(L):
        return lib().orc_diff_T(L)

    def code_embedding(self, latents, T):
        latents = np.ascontiguousarray(latents, np.float32).reshape(-1, 1024)
        out = np.empty((T, 1024), np.float32)
        lib().orc_diff_code_embedding(self.h, latents.reshape(-1), latents.shape[0], T, out.reshape(-1))
        return out

    def forward(self, code_emb, x_t, timestep):
        """x_t [100,T]; returns [200,T]."""
        x_t = np.ascontiguousarray(x_t, np.float32)
        T = x_t.shape[1]
        out = np.empty((200, T), np.float32)
        ce = None if code_emb is None else np.ascontiguousarray(code_emb, np.float32)
        lib().orc_diff_forward(self.h, _ptr(ce), x_t.reshape(-1), T, timestep, out.reshape(-1))
        return out

    def sample(self, latents, n_steps=80, rng=None, noise=None):
        latents = np.ascontiguousarray(latents, np.float32).reshape(-1, 1024)
        L = latents.shape[0]
        T = self.T_of(L)
        mel = np.empty((100, T), np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        lib().orc_diffusion(self.h, latents.reshape(-1), L, n_steps, rng.h if rng else None, _ptr(nz), mel.reshape(-1))
        return mel


class Vocoder:
    def __init__(self, model):
        self.model = model
        self.h = lib().orc_voc_new(model.h)

    def __del__(self):
        if getattr(self, "h", None):
            if lib is not None:  # None while the interpreter shuts down (module globals are cleared before the last objects die)
                lib().orc_voc_free(self.h)
            self.h = None

    def run(self, mel, rng=None, noise=None):
        """mel [100,T] normalised; noise [64,T+10] or drawn from rng. Returns audio."""
        mel = np.ascontiguousarray(mel, np.float32)
        T = mel.shape[1]
        audio = np.empty(lib().orc_voc_audio_len(T), np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        lib().orc_vocoder(self.h, mel.reshape(-1), T, rng.h if rng else None, _ptr(nz), audio)
        return audio


# ---------------------------------------------------------------------------------------------
# The real reference code (only where /root/reference existed at build time).
# ---------------------------------------------------------------------------------------------
_ref = None


class Clvp:
    """CLVP re-ranker, numpy restatement of the UPSTREAM tortoise-tts model (tortoise/models/clvp.py, use_xformers=True;
    its vendored x-transformers Encoder: RMSNorm pre-norm `x / max(|x| d^-1/2, 1e-8) * g`, bias-free q/k/v projections, rotary embedding on
    the first 32 of the 64 head dims of q, k and v (half-split rotate, base 10000), softmax(q k^T / 8) v, to_out with bias, GEGLU feed-forward with
    ff_mult = 2 (erf GELU), final LayerNorm; mean over the sequence, bias-free latent projection, L2 normalise, dot x exp(temperature)).
    The reference has NO CLVP (main.cpp:6575 takes candidate 0), so there is no reference file:line to cite and no fixture: PARITY UNPINNED —
    this class is pinned only against the torch restatement of the same equations (tests/torch_ref.py: TorchCLVP)."""
    ENC = ("text_transformer", "speech_transformer")

    def __init__(self, model, dtype=np.float32):
        self.m, self.dt = model, dtype
        te = model.tensor("text_emb.weight")
        self.dim = te.size // 256
        self.depth = 0
        while True:
            try:
                model.tensor("text_transformer.transformer.attn_layers.layers.%d.0.g" % (2 * self.depth))
            except KeyError:
                break
            self.depth += 1
        self.inner = model.tensor("text_transformer.transformer.attn_layers.layers.0.1.to_q.weight").size // self.dim
        self.heads = self.inner // 64
        self.ff = model.tensor("text_transformer.transformer.attn_layers.layers.1.1.net.3.weight").size // self.dim

    def _t(self, name, *shape):
        return self.m.tensor(name).reshape(shape).astype(self.dt)

    def encode(self, enc, x):
        from scipy.special import erf
        d, H, n = self.dim, self.heads, x.shape[0]
        inv = 1.0 / (10000.0 ** (np.arange(0, 32, 2, dtype=self.dt) / self.dt(32.0)))
        fr = np.outer(np.arange(n, dtype=self.dt), inv).astype(self.dt)
        fr = np.concatenate([fr, fr], axis=-1)
        cs, sn = np.cos(fr), np.sin(fr)

        def rms(t, g):
            nrm = np.sqrt((t * t).sum(-1, keepdims=True)) * self.dt(d ** -0.5)
            return t / np.maximum(nrm, self.dt(1e-8)) * g

        def rope(t):  # [H, n, 64]
            tl = t[..., :32]
            rot = np.concatenate([-tl[..., 16:], tl[..., :16]], axis=-1)
            return np.concatenate([tl * cs + rot * sn, t[..., 32:]], axis=-1)

        for i in range(self.depth):
            a = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i)
            f = "%s.transformer.attn_layers.layers.%d." % (enc, 2 * i + 1)
            y = rms(x, self._t(a + "0.g", d))
            q, k, v = ((y @ self._t(a + "1.%s.weight" % nm, self.inner, d).T).reshape(n, H, 64).transpose(1, 0, 2) for nm in ("to_q", "to_k", "to_v"))
            q, k, v = rope(q), rope(k), rope(v)  # upstream's vendored x-transformers rotates the first 32 dims of q, k AND v
            sc = (q @ k.transpose(0, 2, 1)) * self.dt(0.125)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            att = sc / sc.sum(-1, keepdims=True)
            o = (att @ v).transpose(1, 0, 2).reshape(n, self.inner)
            x = x + o @ self._t(a + "1.to_out.weight", d, self.inner).T + self._t(a + "1.to_out.bias", d)
            y = rms(x, self._t(f + "0.g", d))
            u = y @ self._t(f + "1.net.0.proj.weight", 2 * self.ff, d).T + self._t(f + "1.net.0.proj.bias", 2 * self.ff)
            val, gate = u[:, :self.ff], u[:, self.ff:]
            gl = (gate * self.dt(0.5) * (self.dt(1.0) + erf(gate / np.sqrt(self.dt(2.0))))).astype(self.dt)
            x = x + (val * gl) @ self._t(f + "1.net.3.weight", d, self.ff).T + self._t(f + "1.net.3.bias", d)
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + self.dt(1e-5)) * self._t(enc + ".transformer.norm.weight", d) + self._t(enc + ".transformer.norm.bias", d)

    def latent(self, which, tokens):
        enc = self.ENC[which]
        emb = self._t("text_emb.weight" if which == 0 else "speech_emb.weight", -1, self.dim)
        proj = self._t("to_text_latent.weight" if which == 0 else "to_speech_latent.weight", -1, self.dim)
        z = self.encode(enc, emb[np.asarray(tokens, np.int64)]).mean(0) @ proj.T
        return z / max(float(np.sqrt((z * z).sum())), 1e-12)

    def score(self, text, speech_list):
        zt = self.latent(0, text)
        t = float(np.exp(self.m.tensor("temperature")[0]))
        return np.array([float((zt * self.latent(1, sp)).sum()) * t for sp in speech_list])


class VoiceEncoder:
    """Voice-conditioning encoder, numpy restatement of UPSTREAM tortoise-tts (tortoise/models/autoregressive.py:
    UnifiedVoice.get_conditioning -> ConditioningEncoder(80, 1024, 6 blocks, 16 heads); arch_util.py: AttentionBlock = GroupNorm(32, eps 1e-5)
    -> Conv1d qkv (k = 1) -> QKVAttentionLegacy (channel = head * 192 + {q, k, v}, both q and k scaled by 64^-1/4) -> Conv1d proj_out -> + x;
    output = position 0 of every clip, mean over the clips). The reference only READS the finished 1024-float latent (main.cpp:5179-5184),
    so there is nothing in it to cite or to pin against: PARITY UNPINNED, pinned against the torch restatement (tests/torch_ref.py)."""

    def __init__(self, model, dtype=np.float32):
        self.m, self.dt, self.blocks = model, dtype, 0
        while True:
            try:
                model.tensor("conditioning_encoder.attn.%d.norm.weight" % self.blocks)
            except KeyError:
                break
            self.blocks += 1

    def _t(self, name, *shape):
        return self.m.tensor(name).reshape(shape).astype(self.dt)

    def clip(self, mel):  # [80, T] -> [1024]
        D, H = 1024, 16
        x = np.asarray(mel, self.dt).T  # [T, 80]
        T = x.shape[0]
        h = x @ self._t("conditioning_encoder.init.weight", D, 80).T + self._t("conditioning_encoder.init.bias", D)
        for i in range(self.blocks):
            p = "conditioning_encoder.attn.%d." % i
            g = h.reshape(T, 32, 32)
            mu = g.mean(axis=(0, 2), keepdims=True)
            var = ((g - mu) ** 2).mean(axis=(0, 2), keepdims=True)
            y = ((g - mu) / np.sqrt(var + self.dt(1e-5))).reshape(T, D) * self._t(p + "norm.weight", D) + self._t(p + "norm.bias", D)
            qkv = (y @ self._t(p + "qkv.weight", 3 * D, D).T + self._t(p + "qkv.bias", 3 * D)).reshape(T, H, 3, 64)
            q, k, v = (qkv[:, :, j].transpose(1, 0, 2) for j in range(3))  # [H, T, 64]
            sc = (q * self.dt(64 ** -0.25)) @ (k * self.dt(64 ** -0.25)).transpose(0, 2, 1)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            a = ((sc / sc.sum(-1, keepdims=True)) @ v).transpose(1, 0, 2).reshape(T, D)
            h = h + a @ self._t(p + "proj_out.weight", D, D).T + self._t(p + "proj_out.bias", D)
        return h[0]

    def latent(self, mels):
        return np.mean([self.clip(m).astype(np.float64) for m in mels], axis=0)


class DiffusionConditioning:
    """Diffusion conditioning latent, numpy restatement of UPSTREAM tortoise-tts (tortoise/models/diffusion_decoder.py:
    DiffusionTts.get_conditioning -> contextual_embedder = Conv1d(100, 1024, 3, stride 2, padding 1), Conv1d(1024, 2048, 3, stride 2,
    padding 1), 5 x arch_util.AttentionBlock(2048, 16 heads, relative_pos_embeddings=True): GroupNorm(32, eps 1e-5), qkv (k = 1),
    QKVAttentionLegacy with head dim 128 (q, k scaled by 128^-1/4) + T5 bucket bias (32 buckets, max distance 64, buckets() of this module)
    x sqrt(128), proj_out, + x; mean over the frames of all clips). The reference reads the result as a WEIGHT of ggml-diffusion-model.bin
    (main.cpp:1557-1560): nothing to cite or pin against — PARITY UNPINNED, pinned against the torch restatement (tests/torch_ref.py)."""

    def __init__(self, model, dtype=np.float32):
        self.m, self.dt, self.blocks = model, dtype, 0
        while True:
            try:
                model.tensor("contextual_embedder.%d.norm.weight" % (2 + self.blocks))
            except KeyError:
                break
            self.blocks += 1

    def _t(self, name, *shape):
        return self.m.tensor(name).reshape(shape).astype(self.dt)

    def _conv3s2(self, x, wname, bname, cout, cin):  # x [T, cin] -> [T', cout]
        w, b = self._t(wname, cout, cin, 3), self._t(bname, cout)
        T = x.shape[0]
        To = (T - 1) // 2 + 1
        xp = np.concatenate([np.zeros((1, cin), self.dt), x, np.zeros((2, cin), self.dt)])
        out = np.zeros((To, cout), self.dt)
        for tap in range(3):
            out += xp[tap:tap + 2 * To:2] @ w[:, :, tap].T
        return out + b

    def clip(self, mel):  # [100, T] -> [T2, 2048]
        D, H, dh = 2048, 16, 128
        h = self._conv3s2(np.asarray(mel, self.dt).T, "contextual_embedder.0.weight", "contextual_embedder.0.bias", 1024, 100)
        h = self._conv3s2(h, "contextual_embedder.1.weight", "contextual_embedder.1.bias", D, 1024)
        n = h.shape[0]
        bk = buckets(n).reshape(n, n)  # [query, key], the reference's get_relative_position_buckets (main.cpp:4722-4749) = the upstream T5 rule
        for i in range(self.blocks):
            p = "contextual_embedder.%d." % (2 + i)
            g = h.reshape(n, 32, 64)
            mu = g.mean(axis=(0, 2), keepdims=True)
            var = ((g - mu) ** 2).mean(axis=(0, 2), keepdims=True)
            y = ((g - mu) / np.sqrt(var + self.dt(1e-5))).reshape(n, D) * self._t(p + "norm.weight", D) + self._t(p + "norm.bias", D)
            qkv = (y @ self._t(p + "qkv.weight", 3 * D, D).T + self._t(p + "qkv.bias", 3 * D)).reshape(n, H, 3, dh)
            q, k, v = (qkv[:, :, j].transpose(1, 0, 2) for j in range(3))
            sc = (q * self.dt(dh ** -0.25)) @ (k * self.dt(dh ** -0.25)).transpose(0, 2, 1)
            emb = self._t(p + "relative_pos_embeddings.relative_attention_bias.weight", 32, H)
            sc = sc + emb[bk].transpose(2, 0, 1) * self.dt(dh ** 0.5)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            a = ((sc / sc.sum(-1, keepdims=True)) @ v).transpose(1, 0, 2).reshape(n, D)
            h = h + a @ self._t(p + "proj_out.weight", D, D).T + self._t(p + "proj_out.bias", D)
        return h

    def latent(self, mels):
        return np.concatenate([self.clip(m).astype(np.float64) for m in mels]).mean(axis=0)


def ref():
    global _ref
    if _ref is None:
        build()
        p = os.path.join(HERE, "_ref", "libref.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_uniform.restype = C.c_float
        R.ref_raw_u32.restype = C.c_uint32
        R.ref_normal_fill.argtypes = [_f32p, C.c_int]
        R.ref_process_logits_and_sample.argtypes = [_f32p, _i32p, C.c_int, C.c_int, _i32p, C.c_void_p]
        R.ref_buckets.argtypes = [C.c_int, _i32p]
        R.ref_timestep_embedding.argtypes = [C.c_int, _f32p]
        R.ref_schedule.argtypes = [_i32p, C.c_int] + [_f64p] * 7
        R.ref_diffusion_update.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int] + [C.c_float] * 7 + [C.c_int]
        R.ref_denormalize_mel.argtypes = [_f32p, C.c_int]
        R.ref_apply_padding.argtypes = [_i32p, C.c_int, _i32p]
        R.ref_trim_latents.argtypes = [_f32p, _i32p, C.c_int, _f32p, _i32p]
        R.ref_tokenizer_init.argtypes = [C.c_char_p]
        R.ref_tokenize.argtypes = [C.c_char_p, _i32p, C.c_int]
        R.ref_load_rng_state.argtypes = [C.c_char_p]
        _ref = R
    return _ref
