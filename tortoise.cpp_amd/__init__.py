"""tortoise.cpp_amd — ctypes binding of libtortoise_mi355x.so (the C ABI in include/tortoise_mi355x.h).

This is plumbing only: every stage runs in hand-written HIP kernels inside the shared library. There
is no CPU / PyTorch fallback — importing works without a GPU (for the symbol-export test), but
`Engine()` raises if the library or a HIP device is missing.

Import name: the directory is literally `tortoise.cpp_amd/`; use tortoise_cpp_amd_loader.load().
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTS_LIB_PATH") or os.path.join(HERE, "libtortoise_mi355x.so")  # TTS_LIB_PATH: developer A/B of another build of the same sources
HEADER = os.path.join(os.path.dirname(HERE), "include", "tortoise_mi355x.h")
VOCAB_MEL = 8194
DMODEL = 1024
AR_MASK_STOP = 1
AR_RETIRE = 2
NOISE_REFERENCE, NOISE_DEVICE = 0, 1

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the library + CLI (in-tree, cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", HERE, "-j8", "all"], stdout=None if verbose else subprocess.DEVNULL)


class TtsError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TtsError("libtortoise_mi355x.so is not built (run __graft_entry__.build()); there is no fallback path")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    sig = {
        "tts_version": (ci, []), "tts_create": (vp, [ci]), "tts_destroy": (None, [vp]), "tts_last_error": (C.c_char_p, [vp]),
        "tts_set_option": (ci, [vp, C.c_char_p, C.c_double]),
        "tts_load_ar": (ci, [vp, C.c_char_p]), "tts_load_diffusion": (ci, [vp, C.c_char_p]),
        "tts_load_vocoder": (ci, [vp, C.c_char_p]), "tts_load_clvp": (ci, [vp, C.c_char_p]),
        "tts_load_diffusion_conditioning_encoder": (ci, [vp, C.c_char_p]), "tts_diffusion_conditioning_latent": (ci, [vp, _f32p, _i32p, ci, _f32p]),
        "tts_set_diffusion_conditioning_latent": (ci, [vp, _f32p]),
        "tts_load_voice_encoder": (ci, [vp, C.c_char_p]), "tts_voice_latent": (ci, [vp, _f32p, _i32p, ci, _f32p]),
        "tts_clvp_score": (ci, [vp, _i32p, ci, _i32p, _i32p, ci, ci, _f32p]), "tts_ar_layers": (ci, [vp]), "tts_diffusion_layers": (ci, [vp]),
        "tts_seed": (None, [vp, C.c_uint32]), "tts_rng_load_state": (ci, [vp, C.c_char_p]), "tts_rng_save_state": (ci, [vp, C.c_char_p]),
        "tts_rng_uniform": (cf, [vp]), "tts_rng_normal": (None, [vp, _f32p, C.c_int64]),
        "tts_tokenizer_load": (ci, [vp, C.c_char_p]), "tts_tokenize": (ci, [vp, C.c_char_p, _i32p, ci]),
        "tts_ar_begin": (ci, [vp, _i32p, ci, _f32p, ci, ci]), "tts_ar_prefill": (ci, [vp, _f32p]),
        "tts_ar_step": (ci, [vp, _i32p, ci, _f32p]), "tts_ar_latents": (ci, [vp, _i32p, ci, ci, _f32p]),
        "tts_sample": (ci, [vp, _f32p, _i32p, ci, ci, _i32p]),
        "tts_ar_step_sample": (ci, [vp, _i32p, ci, C.c_uint, _i32p]), "tts_ar_topk_fallbacks": (ci, [vp]), "tts_diffusion_time_mlp_retries": (ci, [vp]), "tts_diffusion_fp16_check": (ci, [vp, C.POINTER(C.c_int64)]),
        "tts_device_numa_node": (ci, [vp, C.c_char_p, ci]), "tts_pin_to_device_numa_node": (ci, [vp]),
        "tts_host_sample_row": (ci, [_f32p, _i32p, ci, cf]), "tts_host_sample_prefiltered": (ci, [_f32p, _i32p, ci, cf, ci]),
        "tts_autoregressive": (ci, [vp, _i32p, ci, _f32p, ci, ci, C.c_uint, _i32p, _i32p, vp, _i32p]),
        "tts_ar_stop_status": (ci, [vp, _i32p, ci]), "tts_ar_set_stop_schedule": (ci, [vp, C.c_void_p, ci]),
        "tts_diffusion_frames": (ci, [ci]),
        "tts_diffusion_forward": (ci, [vp, _f32p, ci, _f32p, ci, ci, _f32p]),
        "tts_diffusion": (ci, [vp, _f32p, _i32p, ci, ci, vp, ci, _f32p]),
        "tts_vocoder_samples": (ci, [ci]),
        "tts_vocoder": (ci, [vp, _f32p, _i32p, ci, vp, ci, _f32p]),
        "tts_vocoder_chunk": (ci, [vp, _f32p, ci, _f32p, ci, ci, _f32p, C.POINTER(ci)]),
        "tts_write_wav": (ci, [C.c_char_p, _f32p, C.c_int64, ci]),
        "tts_host_schedule": (ci, [ci, _i32p] + [_f32p] * 7), "tts_host_timestep_embedding": (None, [ci, _f32p]),
        "tts_host_rel_bucket": (ci, [ci, ci]), "tts_host_pad_codes": (ci, [_i32p, ci, _i32p]), "tts_host_trimmed_rows": (ci, [_i32p]),
        "tts_host_fp8_e4m3": (C.c_uint8, [cf]), "tts_host_mel_frames": (ci, [C.c_int64]),
        "tts_host_mel_diffusion100": (ci, [_f32p, C.c_int64, ci, _f32p]), "tts_host_mel_voice80": (ci, [_f32p, C.c_int64, vp, _f32p]),
        "tts_prof_reset": (ci, [vp, ci]), "tts_prof_get": (ci, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def header_symbols():
    """Function names declared in include/tortoise_mi355x.h (for the export test)."""
    import re
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tts_[a-z0-9_]+)\s*\(", txt)))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One tts_ctx on one GPU. Mirrors the reference's stage drivers (autoregressive / diffusion / vocoder)."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = self.L.tts_create(device)
        if not self.h:
            raise TtsError("tts_create(%d) failed: no usable HIP device (this engine has no CPU path)" % device)

    def close(self):
        if getattr(self, "h", None):
            self.L.tts_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc < 0:
            raise TtsError("%s (status %d)" % (self.L.tts_last_error(self.h).decode(), rc))
        return rc

    # ---- setup ----
    def set_option(self, key, value):
        self._ck(self.L.tts_set_option(self.h, key.encode(), float(value)))

    def load(self, model_dir=None, ar=None, diffusion=None, vocoder=None):
        if model_dir:
            ar = ar or os.path.join(model_dir, "ggml-model.bin")
            diffusion = diffusion or os.path.join(model_dir, "ggml-diffusion-model.bin")
            vocoder = vocoder or os.path.join(model_dir, "ggml-vocoder-model.bin")
        if ar:
            self._ck(self.L.tts_load_ar(self.h, ar.encode()))
        if diffusion:
            self._ck(self.L.tts_load_diffusion(self.h, diffusion.encode()))
        if vocoder:
            self._ck(self.L.tts_load_vocoder(self.h, vocoder.encode()))

    @property
    def ar_layers(self):
        return self.L.tts_ar_layers(self.h)

    def seed(self, s):
        self.L.tts_seed(self.h, s)

    def rng_load_state(self, path):
        self._ck(self.L.tts_rng_load_state(self.h, path.encode()))

    def rng_save_state(self, path):
        self._ck(self.L.tts_rng_save_state(self.h, path.encode()))

    def rng_uniform(self):
        return self.L.tts_rng_uniform(self.h)

    def rng_normal(self, n):
        out = np.empty(n, np.float32)
        self.L.tts_rng_normal(self.h, out, n)
        return out

    def tokenizer_load(self, path):
        return self._ck(self.L.tts_tokenizer_load(self.h, path.encode()))

    def tokenize(self, message):
        out = np.empty(4096, np.int32)
        n = self._ck(self.L.tts_tokenize(self.h, message.encode("utf-8"), out, 4096))
        return out[:n].copy()

    # ---- autoregressive ----
    def ar_begin(self, tokens, voice, B, max_steps):
        self.B = B
        self._ck(self.L.tts_ar_begin(self.h, np.ascontiguousarray(tokens, np.int32), len(tokens),
                                     np.ascontiguousarray(voice, np.float32), B, max_steps))

    def ar_prefill(self):
        out = np.empty((self.B, VOCAB_MEL), np.float32)
        self._ck(self.L.tts_ar_prefill(self.h, out.reshape(-1)))
        return out

    def ar_step(self, prev_ids, i):
        out = np.empty((self.B, VOCAB_MEL), np.float32)
        self._ck(self.L.tts_ar_step(self.h, np.ascontiguousarray(prev_ids, np.int32), i, out.reshape(-1)))
        return out

    def ar_step_sample(self, prev_ids, i, mask_stop=False):
        """tts_ar_step + tts_sample(penalty ids = prev_ids) with the sampler's top-k selected on the device."""
        out = np.empty(self.B, np.int32)
        self._ck(self.L.tts_ar_step_sample(self.h, np.ascontiguousarray(prev_ids, np.int32), i, 1 if mask_stop else 0, out))
        return out

    def topk_fallbacks(self):
        return self.L.tts_ar_topk_fallbacks(self.h)

    def numa_node(self):
        """(NUMA node of this context's GPU or -1, that node's cpulist)"""
        buf = C.create_string_buffer(1024)
        node = self.L.tts_device_numa_node(self.h, buf, 1024)
        return node, buf.value.decode()

    def pin_to_numa_node(self):
        """restrict this process (and the sampler threads created later) to the CPUs of the GPU's NUMA node; returns the number of CPUs, 0 = unchanged"""
        return self.L.tts_pin_to_device_numa_node(self.h)

    def fp16_check(self):
        """(non-finite, saturated) fp16 operand values seen since option fp16_check was set (+ the split weights of the loaded diffusion model)."""
        c = (C.c_int64 * 2)()
        self._ck(self.L.tts_diffusion_fp16_check(self.h, c))
        return int(c[0]), int(c[1])

    def time_mlp_retries(self):
        return self.L.tts_diffusion_time_mlp_retries(self.h)

    def ar_latents(self, codes502, n_mel=502):
        codes502 = np.ascontiguousarray(codes502, np.int32).reshape(-1, 502)
        nb = codes502.shape[0]
        out = np.empty((nb, min(500, n_mel), DMODEL), np.float32)
        self._ck(self.L.tts_ar_latents(self.h, codes502.reshape(-1), nb, n_mel, out.reshape(-1)))
        return out

    def sample(self, logits, penalty_ids):
        logits = np.ascontiguousarray(logits, np.float32)
        ids = np.ascontiguousarray(penalty_ids, np.int32).reshape(logits.shape[0], -1)
        out = np.empty(logits.shape[0], np.int32)
        self._ck(self.L.tts_sample(self.h, logits.reshape(-1), ids.reshape(-1), ids.shape[1], logits.shape[0], out))
        return out

    def set_stop_schedule(self, stop_at=None):
        """candidate b of the following autoregressive() calls samples the stop token after stop_at[b] codes (None clears): a reproducible ragged batch"""
        if stop_at is None:
            self._ck(self.L.tts_ar_set_stop_schedule(self.h, None, 0))
        else:
            a = np.ascontiguousarray(stop_at, np.int32)
            self._ck(self.L.tts_ar_set_stop_schedule(self.h, a.ctypes.data_as(C.c_void_p), len(a)))

    def autoregressive(self, tokens, voice, B, max_steps, mask_stop=False, want_latents=True, retire=False):
        """Returns (codes [B,502], rows [B], list of trimmed latents [rows_c,1024], steps)."""
        codes = np.empty((B, 502), np.int32)
        rows = np.empty(B, np.int32)
        steps = np.zeros(1, np.int32)
        lat = np.empty((B * 500, DMODEL), np.float32) if want_latents else None
        self._ck(self.L.tts_autoregressive(self.h, np.ascontiguousarray(tokens, np.int32), len(tokens),
                                           np.ascontiguousarray(voice, np.float32), B, max_steps,
                                           (AR_MASK_STOP if mask_stop else 0) | (AR_RETIRE if retire else 0), codes.reshape(-1), rows, _ptr(lat), steps))
        lats = None
        if want_latents:
            lats, off = [], 0
            for r in rows:
                lats.append(lat[off:off + r].copy())
                off += r
        return codes, rows, lats, int(steps[0])

    def ar_stop_status(self, B):
        """Per candidate of the last autoregressive() call: 1 = ended in a sampled stop token, 0 = cut at max_steps."""
        out = np.zeros(B, np.int32)
        self._ck(self.L.tts_ar_stop_status(self.h, out, B))
        return out

    # ---- voice-conditioning encoder (not in the reference) ----
    def load_voice_encoder(self, path):
        self._ck(self.L.tts_load_voice_encoder(self.h, path.encode()))

    def voice_latent(self, mels):
        """mels: list of [80, T_c] log-mel spectrograms of the reference clips. Returns the 1024-float voice latent (a --voice file)."""
        frames = np.array([m.shape[1] for m in mels], np.int32)
        mel = np.ascontiguousarray(np.concatenate([np.asarray(m, np.float32).reshape(-1) for m in mels]))
        out = np.empty(1024, np.float32)
        self._ck(self.L.tts_voice_latent(self.h, mel, frames, len(mels), out))
        return out

    def load_diffusion_conditioning_encoder(self, path):
        self._ck(self.L.tts_load_diffusion_conditioning_encoder(self.h, path.encode()))

    def diffusion_conditioning_latent(self, mels):
        """mels: list of [100, T_c] mel spectrograms of the reference clips. Returns the 2048-float diffusion conditioning latent."""
        frames = np.array([m.shape[1] for m in mels], np.int32)
        mel = np.ascontiguousarray(np.concatenate([np.asarray(m, np.float32).reshape(-1) for m in mels]))
        out = np.empty(2048, np.float32)
        self._ck(self.L.tts_diffusion_conditioning_latent(self.h, mel, frames, len(mels), out))
        return out

    def set_diffusion_conditioning_latent(self, latent):
        self._ck(self.L.tts_set_diffusion_conditioning_latent(self.h, np.ascontiguousarray(latent, np.float32).reshape(2048)))

    # ---- candidate re-ranking (not in the reference) ----
    def load_clvp(self, path):
        self._ck(self.L.tts_load_clvp(self.h, path.encode()))

    def clvp_score(self, text_ids, codes_list):
        """codes_list: per candidate its sampled mel codes (< 8192, no start / stop token). Returns scores [n_candidates]."""
        lens = np.array([len(c) for c in codes_list], np.int32)
        stride = int(lens.max())
        codes = np.zeros((len(codes_list), stride), np.int32)
        for i, c in enumerate(codes_list):
            codes[i, :len(c)] = c
        out = np.empty(len(codes_list), np.float32)
        self._ck(self.L.tts_clvp_score(self.h, np.ascontiguousarray(text_ids, np.int32), len(text_ids), codes.reshape(-1), lens, len(codes_list), stride, out))
        return out

    # ---- diffusion ----
    @staticmethod
    def frames(L):
        return lib().tts_diffusion_frames(L)

    def diffusion_forward(self, latents, x_t, timestep, conditioning_free):
        latents = np.ascontiguousarray(latents, np.float32).reshape(-1, DMODEL)
        x_t = np.ascontiguousarray(x_t, np.float32)
        T = x_t.shape[1]
        out = np.empty((200, T), np.float32)
        self._ck(self.L.tts_diffusion_forward(self.h, latents.reshape(-1), latents.shape[0], x_t.reshape(-1),
                                              timestep, 1 if conditioning_free else 0, out.reshape(-1)))
        return out

    def diffusion(self, latents_list, n_steps=80, noise=None, noise_mode=NOISE_REFERENCE):
        """latents_list: list of [L_c,1024]. noise: list of [(n_steps+1), 100*T_c] or None. Returns list of mel [100,T_c]."""
        rows = np.array([len(l) for l in latents_list], np.int32)
        lat = np.ascontiguousarray(np.concatenate([np.asarray(l, np.float32).reshape(-1, DMODEL) for l in latents_list]))
        Ts = [self.frames(int(r)) for r in rows]
        mel = np.empty(sum(100 * t for t in Ts), np.float32)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(np.concatenate([np.asarray(n, np.float32).reshape(-1) for n in noise]))
        self._ck(self.L.tts_diffusion(self.h, lat.reshape(-1), rows, len(rows), n_steps, _ptr(nz), noise_mode, mel))
        out, off = [], 0
        for t in Ts:
            out.append(mel[off:off + 100 * t].reshape(100, t).copy())
            off += 100 * t
        return out

    # ---- vocoder ----
    def vocoder(self, mels, noise=None, noise_mode=NOISE_REFERENCE):
        frames = np.array([m.shape[1] for m in mels], np.int32)
        mel = np.ascontiguousarray(np.concatenate([np.asarray(m, np.float32).reshape(-1) for m in mels]))
        ns = [self.L.tts_vocoder_samples(int(t)) for t in frames]
        audio = np.empty(sum(ns), np.float32)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(np.concatenate([np.asarray(n, np.float32).reshape(-1) for n in noise]))
        self._ck(self.L.tts_vocoder(self.h, mel, frames, len(frames), _ptr(nz), noise_mode, audio))
        out, off = [], 0
        for n in ns:
            out.append(audio[off:off + n].copy())
            off += n
        return out

    def vocoder_chunk(self, mel, noise, frame0, n_frames):
        """Samples of frames [frame0, frame0 + n_frames) of one utterance (mel [100,T], noise [64,T+10] of the whole utterance)."""
        mel = np.ascontiguousarray(mel, np.float32)
        noise = np.ascontiguousarray(noise, np.float32)
        out = np.empty(n_frames * 256, np.float32)
        n = C.c_int(0)
        self._ck(self.L.tts_vocoder_chunk(self.h, mel.reshape(-1), mel.shape[1], noise.reshape(-1), frame0, n_frames, out, C.byref(n)))
        return out[:n.value].copy()

    # ---- profiling ----
    def prof_reset(self, enable=True):
        self.L.tts_prof_reset(self.h, 1 if enable else 0)

    def prof_get(self, family):
        """(device ms, launches, algorithmic work) of a kernel family since prof_reset."""
        ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
        self.L.tts_prof_get(self.h, family.encode(), C.byref(ms), C.byref(n), C.byref(w))
        return ms.value, n.value, w.value


HOST_SCHED_KEYS = ["max_log", "min_log", "cfk", "sqrt_recip", "sqrt_recipm1", "coef1", "coef2"]


def host_schedule(n_steps):
    """The diffusion driver's respaced schedule and per-step scalars (host arithmetic, no device needed)."""
    tm = np.empty(n_steps, np.int32)
    arrs = [np.empty(n_steps, np.float32) for _ in HOST_SCHED_KEYS]
    rc = lib().tts_host_schedule(n_steps, tm, *arrs)
    if rc:
        raise TtsError("tts_host_schedule failed (%d)" % rc)
    return tm, dict(zip(HOST_SCHED_KEYS, arrs))


def host_timestep_embedding(t):
    out = np.empty(1024, np.float32)
    lib().tts_host_timestep_embedding(int(t), out)
    return out


def host_sample_row(row, ids, uniform):
    """The sampler's pure per-candidate function on a full logits row (host_logic.cpp: sample_one)."""
    ids = np.ascontiguousarray(ids, np.int32)
    return lib().tts_host_sample_row(np.ascontiguousarray(row, np.float32), ids, len(ids), float(uniform))


def host_sample_prefiltered(row, ids, uniform, keep=64):
    """The same from a host restatement of the device prefilter's list (the `keep` largest logits + ties of the smallest). -1: the list
    cannot decide and the engine would fetch the full row."""
    ids = np.ascontiguousarray(ids, np.int32)
    return lib().tts_host_sample_prefiltered(np.ascontiguousarray(row, np.float32), ids, len(ids), float(uniform), keep)


def host_rel_buckets(n):
    L = lib()
    return np.array([[L.tts_host_rel_bucket(i, c) for c in range(n)] for i in range(n)], np.int32)


def host_pad_codes(codes):
    codes = np.ascontiguousarray(codes, np.int32)
    out = np.empty(502, np.int32)
    rc = lib().tts_host_pad_codes(codes, len(codes), out)
    if rc:
        raise TtsError("tts_host_pad_codes failed (%d)" % rc)
    return out


def host_trimmed_rows(codes502):
    return lib().tts_host_trimmed_rows(np.ascontiguousarray(codes502, np.int32))


def host_mel_diffusion100(audio24k, normalize=False):
    """[100, frames] log-mel of 24 kHz audio. normalize=False: log(clamp(mel, 1e-5)), the input of Engine.diffusion_conditioning_latent
    (upstream feeds its contextual_embedder the un-normalised mel); normalize=True: mapped to [-1, 1] like the diffusion stage's output."""
    a = np.ascontiguousarray(audio24k, np.float32)
    frames = lib().tts_host_mel_frames(len(a))
    out = np.empty((100, frames), np.float32)
    rc = lib().tts_host_mel_diffusion100(a, len(a), 1 if normalize else 0, out.reshape(-1))
    if rc < 0:
        raise TtsError("tts_host_mel_diffusion100 failed (%d): the clip must be longer than 512 samples" % rc)
    return out


def host_mel_voice80(audio22k, mel_norms=None):
    """[80, frames] log-mel of 22.05 kHz audio (divided per band by mel_norms if given): the input of Engine.voice_latent."""
    a = np.ascontiguousarray(audio22k, np.float32)
    frames = lib().tts_host_mel_frames(len(a))
    out = np.empty((80, frames), np.float32)
    mn = None if mel_norms is None else np.ascontiguousarray(mel_norms, np.float32).reshape(80)
    rc = lib().tts_host_mel_voice80(a, len(a), _ptr(mn), out.reshape(-1))
    if rc < 0:
        raise TtsError("tts_host_mel_voice80 failed (%d): the clip must be longer than 512 samples" % rc)
    return out


def host_fp8_e4m3(values):
    """OCP fp8 e4m3 codes (uint8) of an array of floats: the quantiser of option ar_weights = 2."""
    L = lib()
    v = np.ascontiguousarray(values, np.float32).reshape(-1)
    return np.array([L.tts_host_fp8_e4m3(float(x)) for x in v], np.uint8).reshape(np.shape(values))


def write_wav(path, samples, rate=24000):
    s = np.ascontiguousarray(samples, np.float32)
    return lib().tts_write_wav(path.encode(), s, len(s), rate)
