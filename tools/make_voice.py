#!/usr/bin/env python3
"""Voice files from reference clips, on the MI355X engine (the reference's README.md:54-72 does this offline in PyTorch).

  python tools/make_voice.py --clips a.wav b.wav --conditioning-model ggml-conditioning-model.bin \\
      --diffusion-conditioning-model ggml-diffusion-conditioning-model.bin [--mel-norms mel_norms.npy] --out voices/me

writes voices/me.bin (1024 f32: `tortoise --voice`) and voices/me.diffusion.bin (2048 f32: `tortoise --diffusion-latent`).
Clips: mono WAV, PCM16 or float32, any sample rate (resampled to 22.05 kHz for the AR encoder and 24 kHz for the diffusion encoder with
scipy.signal.resample_poly). The two encoder containers come from upstream tortoise-tts checkpoints through tools/convert_weights.py
(--ar autoregressive.pth --conditioning-encoder / --diffusion-conditioning-encoder diffusion_decoder.pth); --mel-norms is upstream's
data/mel_norms.pth saved as 80 floats (.npy or raw f32): without it the 80-band mel is not divided by the per-band norms.
Clip length follows upstream's get_conditioning_latents: the AR encoder sees exactly 132 300 samples at 22.05 kHz (format_conditioning: zero-padded,
or cropped — upstream crops at a random offset; here at --crop-offset, default 0), the diffusion encoder exactly 102 400 samples at 24 kHz
(pad_or_truncate: zero-padded or the first 102 400) and the UN-normalised log-mel (wav_to_univnet_mel(..., do_normalization=False));
--full-clips feeds whole clips instead."""
import argparse
import os
import struct
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_wav(path):
    b = open(path, "rb").read()
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        sys.exit("%s: not a RIFF/WAVE file" % path)
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        tag, size = b[pos:pos + 4], struct.unpack("<I", b[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b[pos + 8:pos + 24])
        elif tag == b"data":
            data = b[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        sys.exit("%s: fmt / data chunk missing" % path)
    tag, channels, rate, _, _, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(data, np.int16).astype(np.float32) / 32768.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(data, np.float32).copy()
    else:
        sys.exit("%s: PCM16 or float32 expected (format tag %d, %d bits)" % (path, tag, bits))
    if channels > 1:
        x = x.reshape(-1, channels).mean(axis=1)
    return x.astype(np.float32), rate


def resample(x, rate, target):
    if rate == target:
        return x
    from scipy.signal import resample_poly
    fr = Fraction(target, rate)
    return resample_poly(x, fr.numerator, fr.denominator).astype(np.float32)


def pad_or_crop(x, length, offset=0):
    """upstream pad_or_truncate / format_conditioning: zero-pad at the end, or keep `length` samples from `offset`"""
    if len(x) < length:
        return np.concatenate([x, np.zeros(length - len(x), np.float32)])
    offset = max(0, min(offset, len(x) - length))
    return x[offset:offset + length]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--clips", nargs="+", required=True)
    ap.add_argument("--conditioning-model", required=True)
    ap.add_argument("--diffusion-conditioning-model", required=True)
    ap.add_argument("--mel-norms")
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--crop-offset", type=int, default=0, help="first sample (22.05 kHz) of the AR encoder's 132 300-sample window in clips longer than that")
    ap.add_argument("--full-clips", action="store_true", help="do not pad / crop the clips to upstream's 132 300 / 102 400 samples")
    a = ap.parse_args()
    import tortoise_cpp_amd_loader
    pkg = tortoise_cpp_amd_loader.load()
    norms = None
    if a.mel_norms:
        norms = np.load(a.mel_norms) if a.mel_norms.endswith(".npy") else np.fromfile(a.mel_norms, np.float32)
        norms = np.asarray(norms, np.float32).reshape(80)
    clips = [read_wav(p) for p in a.clips]
    fit = (lambda x, n, off=0: x) if a.full_clips else pad_or_crop
    mel80 = [pkg.host_mel_voice80(fit(resample(x, r, 22050), 132300, a.crop_offset), norms) for x, r in clips]
    mel100 = [pkg.host_mel_diffusion100(fit(resample(x, r, 24000), 102400), normalize=False) for x, r in clips]
    e = pkg.Engine(a.device)
    e.load_voice_encoder(a.conditioning_model)
    e.load_diffusion_conditioning_encoder(a.diffusion_conditioning_model)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    e.voice_latent(mel80).astype(np.float32).tofile(a.out + ".bin")
    e.diffusion_conditioning_latent(mel100).astype(np.float32).tofile(a.out + ".diffusion.bin")
    print("%s.bin (1024 f32), %s.diffusion.bin (2048 f32) from %d clip(s)" % (a.out, a.out, len(clips)))


if __name__ == "__main__":
    main()
