"""Host mel front-end of the voice-conditioning encoders (tts_host_mel_diffusion100 / tts_host_mel_voice80). No reference counterpart (the
reference has no audio input); pinned here against torch.stft (centre / reflect / periodic Hann — what torchaudio and upstream's TacotronSTFT
compute) and an independent vectorised restatement of the librosa (Slaney) and torchaudio (HTK + Slaney norm) mel filterbanks."""
import numpy as np
import pytest
import torch


def _fb(n_mels, sr, f_min, f_max, htk, n_fft=1024):
    def h2m(f):
        f = np.asarray(f, np.float64)
        if htk:
            return 2595.0 * np.log10(1.0 + f / 700.0)
        lin = f / (200.0 / 3)
        log = 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0)
        return np.where(f >= 1000.0, log, lin)

    def m2h(m):
        m = np.asarray(m, np.float64)
        if htk:
            return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)

    pts = m2h(np.linspace(h2m(f_min), h2m(f_max), n_mels + 2))
    freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    d = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    lower, upper = -ramps[:-2] / d[:-1, None], ramps[2:] / d[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    return w * (2.0 / (pts[2:] - pts[:-2]))[:, None]


def _stft_mag(audio):
    x = torch.from_numpy(audio.astype(np.float64))
    s = torch.stft(x, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True,
                   pad_mode="reflect", return_complex=True)
    return s.abs().numpy()


@pytest.mark.parametrize("n", [600, 4096, 24000 + 77])
def test_mel_frontends(pkg, n):
    rs = np.random.RandomState(n)
    t = np.arange(n) / 24000.0
    audio = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t + 1) + 0.05 * rs.randn(n)).astype(np.float32)
    mag = _stft_mag(audio)
    assert mag.shape == (513, n // 256 + 1) and pkg.lib().tts_host_mel_frames(n) == n // 256 + 1
    # diffusion side: magnitude, librosa filterbank, log clamp, tacotron normalisation
    want = np.log(np.maximum(_fb(100, 24000.0, 0.0, 12000.0, False) @ mag, 1e-5))
    got = pkg.host_mel_diffusion100(audio)  # un-normalised: what upstream feeds the diffusion conditioning encoder (do_normalization=False)
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5 * 11.6 and got.min() >= np.float32(np.log(1e-5)) - 1e-5
    want = 2 * ((want + 11.512925148010254) / (2.3143386840820312 + 11.512925148010254)) - 1
    got = pkg.host_mel_diffusion100(audio, normalize=True)  # the [-1, 1] scale of the diffusion stage's output
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5
    # AR side: power, HTK scale with Slaney normalisation, log clamp, per-band norms
    norms = (1.0 + rs.rand(80)).astype(np.float32)
    want = np.log(np.maximum(_fb(80, 22050.0, 0.0, 8000.0, True) @ mag ** 2, 1e-5)) / norms[:, None]
    got = pkg.host_mel_voice80(audio, norms)
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    assert np.abs(pkg.host_mel_voice80(audio) - got * norms[:, None]).max() < 1e-4


def test_mel_frontend_rejects_short_clips(pkg):
    with pytest.raises(pkg.TtsError):
        pkg.host_mel_diffusion100(np.zeros(512, np.float32))  # reflect padding of 512 needs more than 512 samples


def test_mel_filterbank_rows_are_slaney_normalised():
    """each triangle integrates to ~1 over frequency (area normalisation), on both mel scales"""
    for htk, sr, fmax, nm in ((False, 24000.0, 12000.0, 100), (True, 22050.0, 8000.0, 80)):
        fb = _fb(nm, sr, 0.0, fmax, htk)
        area = fb.sum(axis=1) * (sr / 1024)
        assert np.abs(area[5:] - 1.0).max() < 0.12  # the discrete sum of a narrow triangle is only roughly its area


def test_make_voice_wav_reader_and_resampler(tmp_path, pkg):
    """tools/make_voice.py's host side: float32 and PCM16 WAV files (the reference writes float32, main.cpp:4821-4868), stereo down-mix,
    rational resampling to the two encoder rates."""
    import importlib.util
    import os
    import struct
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("make_voice", os.path.join(ROOT, "tools", "make_voice.py"))
    mv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mv)
    t = np.arange(8000) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    p32 = str(tmp_path / "f32.wav")
    assert pkg.write_wav(p32, x, 16000) == 0
    y, rate = mv.read_wav(p32)
    assert rate == 16000 and (y == x).all()
    # PCM16 stereo, written by hand
    pcm = (np.stack([x, x], axis=1) * 32767).astype(np.int16)
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 16000, 16000 * 4, 4, 16) + b"data" + struct.pack("<I", len(data))
    p16 = str(tmp_path / "pcm16.wav")
    open(p16, "wb").write(hdr + data)
    y16, rate = mv.read_wav(p16)
    assert rate == 16000 and y16.shape == x.shape and np.abs(y16 - x).max() < 1e-4
    for target in (22050, 24000):
        z = mv.resample(x, 16000, target)
        assert abs(len(z) - len(x) * target / 16000) <= 1
        k = np.argmax(np.abs(np.fft.rfft(z * np.hanning(len(z)))))
        assert abs(k * target / len(z) - 440) < 3  # the tone is still at 440 Hz
