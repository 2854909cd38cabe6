"""GPU parity of the vocoder stage against the oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mel(T, seed):
    return np.clip(np.random.RandomState(seed).randn(100, T) * 0.5, -1, 1).astype(np.float32)


@pytest.mark.parametrize("T", [1, 20, 57])
def test_vocoder_matches_oracle(engine, oracle, small_models, T):
    engine.load(vocoder=small_models + "/ggml-vocoder-model.bin")
    ov = oracle.Vocoder(oracle.Model(small_models + "/ggml-vocoder-model.bin"))
    mel = _mel(T, T)
    noise = np.random.RandomState(9).randn(64, T + 10).astype(np.float32)
    got = engine.vocoder([mel], noise=[noise])[0]
    want = ov.run(mel, noise=noise)
    assert got.shape == want.shape == ((T + 10) * 256 - 6,)
    err = np.abs(got - want).max() / np.abs(want).max()
    print("vocoder T=%d rel err %.2e" % (T, err))
    assert err < 1e-3, err


def test_vocoder_batch_and_reference_noise(engine, oracle, small_models):
    """Two candidates of different length in one batch; noise drawn from the ctx RNG in the reference's order."""
    engine.load(vocoder=small_models + "/ggml-vocoder-model.bin")
    ov = oracle.Vocoder(oracle.Model(small_models + "/ggml-vocoder-model.bin"))
    mels = [_mel(33, 1), _mel(12, 2)]
    engine.seed(42)
    got = engine.vocoder(mels)
    rng = oracle.Rng(42)
    for c, m in enumerate(mels):
        want = ov.run(m, rng=rng)
        err = np.abs(got[c] - want).max() / np.abs(want).max()
        assert err < 1e-3, (c, err)
    assert engine.rng_uniform() == rng.uniform()


@pytest.mark.parametrize("T,chunk", [(150, 40), (57, 16), (9, 5), (300, 128)])
def test_chunked_vocoder_reproduces_the_whole_utterance(engine, small_models, T, chunk):
    """tts_vocoder_chunk (streaming): any partition of the T + 10 frames into chunks gives the samples of the one-shot call —
    the halo covers the stack's receptive field, windows at a sequence end keep the reference's boundary treatment."""
    engine.load(vocoder=small_models + "/ggml-vocoder-model.bin")
    rs = np.random.RandomState(T)
    mel = np.clip(rs.randn(100, T) * 0.5, -1, 1).astype(np.float32)
    nz = rs.randn(64, T + 10).astype(np.float32)
    whole = engine.vocoder([mel], noise=[nz])[0]
    parts = [engine.vocoder_chunk(mel, nz, f0, min(chunk, T + 10 - f0)) for f0 in range(0, T + 10, chunk)]
    got = np.concatenate(parts)
    assert got.shape == whole.shape
    err = float(np.abs(got - whole).max() / np.abs(whole).max())
    print("chunked vocoder T=%d chunk=%d: %d chunks, rel diff vs one-shot %.1e" % (T, chunk, len(parts), err))
    assert err <= 1e-6
