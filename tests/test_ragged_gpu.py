"""GPU: a RAGGED batch — every candidate stopped at its own step, as trained weights do (main.cpp:5188-5249: sequences freeze at their first 8193) — equals
each candidate run alone. Random-init weights never sample a stop token, so the stop SCHEDULE of tts_ar_set_stop_schedule forces one per candidate
(TTS_AR_MASK_STOP | TTS_AR_RETIRE); bench.py times the same construction at full size (`ragged_batch`). What goes ragged: decode steps with retired
candidates, the latent pass (per-candidate trimmed rows), the diffusion row space (32 sequences of 16 different lengths, no unconditioned sequence to share),
the vocoder batch."""
import numpy as np
import pytest

from conftest import DEFAULT_TOKENS

pytestmark = pytest.mark.gpu


def run_batch_and_alone(eng, pkg, voice, toks, B, S, stop_at, n_steps, alone):
    eng.set_stop_schedule(stop_at)
    eng.seed(91)
    codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True, retire=True)
    stopped = eng.ar_stop_status(B)
    mels = eng.diffusion(lats, n_steps=n_steps, noise_mode=pkg.NOISE_DEVICE)
    audio = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
    assert steps == max(min(s + 1, S) for s in stop_at)  # the loop ends with the last candidate's stop (or at max_steps)
    for b in range(B):
        n = min(stop_at[b], S)
        assert (codes[b, 1:1 + n] < 8192).all() and stopped[b] == (1 if stop_at[b] < S else 0)
        if stop_at[b] < S:
            assert codes[b, 1 + n] == 8193
    assert len(set(int(r) for r in rows)) == len(set(stop_at)) and len(set(m.shape[1] for m in mels)) > 1
    worst = 0.0
    try:
        for c in alone:
            eng.set_option("rng_shard_offset", c)
            eng.set_option("rng_shard_total", B)
            eng.set_stop_schedule([stop_at[c]])
            eng.seed(91)
            c1, r1, l1, _ = eng.autoregressive(toks, voice, 1, S, mask_stop=True, retire=True)
            assert (c1[0] == codes[c]).all() and r1[0] == rows[c], c
            # AR: identical codes; latents to f32 round-off (the latent pass of the batch covers the LONGEST candidate's rows: other row tiling, and below 32
            # rows the exact-f32 GEMV path instead of the split-fp16 MFMA one — 2^-22 relative either way)
            e_lat = float(np.abs(l1[0] - lats[c]).max() / np.abs(lats[c]).max())
            # diffusion + vocoder: the candidate alone on the BATCH's latents (the sampling loop amplifies a 1e-5 difference of its input a thousandfold, so
            # the stage is compared on equal inputs): its row in the ragged row space must give what it gives alone
            m1 = eng.diffusion([lats[c]], n_steps=n_steps, noise_mode=pkg.NOISE_DEVICE)[0]
            a1 = eng.vocoder([m1], noise_mode=pkg.NOISE_DEVICE)[0]
            dm, da = float(np.abs(m1 - mels[c]).max()), float(np.abs(a1 - audio[c]).max() / np.abs(audio[c]).max())
            print("ragged batch of %d, candidate %d (%d codes, T=%d): alone vs in the batch: codes identical, latents rel %.1e, mel max abs %.1e, audio rel %.1e"
                  % (B, c, stop_at[c], m1.shape[1], e_lat, dm, da))
            assert e_lat <= 1e-4 and dm <= 1e-5 and da <= 1e-5, (c, e_lat, dm, da)
            worst = max(worst, e_lat, dm, da)
    finally:
        eng.set_option("rng_shard_offset", 0)
        eng.set_option("rng_shard_total", 0)
        eng.set_stop_schedule(None)
    return worst


def test_ragged_batch_equals_each_candidate_alone_small(engine, pkg, small_models, voice):
    engine.load(small_models)
    B, S = 6, 24
    run_batch_and_alone(engine, pkg, voice, DEFAULT_TOKENS, B, S, [13, 24, 17, 15, 22, 19], 6, range(B))


def test_ragged_batch_at_bench_shape(pkg, full_models, voice):
    """bench.py's `ragged_batch` pass itself: full-size weights, the 64-token prompt, 16 candidates stopped after 117 .. 192 codes, 80 diffusion steps; three of
    the candidates against their solo runs."""
    eng = pkg.Engine(0)
    eng.load(full_models)
    toks = np.array([255] + [3 + (7 * j) % 250 for j in range(64)] + [0], np.int32)
    B, S = 16, 192
    stop_at = [int(round(S * (0.61 + 0.39 * b / (B - 1)))) for b in range(B)]
    try:
        run_batch_and_alone(eng, pkg, voice, toks, B, S, stop_at, 80, (0, 7, 15))
    finally:
        eng.close()


def test_stop_schedule_errors(engine, pkg, small_models, voice):
    engine.load(ar=small_models + "/ggml-model.bin")
    engine.set_stop_schedule([3, 4, 5])
    try:
        with pytest.raises(pkg.TtsError, match="stop schedule holds 3 candidates"):
            engine.autoregressive(DEFAULT_TOKENS, voice, 2, 8, mask_stop=True, retire=True)
        with pytest.raises(pkg.TtsError, match="before its first code"):
            engine.set_stop_schedule([0, 4])
        # a schedule only acts on TTS_AR_MASK_STOP | TTS_AR_RETIRE calls: a strict / masked call with a (wrong-sized) schedule still set runs untouched
        engine.seed(3)
        codes, rows, _, steps = engine.autoregressive(DEFAULT_TOKENS, voice, 2, 8, mask_stop=True)
        assert steps == 8 and (codes[:, 1:9] < 8192).all()
    finally:
        engine.set_stop_schedule(None)
    engine.seed(3)
    codes2, _, _, steps = engine.autoregressive(DEFAULT_TOKENS, voice, 2, 8, mask_stop=True)
    assert steps == 8 and (codes2 == codes).all()
