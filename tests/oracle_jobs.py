"""TEST INFRASTRUCTURE: the long oracle computations of the GPU tests whose inputs do not depend on the engine (the full-depth 80-step loop at the
benchmark's length: 160 full-size oracle forwards at T = 870; the 200-step loop of configs[4]; the oracle's own end-to-end chain of configs[1]; the 80-step ragged pair and the 200-step loop on the small weights). tests/conftest.py
(oracle_bg) starts them in worker processes when a GPU session begins, so that they run on the host's idle cores beside the GPU tests instead of in front of them
(round 6: 790 s -> see profiles/r6_gpu_suite.txt). Each function restates exactly the inputs of the test that uses it; the tests fall back to computing in-line when the
background pool is off (TTS_NO_ORACLE_BG=1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(threads):
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ["OMP_WAIT_POLICY"] = "PASSIVE"
    try:
        os.nice(10)  # the foreground tests' own oracle calls come first
    except OSError:
        pass
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _oracle():
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle as O
    O.build()
    return O


def bench_length_inputs(frames_of):
    L = 200
    lat = np.random.RandomState(31).randn(L, 1024).astype(np.float32)
    noise = np.random.RandomState(6).randn(81, 100 * frames_of(L)).astype(np.float32)
    return lat, noise


def bench_length(models):
    """test_full_size_80_steps_at_bench_length: the oracle's mel"""
    O = _oracle()
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    lat, noise = bench_length_inputs(od.T_of)
    return od.sample(lat, n_steps=80, noise=noise)


def config5_inputs(frames_of, steps=200):
    rs = np.random.RandomState(41)
    for _ in range(2 * 16):
        rs.randn(200, 1024)
    lat = rs.randn(9, 1024).astype(np.float32)
    noise = np.random.RandomState(8).randn(steps + 1, 100 * frames_of(9)).astype(np.float32)
    return lat, noise


def config5(models):
    """test_config5_shape_200_steps: the oracle's mel of the short candidate over the 200-step schedule"""
    O = _oracle()
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    lat, noise = config5_inputs(od.T_of)
    return od.sample(lat, n_steps=200, noise=noise)


def small_pair_inputs(frames_of):
    """tests/test_diffusion_gpu.py::test_sampling_loop_matches_oracle and tests/test_latency_mode_gpu.py::test_ragged_pair_latency_mode: two candidates of different length"""
    lats = [np.random.RandomState(s).randn(L, 1024).astype(np.float32) for L, s in ((20, 1), (9, 2))]
    rs = np.random.RandomState(3)
    return lats, [rs.randn(81, 100 * frames_of(len(l))).astype(np.float32) for l in lats]


def small_200_inputs(frames_of):
    """tests/test_diffusion_gpu.py::test_sampling_loop_200_steps_config5"""
    return np.random.RandomState(3).randn(9, 1024).astype(np.float32), np.random.RandomState(8).randn(201, 100 * frames_of(9)).astype(np.float32)


def loop80_inputs(frames_of, L=12):
    """tests/test_fullsize_gpu.py::test_sampling_loop_80_steps, the latency-mode loop and the ablation ladder of tests/test_diffusion_gpu.py: one problem, small and mid weights"""
    return np.random.RandomState(L).randn(L, 1024).astype(np.float32), np.random.RandomState(5).randn(81, 100 * frames_of(L)).astype(np.float32)


def loop80(models):
    O = _oracle()
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    lat, noise = loop80_inputs(od.T_of)
    return od.sample(lat, n_steps=80, noise=noise)


def small_pair(models, c):
    O = _oracle()
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    lats, noise = small_pair_inputs(od.T_of)
    return od.sample(lats[c], n_steps=80, noise=noise[c])


def small_200(models):
    O = _oracle()
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    lat, noise = small_200_inputs(od.T_of)
    return od.sample(lat, n_steps=200, noise=noise)


def config1(models, voice_path, tokens, S, seed):
    """test_config1_end_to_end: the oracle's whole chain with the RNG stream in the reference's order (AR uniforms -> x_T -> 80 noise vectors -> vocoder noise)"""
    O = _oracle()
    voice = np.fromfile(voice_path, np.float32)
    toks = np.asarray(tokens, np.int32)
    ar = O.AR(O.Model(models + "/ggml-model.bin"))
    rng = O.Rng(seed)
    rc, codes_o, steps_o, _ = ar.generate(toks, voice, 1, rng, S, mask_stop=True)
    L = int(O.trimmed_rows(codes_o[0]))
    lat_o = ar.latents(codes_o, L + 1)[0, :L]
    del ar
    od = O.Diffusion(O.Model(models + "/ggml-diffusion-model.bin"))
    mel_o = od.sample(lat_o, n_steps=80, rng=rng)
    del od
    ov = O.Vocoder(O.Model(models + "/ggml-vocoder-model.bin"))
    au_o = ov.run(mel_o, rng=rng)
    u_final = rng.uniform()
    nz = np.random.RandomState(4).randn(64, mel_o.shape[1] + 10).astype(np.float32)
    return {"rc": rc, "codes_o": codes_o, "steps_o": steps_o, "L": L, "lat_o": lat_o, "mel_o": mel_o, "au_o": au_o, "u_final": u_final, "voc_on_mel_o": ov.run(mel_o, noise=nz)}
