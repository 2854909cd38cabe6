import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import tortoise_cpp_amd_loader  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "models")


def _oracle_bg_wanted(config):
    mexpr = config.getoption("-m") or ""
    return os.path.exists("/dev/kfd") and not os.environ.get("TTS_NO_ORACLE_BG") and "gpu" in mexpr and "not gpu" not in mexpr


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _oracle_bg_wanted(config):
        # The long oracle loops of tests/test_fullsize_gpu.py run in four worker processes beside the GPU tests (oracle_bg below). The host's cores are split so that the
        # OpenMP teams never oversubscribe them (a first version let every team take all cores: the spinning teams slowed the foreground oracle calls 20 x): 3/8 of the
        # cores for this process, 1/8 for each of the four workers, passive waiting everywhere. Must be set before liboracle.so (libgomp) is loaded.
        cores = os.cpu_count() or 8
        os.environ.setdefault("OMP_NUM_THREADS", str(max(4, cores * 3 // 8)))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


@pytest.fixture(scope="session")
def pkg():
    return tortoise_cpp_amd_loader.load()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def voice():
    return np.fromfile(os.path.join(MODELS, "mol.bin"), np.float32)


def _synth_dir(pkg, name, **kw):
    """Synthetic weights (reference file format), generated once per machine under /tmp."""
    d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), name)
    stamp = os.path.join(d, ".done")
    if not os.path.exists(stamp):
        from tortoise_cpp_amd import synth_weights as sw
        sw.write_all(d, **kw)
        open(stamp, "w").write("ok")
    return d


@pytest.fixture(scope="session")
def small_models(pkg):
    # 2 GPT-2 layers, 1+1+1+1 diffusion blocks: same tensor shapes, seconds on the CPU oracle
    return _synth_dir(pkg, "small", ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=4321)


@pytest.fixture(scope="session")
def mid_models(pkg):
    return _synth_dir(pkg, "mid", ar_layers=6, diff_main=3, diff_tail=1, diff_integ=1, diff_lc=2, seed=777)


@pytest.fixture(scope="session")
def trained_mid_models(pkg):
    """mid-depth DIFFUSION weights with trained-network statistics (round 6): normalisation gains 0.1 .. 12, heavy-tailed weights with 30-sigma outliers, relative-position
    biases of +-10, unit-scale embeddings (tortoise.cpp_amd/synth_weights.py: _TrainedGen)"""
    d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "trained_mid")
    if not os.path.exists(os.path.join(d, ".done")):
        from tortoise_cpp_amd import synth_weights as sw
        os.makedirs(d, exist_ok=True)
        sw.write_diffusion(os.path.join(d, "ggml-diffusion-model.bin"), 3, 1, 1, 2, seed=9001, stats="trained")
        open(os.path.join(d, ".done"), "w").write("ok")
    return d


@pytest.fixture(scope="session")
def full_models(pkg):
    """The benchmark's own weights: 30 GPT-2 layers, 4 + 3 + 10 + 3 diffusion blocks, UnivNet (2.4 GB, ~1 min to generate).
    Same directory and seed as bench.py, so the weights bench.py times are the weights these tests check."""
    d = os.environ.get("TTS_BENCH_MODELS", "/tmp/tts_bench_models")
    stamp = os.path.join(d, ".done")
    if not os.path.exists(stamp):
        from tortoise_cpp_amd import synth_weights as sw
        sw.write_all(d, seed=1234)
        open(stamp, "w").write("ok")
    return d


@pytest.fixture(scope="session", autouse=True)
def oracle_bg(request):
    """{name: Future}: the long engine-independent oracle computations of tests/test_fullsize_gpu.py (tests/oracle_jobs.py), started in worker processes at the start of
    a GPU session; None when there is no GPU here, when GPU tests are not selected, or with TTS_NO_ORACLE_BG=1 (the tests then compute in-line)."""
    if not _oracle_bg_wanted(request.config):
        yield None
        return
    if not any("test_fullsize_gpu" in item.nodeid for item in request.session.items):
        yield None
        return
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    import oracle_jobs
    models = request.getfixturevalue("full_models")
    request.getfixturevalue("oracle")  # liboracle.so is built before the workers look for it
    small = request.getfixturevalue("small_models")
    threads = max(2, (os.cpu_count() or 8) // 8)
    ex = ProcessPoolExecutor(4, mp_context=multiprocessing.get_context("spawn"), initializer=oracle_jobs._init, initargs=(threads,))
    # four workers: the longest job and the three small-weight loops (tests/test_diffusion_gpu.py asks for them a minute or two into the session) start at once, the
    # other two full-size jobs (wanted after ~4 minutes) follow on the workers the small ones free
    futs = {"bench_length": ex.submit(oracle_jobs.bench_length, models)}
    futs.update({"small_pair0": ex.submit(oracle_jobs.small_pair, small, 0), "small_pair1": ex.submit(oracle_jobs.small_pair, small, 1),
                 "small_200": ex.submit(oracle_jobs.small_200, small)})
    futs.update({"loop80:" + d: ex.submit(oracle_jobs.loop80, d) for d in (request.getfixturevalue("mid_models"), small)})
    futs.update({"config5": ex.submit(oracle_jobs.config5, models),
                 "config1": ex.submit(oracle_jobs.config1, models, os.path.join(MODELS, "mol.bin"), [int(t) for t in DEFAULT_TOKENS], 40, 0)})
    yield futs
    ex.shutdown(wait=False, cancel_futures=True)


@pytest.fixture(scope="session")
def oracle_sample(oracle, oracle_bg):
    """run(model_dir, latents, noise, steps, bg=None) -> the oracle's mel of one sampling loop with explicit noise, computed once per session: several GPU tests hold
    different engine settings (arithmetic modes, latency mode, the ablation ladder) against the oracle on the SAME problem. `bg` names the background job
    (tests/oracle_jobs.py) that computes this very problem when the pool is on."""
    import zlib
    memo, models = {}, {}

    def run(model_dir, lat, noise, steps, bg=None):
        key = (model_dir, steps, zlib.crc32(np.ascontiguousarray(lat).tobytes()), zlib.crc32(np.ascontiguousarray(noise).tobytes()))
        if key not in memo:
            if bg is None and steps == 80 and lat.shape[0] == 12:
                bg = "loop80:" + model_dir
            if oracle_bg is not None and bg in oracle_bg:
                memo[key] = oracle_bg[bg].result()
            else:
                if model_dir not in models:
                    models[model_dir] = oracle.Diffusion(oracle.Model(model_dir + "/ggml-diffusion-model.bin"))
                memo[key] = models[model_dir].sample(lat, n_steps=steps, noise=noise)
        return memo[key]
    return run


@pytest.fixture(scope="session")
def engine(pkg):
    eng = pkg.Engine(0)
    yield eng
    eng.close()


def loop_gate(kind, attn_f32=False):
    """Gate (max abs on the +-1 mel range) of an 80-/200-step sampling-loop comparison between the engine and the oracle on the `kind` = small | mid | full
    weights (record: tests/golden/parity_floor.json, measured on the CPU with torch by tools/regen_parity_floor.py / tests/test_parity_floor.py).

    ONE gate for both arithmetic modes of the diffusion stage since round 5: the distance between the oracle and a torch-f32 evaluation of the
    reference's graph (`oracle_vs_t32`: two correct f32 evaluations, measured on the class's samples AND on the very problems the GPU tests run):
    max(1e-3 [north star], 1.5 x the largest recorded maximum) — the maximum over 100 x T chaotic values moves by +-30 % under an arithmetic-neutral change, see
    loop_gate_mean for the stable statistic. The trajectory is chaotic at the 1e-3 level (fp16 rounding of every convolution operand), so the distance between two
    correct f32 evaluations IS the tolerance an engine can be held to. The reference's own gate is 0.01 (main.cpp:6223).
      mode 0  the default: q, k, v, softmax numerators and attention output are fp16 MFMA operands; proj_out multiplies on its F32 weight as a split pair and the
              latent conditioner (once per utterance) runs in reference precision — the two roundings that are the SAME perturbation at every step; what is left
              averages out over the loop (tests/golden/parity_floor.json "ablation", profiles/r5_attention_ablation.txt);
      mode 1  option attn_f32: every product of the AttentionBlock on split-fp16 pairs + exact SiLU (main.cpp:3848-3875's F32 arithmetic).
    (Rounds 1-4 gated mode 0 at max(1e-3, 2 x an f32 emulation of the engine's own all-fp16 block): 2.8e-3 .. 5.0e-3. That block is still there as option
    attn_proj_f16 = 1 / lc_attn_f32 = 0 for A/B and is not gated.)"""
    import json
    rec = json.load(open(os.path.join(GOLDEN, "parity_floor.json")))[kind]
    return float(rec["gate_f32"])


MEAN_RATIO_MAX = 1.30
"""Largest allowed ratio between the engine's mean abs distance from the oracle and the distance a torch-f32 evaluation of the reference's graph keeps from the oracle on
the SAME problem (round 6, VERDICT r5 item 3c). Measured ratios: default arithmetic 1.09 .. 1.24 (the four fp16 activation roundings of the AttentionBlock, which average
out over the loop but not to nothing), option attn_f32 1.04 .. 1.17, option latency_mode the same as the mode it runs in. The floor itself is one sample of a
distribution whose relative sigma over (latents, noise) seeds is 2.5 % (tests/golden/parity_floor.json "seed_distribution", tools/regen_parity_floor.py --seeds)."""


def loop_gate_mean(kind, problem=None):
    """Gate on the MEAN abs distance from the oracle (a stable statistic, unlike the maximum). `problem` = the name under which tools/regen_parity_floor.py --problems
    recorded the torch-f32-vs-oracle distance of exactly these inputs: the gate is MEAN_RATIO_MAX x that problem's own floor. Without a recorded problem: MEAN_RATIO_MAX x
    max(largest recorded problem mean, mu + 3 sigma of the class's seed distribution) (`gate_f32_mean` in the record). Rounds 4-5 used 1.25 x the largest recorded mean of
    the class for every problem and passed with 2-3 % of air on the problems that defined it."""
    import json
    rec = json.load(open(os.path.join(GOLDEN, "parity_floor.json")))[kind]
    if problem is not None:
        return MEAN_RATIO_MAX * float(rec["problems"][problem]["oracle_vs_t32_mean"])
    return float(rec["gate_f32_mean"])


def check_loop(err, kind, mode, what="", problem=None):
    """assert the loop gates of one comparison (err = |engine - oracle|), the same for both modes; returns the text for the log (incl. the engine / floor ratio when the
    problem's own floor is on record)"""
    g, gm = loop_gate(kind, mode), loop_gate_mean(kind, problem)
    assert err.max() <= g, (what, mode, float(err.max()), float(err.mean()), g)
    assert err.mean() <= gm, (what, mode, float(err.mean()), gm)
    ratio = " = %.2f x this problem's f32-vs-f32 mean (limit %.2f)" % (err.mean() / (gm / MEAN_RATIO_MAX), MEAN_RATIO_MAX) if problem else ""
    return "max %.2e (gate %.2e) mean %.2e (gate %.2e)%s" % (err.max(), g, err.mean(), gm, ratio)


ATTN_MODES = ((0, "default: fp16 attention operands, F32-accurate proj_out weight, f32 conditioner"), (1, "reference precision: attn_f32"))


DEFAULT_TOKENS = np.array([255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0], np.int32)
