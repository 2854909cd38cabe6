"""CLVP candidate re-ranking. The reference has none (main.cpp:6575 keeps candidate 0); upstream tortoise-tts scores
the candidates with CLVP and keeps the best. No upstream weights or fixtures exist offline, so the chain of evidence is:
  torch restatement of the upstream equations (tests/torch_ref.py: TorchCLVP, f64)  ==  numpy oracle (oracle.Clvp)   [CPU, here]
  numpy oracle  ~  HIP engine (tts_load_clvp / tts_clvp_score) on synthetic weights                                  [GPU]
i.e. parity UNPINNED against upstream, pinned between three independent implementations of the same architecture."""
import os

import numpy as np
import pytest
import torch

import torch_ref as TR


@pytest.fixture(scope="session")
def clvp_models(pkg):
    d = os.path.join(os.environ.get("TTS_SYNTH_DIR", "/tmp/tts_synth"), "clvp")
    os.makedirs(d, exist_ok=True)
    from tortoise_cpp_amd import synth_weights as sw
    out = {}
    for name, depth in (("small", 2), ("full", 20)):
        p = os.path.join(d, "ggml-clvp-model-%s.bin" % name)
        if not os.path.exists(p + ".done"):
            sw.write_clvp(p, depth=depth, seed=99 + depth)
            open(p + ".done", "w").write("ok")
        out[name] = p
    return out


def _inputs(seed=0, lens=(40, 57, 13, 200)):
    rs = np.random.RandomState(seed)
    return rs.randint(0, 256, 30).astype(np.int32), [rs.randint(0, 8192, n).astype(np.int32) for n in lens]


def test_clvp_oracle_vs_torch(oracle, clvp_models):
    text, sp = _inputs()
    so = oracle.Clvp(oracle.Model(clvp_models["small"])).score(text, sp)
    t64 = TR.TorchCLVP(clvp_models["small"], torch.float64).score(text, sp)
    t32 = TR.TorchCLVP(clvp_models["small"]).score(text, sp)
    print("CLVP scores: oracle", so, "torch f64", t64)
    assert np.abs(so - t64).max() < 2e-6 and np.abs(t32 - t64).max() < 2e-6
    assert np.ptp(t64) > 1e-3  # the candidates are told apart


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["small", "full"])
def test_clvp_engine_vs_oracle(pkg, oracle, clvp_models, which):
    """fp16-operand GEMMs with f32 accumulation against the f32 oracle: the score is a cosine x e, tolerance 1e-3 absolute (north star; measured
    ~1e-4 at depth 2, ~2-5e-4 at depth 20), and the ranking of candidates whose scores differ by more than that is the oracle's."""
    text, sp = _inputs(seed=3, lens=(40, 57, 13, 200, 301, 7) if which == "small" else (60, 187, 200))
    e = pkg.Engine(0)
    e.load_clvp(clvp_models[which])
    got = e.clvp_score(text, sp)
    again = e.clvp_score(text, sp)
    want = oracle.Clvp(oracle.Model(clvp_models[which])).score(text, sp)
    print("CLVP %s: engine" % which, got, "oracle", want, "max abs diff %.1e" % np.abs(got - want).max())
    assert (got == again).all(), "not deterministic"
    assert np.abs(got - want).max() < 1e-3
    order_o = np.argsort(-want)
    if want[order_o[0]] - want[order_o[1]] > 6e-3:
        assert int(np.argmax(got)) == int(order_o[0])
    # a candidate's score does not depend on who shares the batch
    solo = e.clvp_score(text, [sp[1]])
    assert abs(float(solo[0]) - float(got[1])) < 1e-6
    if which == "full":  # the bench's shape: 16 candidates x 200 codes against a 66-id prompt (printed for DESIGN.md section 3; no gate)
        import time
        rs = np.random.RandomState(9)
        t66, c16 = rs.randint(0, 256, 66).astype(np.int32), [rs.randint(0, 8192, 200).astype(np.int32) for _ in range(16)]
        e.clvp_score(t66, c16)
        t0 = time.time()
        for _ in range(3):
            e.clvp_score(t66, c16)
        print("CLVP re-ranking of 16 candidates x 200 codes, depth 20: %.1f ms per call" % ((time.time() - t0) / 3 * 1e3))
    e.close()


@pytest.mark.gpu
def test_clvp_errors(pkg, clvp_models, small_models):
    e = pkg.Engine(0)
    with pytest.raises(pkg.TtsError, match="tts_load_clvp not called"):
        e.clvp_score(np.array([1, 2], np.int32), [np.array([5, 6], np.int32)])
    with pytest.raises(pkg.TtsError, match="not a CLVP model file"):
        e.load_clvp(small_models + "/ggml-vocoder-model.bin")
    e.load_clvp(clvp_models["small"])
    with pytest.raises(pkg.TtsError, match="out of range"):
        e.clvp_score(np.array([1, 2], np.int32), [np.array([5, 8192], np.int32)])  # the start token is not a speech code
    with pytest.raises(pkg.TtsError, match="out of range"):
        e.clvp_score(np.array([1, 256], np.int32), [np.array([5, 6], np.int32)])
    e.close()
