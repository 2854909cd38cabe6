"""Candidate-parallel sharding on real devices. The single-device test proves what the multi-GPU design rests on: a
context that holds candidates [c0, c0 + b) of a batch of B (options rng_shard_offset / rng_shard_total) produces exactly the
codes, latents, mel and audio those candidates get in the unsharded batch — ids bit-identical (RNG stream partition), device
noise keyed by the global candidate id. The two-device test (skipped on a one-GPU box) runs bench.py's own launch path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DEFAULT_TOKENS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_contexts_reproduce_the_unsharded_batch(pkg, small_models, voice):
    B, S, n_steps = 6, 14, 3
    full = pkg.Engine(0)
    full.load(small_models)
    full.seed(5)
    codes, rows, lats, _ = full.autoregressive(DEFAULT_TOKENS, voice, B, S, mask_stop=True)
    mels = full.diffusion(lats, n_steps=n_steps, noise_mode=pkg.NOISE_DEVICE)
    audio = full.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
    full.close()
    for G in (2, 3):
        b = B // G
        for r in range(G):
            e = pkg.Engine(0)
            e.load(small_models)
            e.set_option("rng_shard_offset", r * b)
            e.set_option("rng_shard_total", B)
            e.seed(5)
            c2, r2, l2, _ = e.autoregressive(DEFAULT_TOKENS, voice, b, S, mask_stop=True)
            assert (c2 == codes[r * b:(r + 1) * b]).all(), (G, r)
            m2 = e.diffusion(l2, n_steps=n_steps, noise_mode=pkg.NOISE_DEVICE)
            a2 = e.vocoder(m2, noise_mode=pkg.NOISE_DEVICE)
            for k in range(b):
                assert np.abs(l2[k] - lats[r * b + k]).max() <= 1e-5 * np.abs(lats[r * b + k]).max()
                assert np.abs(m2[k] - mels[r * b + k]).max() < 1e-4, (G, r, k)  # same noise stream, same arithmetic per row
                assert np.abs(a2[k] - audio[r * b + k]).max() <= 1e-3 * np.abs(audio[r * b + k]).max()
            e.close()


def test_retire_mode_matches_strict_sequences(pkg, small_models, voice):
    """TTS_AR_RETIRE (throughput stop rule): every candidate's sequence is EXACTLY the one the reference's rule produces for it.
    The reference's rule for one candidate is "stop at the first 8193" (main.cpp:5214-5222 with B = 1), so the direct comparison is the
    batch of 4 in retire mode against four single-candidate contexts holding candidate c of the same batch (RNG stream partition):
    identical codes, identical stop status; and for B > 1 the strict rule fails where retire returns."""
    B, M = 4, 40
    e = pkg.Engine(0)
    e.load(ar=small_models + "/ggml-model.bin")
    e.seed(11)
    c_ret, rows, lats, steps = e.autoregressive(DEFAULT_TOKENS, voice, B, M, retire=True)
    stopped = e.ar_stop_status(B)
    assert steps <= M and len(lats) == B
    for c in range(B):
        seq = list(c_ret[c, 1:1 + M])
        assert bool(stopped[c]) == (8193 in seq), (c, stopped, seq)
    if not stopped.all():
        with pytest.raises(pkg.TtsError):  # the reference's rule for B > 1: no common stop within max_steps is a failure
            e.seed(11)
            e.autoregressive(DEFAULT_TOKENS, voice, B, M)
    e.close()
    for c in range(B):
        s1 = pkg.Engine(0)
        s1.load(ar=small_models + "/ggml-model.bin")
        s1.set_option("rng_shard_offset", c)
        s1.set_option("rng_shard_total", B)
        s1.seed(11)
        if stopped[c]:  # strict mode IS the reference's rule for one candidate
            c1, r1, _, _ = s1.autoregressive(DEFAULT_TOKENS, voice, 1, M, want_latents=False)
        else:           # it never stops within M steps: strict fails, retire cuts — at the same codes
            with pytest.raises(pkg.TtsError):
                s1.autoregressive(DEFAULT_TOKENS, voice, 1, M, want_latents=False)
            s1.seed(11)
            c1, r1, _, _ = s1.autoregressive(DEFAULT_TOKENS, voice, 1, M, want_latents=False, retire=True)
        assert (c1[0] == c_ret[c]).all() and r1[0] == rows[c], c
        assert s1.ar_stop_status(1)[0] == stopped[c]
        s1.close()


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two MI355X devices (the round-end GPU box has one)")
def test_bench_two_ranks_on_two_gpus():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--quick", "--config", "4", "--candidates", "4", "--steps", "1",
           "--warmup", "0", "--decode-steps", "8", "--diff-steps", "4", "--no-cpu-baseline", "--no-ab"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_bench_two_ranks_share_one_gpu():
    """bench.py's whole N > 1 path with REAL engines on a one-GPU box: it launches its own two ranks (torch.distributed.run, 127.0.0.1), both
    ranks create their context on device 0 (`--device-map 0,0`), gloo carries the prompt / voice broadcast and the audio gather (RCCL needs one
    GPU per rank). configs[3] shape: one batch of 4 candidates sharded 2 + 2 with the RNG stream partition."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-map", "0,0", "--allow-shared-device", "--quick", "--config", "4",
           "--candidates", "4", "--steps", "1", "--warmup", "0", "--decode-steps", "8", "--diff-steps", "4", "--no-cpu-baseline", "--no-ab"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["candidates_per_gpu"] == 2
    assert "roofline" in out and out["roofline"]["launches_timed"] > 0


def test_cli_devices_shards_reproduce_the_single_process_batch(small_models, tmp_path):
    """The product's own multi-GPU entry: `tortoise --candidates 4 --devices 2` re-executes itself once per device (here both workers
    are mapped onto device 0 with --device-map 0,0: the GPU box has one), worker r takes candidates [2r, 2r + 2) of the ONE batch.
    The four WAV files must be the ones a single process writes for --candidates 4 with the same seed."""
    import shutil
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    d = tmp_path / "models"
    d.mkdir()
    for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
        os.symlink(os.path.join(small_models, f), d / f)
    shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), d / "tokenizer.json")
    base = [exe, "--models", str(d), "--message", "this is a test message.", "--voice", os.path.join(ROOT, "models", "mol.bin"), "--seed", "3",
            "--codes", "16", "--steps", "4", "--candidates", "4"]
    outs, errs = {}, {}
    for tag, extra in (("one", []), ("two", ["--devices", "2", "--device-map", "0,0", "--allow-shared-device", "1"])):
        out = tmp_path / (tag + ".wav")
        r = subprocess.run(base + ["--output", str(out)] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        errs[tag] = r.stderr
        files = [out] + [tmp_path / ("%s.wav.%d.wav" % (tag, c)) for c in range(1, 4)]
        assert all(f.exists() for f in files), [f.name for f in files if not f.exists()]
        outs[tag] = [np.frombuffer(f.read_bytes()[44:], np.float32) for f in files]
    # (two engine processes on ONE GPU is this test's vehicle, not a deployment: while they overlap, the diffusion stage's timestep MLP may be re-evaluated —
    #  the CLI says so on stderr — see DESIGN.md section 6 and profiles/r4_two_process_determinism.txt)
    for c in range(4):
        a, b = outs["one"][c], outs["two"][c]
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-4 * max(1e-6, np.abs(a).max()), (c, errs["two"][-600:])


@pytest.mark.parametrize("models", ["small", "full"])
def test_cli_round6_load_and_noise_paths_write_the_same_file(small_models, full_models, tmp_path, models):
    """One utterance through the CLI (one candidate: the reference's RNG order) with the round-6 defaults — AR layouts built by kernels from pinned uploads on worker
    threads, the diffusion and vocoder models loaded on a second thread while the AR stage loads and runs, the noise of step k + 1 drawn beside step k in the two-phase
    form of the normal distribution — and with all of that switched off (the loaders and draws of rounds 1-5): the two WAV files are equal byte for byte, three times
    in a row for the default (the second thread races the AR stage's graph capture; profiles/r6_cli_wall.txt)."""
    import shutil
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    src = small_models if models == "small" else full_models
    d = tmp_path / "models"
    d.mkdir()
    for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
        os.symlink(os.path.join(src, f), d / f)
    shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), d / "tokenizer.json")
    base = [exe, "--models", str(d), "--message", "this is a test message.", "--voice", os.path.join(ROOT, "models", "mol.bin"), "--seed", "3", "--codes", "48", "--timing", "1"]
    old = ["--option", "load_threads=1", "--option", "load_device_pack=0", "--option", "noise_pipeline=0", "--option", "rng_fast_normal=0"]
    got = {}
    for tag, extra in (("old", old), ("new1", []), ("new2", []), ("new3", [])):
        out = tmp_path / (tag + ".wav")
        r = subprocess.run(base + ["--output", str(out)] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and out.exists(), r.stdout + r.stderr
        got[tag] = out.read_bytes()
        if tag in ("old", "new1"):
            print("%s weights, %s: %s" % (models, tag, " | ".join(l.split("]")[1].strip() for l in r.stderr.splitlines() if l.startswith("[timing]") and ("load" in l or "wait" in l))))
    assert len(got["old"]) > 44 + 4 * 1000
    for tag in ("new1", "new2", "new3"):
        assert got[tag] == got["old"], tag


def test_cli_clvp_reranking_single_process_and_shards(small_models, tmp_path):
    """`tortoise --clvp <file>` (extension): the candidates are scored with CLVP, only the best one is carried through diffusion +
    vocoder and written to --output. A single process and two --devices workers (both on device 0) must keep the SAME candidate (the codes of
    a sharded batch are the single batch's) and write the same audio; that audio is the plain run's WAV of that candidate."""
    import re
    import shutil
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    from tortoise_cpp_amd import synth_weights as sw
    d = tmp_path / "models"
    d.mkdir()
    for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
        os.symlink(os.path.join(small_models, f), d / f)
    shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), d / "tokenizer.json")
    clvp = str(tmp_path / "ggml-clvp-model.bin")
    sw.write_clvp(clvp, depth=2, seed=5)
    base = [exe, "--models", str(d), "--message", "this is a test message.", "--voice", os.path.join(ROOT, "models", "mol.bin"), "--seed", "3",
            "--codes", "16", "--steps", "4", "--candidates", "4"]
    plain = tmp_path / "plain.wav"
    r = subprocess.run(base + ["--output", str(plain)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    kept, audio = {}, {}
    for tag, extra in (("one", []), ("two", ["--devices", "2", "--device-map", "0,0", "--allow-shared-device", "1"])):
        out = tmp_path / (tag + ".wav")
        r = subprocess.run(base + ["--clvp", clvp, "--output", str(out)] + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.findall(r"clvp: candidate (\d+) kept", r.stdout)
        assert m, r.stdout
        kept[tag] = int(m[-1])
        audio[tag] = np.frombuffer(out.read_bytes()[44:], np.float32)
        leftovers = [f.name for f in tmp_path.iterdir() if f.name.startswith(tag + ".wav.")]
        assert not leftovers, leftovers  # no sidecar score files, no losing workers' WAVs
    assert kept["one"] == kept["two"]
    c = kept["one"]
    ref = np.frombuffer((plain if c == 0 else tmp_path / ("plain.wav.%d.wav" % c)).read_bytes()[44:], np.float32)
    for tag in ("one", "two"):
        a = audio[tag]
        assert a.shape == ref.shape and np.abs(a - ref).max() <= 1e-4 * max(1e-6, np.abs(ref).max()), tag


def test_bench_rccl_one_rank():
    """RCCL itself, on the one GPU the box has: `bench.py --force-dist` initialises torch.distributed with backend nccl (= RCCL) for a single
    rank and runs every collective of the N > 1 path on the device — all_reduce (rank count), broadcast (prompt ids, voice), all_gather
    (sizes), gather (audio), the timing all_reduces and the barrier. What a one-GPU box cannot show is xGMI traffic; what it does show is that
    the RCCL code path of the bench initialises and completes on this software stack."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--quick", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-ab"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["collective_backend"] == "nccl" and out["collective_ranks"] == 1 and out["n_gpus"] == 1
    assert out["gathered_samples"] and out["gathered_samples"] > 16 * 1000
    assert out["value"] > 0


def test_cli_rccl_exchange_one_worker(small_models, tmp_path):
    """`tortoise --exchange rccl`: the worker processes form an RCCL communicator (csrc/cli_rccl.h: librccl.so through dlopen) — rank 0
    broadcasts the conditioning, sizes / CLVP scores are all-gathered, the audio is sent to rank 0, which writes every WAV. The GPU box has
    one device and RCCL wants a distinct GPU per rank, so what runs here is ONE worker: communicator init from the parent's unique id,
    broadcast, all-gather, the rank-0 writer — and the files must be the plain run's."""
    import re
    import shutil
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    from tortoise_cpp_amd import synth_weights as sw
    d = tmp_path / "models"
    d.mkdir()
    for f in ("ggml-model.bin", "ggml-diffusion-model.bin", "ggml-vocoder-model.bin"):
        os.symlink(os.path.join(small_models, f), d / f)
    shutil.copy(os.path.join(ROOT, "models", "tokenizer.json"), d / "tokenizer.json")
    clvp = str(tmp_path / "ggml-clvp-model.bin")
    sw.write_clvp(clvp, depth=2, seed=5)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base = [exe, "--models", str(d), "--message", "this is a test message.", "--voice", os.path.join(ROOT, "models", "mol.bin"), "--seed", "3",
            "--codes", "16", "--steps", "4", "--candidates", "2"]
    for tag, extra in (("plain", []), ("rccl", ["--devices", "1", "--exchange", "rccl"])):
        r = subprocess.run(base + ["--output", str(tmp_path / (tag + ".wav"))] + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
    for suffix in ("", ".1.wav"):
        a, b = (tmp_path / ("plain.wav" + suffix)).read_bytes(), (tmp_path / ("rccl.wav" + suffix)).read_bytes()
        assert a == b, suffix
    kept = {}
    for tag, extra in (("plainc", []), ("rcclc", ["--devices", "1", "--exchange", "rccl"])):
        r = subprocess.run(base + ["--clvp", clvp, "--output", str(tmp_path / (tag + ".wav"))] + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        kept[tag] = re.findall(r"clvp: candidate (\d+) kept", r.stdout)[-1]
    assert kept["plainc"] == kept["rcclc"]
    assert (tmp_path / "plainc.wav").read_bytes() == (tmp_path / "rcclc.wav").read_bytes()
    # an unknown exchange is an error; two workers on ONE device cannot form an RCCL communicator and say so instead of hanging
    r = subprocess.run(base + ["--output", str(tmp_path / "x.wav"), "--exchange", "mpi"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "files or rccl" in r.stderr
    r = subprocess.run(base + ["--output", str(tmp_path / "y.wav"), "--devices", "2", "--device-map", "0,0", "--allow-shared-device", "1", "--exchange", "rccl"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    # two workers on one GPU are refused outright unless asked for (unsupported form: DESIGN.md section 6)
    r = subprocess.run(base + ["--output", str(tmp_path / "w.wav"), "--devices", "2", "--device-map", "0,0"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode != 0 and "share a GPU" in r.stderr
