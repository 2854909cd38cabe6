"""world_size-2 gloo test of the candidate-parallel sharding used by bench.py --gpus N: prompt/voice broadcast
from rank 0, per-rank RNG streams, gather of per-candidate results on rank 0. No device compute: each rank's
'stage' is the product's host logic (tokenizer + sampler on a host-only context)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
L = pkg.lib()
eng = pkg.Engine.__new__(pkg.Engine); eng.L = L; eng.h = L.tts_create(-1)
eng.tokenizer_load(os.path.join(%(root)r, "models", "tokenizer.json"))
# rank 0 owns the prompt and the voice latent; everyone else receives them
if rank == 0:
    toks = torch.from_numpy(eng.tokenize("this is a test message.").astype(np.int32))
    voice = torch.from_numpy(np.fromfile(os.path.join(%(root)r, "models", "mol.bin"), np.float32))
    n = torch.tensor([toks.numel()])
else:
    n = torch.zeros(1, dtype=torch.int64); voice = torch.zeros(1024)
dist.broadcast(n, 0)
if rank != 0: toks = torch.zeros(int(n), dtype=torch.int32)
dist.broadcast(toks, 0); dist.broadcast(voice, 0)
assert toks.tolist() == [255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0]
# candidates [rank*B, (rank+1)*B) with their own RNG stream
B = 3
eng.seed(1000 + rank)
logits = np.random.RandomState(7).randn(B, 8194).astype(np.float32) * 3   # same logits everywhere
ids = np.tile(np.array([1] * 17 + [8192], np.int32), (B, 1))
mine = torch.from_numpy(eng.sample(logits, ids).astype(np.int64))
outs = [torch.zeros(B, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
dist.gather(mine, outs, dst=0)
# throughput aggregation exactly as bench.py does it
t = torch.tensor([1.0 + rank]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
s = torch.tensor([10.0]); dist.all_reduce(s, op=dist.ReduceOp.SUM)
if rank == 0:
    assert float(t) == float(world) and float(s) == 10.0 * world
    flat = torch.stack(outs)
    assert flat.shape == (world, B)
    assert not torch.equal(flat[0], flat[1])  # independent per-rank streams
    eng.seed(1001)                             # rank 1's candidates are reproducible from its seed alone
    assert eng.sample(logits, ids).tolist() == flat[1].tolist()
    print("DIST_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_OK" in r.stdout


def _host_engine(pkg):
    eng = pkg.Engine.__new__(pkg.Engine)
    eng.L = pkg.lib()
    eng.h = eng.L.tts_create(-1)
    return eng


def test_rng_stream_partition_reproduces_unsharded_ids(pkg):
    """the used uniform of (step s, candidate c) is output 2 (s B + c) + 1 of the one mt19937 stream. With the options
    rng_shard_offset / rng_shard_total a rank that holds candidates [c0, c0 + b) skips the other ranks' draws, so G ranks x B/G
    candidates sample exactly the ids one rank x B samples — for every split, step after step."""
    B, steps = 12, 7
    rs = np.random.RandomState(3)
    logits = [rs.randn(B, 8194).astype(np.float32) * 3 for _ in range(steps)]
    prev = rs.randint(0, 8192, (B, 1)).astype(np.int32)
    ref = _host_engine(pkg)
    ref.seed(99)
    want = np.stack([ref.sample(lg, prev) for lg in logits])
    tail = ref.rng_uniform()
    for G in (2, 3, 4, 12):
        b = B // G
        got = np.zeros_like(want)
        for r in range(G):
            e = _host_engine(pkg)
            e.set_option("rng_shard_offset", r * b)
            e.set_option("rng_shard_total", B)
            e.seed(99)
            for s in range(steps):
                got[s, r * b:(r + 1) * b] = e.sample(logits[s][r * b:(r + 1) * b], prev[r * b:(r + 1) * b])
            assert e.rng_uniform() == tail  # every rank leaves the stream where the unsharded run leaves it
            e.close()
        assert (got == want).all(), G
    # a shard that does not fit its batch is an argument error, not a silent mis-draw
    e = _host_engine(pkg)
    e.set_option("rng_shard_offset", 8)
    e.set_option("rng_shard_total", 10)
    with pytest.raises(pkg.TtsError):
        e.sample(logits[0][:4], prev[:4])
    e.close()
    ref.close()


def test_bench_self_launches_its_ranks(pkg):
    """`python bench.py --gpus 2` (no torchrun around it) spawns its own two ranks, runs the broadcast / per-rank work / audio gather /
    max-over-ranks timing and prints ONE JSON line with n_gpus = 2. Device work is replaced by the host sampler (--dry-engine,
    gloo): config 4 shards ONE batch of 6 candidates 3 + 3 with the RNG stream partition, and the two ranks' ids must be the ids an
    unsharded context samples."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--candidates", "6", "--backend", "gloo",
           "--dry-engine", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["workload"].startswith("configs[3]")
    ids = {}
    for l in r.stderr.splitlines():
        if "DRY_IDS rank" in l and "pass 0" in l:
            ids[int(l.split("rank ")[1].split()[0])] = [int(x) for x in l.split(": ")[1].split()]
    assert sorted(ids) == [0, 1] and len(ids[0]) == len(ids[1]) == 3
    ref = _host_engine(pkg)
    ref.seed(0)  # bench.py: seed 1000 * pass + 17 * prompt
    logits = np.random.RandomState(7).randn(6, 8194).astype(np.float32) * 3
    want = ref.sample(logits, np.tile(np.array([1] * 17 + [8192], np.int32), (6, 1)))
    ref.close()
    # every rank draws on the same 3 rows of logits (its own slice of the fake stage is rows 0..2): compare with those rows
    want0 = _host_engine(pkg); want0.seed(0)
    lg3 = logits[:3]
    e0 = _host_engine(pkg); e0.set_option("rng_shard_total", 6); e0.seed(0)
    e1 = _host_engine(pkg); e1.set_option("rng_shard_offset", 3); e1.set_option("rng_shard_total", 6); e1.seed(0)
    pen = np.tile(np.array([1] * 17 + [8192], np.int32), (3, 1))
    assert ids[0] == e0.sample(lg3, pen).tolist() and ids[1] == e1.sample(lg3, pen).tolist()
    for e in (want0, e0, e1):
        e.close()
    assert out["dry_engine"]["gathered_samples"] > 0


def test_bench_force_dist_one_rank_gloo():
    """`bench.py --force-dist` (the switch the GPU suite uses to run RCCL with one rank) on the CPU: a one-rank gloo group of its own
    rendezvous, every collective of the N > 1 path executed, the line carries the backend, the rank count and the gathered sample count."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--backend", "gloo", "--dry-engine", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["collective_backend"] == "gloo" and out["collective_ranks"] == 1
    assert out["gathered_samples"] == out["dry_engine"]["gathered_samples"] > 0


def test_cli_devices_worker_bookkeeping_without_a_gpu(tmp_path):
    """`tortoise --devices 2 --exchange files` on a machine without a GPU (--dry-run 1: host-only contexts, the host sampler on fixed synthetic
    logits stands in for the stages): the parent re-executes itself once per shard with --shard r/2 and ONE seed, worker r draws exactly its
    candidates' uniforms from the one mt19937 stream and writes its candidates under their GLOBAL names — the four files of the two-worker run are
    the four files of the single process; with --clvp the parent keeps the best of the workers' winners (the single process's winner); a failing
    worker makes the parent fail."""
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    models = os.path.join(ROOT, "models")
    base = [exe, "--dry-run", "1", "--models", models, "--voice", os.path.join(models, "mol.bin"), "--seed", "11", "--codes", "9", "--candidates", "4"]

    def wav(p):
        return np.frombuffer(open(p, "rb").read()[44:], np.float32)

    runs = {}
    for tag, extra in (("one", []), ("two", ["--devices", "2"]), ("four", ["--devices", "4"])):
        out = str(tmp_path / (tag + ".wav"))
        r = subprocess.run(base + ["--output", out] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        files = [out] + ["%s.%d.wav" % (out, c) for c in range(1, 4)]
        assert all(os.path.exists(f) for f in files), (tag, os.listdir(tmp_path))
        runs[tag] = [wav(f) for f in files]
        assert all(len(x) == 9 and (x >= 0).all() and (x < 8194).all() for x in runs[tag])
    for tag in ("two", "four"):
        for c in range(4):
            assert (runs[tag][c] == runs["one"][c]).all(), (tag, c)
    assert len({tuple(x) for x in runs["one"]}) > 1  # the candidates differ (their draws do)
    # re-ranking: the parent's pick among the workers' winners = the single process's winner; no per-candidate files are left behind
    picks = {}
    for tag, extra in (("one", []), ("two", ["--devices", "2"])):
        out = str(tmp_path / ("rr_" + tag + ".wav"))
        r = subprocess.run(base + ["--output", out, "--clvp", "unused-in-dry-run"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "clvp: candidate" in r.stdout, r.stdout + r.stderr
        picks[tag] = (r.stdout.split("clvp: candidate")[1].split()[0], tuple(wav(out)))
        assert not [f for f in os.listdir(tmp_path) if f.startswith("rr_" + tag + ".wav.")], os.listdir(tmp_path)
    assert picks["one"] == picks["two"]
    # a worker that cannot start its work (unreadable --voice) fails the parent
    r = subprocess.run(base[:6] + [str(tmp_path / "missing.bin")] + base[7:] + ["--devices", "2", "--output", str(tmp_path / "x.wav")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    # candidates must divide over the devices
    r = subprocess.run(base + ["--devices", "3", "--output", str(tmp_path / "y.wav")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not divide" in r.stderr


def test_cli_option_flag(tmp_path):
    """`--option key=value` forwards engine options (tts_set_option) before the models load; workers inherit it; a malformed or unknown option fails loudly."""
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    models = os.path.join(ROOT, "models")
    base = [exe, "--dry-run", "1", "--models", models, "--voice", os.path.join(models, "mol.bin"), "--seed", "11", "--codes", "5", "--candidates", "2",
            "--output", str(tmp_path / "o.wav")]
    r = subprocess.run(base + ["--option", "device_topk=0", "--option", "attn_f32=1", "--devices", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(base + ["--option", "attn_f32"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "key=value" in r.stderr
    r = subprocess.run(base + ["--option", "no_such_option=1"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0, r.stdout + r.stderr


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    """tests/fake_rccl.cpp -> a shared library with librccl's ten entry points that moves host buffers over Unix sockets (TTS_RCCL_LIB)."""
    src = os.path.join(ROOT, "tests", "fake_rccl.cpp")
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I/opt/rocm/include", src, "-o", so], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        pytest.skip("fake rccl does not build here: " + r.stderr[-500:])
    return so


def test_cli_rccl_exchange_four_ranks_on_the_cpu(tmp_path, fake_rccl):
    """`tortoise --devices 4 --exchange rccl` with FOUR ranks on a machine without a GPU: --dry-run keeps the staging buffers of
    csrc/cli_rccl.h on the host and TTS_RCCL_LIB points its dlopen at tests/fake_rccl.cpp, so the N > 1 code of the RCCL exchange executes for the first
    time anywhere — unique id created by the parent and handed to the workers, conditioning broadcast from rank 0, all-gather of the per-candidate
    sample counts, one send / receive pair per rank to rank 0, which writes every file. The files equal the `--exchange files` run's and the
    single process's; with --clvp the winner is the single process's and no per-candidate file is left."""
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    models = os.path.join(ROOT, "models")
    env = dict(os.environ, TTS_RCCL_LIB=fake_rccl, TMPDIR=str(tmp_path))
    base = [exe, "--dry-run", "1", "--models", models, "--voice", os.path.join(models, "mol.bin"), "--seed", "23", "--codes", "7", "--candidates", "8"]

    def wav(p):
        return np.frombuffer(open(p, "rb").read()[44:], np.float32)

    runs = {}
    for tag, extra in (("one", []), ("files4", ["--devices", "4"]), ("rccl4", ["--devices", "4", "--exchange", "rccl"]), ("rccl2", ["--devices", "2", "--exchange", "rccl"])):
        out = str(tmp_path / (tag + ".wav"))
        r = subprocess.run(base + ["--output", out] + extra, capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, (tag, r.stdout + r.stderr)
        files = [out] + ["%s.%d.wav" % (out, c) for c in range(1, 8)]
        assert all(os.path.exists(f) for f in files), (tag, sorted(os.listdir(tmp_path)))
        runs[tag] = [wav(f) for f in files]
    for tag in ("files4", "rccl4", "rccl2"):
        for c in range(8):
            assert len(runs[tag][c]) == 7 and (runs[tag][c] == runs["one"][c]).all(), (tag, c)
    picks = {}
    for tag, extra in (("one", []), ("rccl4", ["--devices", "4", "--exchange", "rccl"])):
        out = str(tmp_path / ("rr_" + tag + ".wav"))
        r = subprocess.run(base + ["--output", out, "--clvp", "unused-in-dry-run"] + extra, capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0 and "clvp: candidate" in r.stdout, (tag, r.stdout + r.stderr)
        picks[tag] = (r.stdout.split("clvp: candidate")[1].split()[0], tuple(wav(out)))
        assert not [f for f in os.listdir(tmp_path) if f.startswith("rr_" + tag + ".wav.")], os.listdir(tmp_path)
    assert picks["one"] == picks["rccl4"]
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".sock")]  # rank 0 removed its socket
    # a library that is not there fails loudly, in the parent, before any worker starts
    r = subprocess.run(base + ["--output", str(tmp_path / "z.wav"), "--devices", "2", "--exchange", "rccl"], capture_output=True, text=True, timeout=120,
                       env=dict(env, TTS_RCCL_LIB=str(tmp_path / "no_such_lib.so")))
    assert r.returncode != 0 and "TTS_RCCL_LIB" in r.stderr


def test_cli_rccl_exchange_eight_ranks_slow_rank_and_failing_rank(tmp_path, fake_rccl):
    """The rank count of the first 8-GPU lease, on the CPU: `tortoise --devices 8 --exchange rccl` through the stand-in library.
    (1) eight ranks, files equal the single process's; (2) one rank arrives 2 s late at the final exchange (--test-slow-shard): same files; (3) one rank dies after the
    conditioning broadcast (--test-fail-shard): with a stand-in that FAILS on a lost peer every other rank ends with an error by itself; with one that BLOCKS for ever
    (FAKE_RCCL_HANG_ON_PEER_LOSS — what librccl does when a rank dies inside a collective) the parent ends the waiting workers (SIGTERM): either way the command returns a
    non-zero status within 10 s and leaves no worker behind; (4) --dry-run without the stand-in refuses to hand host buffers to the real library."""
    import time
    exe = os.path.join(ROOT, "tortoise.cpp_amd", "tortoise")
    if not os.path.exists(exe):
        pytest.skip("CLI binary not built")
    models = os.path.join(ROOT, "models")
    env = dict(os.environ, TTS_RCCL_LIB=fake_rccl, TMPDIR=str(tmp_path))
    base = [exe, "--dry-run", "1", "--models", models, "--voice", os.path.join(models, "mol.bin"), "--seed", "29", "--codes", "5", "--candidates", "16"]

    def wavs(out):
        files = [out] + ["%s.%d.wav" % (out, c) for c in range(1, 16)]
        assert all(os.path.exists(f) for f in files), sorted(os.listdir(tmp_path))
        return [np.frombuffer(open(f, "rb").read()[44:], np.float32) for f in files]

    runs = {}
    for tag, extra in (("one", []), ("rccl8", ["--devices", "8", "--exchange", "rccl"]), ("slow", ["--devices", "8", "--exchange", "rccl", "--test-slow-shard", "5"])):
        out = str(tmp_path / (tag + ".wav"))
        r = subprocess.run(base + ["--output", out] + extra, capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, (tag, r.stdout + r.stderr)
        runs[tag] = wavs(out)
    for tag in ("rccl8", "slow"):
        for c in range(16):
            assert (runs[tag][c] == runs["one"][c]).all(), (tag, c)
    for hang in (False, True):
        e = dict(env, FAKE_RCCL_HANG_ON_PEER_LOSS="1") if hang else env
        t0 = time.time()
        r = subprocess.run(base + ["--output", str(tmp_path / ("fail%d.wav" % hang)), "--devices", "8", "--exchange", "rccl", "--test-fail-shard", "3"],
                           capture_output=True, text=True, timeout=60, env=e)
        dt = time.time() - t0
        assert r.returncode != 0 and "--test-fail-shard" in r.stderr, (hang, r.returncode, r.stderr[-500:])
        assert dt < 10.0, (hang, dt)
        # the parent reaps every worker before it returns: nothing of this command is left running
        left = subprocess.run(["pgrep", "-f", str(tmp_path / ("fail%d.wav" % hang))], capture_output=True, text=True).stdout.split()
        assert not left, (hang, left)
    e = {k: v for k, v in env.items() if k != "TTS_RCCL_LIB"}
    r = subprocess.run(base + ["--output", str(tmp_path / "nolib.wav"), "--devices", "2", "--exchange", "rccl"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode != 0 and "TTS_RCCL_LIB" in r.stderr, r.stderr[-500:]


def test_bench_eight_ranks_gloo_dry_engine():
    """The rank count the driver's scaling run uses: `python bench.py --gpus 8` launches its own eight ranks (torch.distributed.run, 127.0.0.1), every rank a host-only
    context (--dry-engine), gloo for the collectives: rendezvous, the all_reduce of ones (collective_ranks = 8), prompt / voice broadcast, size all_gather, gather of the
    results on rank 0, MAX-over-ranks timing — for the strong-scaling workload (configs[3]: 64 candidates, 8 per rank) and the 8-prompt one (configs[4])."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for config, per_gpu, prompts in ((4, 8, 1), (5, 16, 8)):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-engine", "--config", str(config), "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 8 and out["collective_ranks"] == 8 and out["collective_backend"] == "gloo" and out["scaling"] == "strong"
        assert out["config"]["candidates_per_gpu"] == per_gpu and out["config"]["prompts"] == prompts and out["gathered_samples"] > 0
        # round 6: every rank's own time and host placement are in the line (a straggler shows up in the first SCALE_r*.json, not only as the maximum)
        assert [r["rank"] for r in out["per_rank"]] == list(range(8)) and all(r["ms_per_step"] > 0 for r in out["per_rank"])
        assert max(r["ms_per_step"] for r in out["per_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-3, abs=0.02)
        assert all(r["numa_node"] == -1 and r["pinned_cpus"] == 0 for r in out["per_rank"])  # host-only contexts: nothing to pin to


def test_numa_placement_api_on_a_host_only_context(pkg):
    """tts_device_numa_node / tts_pin_to_device_numa_node (round 6: each rank of a multi-GPU run keeps its sampler threads on the CPUs next to its GPU): a host-only context has
    no device — node -1, empty CPU list, nothing pinned, the process's affinity untouched."""
    eng = pkg.Engine.__new__(pkg.Engine)
    eng.L = pkg.lib()
    eng.h = eng.L.tts_create(-1)
    try:
        before = os.sched_getaffinity(0)
        assert eng.numa_node() == (-1, "")
        assert eng.pin_to_numa_node() == 0
        assert os.sched_getaffinity(0) == before
    finally:
        eng.close()
