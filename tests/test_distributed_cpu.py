"""world_size-2 gloo test of the candidate-parallel sharding used by bench.py --gpus N: prompt/voice broadcast
from rank 0, per-rank RNG streams, gather of per-candidate results on rank 0. No device compute: each rank's
'stage' is the product's host logic (tokenizer + sampler on a host-only context)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
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
