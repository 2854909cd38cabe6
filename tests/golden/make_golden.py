"""Generates the committed golden vectors from the REAL reference code (oracle/_ref/libref.so, built by
oracle/build_ref.sh from /root/reference). Run in the CPU container only:  python tests/golden/make_golden.py
Outputs: host_golden.json, sampler_golden.npz, schedule_golden.npz (data only — inputs and expected outputs)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

R = O.ref()
assert R is not None, "needs /root/reference"
out = {}
R.ref_seed(245645656)
out["uniform_seed_245645656"] = [float(R.ref_uniform()) for _ in range(8)]
R.ref_load_rng_state(os.path.join(HERE, "reference_assets", "test_diffusion_seed.bin").encode())
a = np.empty(64, np.float32)
R.ref_normal_fill(a, 64)
out["normal_diffusion_seed"] = [float(x) for x in a]
R.ref_tokenizer_init(os.path.join(ROOT, "models", "tokenizer.json").encode())
tok = {}
for msg in ["this is a test message.", "based... dr freeman?", "congratulations! autoregressive model complete!",
            "the united states must not adopt the tactics of the enemy.", "it's 3 o'clock, we're late", ""]:
    o = np.empty(1024, np.int32)
    n = R.ref_tokenize(msg.encode(), o, 1024)
    tok[msg] = [int(x) for x in o[:n]]
out["tokenizer"] = tok
bk = {}
for n in (1, 6, 43, 100):
    o = np.empty((n, n), np.int32)
    R.ref_buckets(n, o.reshape(-1))
    bk[str(n)] = o.tolist()
out["buckets"] = bk
json.dump(out, open(os.path.join(HERE, "host_golden.json"), "w"))

# sampler
rs = np.random.RandomState(2024)
d = {"seeds": []}
for k in range(6):
    B = [1, 4, 2, 3, 16, 1][k]
    logits = (rs.randn(B, 8194) * [1.0, 3.0, 0.3, 6.0, 2.0, 2.0][k]).astype(np.float32)
    if k == 5:
        logits = np.round(logits * 2) / 2
    ids = rs.randint(0, 8194, (B, 1)).astype(np.int32) if k % 2 else np.tile(np.array([1] * 17 + [8192], np.int32), (B, 1))
    seed = int(rs.randint(1 << 30))
    R.ref_seed(seed)
    o = np.empty(B, np.int32)
    R.ref_process_logits_and_sample(logits.reshape(-1), np.ascontiguousarray(ids).reshape(-1), ids.size, B, o, None)
    d["seeds"].append(seed)
    d["logits_%d" % k], d["ids_%d" % k], d["samples_%d" % k] = logits, ids, o
d["seeds"] = np.array(d["seeds"], np.int64)
np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **d)

# schedule + timestep embedding + one ancestral update
tm = O.default_timestep_map(80)
arrs = [np.empty(80) for _ in range(7)]
R.ref_schedule(tm, 80, *arrs)
s = dict(zip(O.SCHED_KEYS, arrs))
te = {}
for t in (0, 51, 2025, 3999):
    o = np.empty(1024, np.float32)
    R.ref_timestep_embedding(t, o)
    te["temb_%d" % t] = o
T = 23
upd = {}
for t in (79, 40, 1, 0):
    oc, ou = rs.randn(200 * T).astype(np.float32), rs.randn(200 * T).astype(np.float32)
    x, nz = (rs.randn(100 * T) * 2).astype(np.float32), rs.randn(100 * T).astype(np.float32)
    xr = x.copy()
    cfk = np.float32(2.0) * (np.float32(1) - np.float32(t) / np.float32(80))
    R.ref_diffusion_update(oc, ou, xr, nz, T, np.float32(np.log(s["betas"][t])), np.float32(s["post_logvar"][t]), cfk,
                           np.float32(s["sqrt_recip"][t]), np.float32(s["sqrt_recipm1"][t]), np.float32(s["coef1"][t]),
                           np.float32(s["coef2"][t]), int(t == 0))
    upd.update({"upd%d_oc" % t: oc, "upd%d_ou" % t: ou, "upd%d_x" % t: x, "upd%d_nz" % t: nz, "upd%d_out" % t: xr})
np.savez_compressed(os.path.join(HERE, "schedule_golden.npz"), timestep_map=tm, **s, **te, **upd)
print("golden vectors written")
