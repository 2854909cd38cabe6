#!/usr/bin/env python3
"""bench.py — end-to-end throughput of the hot path (AR decode -> diffusion -> vocoder) on MI355X.

Metric (BASELINE.json): audio-seconds/sec end-to-end, 16 AR candidates x 80 diffusion steps per GPU.
One "step" = one full pass over one batch: tts_autoregressive (prefill + 192 sampled codes + latent pass)
-> tts_diffusion (80 steps, cond+uncond) -> tts_vocoder for 16 candidates, weights resident in HBM,
synthetic 64-token prompt, stock mol.bin voice, full-size synthetic weights in the reference file format
(the trained weights are not available offline).  N>1: one process per GPU (torchrun), every rank runs
its own 16 candidates (weak scaling); RCCL broadcasts the prompt/voice and gathers the audio on rank 0.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the fp16 MFMA GEMM of the diffusion
stage, timed with HIP events on the engine's stream inside the timed region) and `cpu_baseline`
(the oracle = CPU restatement of the reference path, timed on a bounded sample and extrapolated).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tortoise_cpp_amd_loader  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16


def synthetic_prompt():
    # ids 255, (3 + 7j mod 250) for j < 64, 0  -> n = 66 text ids, P = 68 prompt positions
    return np.array([255] + [3 + (7 * j) % 250 for j in range(64)] + [0], np.int32)


def ensure_models(path, quick, rank_is_writer):
    stamp = os.path.join(path, ".done")
    if rank_is_writer and not os.path.exists(stamp):
        from tortoise_cpp_amd import synth_weights as sw
        if quick:
            sw.write_all(path, ar_layers=2, diff_main=1, diff_tail=1, diff_integ=1, diff_lc=1, seed=1234)
        else:
            sw.write_all(path, seed=1234)  # 30-layer GPT-2, 4+3+10+3 diffusion blocks, UnivNet
        open(stamp, "w").write("ok")


def cpu_baseline(model_dir, voice, toks, S, L_bench, n_diff_steps, quick):
    """Oracle (CPU restatement) on a bounded sample of the same workload, extrapolated with the
    algorithmic-work formulae of SURVEY §8d to one candidate of the bench workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    n = len(toks)
    t_all = time.time()
    # --- AR: prefill + a few decode steps at B=1
    ar = O.AR(O.Model(os.path.join(model_dir, "ggml-model.bin")))
    nstep = 48 if not quick else 4
    ar.start(toks, voice, 1, n + 2 + nstep + 1)
    t0 = time.time(); ar.prefill(); t_prefill = time.time() - t0
    t0 = time.time()
    for i in range(nstep):
        ar.step(np.array([100 + i], np.int32), i)
    t_step = (time.time() - t0) / nstep
    row_s = t_prefill / (n + 2)                                   # seconds per transformer row (dense part)
    t_ar = t_prefill + S * t_step + (n + 1 + L_bench) * row_s     # + latent pass over the needed prefix
    del ar
    # --- diffusion: conditioner + one cond and one uncond forward at the bench's own size (L=200, T=870): only the
    # number of repetitions (80 steps) is extrapolated
    od = O.Diffusion(O.Model(os.path.join(model_dir, "ggml-diffusion-model.bin")))
    Ls = L_bench if not quick else 12
    Ts = od.T_of(Ls)
    lat = np.random.RandomState(0).randn(Ls, 1024).astype(np.float32)
    x = np.random.RandomState(1).randn(100, Ts).astype(np.float32)
    t0 = time.time(); ce = od.code_embedding(lat, Ts); t_cond = time.time() - t0
    npair = 3 if not quick else 1                                 # cond + uncond forward at three timesteps
    t0 = time.time()
    for ts in (3999, 2025, 51)[:npair]:
        od.forward(ce, x, ts); od.forward(None, x, ts)
    t_pair = (time.time() - t0) / npair
    fl = lambda T: 249307136.0 * T + 53248.0 * T * T              # per forward
    Tb = od.T_of(L_bench)
    t_diff = t_cond + n_diff_steps * t_pair * fl(Tb) / fl(Ts)     # conditioner once per utterance
    del od
    # --- vocoder at the same T
    ov = O.Vocoder(O.Model(os.path.join(model_dir, "ggml-vocoder-model.bin")))
    mel = np.clip(np.random.RandomState(2).randn(100, Ts) * 0.5, -1, 1).astype(np.float32)
    t0 = time.time(); ov.run(mel, rng=O.Rng(0)); t_voc_s = time.time() - t0
    t_voc = t_voc_s * (Tb + 10) / (Ts + 10)
    audio_s = ((Tb + 10) * 256 - 6) / 24000.0
    return {
        "value": round(audio_s / (t_ar + t_diff + t_voc), 5), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
        "sample": "oracle (f32 C++/OpenMP restatement of the ggml graphs; the reference itself cannot be built: ggml "
                  "submodule absent). Measured B=1: prefill(P=%d) %.2fs, %d decode steps %.3fs/step, diffusion cond+uncond "
                  "forward pair (mean of %d timesteps) at L=%d/T=%d %.2fs, vocoder T=%d %.2fs (%.0fs total); extrapolated per candidate to S=%d steps, L=%d/T=%d, "
                  "%d diffusion steps (repetition counts; SURVEY 8d work formulae where a sample is smaller than the workload): AR %.1fs + diffusion %.1fs + vocoder %.1fs for %.2fs of audio"
                  % (n + 2, t_prefill, nstep, t_step, npair, Ls, Ts, t_pair, Ts, t_voc_s, time.time() - t_all, S, L_bench, Tb, n_diff_steps,
                     t_ar, t_diff, t_voc, audio_s),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--candidates", type=int, default=16, help="AR candidates per GPU")
    ap.add_argument("--diff-steps", type=int, default=80)
    ap.add_argument("--decode-steps", type=int, default=192, help="sampled codes per candidate (stop token masked) -> L=200, T=870")
    ap.add_argument("--quick", action="store_true", help="tiny layer counts (plumbing check only; NOT the benchmark)")
    ap.add_argument("--prof-stride", type=int, default=13, help="every Nth GEMM launch is bracketed by a HIP event pair (roofline timing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-share-uncond", action="store_true", help="evaluate the unconditioned integrator layers once per candidate")
    ap.add_argument("--models", default=None)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world)

    pkg = tortoise_cpp_amd_loader.load()
    model_dir = a.models or ("/tmp/tts_bench_models_quick" if a.quick else "/tmp/tts_bench_models")
    ensure_models(model_dir, a.quick, local_rank == 0)
    if dist:
        dist.barrier()

    eng = pkg.Engine(local_rank)  # raises without the HIP library/device: there is no fallback path
    if world > 1:  # N processes share the host: leave each rank's sampler pool its share of the cores
        eng.set_option("sampler_threads", max(0, min(7, (os.cpu_count() or 8) // world - 2)))
    eng.load(model_dir)
    if a.no_share_uncond:
        eng.set_option("share_uncond", 0)
    B, S = a.candidates, a.decode_steps
    toks = synthetic_prompt()
    voice = np.fromfile(os.path.join(ROOT, "models", "mol.bin"), np.float32)
    if dist:  # prompt + conditioning come from rank 0 (RCCL broadcast), results are gathered on rank 0
        import torch
        tt = torch.from_numpy(toks.copy()).cuda()
        tv = torch.from_numpy(voice.copy()).cuda()
        dist.broadcast(tt, 0)
        dist.broadcast(tv, 0)
        toks, voice = tt.cpu().numpy(), tv.cpu().numpy()

    stage_ms = {"ar": 0.0, "diffusion": 0.0, "vocoder": 0.0}

    def one_pass(it):
        eng.seed(1000 * it + rank)  # distinct candidates per rank and per pass
        t_a = time.time()
        codes, rows, lats, steps = eng.autoregressive(toks, voice, B, S, mask_stop=True)
        t_b = time.time()
        mels = eng.diffusion(lats, n_steps=a.diff_steps, noise_mode=pkg.NOISE_DEVICE)
        t_c = time.time()
        audio = eng.vocoder(mels, noise_mode=pkg.NOISE_DEVICE)
        t_d = time.time()
        if it >= 0:
            stage_ms["ar"] += 1e3 * (t_b - t_a); stage_ms["diffusion"] += 1e3 * (t_c - t_b); stage_ms["vocoder"] += 1e3 * (t_d - t_c)
        if dist:
            import torch
            flat = torch.from_numpy(np.concatenate(audio)).cuda()
            sizes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([flat.numel()], dtype=torch.int64, device="cuda"))
            mx = int(max(s.item() for s in sizes))
            pad = torch.zeros(mx, device="cuda")
            pad[:flat.numel()] = flat
            outs = [torch.zeros(mx, device="cuda") for _ in range(world)] if rank == 0 else None
            dist.gather(pad, outs, dst=0)
        return sum(len(x) for x in audio) / 24000.0, rows, [m.shape[1] for m in mels]

    def sync():
        if dist:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        # every engine call is synchronous (ends with hipStreamSynchronize on its stream)

    for w in range(a.warmup):
        one_pass(-1 - w)
    eng.set_option("prof_only:diff_gemm", 1)
    # every 13th GEMM launch is bracketed by a HIP event pair (60 launches per diffusion step, 13 is coprime: every launch
    # position is sampled equally often over the 80 steps). An event pair drains the pipeline around its launch: bracketing
    # all ~9 600 launches of the timed region cost 5 % of the pass, every 7th 0.5 %, every 29th nothing measurable.
    eng.set_option("prof_stride", a.prof_stride)
    eng.prof_reset(True)
    sync()
    t0 = time.time()
    audio_s = 0.0
    for k in range(a.steps):
        s, rows, Ts = one_pass(k)
        audio_s += s
    sync()
    dt = time.time() - t0
    g_ms, g_n, g_flops = eng.prof_get("diff_gemm")
    eng.prof_reset(False)
    if dist:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([audio_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        audio_s = float(s.item())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    # HBM bytes per GEMM launch from the committed PMC profile (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes,
    # calibration in the file's note); null when the profile is absent. bench.py does not run rocprofv3 itself.
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.json")))["kernels"]
        gk = [v for k, v in prof.items() if "gemm_f16" in k]
        traffic = int(sum(v["dispatches"] * v["hbm_bytes_per_launch"] for v in gk) / max(1, sum(v["dispatches"] for v in gk)))
    except Exception:
        pass
    out = {
        "metric": "audio-seconds/sec end-to-end (AR+diffusion+vocoder), 16 cands x 80 steps",
        "value": round(audio_s / dt, 3), "unit": "audio-seconds/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1000.0 * dt / a.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 MFMA inputs / f32 accumulate (diffusion, vocoder convs); f32 weights, f16 KV (AR)", "data": "synthetic",
        "config": {"workload": "configs[2]: synthetic 64-token prompt (n=66), mol.bin voice, %d AR candidates/GPU x %d sampled codes "
                               "(L=%d latent rows, T=%d mel frames, %.2f s audio each), %d diffusion steps (cond+uncond batched), "
                               "UnivNet vocoder; full-size synthetic weights%s" % (B, S, int(rows[0]), int(Ts[0]),
                                                                                  ((Ts[0] + 10) * 256 - 6) / 24000.0, a.diff_steps,
                                                                                  " [QUICK: reduced layer counts]" if a.quick else ""),
                   "candidates_per_gpu": B, "diffusion_steps": a.diff_steps, "decode_steps": S, "parallelism": "candidate-parallel x%d" % world,
                   # the unconditioned branch's integrator layers (input independent of the candidate) are evaluated once per distinct
                   # sequence length; with the stop token masked all candidates have one length (option share_uncond)
                   "uncond_integrator_shared": not a.no_share_uncond},
        "stage_ms_per_step": {k: round(v / a.steps, 1) for k, v in stage_ms.items()},
        "roofline": {"kernel": "gemm_f16_glds_kernel + gemm_f16_conv3_kernel (diffusion convs/projections)", "bound": "mfma", "achieved": round(achieved, 1),
                     "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                     "traffic": traffic, "launches_timed": int(g_n), "launch_sampling": "every %dth launch of the family is bracketed by HIP events" % a.prof_stride, "avg_launch_us": round(1000.0 * g_ms / max(g_n, 1), 2),
                     "algorithmic_gflop_per_launch": round(g_flops / max(g_n, 1) / 1e9, 2)},
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model_dir, voice, toks, S, int(rows[0]), a.diff_steps, a.quick)
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
