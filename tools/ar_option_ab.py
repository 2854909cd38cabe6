"""Developer A/B of one engine option on the AR stage of the bench workload (full-depth synthetic weights, 192 masked steps).
   python tools/ar_option_ab.py dec_prefetch [B ...]      -> per setting: stage ms (best of 3), decode-step us from the HIP-event family, codes identical?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tortoise_cpp_amd_loader  # noqa: E402
pkg = tortoise_cpp_amd_loader.load()

opt = sys.argv[1]
Bs = [int(x) for x in sys.argv[2:]] or [16, 1]
d = os.environ.get("TTS_BENCH_MODELS", "/tmp/tts_bench_models")
if not os.path.exists(os.path.join(d, ".done")):
    from tortoise_cpp_amd import synth_weights as sw
    sw.write_all(d, seed=1234)
    open(os.path.join(d, ".done"), "w").write("ok")
eng = pkg.Engine(0)
eng.load(ar=os.path.join(d, "ggml-model.bin"))
toks = np.array([255, 147, 2, 54, 2, 14, 2, 136, 63, 2, 80, 32, 150, 112, 9, 0], np.int32)
voice = np.random.RandomState(0).randn(1024).astype(np.float32) * 0.1
S = 192
for B in Bs:
    ref = None
    for rnd in range(2):
        for on in (0, 1):
            eng.set_option(opt, on)
            eng.seed(4242)
            eng.autoregressive(toks, voice, B, S, mask_stop=True)  # re-captures the graph
            eng.set_option("prof_only:ar_decode_step", 1)
            eng.prof_reset(True)
            best = 1e9
            for rep in range(3):
                eng.seed(4242)
                t0 = time.time()
                codes, _, _, _ = eng.autoregressive(toks, voice, B, S, mask_stop=True)
                best = min(best, time.time() - t0)
            ms, n, nbytes = eng.prof_get("ar_decode_step")
            eng.prof_reset(False)
            if ref is None:
                ref = codes
            print("B=%2d %s=%d: AR stage %.1f ms, decode step %.1f us (%d timed, %.0f GB/s), codes identical to the first run: %s"
                  % (B, opt, on, 1e3 * best, 1e3 * ms / max(n, 1), n, nbytes / max(ms, 1e-9) / 1e6, bool((codes == ref).all())), flush=True)
eng.close()
