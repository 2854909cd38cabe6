#!/bin/bash
# round 6: wall clock of the drop-in CLI for one utterance (process start -> WAV on disk) with full-size synthetic weights, and the three loads timed on their own
cd "$(dirname "$0")/../.." || exit 1
R=$(pwd); out=gpurun_out/r6cli; mkdir -p $out
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, ".")
import tortoise_cpp_amd_loader
tortoise_cpp_amd_loader.load()
import bench, shutil, os
d = "/tmp/tts_bench_models"
bench.ensure_models(d, False, True)
shutil.copy("models/tokenizer.json", d + "/tokenizer.json")
PY
for i in 1 2 3; do
  t0=$(date +%s%N)
  ./tortoise.cpp_amd/tortoise --models /tmp/tts_bench_models --message "this is a test message." --voice models/mol.bin --seed 3 --codes 192 --output /tmp/o$i.wav --timing 1 $EXTRA 2>&1 | tail -16
  echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
done
python - <<'PY'
import time, sys
sys.path.insert(0, ".")
import tortoise_cpp_amd_loader
pkg = tortoise_cpp_amd_loader.load()
t0 = time.time(); eng = pkg.Engine(0); t1 = time.time()
print("tts_create %.3f s" % (t1 - t0))
d = "/tmp/tts_bench_models"
for k, f in (("ar", "ggml-model.bin"), ("diffusion", "ggml-diffusion-model.bin"), ("vocoder", "ggml-vocoder-model.bin")):
    for rep in range(2):
        t0 = time.time(); eng.load(**{k: d + "/" + f}); t1 = time.time()
        print("load %-10s %.3f s" % (k, t1 - t0))
PY
