#!/usr/bin/env bash
# TEST INFRASTRUCTURE. Builds oracle/_ref/libref.so from the reference's own sources *where they
# lie* under /root/reference: common.cpp (tokenizer; ggml-free) is compiled as-is, and the
# ggml-free line ranges of main.cpp (RNG globals, sampler, rel-pos buckets, sequence
# bookkeeping, diffusion schedule/update math) are streamed by sed straight into the compiler
# together with oracle/ref_shim.cpp. Nothing from /root/reference is written into this repo;
# the only output is the shared object under oracle/_ref/ (git-ignored, NOT gpurun-ignored).
#
# The rest of main.cpp (the ggml graphs) cannot be built here: the ggml/ submodule directory is
# empty in the checkout — the network stages are therefore "parity unpinned".
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${REF_ROOT:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -f "$REF/main.cpp" ] || [ ! -f "$REF/common.cpp" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt $OUT/libref.so" >&2
  exit 0
fi
mkdir -p "$OUT"
CXX="${CXX:-g++}"
"$CXX" -O2 -fPIC -std=c++17 -w -c "$REF/common.cpp" -o "$OUT/common_ref.o"
{
  echo '#include "common.h"'
  echo '#include <numeric>'
  echo '#include <stdexcept>'
  echo '#include <limits>'
  sed -n '15,27p;39,57p;4510,4532p;4562,4749p;4809,4817p;4873,4915p;5369,5612p' "$REF/main.cpp"
  cat "$HERE/ref_shim.cpp"
} | "$CXX" -O2 -fPIC -std=c++17 -w -I"$REF" -x c++ -c - -o "$OUT/main_ranges_ref.o"
"$CXX" -shared -o "$OUT/libref.so" "$OUT/main_ranges_ref.o" "$OUT/common_ref.o"
rm -f "$OUT/main_ranges_ref.o" "$OUT/common_ref.o"
echo "built $OUT/libref.so"
