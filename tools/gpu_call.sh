#!/bin/bash
# Developer helper: one gpurun call = a list of steps, each logged under gpurun_out/<tag>/ (the only directory that comes back).
# usage: tools/gpu_call.sh <tag> <step> [<step> ...]   steps: tests | tests:<pytest args> | diag | bench[:args] | cmd:<shell command>
cd "$(dirname "$0")/.." || exit 1
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  t0=$(date +%s)
  case $name in
    tests) timeout 2400 python -m pytest tests -m gpu -q -s $arg > "$out/tests.log" 2>&1; echo "tests rc=$? $(tail -1 "$out/tests.log")";;
    diag) for b in gemm_diag gemm_diag_trace; do timeout 300 tools/bin/$b > "$out/$b.log" 2>&1; echo "$b rc=$?"; done;;
    bench) timeout 1200 python bench.py $arg > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$? $(head -c 300 "$out/bench.json")";;
    cmd) bash -c "$arg" > "$out/cmd_$(echo "$arg" | md5sum | head -c 6).log" 2>&1; echo "cmd rc=$?";;
  esac
  echo "  [$name took $(( $(date +%s) - t0 )) s]"
done
