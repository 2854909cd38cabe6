#!/bin/bash
# round 5: the whole GPU suite + smoke with the unified loop gates
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/r5suite; mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 2000 python -m pytest tests -m gpu -q -s > $out/tests.log 2>&1; echo "tests rc=$? [$(( $(date +%s) - t0 )) s]"
grep -E "passed|failed" $out/tests.log | tail -3; grep -E "^FAILED|^ERROR|AssertionError" $out/tests.log | head -20
grep -E "loop|sampling|configs\[" $out/tests.log | cut -c1-260 | head -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log
