#!/bin/bash
# round 4, GPU call 5: the whole GPU suite + smoke
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r4c5; mkdir -p $out
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
grep -E "passed|failed|rc=" $out/tests.log | tail -5; tail -3 $out/smoke.log
grep -E "\[reference precision|FAILED|Error" $out/tests.log | tail -40
