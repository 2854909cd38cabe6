"""Developer aid: condense bench.py's JSON line (stdin) to the few numbers compared in A/B runs."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(" ".join(sys.argv[1:]), d["value"], d["ms_per_step"], d["stage_ms_per_step"], r["achieved"], r.get("launches_timed"))
