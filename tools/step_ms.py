"""ms_per_step (and the forward's) of one bench.py line read from stdin: python bench.py ... | python tools/step_ms.py"""
import json
import sys

line = [x for x in sys.stdin if x.strip().startswith("{")][-1]
rec = json.loads(line)
print("ms_per_step", rec["ms_per_step"], "fwd", rec.get("fwd_only", {}).get("ms_per_step"), "l1", rec.get("trained_weights_l1"))
