"""Development helper: `python bench.py ... | python scripts/_pp.py` prints `value` and the pipelined region's value of
the bench line (A/B runs of environment switches on one box)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(d["value"], d.get("pipelined", {}).get("value"))
