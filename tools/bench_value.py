"""Print value / ms_per_step / host_cores_busy of a bench.py JSON line read from stdin, prefixed by argv[1] (for sweeps)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d.get("host_cores_busy"))
