"""print selected keys of the last JSON line of a bench log: python tools/show_bench.py <log> key [key ...]"""
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("no JSON line in", sys.argv[1]); sys.exit(0)
d = json.loads(lines[-1])
for k in sys.argv[2:]:
    print(k, json.dumps(d.get(k))[:3000])
