#!/usr/bin/env python3
"""tools/jline.py FILE [key.path ...] -- print chosen fields of the last JSON line of a bench.py log."""
import json
import sys

d = None
for line in open(sys.argv[1]):
    if line.startswith("{"):
        try:
            d = json.loads(line)
        except ValueError:
            pass
if d is None:
    print(sys.argv[1], "no JSON line")
    sys.exit(0)
out = []
for path in sys.argv[2:]:
    v = d
    for k in path.split("."):
        v = v.get(k) if isinstance(v, dict) else None
    out.append("%s=%s" % (path, v))
print(sys.argv[1].split("/")[-1], " ".join(out))
