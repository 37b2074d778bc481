#!/usr/bin/env python3
"""Print selected fields of the last JSON line of a bench log: tools/jget.py <log> value ms_per_step config.batch_per_gpu"""
import json
import sys

line = [ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1]
j = json.loads(line)
out = []
for key in sys.argv[2:]:
    v = j
    for part in key.split("."):
        v = v[part]
    out.append(str(v))
print(" ".join(out))
