#!/usr/bin/env python3
"""One compact line from a bench.py JSON line: tools/show_bench.py <file> [label]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
k = d["kernels_ms_per_step"]
print(sys.argv[2] if len(sys.argv) > 2 else d["config"]["name"], d["value"], "Msamples/s", {n: round(x["ms"], 1) for n, x in k.items() if x["ms"] >= 0.05})
