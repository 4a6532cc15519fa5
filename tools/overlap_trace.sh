#!/bin/bash
# Do the kernels of two renders that share a device overlap? rocprofv3 --kernel-trace of rank 0's shard of an 8-rank C2 job rendered as two half-shards in flight
# (pt_multi_render, replicas 0,0), then per queue: busy time, and the time both queues had a kernel running. usage (through gpurun): tools/overlap_trace.sh <tag>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --config C2 --sim-world 8 --steps 2 --warmup 1 --projection off --other-configs off --cpu-seconds 0 --gpus 2 --devices 0,0 --in-process > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "kt", "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f, newline="")))
print("columns:", list(rows[0].keys()))
q = defaultdict(list)
for r in rows:
    q[r.get("Queue_Id") or r.get("Stream_Id")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
t_end = max(e for v in q.values() for _, e, _ in v)
ev = []
for k, v in q.items():
    v = [x for x in v if x[0] > t_end - 200e6]      # the last 200 ms: the timed steps
    print("queue", k, "kernels", len(v), "busy ms", round(sum(e - s for s, e, _ in v) / 1e6, 2))
    for s, e, _ in v: ev += [(s, 1), (e, -1)]
ev.sort()
depth = 0; last = None; t = defaultdict(float)
for ts, d in ev:
    if last is not None: t[depth] += ts - last
    depth += d; last = ts
print("ms with 0 / 1 / 2+ kernels running:", round(t[0] / 1e6, 2), round(t[1] / 1e6, 2), round(sum(v for k, v in t.items() if k >= 2) / 1e6, 2))
PY
find $OUT -name '*kernel_trace.csv' -size +8M -delete
