#!/bin/bash
# FETCH_SIZE calibration on 64-byte gathers (tools/microbench/gather_bench.hip): separate PMC passes, --kernel-trace only.
TAG=${1:-gather}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN=$REPO/tools/microbench/gather_bench
for MB in 200 2048; do
  $BIN $MB 256 > $OUT/plain_$MB.jsonl 2> $OUT/plain_$MB.err
  i=0
  for CTRS in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p${i}_$MB -- $BIN $MB 256 > $OUT/p${i}_$MB.jsonl 2> $OUT/p${i}_$MB.err || { echo "pass $i ($CTRS) failed"; tail -5 $OUT/p${i}_$MB.err; }
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for mb in (200, 2048):
    req = [json.loads(l) for l in open(os.path.join(out, f"plain_{mb}.jsonl")) if l.startswith("{")]
    ctr = {}
    for f in glob.glob(os.path.join(out, f"p*_{mb}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip(); d = int(r["Dispatch_Id"])
            e = ctr.setdefault((k, d), {}); e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    byk = {}
    for (k, d) in sorted(ctr, key=lambda kd: kd[1]): byk.setdefault(k, []).append(ctr[(k, d)])
    print(f"== table {mb} MB")
    seen = {}
    for r in req:
        k = r["kernel"]; i = seen.get(k, 0); seen[k] = i + 1
        # the three PMC passes each dispatched every kernel once more: merge the i-th dispatch of each pass
        c = {}
        lst = byk.get(k, [])
        per = max(1, len(lst) // 3) if k != "k_stream" else max(1, len(lst) // 3)
        for e in lst[i::per] if per else []: c.update(e)
        fs = c.get("FETCH_SIZE", float("nan")) * 1024.0
        line = dict(kernel=k, table_MB=mb, requested_bytes=r["requested_bytes"], touched_64B_lines_bytes=r["touched_64B_lines_bytes"], ms=r["ms"], FETCH_SIZE_bytes=fs,
                    fetch_over_requested=fs / r["requested_bytes"], fetch_over_lines=fs / r["touched_64B_lines_bytes"], **{n: v for n, v in c.items() if n != "FETCH_SIZE"})
        print(json.dumps(line)); res.setdefault(str(mb), []).append(line)
json.dump(res, open(os.path.join(out, "gather_calibration.json"), "w"), indent=1)
PY
