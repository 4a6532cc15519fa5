#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a compact text/JSON report.
FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide streaming reads (MI355X_MICROARCH.md, HBM):
reported raw AND doubled; WRITE_SIZE is exact for 16-B streaming stores and float atomics. Units: KiB."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from code_hash import code_hash
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>()]*>)?)", name)
    return m.group(1) if m else name.split("(")[0].replace("void ", "").strip()


report = {}
print("== kernel stats (rocprofv3 --kernel-trace --stats)")
stats = list(rows("stats/**/*kernel_stats.csv"))
for r in stats[:16]:
    name = short(r.get("Name", ""))
    calls, total, avg, pct = r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")
    print(f"{name:32s} calls={calls:>6s} total_ms={float(total)/1e6:10.3f} avg_us={float(avg)/1e3:10.2f} pct={pct}")
    report.setdefault("kernel_stats", []).append(dict(name=name, calls=int(calls), total_ms=float(total) / 1e6, avg_us=float(avg) / 1e3, pct=float(pct)))

for tag, counters in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_rd", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum"]),
                      ("pmc_write", ["WRITE_SIZE"]), ("pmc_l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
    for r in rows(f"{tag}/**/*counter_collection.csv"):
        k = short(r.get("Kernel_Name", ""))
        c = r.get("Counter_Name"); v = float(r.get("Counter_Value", 0))
        acc[k][c] += v
        if c == counters[0]:
            n[k] += 1
    if not acc:
        continue
    print(f"== {tag}")
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        line = f"{k:32s} dispatches={n[k]:5d} " + " ".join(f"{c}={acc[k][c]:.4g} (per dispatch {acc[k][c]/max(1,n[k]):.4g})" for c in counters)
        print(line)
        report.setdefault(tag, {})[k] = dict(dispatches=n[k], **{c: acc[k][c] for c in counters})

# --- HBM traffic per launch for bench.py's roofline.traffic.
# Reads: the memory-side read requests of the L2 by size, 128 * RDREQ_128B + 64 * RDREQ_64B + 32 * RDREQ_32B -- exact for any mix of
# wide streaming reads (128-byte requests, which FETCH_SIZE tallies at 64 bytes: the guide's "double it") and 64-byte gathers (which
# FETCH_SIZE counts exactly; profiles/r2_gather_calibration.json). Fallback when that pass is missing: FETCH_SIZE x 2 (upper bound).
# Writes: WRITE_SIZE (exact for 16-byte streaming stores and float atomics). FETCH_SIZE / WRITE_SIZE are in KiB.
ppass = int(sys.argv[2]) if len(sys.argv) > 2 else None
config = sys.argv[3] if len(sys.argv) > 3 else "C2"
traffic = {}
for k, w in report.get("pmc_write", {}).items():
    rd = report.get("pmc_rd", {}).get(k); fe = report.get("pmc_fetch", {}).get(k); l2 = report.get("pmc_l2", {}).get(k)
    if not w["dispatches"]:
        continue
    if rd and rd["dispatches"] and rd.get("TCC_EA0_RDREQ_sum"):
        read_b = (128.0 * rd["TCC_EA0_RDREQ_128B_sum"] + 64.0 * rd["TCC_EA0_RDREQ_64B_sum"] + 32.0 * rd["TCC_EA0_RDREQ_32B_sum"]) / rd["dispatches"]
        src = "rocprofv3 --pmc TCC_EA0_RDREQ_{128B,64B,32B}_sum (read bytes by request size) + WRITE_SIZE, separate passes"
    elif fe and fe["dispatches"]:
        read_b = fe["FETCH_SIZE"] * 1024.0 * 2.0 / fe["dispatches"]
        src = "rocprofv3 --pmc FETCH_SIZE x 2 (upper bound: exact for 128-byte streaming requests, 2x over for 64-byte gathers) + WRITE_SIZE"
    else:
        continue
    write_b = w["WRITE_SIZE"] * 1024.0 / w["dispatches"]
    traffic[k] = dict(hbm_bytes_per_launch=int(read_b + write_b), read_bytes_per_launch=int(read_b), write_bytes_per_launch=int(write_b),
                      fetch_size_kib_raw=fe["FETCH_SIZE"] if fe else None, rdreq=({c: rd[c] for c in rd if c != "dispatches"} if rd else None),
                      l2_hit_rate=(round(l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), 4) if l2 else None),
                      dispatches=w["dispatches"], spp_per_pass=ppass, workload=[1466, 1920, 1080], source=src, code_hash=code_hash())
report["traffic"] = traffic
json.dump({config: traffic}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
json.dump(report, open(os.path.join(out, "summary.json"), "w"), indent=1)
