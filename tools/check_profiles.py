#!/usr/bin/env python3
"""Profiles that cannot go stale silently.

The current round's record lives in profiles/r<N>/final/ with a MANIFEST.json written by `tools/check_profiles.py --write-manifest <dir>` on the box that
took it: {"code_hash": tools/code_hash.py's hash of the device sources the record was taken on, "files": {name: sha256}}. This check fails when
  * a file under that final/ is not in the manifest, is missing, or changed since (a file copied in from another run: round 5's stale kernel_stats CSV);
  * bench*.json / stats_bench.json in it carry another `code_hash` than the manifest's;
  * a record of profiles/pmc_traffic.json carries another code hash than the manifest's (--strict: than the LIVE tree's);
  * DESIGN.md section 7 quotes another hash, or its "Dominant kernel `K` (X ms per launch by HIP events, Y in `<csv>`)" sentence differs from the
    CSV's average for K (Y) or from bench.json's roofline.avg_launch_ms (X) by more than 1 %;
  * (--strict, what tests/test_profiles_fresh.py runs) the manifest's hash is not the hash of the tree as it is now: the record is of other code.
Exit code 0 = fresh, 1 = stale (reasons on stdout)."""
import csv
import glob
import hashlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from code_hash import code_hash

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = 6


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def final_dir(rnd=ROUND):
    return os.path.join(ROOT, "profiles", f"r{rnd}", "final")


def write_manifest(d):
    files = {}
    for f in sorted(glob.glob(os.path.join(d, "**", "*"), recursive=True)):
        if os.path.isfile(f) and os.path.basename(f) != "MANIFEST.json":
            files[os.path.relpath(f, d)] = sha(f)
    json.dump({"code_hash": code_hash(ROOT), "files": files}, open(os.path.join(d, "MANIFEST.json"), "w"), indent=1)
    print(f"manifest: {len(files)} files, code hash {code_hash(ROOT)}")


def last_json_line(path):
    lines = [l for l in open(path).read().strip().split("\n") if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def check(strict=False, rnd=ROUND):
    bad = []
    d = final_dir(rnd)
    mpath = os.path.join(d, "MANIFEST.json")
    if not os.path.exists(mpath):
        return [f"{os.path.relpath(mpath, ROOT)} does not exist: round {rnd} has no final record yet (tools/r{rnd}_profiles.sh final)"]
    man = json.load(open(mpath))
    h = man["code_hash"]
    live = code_hash(ROOT)
    if strict and h != live:
        bad.append(f"profiles/r{rnd}/final was taken on code {h}, the tree is {live}: re-take it (tools/r{rnd}_profiles.sh final + pmc) or the numbers quoted from it describe other code")
    on_disk = {os.path.relpath(f, d) for f in glob.glob(os.path.join(d, "**", "*"), recursive=True) if os.path.isfile(f)} - {"MANIFEST.json"}
    for f in sorted(on_disk - set(man["files"])):
        bad.append(f"final/{f} is not in the manifest (copied in from another run?)")
    for f, digest in man["files"].items():
        p = os.path.join(d, f)
        if not os.path.exists(p):
            bad.append(f"final/{f} is in the manifest but missing")
        elif sha(p) != digest:
            bad.append(f"final/{f} changed since the manifest was written")
    for f in sorted(on_disk):
        if re.fullmatch(r"(bench.*|stats_bench)\.json", os.path.basename(f)):
            line = last_json_line(os.path.join(d, f))
            if line is not None and line.get("code_hash") != h:
                bad.append(f"final/{f} carries code hash {line.get('code_hash')}, the manifest {h}")
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    want = live if strict else h
    for cfg, recs in pmc.items():
        stale = sorted(k for k, r in recs.items() if r.get("code_hash") != want)
        if stale:
            bad.append(f"profiles/pmc_traffic.json {cfg}: {len(stale)} of {len(recs)} kernel records carry another code hash than {want} (e.g. {stale[0]}: {recs[stale[0]].get('code_hash')})")
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec7 = design[design.index("## 7."):]
    m = re.search(r"hash `([0-9a-f]{16})`", sec7)
    if not m or m.group(1) != h:
        bad.append(f"DESIGN.md section 7 quotes hash {m.group(1) if m else None}, profiles/r{rnd}/final is {h}")
    m = re.search(r"Dominant kernel\s+`([^`]+)`\s+\(([\d.]+)\s+ms\s+per\s+launch\s+by\s+HIP\s+events,\s+([\d.]+)\s+in\s+`([^`]+)`\)", sec7)
    if not m:
        bad.append("DESIGN.md section 7 has no \"Dominant kernel `K` (X ms per launch by HIP events, Y in `<csv>`)\" sentence to check")
    else:
        kern, x, y, rel = m.group(1), float(m.group(2)), float(m.group(3)), m.group(4)
        cpath = os.path.join(ROOT, rel)
        avg = None
        if os.path.exists(cpath):
            for r in csv.DictReader(open(cpath)):
                if kern in r.get("Name", ""):
                    avg = float(r["AverageNs"]) / 1e6
                    break
        if avg is None:
            bad.append(f"DESIGN.md section 7 quotes {rel} for {kern}: no such file or kernel row")
        elif abs(avg - y) > 0.01 * avg:
            bad.append(f"DESIGN.md section 7 quotes {y} ms per launch of {kern} from {rel}; the file says {avg:.3f}")
        bpath = os.path.join(d, "bench.json")
        line = last_json_line(bpath) if os.path.exists(bpath) else None
        if line is None:
            bad.append("final/bench.json missing")
        else:
            ev = line["roofline"]["avg_launch_ms"]
            if abs(ev - x) > 0.01 * ev or kern not in line["roofline"]["kernel"]:
                bad.append(f"DESIGN.md section 7 quotes {x} ms by HIP events for {kern}; final/bench.json says {ev} for {line['roofline']['kernel']}")
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--write-manifest":
        write_manifest(sys.argv[2]); sys.exit(0)
    problems = check(strict="--strict" in sys.argv)
    for p in problems:
        print("STALE:", p)
    print("profiles fresh" if not problems else f"{len(problems)} problem(s)")
    sys.exit(1 if problems else 0)
