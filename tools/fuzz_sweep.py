"""One-off wide sweep of tests/test_fuzz_parity.py's generator: python tools/fuzz_sweep.py <first> <last>"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np
from _pkg import import_pkg
pkg = import_pkg()
from oracle.oracle_binding import Oracle
import test_fuzz_parity as T
lib = pkg.load_library(); lib.init(0)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        T.test_gpu_matches_oracle_on_random_scenes.__wrapped__ if False else None
        T.test_gpu_matches_oracle_on_random_scenes(pkg, lib, orc, seed, "quad")
    except Exception as e:
        bad.append(seed); print("seed", seed, "FAILED:", str(e).splitlines()[0][:200], flush=True)
    if seed % 500 == 499: print("..", seed + 1, "checked,", len(bad), "failures so far", flush=True)   # (a silent GPU run is taken to be hung)
print("checked", sys.argv[1], "..", sys.argv[2], "failures:", bad)
