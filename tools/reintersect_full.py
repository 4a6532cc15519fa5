#!/usr/bin/env python3
"""tests/shapes.rs:173-224 triangle_reintersect at the reference's full count: RNG::new(0..999), 10 000 spawned ray pairs per hit
triangle (tests/test_oracle_kats.py runs 2 000 per triangle to stay fast). Prints one JSON line; a round's result is kept in profiles/."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _pkg import import_pkg
from oracle.oracle_binding import Oracle
pkg = import_pkg()
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
orc.lib.orc_test_triangle_reintersect.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
n = C.c_int(); t0 = time.time()
failures = orc.lib.orc_test_triangle_reintersect(1000, 10000, C.byref(n))
print(json.dumps(dict(test="tests/shapes.rs:173-224 triangle_reintersect", seeds=1000, rays_per_triangle=10000, triangles_hit=n.value, failures=failures, seconds=round(time.time() - t0, 1))))
sys.exit(1 if failures else 0)
