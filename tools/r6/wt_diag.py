"""Diagnosis: where does pt_trace_closest differ from the oracle on the triangle_watertight rays (ties at shared vertices)?"""
import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _pkg import import_pkg
pkg = import_pkg()
from oracle.oracle_binding import Oracle
from test_oracle_kats import _watertight
orc_lib = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
lib = pkg.load_library(); lib.init(0)
failures, v, idx, ro, rd, nh = _watertight(orc_lib, 100000, 0)
b = pkg.host.SceneBuilder(); b.trianglemesh(v, idx); sd, _ = b.world_end()
g = pkg.Scene(lib, sd); orc = orc_lib.scene(sd)
tmax = np.full(len(ro), np.inf, np.float32)
op, ot, ob = orc.trace_closest(ro, rd, tmax)
nodes, ordered = orc.bvh()
pos = np.zeros(len(ordered), np.int64); pos[ordered] = np.arange(len(ordered))
for exact in (False, True):
    lib.set_trace_exact(exact)
    gp, gt, gb = g.trace_closest(ro, rd, tmax)
    bad = np.nonzero((gp != op) | (gt.view(np.uint32) != ot.view(np.uint32)) | (gb.view(np.uint32) != ob.view(np.uint32)).any(axis=1))[0]
    print("exact" if exact else "quad", "mismatches", len(bad), "of", len(ro), "vertex rays among them", int((bad % 2 == 1).sum()))
    for i in bad[:12]:
        print(" ray", i, "o", ro[i], "d", rd[i], "nhits", nh[i])
        print("   gpu prim", gp[i], "t", gt[i].view(np.uint32), float(gt[i]), "b", gb[i], "leaf pos", pos[gp[i]] if gp[i] < len(pos) else -1)
        print("   orc prim", op[i], "t", ot[i].view(np.uint32), float(ot[i]), "b", ob[i], "leaf pos", pos[op[i]] if op[i] < len(pos) else -1)
    tp = np.equal(gp, op).mean(); print("  same prim frac", tp, "same t frac", np.equal(gt.view(np.uint32), ot.view(np.uint32)).mean())
lib.set_trace_exact(False)
