import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _pkg import import_pkg
pkg = import_pkg(); A = pkg._abi
from oracle.oracle_binding import Oracle
lib = pkg.load_library(); lib.init(0)
orc = Oracle(A, pkg.runtime.TABLES_PATH)
for sb in ((-1, -1, 57, 33), (-1, -1, 65, 33)):
    rng = np.random.default_rng(1); n, nd = 4096, 64
    xy = np.stack([rng.integers(sb[0], sb[2], n), rng.integers(sb[1], sb[3], n)], axis=1).astype(np.int32)
    sn = rng.integers(0, 6, n).astype(np.uint32)
    outs = []
    for fn in (lib.lib.pt_sobol_samples, orc.lib.orc_sobol_samples):
        out = np.zeros((n, nd), np.float32); idx = np.zeros(n, np.uint64)
        assert fn((C.c_int32 * 4)(*sb), n, xy.ctypes.data_as(A.i32p), sn.ctypes.data_as(A.u32p), nd, out.ctypes.data_as(A.fp), idx.ctypes.data_as(A.u64p)) == 0
        outs.append((out, idx))
    print(sb, "index equal", np.array_equal(outs[0][1], outs[1][1]), "values equal", np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32)), "max index", int(outs[1][1].max()))
