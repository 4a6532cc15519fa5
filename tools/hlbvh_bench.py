"""Time scene creation (accelerator build + upload) with the host SAH builder and the GPU HLBVH builder on the headline
scene, then render a few passes through each tree. usage: python tools/hlbvh_bench.py [n] [spp]"""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _pkg import import_pkg
pkg = import_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1466
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = pkg.load_library(); lib.init(0)
sd, rp = pkg.scenes.ganesha_scale(n=n, spp=spp).world_end()
out = {}
for name, sm in (("sah", pkg._abi.PT_SPLIT_SAH), ("hlbvh", pkg._abi.PT_SPLIT_HLBVH), ("hlbvh_again", pkg._abi.PT_SPLIT_HLBVH)):
    sd.split_method = sm
    t0 = time.time(); g = pkg.Scene(lib, sd); t1 = time.time()
    g.render(rp)   # warm
    t2 = time.time(); g.render(rp); t3 = time.time()
    c = g.counters()
    nn, _ = g.bvh()
    out[name] = dict(create_s=round(t1 - t0, 3), render_s=round(t3 - t2, 3), nodes=len(nn), nodes_visited=int(c["bvh_nodes_visited"]), tri_tests=int(c["triangle_tests"]))
    g.close()
print(json.dumps(out))
