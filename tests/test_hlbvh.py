"""HLBVH built on the GPU (csrc/gpu_bvh.hip) against its CPU restatement (oracle/ref_hlbvh.h).
The reference's own HLBVH path is broken / timing dependent (bvh.rs:424-455,488-493), so the statements pinned here are:
  * the oracle's tree is a valid BVH over every primitive and traces to exactly the SAH tree's hits (CPU tests);
  * the GPU builder returns that same tree, bit for bit (index work: bit-exact);
  * rendering through it matches the oracle (bit-exact counters) and the SAH render (same hits)."""
import numpy as np
import pytest


def _hlbvh(sd, pkg):
    sd.split_method = pkg._abi.PT_SPLIT_HLBVH
    return sd


def _scenes(pkg):
    """name -> (SceneData, RenderParams): triangles only, two-level with instances, spheres, degenerate layouts."""
    S = pkg.scenes
    out = {
        "ganesha": S.ganesha_scale(n=24, xres=64, yres=48, spp=4).world_end(),
        "instances": S.instanced_garden(xres=64, yres=48, spp=4).world_end(),
        "spheres": S.spheres_c1(xres=48, yres=48, spp=4).world_end(),
    }
    # 300 copies of one triangle (identical Morton codes -> split by position) + a few distinct ones
    b = S.ganesha_scale(n=6, xres=32, yres=24, spp=2)
    tri = np.array([[0, 0.2, 0], [0.3, 0.2, 0], [0, 0.5, 0]], np.float32)
    b.trianglemesh(np.tile(tri, (300, 1)), np.arange(900).reshape(-1, 3))
    out["duplicates"] = b.world_end()
    return out


def _check_tree(nodes, ordered, bounds, max_prims):
    """Structure of a LinearBVHNode array (bvh.rs:89-95,662-693): depth-first, every primitive in exactly one leaf,
    every node's box the union of what it holds."""
    n = len(nodes)
    seen = np.zeros(len(ordered), bool)
    stack = [0]
    expect_next = 0
    boxes = {}
    order = []
    while stack:
        i = stack.pop()
        assert i == expect_next, "nodes are not in depth-first order"
        expect_next += 1
        nd = nodes[i]
        order.append(i)
        if nd.n_prims:
            assert nd.n_prims <= max_prims
            sl = slice(nd.offset, nd.offset + nd.n_prims)
            assert not seen[sl].any()
            seen[sl] = True
            pb = bounds[ordered[sl]]
            assert np.array_equal(np.array(nd.bmin[:], np.float32), pb[:, :3].min(0)) and np.array_equal(np.array(nd.bmax[:], np.float32), pb[:, 3:].max(0))
        else:
            assert nd.axis < 3 and i + 1 < nd.offset < n
            stack.append(nd.offset); stack.append(i + 1)
    assert expect_next == n and seen.all()
    assert sorted(ordered.tolist()) == list(range(len(ordered)))
    for i in reversed(order):   # interior boxes = union of the children's boxes
        nd = nodes[i]
        lo, hi = np.array(nd.bmin[:], np.float32), np.array(nd.bmax[:], np.float32)
        if not nd.n_prims:
            (l0, h0), (l1, h1) = boxes[i + 1], boxes[nd.offset]
            assert np.array_equal(lo, np.minimum(l0, l1)) and np.array_equal(hi, np.maximum(h0, h1))
        boxes[i] = (lo, hi)


def _tri_bounds(sd):
    P = sd.P.reshape(-1, 3); idx = sd.idx.reshape(-1, 3)
    v = P[idx]
    return np.concatenate([v.min(1), v.max(1)], axis=1).astype(np.float32)


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3)) * 6 - 3).astype(np.float32); o[:, 1] = np.abs(o[:, 1]) + 0.1
    t = (rng.random((n, 3)) * 2 - 1).astype(np.float32) * 0.8
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("name", ["ganesha", "duplicates"])
def test_oracle_hlbvh_tree_is_valid(pkg, oracle, name):
    sd, rp = _scenes(pkg)[name]
    nodes, ordered = oracle.scene(_hlbvh(sd, pkg)).bvh()
    _check_tree(nodes, ordered, _tri_bounds(sd), 4)
    sd.split_method = pkg._abi.PT_SPLIT_SAH
    sah_nodes, _ = oracle.scene(sd).bvh()
    assert np.array_equal(np.array(nodes[0].bmin[:]), np.array(sah_nodes[0].bmin[:])) and np.array_equal(np.array(nodes[0].bmax[:]), np.array(sah_nodes[0].bmax[:]))


def test_oracle_hlbvh_morton_order_and_leaf_limit(pkg, oracle):
    """ordered_prims is the stable Morton order of the centroids (bvh.rs:392-403,832-857); maxnodeprims is honoured."""
    sd, rp = _scenes(pkg)["ganesha"]
    sd.max_node_prims = 2
    nodes, ordered = oracle.scene(_hlbvh(sd, pkg)).bvh()
    assert max(nd.n_prims for nd in nodes) <= 2
    b = _tri_bounds(sd)
    c = np.float32(0.5) * b[:, :3] + np.float32(0.5) * b[:, 3:]
    lo, hi = c.min(0), c.max(0)
    q = np.minimum(((c - lo) / (hi - lo) * np.float32(1024.0)).astype(np.uint32), 1023)

    def spread(x):
        x = x.astype(np.uint32)
        x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
        return x
    code = (spread(q[:, 2]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 0])
    assert np.array_equal(ordered, np.argsort(code, kind="stable"))


@pytest.mark.parametrize("name", ["ganesha", "instances", "spheres", "duplicates"])
def test_oracle_hlbvh_hits_equal_sah_hits(pkg, oracle, name):
    """Closest hits do not depend on the tree (apart from equal-t ties): same primitive, t and barycentrics."""
    sd, rp = _scenes(pkg)[name]
    sah = oracle.scene(sd)
    hl = oracle.scene(_hlbvh(sd, pkg))
    o, d = _rays(20000, 5)
    if name == "instances": o[:, 1] += 2.0
    tmax = np.full(len(o), np.inf, np.float32)
    a, b = sah.trace_closest(o, d, tmax), hl.trace_closest(o, d, tmax)
    assert (a[0] != 0xFFFFFFFF).mean() > 0.05
    if name == "duplicates":   # 300 coincident triangles tie at equal t: which copy wins is the traversal order's choice
        assert np.array_equal(a[0] == 0xFFFFFFFF, b[0] == 0xFFFFFFFF)
    else: assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    tm = np.full(len(o), 4.0, np.float32)
    assert np.array_equal(sah.trace_any(o, d, tm), hl.trace_any(o, d, tm))


@pytest.mark.parametrize("k", [1, 2, 5])
def test_oracle_hlbvh_tiny_scenes(pkg, oracle, k):
    rng = np.random.default_rng(k)
    b = pkg.host.SceneBuilder()
    b.trianglemesh(rng.random((3 * k, 3)).astype(np.float32), np.arange(3 * k).reshape(-1, 3))
    sd, rp = b.world_end()
    nodes, ordered = oracle.scene(_hlbvh(sd, pkg)).bvh()
    assert len(ordered) == k and (k > 1 or len(nodes) == 1)   # treelets never merge: two primitives in different cells are two leaves
    _check_tree(nodes, ordered, _tri_bounds(sd), 4)


def test_front_end_accelerator_directive(pkg):
    """api.rs make_accelerator / bvh.rs:918-940: "splitmethod" sah | hlbvh are honoured, the others are refused loudly."""
    world = 'WorldBegin\nShape "sphere"\nWorldEnd\n'
    A = pkg._abi
    assert pkg.frontend.FrontScene(text=world).desc().split_method == A.PT_SPLIT_SAH
    fs = pkg.frontend.FrontScene(text='Accelerator "bvh" "string splitmethod" "hlbvh" "integer maxnodeprims" 3\n' + world)
    assert fs.desc().split_method == A.PT_SPLIT_HLBVH and fs.desc().max_node_prims == 3
    with pytest.raises(Exception, match="splitmethod"):
        pkg.frontend.FrontScene(text='Accelerator "bvh" "string splitmethod" "middle"\n' + world)


# ---------------------------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ganesha", "instances", "spheres", "duplicates"])
def test_gpu_hlbvh_tree_equals_oracle_tree(pkg, gpu, oracle, name):
    sd, rp = _scenes(pkg)[name]
    _hlbvh(sd, pkg)
    gn, go = pkg.Scene(gpu, sd).bvh(); on, oo = oracle.scene(sd).bvh()
    assert len(gn) == len(on) and np.array_equal(go, oo)
    assert bytes(gn) == bytes(on)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 5])
def test_gpu_hlbvh_tiny_scenes(pkg, gpu, oracle, k):
    rng = np.random.default_rng(k)
    b = pkg.host.SceneBuilder()
    b.trianglemesh(rng.random((3 * k, 3)).astype(np.float32), np.arange(3 * k).reshape(-1, 3))
    sd, rp = b.world_end()
    gn, go = pkg.Scene(gpu, _hlbvh(sd, pkg)).bvh(); on, oo = oracle.scene(sd).bvh()
    assert np.array_equal(go, oo) and bytes(gn) == bytes(on)


@pytest.mark.gpu
@pytest.mark.parametrize("maxp", [1, 4, 9])
def test_gpu_hlbvh_large_mesh_and_leaf_sizes(pkg, gpu, oracle, maxp):
    """~80k triangles: several radix tiles per pass, thousands of treelets, the SAH upper levels."""
    sd, rp = pkg.scenes.ganesha_scale(n=200, xres=64, yres=48, spp=2).world_end()
    sd.max_node_prims = maxp
    _hlbvh(sd, pkg)
    g = pkg.Scene(gpu, sd)
    gn, go = g.bvh(); on, oo = oracle.scene(sd).bvh()
    assert np.array_equal(go, oo) and bytes(gn) == bytes(on)
    if maxp == 4: _check_tree(gn, go, _tri_bounds(sd), maxp)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ganesha", "instances", "spheres"])
def test_gpu_hlbvh_render_matches_oracle_and_sah(pkg, gpu, oracle, name):
    from test_gpu_parity import _compare_render
    sd, rp = _scenes(pkg)[name]
    sah_film = pkg.Scene(gpu, sd).render(rp)
    film, ref = _compare_render(pkg, gpu, oracle, _hlbvh(sd, pkg), rp)
    assert np.array_equal(film[..., 3], sah_film[..., 3])
    np.testing.assert_allclose(film[..., :3], sah_film[..., :3], rtol=2e-5, atol=1e-6)   # same hits; only the film's atomic order differs


@pytest.mark.gpu
def test_gpu_hlbvh_through_the_front_end(pkg, gpu, tmp_path):
    """Accelerator "bvh" "string splitmethod" "hlbvh" in a .pbrt file selects the GPU builder."""
    import os
    src = open(os.path.join(os.path.dirname(__file__), "scenes", "spheres_c1.pbrt")).read()
    assert "WorldBegin" in src
    text = src.replace("WorldBegin", 'Accelerator "bvh" "string splitmethod" "hlbvh"\nWorldBegin', 1)
    fs_h = pkg.frontend.FrontScene(text=text, base_dir=os.path.join(os.path.dirname(__file__), "scenes"))
    fs_s = pkg.frontend.FrontScene(text=src, base_dir=os.path.join(os.path.dirname(__file__), "scenes"))
    assert fs_h.desc().split_method == pkg._abi.PT_SPLIT_HLBVH and fs_s.desc().split_method == pkg._abi.PT_SPLIT_SAH
    rp = fs_s.render_params()
    a = pkg.Scene(gpu, fs_s).render(rp); b = pkg.Scene(gpu, fs_h).render(fs_h.render_params())
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=2e-5, atol=1e-6)
