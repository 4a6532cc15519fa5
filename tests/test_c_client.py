"""include/mi355pt.h as a C compiler sees it. (1) A probe program built by this test (gcc -std=c99 -pedantic) prints sizeof / offsetof of every struct and field and the value of
every PT_* constant the ctypes mirror (pbrt-rust_amd/_abi.py) declares; both must agree -- the header is what bindgen will read, the mirror is what
every test of this repo drives the library through. (2) A plain-C client (tests/c_client/client.c) does INTEGRATION.md section 2's call sequence; on the GPU its film must be the
film the ctypes path renders, bit for bit in the weights, and the oracle's."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _structs(A):
    return [(n, t) for n, t in vars(A).items() if isinstance(t, type) and issubclass(t, C.Structure) and t is not C.Structure and n.startswith("Pt")]


def test_struct_layouts_and_constants_equal_what_a_c_compiler_sees(pkg, tmp_path):
    A = pkg._abi
    structs = _structs(A)
    assert {"PtSceneDesc", "PtRenderParams", "PtCounters", "PtKernelStat", "PtMaterial", "PtLight", "PtSphere", "PtBVHNode", "PtInstance", "PtTexture", "PtImage", "PtMedium",
            "PtObject", "PtBSSRDFTable"} <= {n for n, _ in structs}
    consts = sorted(n for n, v in vars(A).items() if re.fullmatch(r"PT_[A-Z0-9_]+", n) and isinstance(v, int))
    assert len(consts) > 60
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mi355pt.h"', 'int main(void) {']
    for name, t in structs:
        src.append(f'    printf("S {name} %zu\\n", sizeof({name}));')
        for f in t._fields_:
            src.append(f'    printf("F {name}.{f[0]} %zu %zu\\n", offsetof({name}, {f[0]}), sizeof((({name} *)0)->{f[0]}));')
    for c in consts:
        src.append(f'    printf("C {c} %lld\\n", (long long){c});')
    src += ['    return 0;', '}']
    cfile = tmp_path / "probe.c"; cfile.write_text("\n".join(src))
    exe = tmp_path / "probe"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", INC, str(cfile), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]          # (a field or constant the mirror has and the header lacks fails here, by name)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    seen = {}
    for line in out:
        if line:
            p = line.split(); seen[(p[0], p[1])] = [int(x) for x in p[2:]]
    for name, t in structs:
        assert seen[("S", name)] == [C.sizeof(t)], name
        for f in t._fields_:
            d = getattr(t, f[0])
            assert seen[("F", f"{name}.{f[0]}")] == [d.offset, d.size], f"{name}.{f[0]}"
    for c in consts:
        assert seen[("C", c)] == [getattr(A, c)], c
    # the other direction: every struct member and every enumerator of the header is in the mirror (a field appended to the header only would not shift anything above)
    hdr = open(os.path.join(INC, "mi355pt.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", body, flags=re.S):
        name = m.group(1); t = getattr(A, name)
        decl = [re.sub(r"\[[^\]]*\]", "", x.strip().split()[-1]).lstrip("*") for x in m.group(2).split(";") if x.strip()]
        # `float a, b;` style declarations list several members
        members = []
        for x in m.group(2).split(";"):
            x = x.strip()
            if not x: continue
            parts = x.split(",")
            members.append(re.sub(r"\[[^\]]*\]", "", parts[0].split()[-1]).lstrip("*"))
            members += [re.sub(r"\[[^\]]*\]", "", q.strip()).lstrip("*") for q in parts[1:]]
        assert members == [f[0] for f in t._fields_], name
        del decl
    for m in re.finditer(r"\b(PT_[A-Z0-9_]+)\s*=", body):
        assert hasattr(A, m.group(1)), m.group(1)


def _build_client(tmp_path, pkg):
    exe = tmp_path / "client"
    libdir = os.path.dirname(pkg.runtime.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", INC, os.path.join(ROOT, "tests", "c_client", "client.c"), "-o", str(exe),
                        "-L", libdir, "-lmi355pt", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_plain_c_client_builds_against_the_header_and_links_the_library(pkg, tmp_path):
    _build_client(tmp_path, pkg)


@pytest.mark.gpu
def test_plain_c_client_renders_what_the_ctypes_path_renders(pkg, gpu, oracle, tmp_path):
    from conftest import trace_env
    exe = _build_client(tmp_path, pkg)
    b = pkg.scenes.ganesha_scale(n=16, xres=64, yres=48, spp=4, env=False)
    sd, rp = b.world_end()
    d = sd.desc()
    assert d.n_spheres == 0 and d.n_textures == 0 and d.n_instances == 0 and not d.env_texels   # what the client fills in
    np.asarray(sd.P, np.float32).tofile(tmp_path / "P.bin"); np.asarray(sd.idx, np.uint32).tofile(tmp_path / "indices.bin")
    np.asarray(sd.tri_flags, np.uint8).tofile(tmp_path / "tri_flags.bin")
    for name in ("prim_shape", "prim_material", "prim_light"):
        np.asarray(getattr(sd, name), np.uint32).tofile(tmp_path / f"{name}.bin")
    (tmp_path / "materials.bin").write_bytes(bytes(sd.materials)); (tmp_path / "lights.bin").write_bytes(bytes(sd.lights))
    (tmp_path / "render_params.bin").write_bytes(bytes(rp))
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=trace_env())
    assert r.returncode == 0, (r.stdout, r.stderr)
    g = pkg.Scene(gpu, sd)
    film = g.render(rp); gc = g.counters()
    cfilm = np.fromfile(tmp_path / "film.bin", np.float32).reshape(film.shape)
    assert np.array_equal(cfilm[..., 3], film[..., 3])
    np.testing.assert_allclose(cfilm, film, rtol=2e-6, atol=1e-7)
    words = r.stdout.split()
    cc = {words[i]: int(words[i + 1]) for i in range(0, len(words), 2)}
    for k, v in cc.items():
        assert gc[k] == v, k
    orc = oracle.scene(sd)
    ofilm = orc.render(rp, nthreads=4)
    np.testing.assert_allclose(cfilm, ofilm, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(np.fromfile(tmp_path / "rgb.bin", np.float32).reshape(48, 64, 3), orc.resolve(ofilm, rp.scale), rtol=2e-5, atol=1e-6)
