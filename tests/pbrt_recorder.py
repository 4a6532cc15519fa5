"""Test infrastructure: a SceneBuilder that also writes the .pbrt text of everything it is told, so that one seeded scene exists
twice -- through the Python mirror of api.rs (pbrt_rust_amd/host.py) and through the C++ front end (frontend/frontend.cpp)."""
import os
import numpy as np


def _f(x): return "%.9g" % float(np.float32(x))
def _vec(v): return "[" + " ".join(_f(x) for x in np.asarray(v, dtype=np.float32).reshape(-1)) + "]"


def make_recorder(pkg, out_dir):
    Base = pkg.host.SceneBuilder

    class Recorder(Base):
        def __init__(self):
            self.lines = []; self.out_dir = out_dir; self._n_files = 0; self._named = {}
            super().__init__()

        def _emit(self, s): self.lines.append(s)
        def _pfm(self, img):
            img = np.asarray(img, dtype=np.float32); name = "img%d.pfm" % self._n_files; self._n_files += 1
            with open(os.path.join(self.out_dir, name), "wb") as f:
                f.write(b"PF\n%d %d\n-1\n" % (img.shape[1], img.shape[0])); f.write(np.ascontiguousarray(img[::-1]).tobytes())
            return name

        def _params(self, kw, spectrum=(), textures_spec=(), textures_float=()):
            out = []
            for k, v in kw.items():
                if isinstance(v, bool): out.append('"bool %s" "%s"' % (k, "true" if v else "false"))
                elif isinstance(v, str):
                    if k in textures_spec or k in textures_float: out.append('"texture %s" "%s"' % (k, v))
                    else: out.append('"string %s" "%s"' % (k, v))
                elif isinstance(v, (int, np.integer)) and k in ("octaves", "dimension"): out.append('"integer %s" %d' % (k, int(v)))
                elif np.isscalar(v):
                    if k in spectrum: out.append('"rgb %s" %s' % (k, _vec([v] * 3)))
                    else: out.append('"float %s" %s' % (k, _f(v)))
                else: out.append('"%s %s" %s' % ("rgb" if k in spectrum else "vector" if k in ("v1", "v2") else "float", k, _vec(v)))
            return " ".join(out)

        # transforms / attributes
        def translate(self, x, y, z): self._emit("Translate %s %s %s" % (_f(x), _f(y), _f(z))); super().translate(x, y, z)
        def scale(self, x, y, z): self._emit("Scale %s %s %s" % (_f(x), _f(y), _f(z))); super().scale(x, y, z)
        def rotate(self, deg, x, y, z): self._emit("Rotate %s %s %s %s" % (_f(deg), _f(x), _f(y), _f(z))); super().rotate(deg, x, y, z)
        def look_at(self, e, l, u): self._emit("LookAt " + " ".join(_f(x) for x in (*e, *l, *u))); super().look_at(e, l, u)
        def attribute_begin(self): self._emit("AttributeBegin"); super().attribute_begin()
        def attribute_end(self): self._emit("AttributeEnd"); super().attribute_end()
        def transform_begin(self): self._emit("TransformBegin"); super().transform_begin()
        def transform_end(self): self._emit("TransformEnd"); super().transform_end()
        def toggle_reverse_orientation(self): self._emit("ReverseOrientation"); self.reverse_orientation = not self.reverse_orientation

        def make_named_medium(self, name, **kw):
            if kw.get("density") is not None:   # "heterogeneous": nx ny nz, the density in x-fastest order, the box corners
                d = np.asarray(kw["density"], dtype=np.float32); rest = {k: v for k, v in kw.items() if k not in ("density", "p0", "p1")}
                self._emit('MakeNamedMedium "%s" "string type" "heterogeneous" %s "integer nx" %d "integer ny" %d "integer nz" %d "point p0" %s "point p1" %s "float density" %s'
                           % (name, self._params(rest, spectrum=("sigma_a", "sigma_s")), d.shape[2], d.shape[1], d.shape[0], _vec(kw.get("p0", (0, 0, 0))), _vec(kw.get("p1", (1, 1, 1))), _vec(d.ravel())))
                return super().make_named_medium(name, **kw)
            self._emit('MakeNamedMedium "%s" "string type" "homogeneous" %s' % (name, self._params(kw, spectrum=("sigma_a", "sigma_s"))))
            super().make_named_medium(name, **kw)
        def medium_interface(self, inside="", outside=""): self._emit('MediumInterface "%s" "%s"' % (inside, outside)); super().medium_interface(inside, outside)

        def camera(self, **kw):
            self._emit('Camera "perspective" ' + self._params(kw)); super().camera(**kw)

        def world_begin(self):
            f = self.film
            self._emit('Film "image" "integer xresolution" %d "integer yresolution" %d "float cropwindow" %s "float scale" %s "string filename" "out.pfm"' % (f["xres"], f["yres"], _vec(f["crop"]), _f(f["scale"])))
            self._emit('Sampler "%s" "integer pixelsamples" %d %s' % (self.sampler, self.spp, '"bool samplepixelcenter" "true"' if self.sample_at_pixel_center and self.sampler == "halton" else ""))
            fl = self.filter
            extra = ' "float alpha" %s' % _f(fl["alpha"]) if fl["kind"] == "gaussian" else ""
            self._emit('PixelFilter "%s" "float xwidth" %s "float ywidth" %s%s' % (fl["kind"], _f(fl["radius"][0]), _f(fl["radius"][1]), extra))
            it = self.integ
            self._emit('Integrator "%s" "integer maxdepth" %d "float rrthreshold" %s "string lightsamplestrategy" "%s"' % (it.get("kind", "path"), it["maxdepth"], _f(it["rrthreshold"]), it["strategy"]))
            self._emit('Accelerator "bvh" "string splitmethod" "%s" "integer maxnodeprims" %d' % (self.split_method, self.max_node_prims))
            self._emit("WorldBegin"); super().world_begin()

        def material(self, kind, **kw):
            super().material(kind, **kw)
            if kind in ("none", ""): self._emit('Material "none"'); return
            name = "m%d" % self.material_id
            if len(self.materials) == 1 and not self.lines: return   # the constructor's default matte
            kw2 = dict(kw)
            if kind == "mix":
                kw2["namedmaterial1"] = "m%d" % kw["namedmaterial1"]; kw2["namedmaterial2"] = "m%d" % kw["namedmaterial2"]
            spec = ("Kd", "Ks", "Kr", "Kt", "opacity", "reflect", "transmit", "amount", "color", "sigma_a", "sigma_s", "mfp", "scatterdistance", "eta_rgb", "k")
            flt = ("sigma", "roughness", "uroughness", "vroughness", "eta", "bumpmap")
            text = self._params({k: v for k, v in kw2.items()}, spectrum=spec, textures_spec=spec, textures_float=flt)
            self._emit('MakeNamedMaterial "%s" "string type" "%s" %s' % (name, kind, text)); self._emit('NamedMaterial "%s"' % name)

        def texture(self, name, kind, cls, **kw):
            kw2 = dict(kw)
            if "pixels" in kw2: kw2["filename"] = self._pfm(kw2.pop("pixels"))
            children = ("tex1", "tex2", "inside", "outside", "amount")
            spec = () if kind == "float" else ("tex1", "tex2", "inside", "outside", "v00", "v01", "v10", "v11")
            parts = []
            for k, v in kw2.items():
                if k in children and isinstance(v, str): parts.append('"texture %s" "%s"' % (k, v))
                elif k == "amount" and not isinstance(v, str): parts.append('"float amount" %s' % _f(v))
                else: parts.append(self._params({k: v}, spectrum=spec))
            self._emit('Texture "%s" "%s" "%s" %s' % (name, kind, cls, " ".join(parts)))
            super().texture(name, kind, cls, **kw)

        def area_light_source(self, L=(1, 1, 1), twosided=False):
            self._emit('AreaLightSource "diffuse" "rgb L" %s "bool twosided" "%s"' % (_vec(L), "true" if twosided else "false")); super().area_light_source(L=L, twosided=twosided)

        def light_source(self, kind, **kw):
            kw2 = {("from" if k == "from_" else k): v for k, v in kw.items()}
            parts = []
            for k, v in kw2.items():
                if k == "texels": parts.append('"string mapname" "%s"' % self._pfm(v))
                elif k in ("from", "to"): parts.append('"point %s" %s' % (k, _vec(v)))
                elif k in ("L", "I"): parts.append('"rgb %s" %s' % (k, _vec(v)))
                else: parts.append('"float %s" %s' % (k, _f(v)))
            self._emit('LightSource "%s" %s' % (kind, " ".join(parts))); super().light_source(kind, **kw)

        def trianglemesh(self, P, indices, N=None, UV=None, S=None, alpha=None, shadowalpha=None):
            parts = ['"integer indices" [' + " ".join(str(int(i)) for i in np.asarray(indices).reshape(-1)) + "]", '"point P" ' + _vec(P)]
            if N is not None: parts.append('"normal N" ' + _vec(N))
            if UV is not None: parts.append('"float uv" ' + _vec(UV))
            if S is not None: parts.append('"vector S" ' + _vec(S))
            for nm, v in (("alpha", alpha), ("shadowalpha", shadowalpha)):
                if isinstance(v, str): parts.append('"texture %s" "%s"' % (nm, v))
                elif v is not None: parts.append('"float %s" %s' % (nm, _f(v)))
            self._emit('Shape "trianglemesh" ' + " ".join(parts)); super().trianglemesh(P, indices, N=N, UV=UV, S=S, alpha=alpha, shadowalpha=shadowalpha)

        def sphere(self, **kw): self._emit('Shape "sphere" ' + self._params(kw)); super().sphere(**kw)
        def disk(self, **kw): self._emit('Shape "disk" ' + self._params(kw)); super().disk(**kw)
        def object_begin(self, name): self._emit('ObjectBegin "%s"' % name); super().object_begin(name); self.lines.pop()   # the base class' AttributeBegin is implied
        def object_end(self): super().object_end(); self.lines.pop(); self._emit("ObjectEnd")
        def object_instance(self, name): self._emit('ObjectInstance "%s"' % name); super().object_instance(name)

        def text(self): return "\n".join(self.lines + ["WorldEnd", ""])

    return Recorder()
