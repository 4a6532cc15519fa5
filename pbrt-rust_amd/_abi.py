"""ctypes mirror of include/mi355pt.h (the C-ABI drop-in boundary).

Only plain structs, enums and argtypes live here; no compute.  The same structs are
accepted by the product library (csrc -> libmi355pt.so) and, in tests only, by the CPU oracle.
"""
import ctypes as C

PT_OK = 0
PT_ERR_INVALID_ARG, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_UNSUPPORTED = 1, 2, 3, 4
PT_ERR_SOBOL_DIMENSIONS, PT_ERR_STACK_OVERFLOW, PT_ERR_OUT_OF_MEMORY, PT_ERR_PROBE_CHAIN = 5, 6, 7, 8

PT_TRI_REVERSE_ORIENTATION = 1
PT_TRI_SWAPS_HANDEDNESS = 2
PT_TRI_HAS_N = 4
PT_TRI_HAS_S = 8
PT_TRI_HAS_UV = 16

PT_SHAPE_TRIANGLE, PT_SHAPE_SPHERE = 0, 1
PT_NONE = 0xFFFFFFFF

PT_MAT_MATTE, PT_MAT_MIRROR, PT_MAT_GLASS, PT_MAT_PLASTIC, PT_MAT_METAL, PT_MAT_UBER, PT_MAT_SUBSTRATE, PT_MAT_SUBSURFACE, PT_MAT_TRANSLUCENT, PT_MAT_MIX, PT_MAT_DISNEY = range(11)
(PT_DS_METALLIC, PT_DS_SPECULARTINT, PT_DS_ANISOTROPIC, PT_DS_SHEEN, PT_DS_SHEENTINT, PT_DS_CLEARCOAT, PT_DS_CLEARCOATGLOSS, PT_DS_SPECTRANS,
 PT_DS_FLATNESS, PT_DS_DIFFTRANS) = range(10)
PT_LIGHT_DIFFUSE_AREA, PT_LIGHT_DISTANT, PT_LIGHT_POINT, PT_LIGHT_INFINITE, PT_LIGHT_SPOT = range(5)
PT_LS_UNIFORM, PT_LS_POWER, PT_LS_SPATIAL, PT_LS_SPATIAL_EAGER, PT_LS_SPATIAL_LAZY = range(5)
PT_SPLIT_SAH, PT_SPLIT_HLBVH = range(2)
PT_SAMPLER_SOBOL, PT_SAMPLER_HALTON = range(2)
PT_INTEGRATOR_PATH, PT_INTEGRATOR_VOLPATH = range(2)
PT_QUADRIC_SPHERE, PT_QUADRIC_DISK = range(2)


def shape_ref(kind, index):
    return (kind << 30) | index


f32, u32, i32, u8, u16, u64 = C.c_float, C.c_uint32, C.c_int32, C.c_uint8, C.c_uint16, C.c_uint64
fp, u32p, u8p, i32p, u64p = C.POINTER(f32), C.POINTER(u32), C.POINTER(u8), C.POINTER(i32), C.POINTER(u64)


class PtSphere(C.Structure):
    _fields_ = [("object_to_world", f32 * 16), ("world_to_object", f32 * 16),
                ("radius", f32), ("z_min", f32), ("z_max", f32), ("theta_min", f32), ("theta_max", f32),
                ("phi_max", f32), ("reverse_orientation", u32), ("transform_swaps_handedness", u32), ("kind", u32), ("inner_radius", f32)]


class PtMaterial(C.Structure):
    _fields_ = [("type", u32), ("kd", f32 * 3), ("ks", f32 * 3), ("kr", f32 * 3), ("kt", f32 * 3),
                ("opacity", f32 * 3), ("eta_rgb", f32 * 3), ("k_rgb", f32 * 3), ("sigma", f32), ("eta", f32),
                ("roughness", f32), ("u_roughness", f32), ("v_roughness", f32), ("remap_roughness", u32),
                ("sigma_a", f32 * 3), ("sigma_s", f32 * 3), ("scale", f32), ("bssrdf_table", u32), ("tex", C.c_int32 * 16), ("mix", u32 * 2), ("disney", f32 * 10), ("disney_thin", u32), ("disney_scatter", f32 * 3), ("mfp", f32 * 3), ("kd_subsurface", u32)]


PT_MEDIUM_HOMOGENEOUS, PT_MEDIUM_GRID = 0, 1


class PtMedium(C.Structure):
    _fields_ = [("sigma_a", f32 * 3), ("sigma_s", f32 * 3), ("g", f32), ("type", u32), ("nx", u32), ("ny", u32), ("nz", u32),
                ("world_to_medium", f32 * 16), ("density", fp)]


class PtLight(C.Structure):
    _fields_ = [("type", u32), ("L", f32 * 3), ("two_sided", u32), ("prim", u32), ("pos", f32 * 3),
                ("dir", f32 * 3), ("cos_total_width", f32), ("cos_falloff_start", f32),
                ("light_to_world", f32 * 16), ("world_to_light", f32 * 16)]


class PtBVHNode(C.Structure):
    _fields_ = [("bmin", f32 * 3), ("bmax", f32 * 3), ("offset", u32), ("n_prims", u16), ("axis", u8), ("pad", u8)]


class PtObject(C.Structure):
    _fields_ = [("first_prim", u32), ("n_prims", u32)]


class PtInstance(C.Structure):
    _fields_ = [("object", u32), ("instance_to_world", f32 * 16), ("world_to_instance", f32 * 16)]


PT_TOP_INSTANCE = 0x80000000


(PT_TEX_CONSTANT, PT_TEX_SCALE, PT_TEX_MIX, PT_TEX_CHECKERBOARD2D, PT_TEX_CHECKERBOARD3D, PT_TEX_IMAGEMAP, PT_TEX_UV, PT_TEX_BILERP,
 PT_TEX_FBM, PT_TEX_WRINKLED, PT_TEX_WINDY, PT_TEX_MARBLE, PT_TEX_DOTS) = range(13)
PT_MAP_UV, PT_MAP_PLANAR, PT_MAP_SPHERICAL, PT_MAP_CYLINDRICAL = range(4)
PT_WRAP_REPEAT, PT_WRAP_BLACK = range(2)
(PT_MP_KD, PT_MP_KS, PT_MP_KR, PT_MP_KT, PT_MP_OPACITY, PT_MP_ETA_RGB, PT_MP_K_RGB, PT_MP_SIGMA_A, PT_MP_SIGMA_S,
 PT_MP_SIGMA, PT_MP_ROUGHNESS, PT_MP_U_ROUGHNESS, PT_MP_V_ROUGHNESS, PT_MP_ETA, PT_MP_BUMP, PT_MP_MFP) = range(16)
PT_MP_COUNT = 16
PT_PEER_SAME_DEVICE, PT_PEER_ENABLED, PT_PEER_STAGED = range(3)


class PtTexture(C.Structure):
    _fields_ = [("type", u32), ("child", C.c_int32 * 3), ("value", f32 * 3), ("v00", f32 * 3), ("v01", f32 * 3), ("v10", f32 * 3), ("v11", f32 * 3),
                ("mapping", u32), ("su", f32), ("sv", f32), ("du", f32), ("dv", f32), ("vs", f32 * 3), ("vt", f32 * 3),
                ("world_to_texture", f32 * 16), ("aa_closedform", u32), ("image", u32), ("trilinear", u32), ("max_anisotropy", f32), ("wrap", u32),
                ("octaves", u32), ("omega", f32), ("marble_scale", f32), ("variation", f32)]


class PtImage(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("n_levels", u32), ("channels", u32), ("texels", fp)]


class PtBSSRDFTable(C.Structure):
    _fields_ = [("n_rho", u32), ("n_radius", u32), ("rho_samples", fp), ("radius_samples", fp), ("profile", fp),
                ("rhoeff", fp), ("profile_cdf", fp)]


class PtSceneDesc(C.Structure):
    _fields_ = [("n_vertices", u32), ("P", fp), ("N", fp), ("S", fp), ("UV", fp),
                ("n_triangles", u32), ("indices", u32p), ("tri_flags", u8p),
                ("n_spheres", u32), ("spheres", C.POINTER(PtSphere)),
                ("n_prims", u32), ("prim_shape", u32p), ("prim_material", u32p), ("prim_light", u32p),
                ("n_materials", u32), ("materials", C.POINTER(PtMaterial)),
                ("n_lights", u32), ("lights", C.POINTER(PtLight)),
                ("env_width", u32), ("env_height", u32), ("env_texels", fp), ("env_importance", fp), ("env_power_lookup", f32 * 3),
                ("max_node_prims", u32), ("n_nodes", u32), ("nodes", C.POINTER(PtBVHNode)), ("ordered_prims", u32p),
                ("n_objects", u32), ("objects", C.POINTER(PtObject)), ("n_instances", u32), ("instances", C.POINTER(PtInstance)),
                ("n_top", u32), ("top_refs", u32p), ("n_bssrdf_tables", u32), ("bssrdf_tables", C.POINTER(PtBSSRDFTable)),
                ("n_textures", u32), ("textures", C.POINTER(PtTexture)), ("tri_alpha", i32p), ("tri_shadow_alpha", i32p), ("n_images", u32), ("images", C.POINTER(PtImage)), ("ewa_weight_lut", fp), ("n_media", u32), ("media", C.POINTER(PtMedium)), ("prim_medium_inside", u32p), ("prim_medium_outside", u32p), ("split_method", u32)]


class PtRenderParams(C.Structure):
    _fields_ = [("full_resolution", i32 * 2), ("cropped_pixel_bounds", i32 * 4), ("filter_radius", f32 * 2),
                ("filter_table", f32 * 256), ("max_sample_luminance", f32), ("scale", f32),
                ("spp", u32), ("sample_bounds", i32 * 4),
                ("raster_to_camera", f32 * 16), ("camera_to_world", f32 * 16), ("lens_radius", f32),
                ("focal_distance", f32), ("shutter_open", f32), ("shutter_close", f32),
                ("max_depth", u32), ("rr_threshold", f32), ("pixel_bounds", i32 * 4), ("light_strategy", u32),
                ("tile_rank", u32), ("tile_world", u32), ("spp_per_pass", u32), ("profile", u32), ("sampler_type", u32), ("sample_at_pixel_center", u32), ("integrator", u32), ("camera_medium", u32)]


class PtCounters(C.Structure):
    _fields_ = [("camera_rays", u64), ("intersect_tests", u64), ("shadow_tests", u64), ("bvh_nodes_visited", u64),
                ("triangle_tests", u64), ("sphere_tests", u64), ("zero_radiance_paths_num", u64),
                ("zero_radiance_paths_den", u64), ("path_length_hist", u64 * 16), ("sanitized_nan", u64),
                ("sanitized_negative", u64), ("sanitized_infinite", u64), ("film_splats", u64),
                ("wavefront_stages", u64), ("reference_asserts", u64)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


class PtKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", u64), ("total_ms", C.c_double), ("items", u64), ("bvh_nodes", u64), ("triangle_tests", u64), ("kernel", C.c_char * 48)]


# Every symbol include/mi355pt.h declares, with its signature (restype, argtypes).
VP = C.c_void_p
ENTRY_POINTS = {
    "pt_init": (C.c_int, [C.c_int]),
    "pt_last_error": (C.c_char_p, []),
    "pt_scene_create": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(VP)]),
    "pt_scene_destroy": (None, [VP]),
    "pt_scene_bvh_info": (C.c_int, [VP, u32p, u32p]),
    "pt_scene_bvh_read": (C.c_int, [VP, C.POINTER(PtBVHNode), u32p]),
    "pt_render": (C.c_int, [VP, C.POINTER(PtRenderParams), VP, C.c_int]),
    "pt_pass_size": (C.c_int, [VP, C.POINTER(PtRenderParams), u32p]),
    "pt_film_resolve": (C.c_int, [fp, u32, f32, fp]),
    "pt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pt_set_trace_exact": (C.c_int, [C.c_int]),
    "pt_multi_scene_create": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(C.c_int), u32, C.POINTER(VP)]),
    "pt_multi_scene_destroy": (None, [VP]),
    "pt_multi_render": (C.c_int, [VP, C.POINTER(PtRenderParams), VP, C.c_int]),
    "pt_multi_get_counters": (C.c_int, [VP, C.POINTER(PtCounters)]),
    "pt_multi_get_kernel_stats": (C.c_int, [VP, u32, C.POINTER(PtKernelStat), u32, u32p]),
    "pt_multi_get_timing": (C.c_int, [VP, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), u32]),
    "pt_multi_get_peer_access": (C.c_int, [VP, C.POINTER(C.c_int), u32]),
    "pt_multi_get_create_timing": (C.c_int, [VP, C.POINTER(C.c_double), C.POINTER(C.c_double), u32]),
    "pt_multi_tile_shard": (None, [u32, u32, u32, u32, u32p, u32p]),
    "pt_get_counters": (C.c_int, [VP, C.POINTER(PtCounters)]),
    "pt_get_kernel_stats": (C.c_int, [VP, C.POINTER(PtKernelStat), u32, u32p]),
    "pt_trace_closest": (C.c_int, [VP, u32, fp, fp, fp, u32p, fp, fp]),
    "pt_trace_any": (C.c_int, [VP, u32, fp, fp, fp, u8p]),
    "pt_sobol_samples": (C.c_int, [i32p, u32, i32p, u32p, u32, fp, u64p]),
    "pt_halton_samples": (C.c_int, [i32p, u32, u32, i32p, u32p, u32, fp, u64p]),
    "pt_camera_rays": (C.c_int, [C.POINTER(PtRenderParams), u32, fp, fp, fp]),
    "pt_dist1d_sample": (C.c_int, [fp, u32, C.c_int, u32, fp, fp, fp, i32p]),
}


def bind(lib, table=ENTRY_POINTS, prefix_from="pt_", prefix_to="pt_", strict=True):
    """Attach restype/argtypes; raises AttributeError if a declared symbol is missing (strict=False: an older library variant of an A/B run may lack the newest entry points)."""
    for name, (res, args) in table.items():
        fn = getattr(lib, name.replace(prefix_from, prefix_to, 1), None)
        if fn is None:
            if strict:
                raise AttributeError(f"{name}: symbol declared in include/mi355pt.h is missing from the library")
            continue
        fn.restype = res
        fn.argtypes = args
    return lib
