// tu_shade.hip -- the surface shade kernels of one MODE (0 triangle-only, 1 general geometry, 2 + textures, 3 volpath) and one
// lobe budget; compiled once per (PT_TU_MODE, PT_TU_MAXL) pair so the sixteen instantiations build in parallel.
#include "kern_shade.h"
#if !defined(PT_TU_MODE) || !defined(PT_TU_MAXL)
#error "compile with -DPT_TU_MODE=0..3 -DPT_TU_MAXL=1|2|5"
#endif
#define PT_INST_SHADE(L, S, D) template __global__ void k_shade<L, S, D>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, ShadeJob);
#if PT_TU_MAXL == 1
PT_INST_SHADE(1, PT_TU_MODE, 1) PT_INST_SHADE(1, PT_TU_MODE, 0)
#if PT_TU_MODE != 3   // (the volumetric integrator has no specular-only class: it estimates direct light at every vertex)
PT_INST_SHADE(1, PT_TU_MODE, 2)
#endif
#if PT_TU_MODE < 3     // the metal class (round 6: textured scenes too -- a texture changes a metal's parameters, not its lobe set)
PT_INST_SHADE(1, PT_TU_MODE, 3)
#endif
#if PT_TU_MODE < 2     // the smooth-subsurface class: untextured scenes only (its kernel takes sigma_a / sigma_s from the material)
PT_INST_SHADE(1, PT_TU_MODE, 6)
#endif
#else
PT_INST_SHADE(PT_TU_MAXL, PT_TU_MODE, 0)
#if PT_TU_MAXL == 5 && PT_TU_MODE < 3   // the uber class
PT_INST_SHADE(5, PT_TU_MODE, 5)
#endif
#if PT_TU_MAXL == 2 && PT_TU_MODE < 3   // the plastic-like class (plastic, opaque uber without specular terms)
PT_INST_SHADE(2, PT_TU_MODE, 4)
#endif
#endif
