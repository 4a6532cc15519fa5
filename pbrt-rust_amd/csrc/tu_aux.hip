// tu_aux.hip -- escaped rays / dead paths (k_shade_miss) and the volumetric integrator's kernels (k_medium_route, k_shade_medium).
#include "kern_aux.h"
template __global__ void k_shade_miss<false, false>(DeviceScene, RenderConst, PathSoA, ShadeJob);
template __global__ void k_shade_miss<true, false>(DeviceScene, RenderConst, PathSoA, ShadeJob);
template __global__ void k_shade_miss<true, true>(DeviceScene, RenderConst, PathSoA, ShadeJob);
template __global__ void k_film_final<false>(DeviceScene, RenderConst, PathSoA, const float *, float *, DevCounters *);
template __global__ void k_film_final<true>(DeviceScene, RenderConst, PathSoA, const float *, float *, DevCounters *);
