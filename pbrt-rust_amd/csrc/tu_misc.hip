// tu_misc.hip -- scene preparation, camera-ray generation, routing, film, light grid and the parity helper kernels.
#include "kern_misc.h"
template __global__ void k_light_touch<false>(DeviceScene, LightGrid, PathSoA, const uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *);
template __global__ void k_light_touch<true>(DeviceScene, LightGrid, PathSoA, const uint32_t *, const uint32_t *, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *);
template __global__ void k_route<6, 2048>(DeviceScene, const uint32_t *, const uint32_t *, PathSoA, uint32_t *, RouteJob);
template __global__ void k_route<12, 1024>(DeviceScene, const uint32_t *, const uint32_t *, PathSoA, uint32_t *, RouteJob);
