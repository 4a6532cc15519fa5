// tu_bssrdf.hip -- subsurface exit-point kernel (path.rs:177-204, bssrdf.rs:334-410,559-574).
#include "kern_bssrdf.h"
template __global__ void k_bssrdf<false, false>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, BssrdfJob);
template __global__ void k_bssrdf<true, false>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, BssrdfJob);
template __global__ void k_bssrdf<true, true>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, BssrdfJob);   // volpath (general geometry, like every volpath kernel)
