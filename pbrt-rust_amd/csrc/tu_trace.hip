// tu_trace.hip -- BVH traversal kernels; compiled once per PT_TU_ANY (closest-hit / any-hit) so the two halves build in parallel.
#include "kern_trace.h"
#ifndef PT_TU_ANY
#error "compile with -DPT_TU_ANY=0 (closest hit), 1 (any hit) or 2 (probe chains)"
#endif
#if PT_TU_ANY == 2   // BSSRDF probe chains (closest hit, chain walked inside the kernel)
template __global__ void k_trace<false, 0, true>(DeviceScene, TraceJob);
template __global__ void k_trace<false, 1, true>(DeviceScene, TraceJob);
template __global__ void k_trace<false, 2, true>(DeviceScene, TraceJob);
template __global__ void k_trace<false, 3, true>(DeviceScene, TraceJob);
#else
template __global__ void k_trace<PT_TU_ANY != 0, 0, false>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY != 0, 1, false>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY != 0, 2, false>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY != 0, 3, false>(DeviceScene, TraceJob);
#endif
