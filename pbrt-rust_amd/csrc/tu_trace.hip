// tu_trace.hip -- BVH traversal kernels; compiled once per PT_TU_ANY (closest-hit / any-hit) so the two halves build in parallel.
#include "kern_trace.h"
#ifndef PT_TU_ANY
#error "compile with -DPT_TU_ANY=0 (closest hit) or 1 (any hit)"
#endif
template __global__ void k_trace<PT_TU_ANY != 0, 0>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY != 0, 1>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY != 0, 2>(DeviceScene, TraceJob);
