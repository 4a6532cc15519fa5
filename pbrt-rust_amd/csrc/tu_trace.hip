// tu_trace.hip -- BVH traversal kernels; compiled once per (PT_TU_ANY, PT_TU_QUAD) so that the variants build in parallel.
#include "kern_trace.h"
#ifndef PT_TU_ANY
#error "compile with -DPT_TU_ANY=0 (closest hit), 1 (any hit), 2 (mixed: the three ray kinds of one wavefront iteration) or 3 (probe chains)"
#endif
#ifndef PT_TU_QUAD
#error "compile with -DPT_TU_QUAD=1 (production: four-wide records), 2 (the same for a record / packet pool beyond 4 GB) or 0 (two-wide records, the reference's node-visit counter)"
#endif
#if PT_TU_ANY == 3   // BSSRDF probe chains (closest hit, chain walked inside the kernel)
template __global__ void k_trace<0, 0, true, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<0, 1, true, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<0, 2, true, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<0, 3, true, PT_TU_QUAD>(DeviceScene, TraceJob);
#else
template __global__ void k_trace<PT_TU_ANY, 0, false, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY, 1, false, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY, 2, false, PT_TU_QUAD>(DeviceScene, TraceJob);
template __global__ void k_trace<PT_TU_ANY, 3, false, PT_TU_QUAD>(DeviceScene, TraceJob);
#endif
