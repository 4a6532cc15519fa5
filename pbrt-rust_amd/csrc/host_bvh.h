// host_bvh.h -- host SAH BVH builder interface (see host_bvh.cpp).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/mi355pt.h"

namespace pth {
struct PrimBound { float lo[3], hi[3]; };  // Primitive::world_bound()
void build_sah_bvh(const std::vector<PrimBound> &prims, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered);
// HLBVH on the GPU (gpu_bvh.hip): same outputs. Returns 0, or non-zero with *err pointing at a static message.
int build_hlbvh_gpu(const std::vector<PrimBound> &prims, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered, const char **err);
}
