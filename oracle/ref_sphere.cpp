// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_sphere.cpp: Sphere (shapes/sphere.rs) -- NOT YET RESTATED; scenes with spheres are rejected by the
// callers (orc_scene_create accepts them but every sphere test misses). Row a14 of SURVEY section 8.
#include "ref_scene.h"
namespace ref {
Bounds3 Scene::sphere_world_bound(uint32_t) const { return Bounds3(); }
bool Scene::sphere_intersect(uint32_t, const Ray &, Float &, SurfaceInteraction &, bool) const { return false; }
bool Scene::sphere_intersect_p(uint32_t, const Ray &) const { return false; }
}
