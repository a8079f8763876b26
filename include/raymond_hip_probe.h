/*
 * raymond_hip_probe.h — diagnostic entry points, in libraymond_hip_probe.so (a separate library that links against
 * libraymond_hip.so and takes its rmd_context / rmd_scene handles; the product library exports none of these).
 *
 * Each probe runs ONE device function of the hot path on the GPU over a batch of host-supplied
 * inputs, so the parity tests can compare it with the CPU oracle function by function
 * (known-answer level of the test pyramid).  They are not part of the drop-in boundary; a host
 * renderer never calls them.  All pointers are HOST pointers; the library stages them through HBM.
 * Argument layouts match the oracle's batched functions (oracle/oracle.h): rays are 6 doubles
 * (origin xyz, direction xyz); hit[i] is 1/0 and t[i] is meaningful only where hit[i] = 1.
 * Each cites the reference function whose device implementation it exercises.
 */
#ifndef RAYMOND_HIP_PROBE_H
#define RAYMOND_HIP_PROBE_H

#include "raymond_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RNG (replaces rand::random::<f64>(), src/trace.rs:260 etc.) */
rmd_status rmd_probe_philox4x32_10(rmd_context *ctx, size_t n, const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4);
/* per entry 5 doubles: the two 53-bit uniforms and the 22-bit uniform of Philox block `block` of (pixel, sample) as the jitter /
 * lens draws take them (next2), then (r, r1, r2) as a shaded depth takes them (next3) — include/raymond_hip.h "RNG" */
rmd_status rmd_probe_block_uniforms(rmd_context *ctx, uint64_t seed, size_t n, const uint32_t *pixel, const uint32_t *sample,
                                    const uint32_t *block, double *out5);
/* core/src/geometry/primitives/sphere.rs:11-27, :31-35 */
rmd_status rmd_probe_sphere_intersect(rmd_context *ctx, size_t n, const double *sphere4, const double *ray6, int32_t *hit, double *t);
rmd_status rmd_probe_sphere_normal(rmd_context *ctx, size_t n, const double *sphere4, const double *ray6, const double *t, double *n3);
/* core/src/geometry/primitives/plane.rs:11-24 */
rmd_status rmd_probe_plane_intersect(rmd_context *ctx, size_t n, const double *plane6, const double *ray6, int32_t *hit, double *t);
/* core/src/geometry/primitives/aabb.rs:10-31 */
rmd_status rmd_probe_aabb_intersect(rmd_context *ctx, size_t n, const double *aabb6, const double *ray6, int32_t *hit, double *t);
/* core/src/geometry/primitives/triangle.rs:11-44, :47-68 */
rmd_status rmd_probe_triangle_intersect(rmd_context *ctx, size_t n, const double *pos9, const double *ray6, int32_t *hit, double *t);
rmd_status rmd_probe_triangle_normal(rmd_context *ctx, size_t n, const double *pos9, const double *nrm9, const double *ray6,
                                     const double *t, double *n3);
/* src/trace.rs:408-416, :396-406, :286-296, :362-370, :380-382, :384-386 */
rmd_status rmd_probe_onb(rmd_context *ctx, size_t n, const double *n3, double *t3, double *b3);
rmd_status rmd_probe_cosine_hemisphere(rmd_context *ctx, size_t n, const double *r1, const double *r2, double *dir3, double *pdf);
rmd_status rmd_probe_importance_sample_ggx(rmd_context *ctx, size_t n, const double *reflect3, const double *rough, const double *r1,
                                           const double *r2, double *dir3);
rmd_status rmd_probe_ggx_distribution(rmd_context *ctx, size_t n, const double *n3, const double *h3, const double *rough, double *out);
rmd_status rmd_probe_geometry_smith(rmd_context *ctx, size_t n, const double *n3, const double *v3, const double *l3,
                                    const double *rough, double *out);
rmd_status rmd_probe_fresnel_schlick(rmd_context *ctx, size_t n, const double *cos_theta, const double *f0_3, double *out3);
/* The device's elementary functions as the kernels use them: IEEE sqrt (bit-exact) and the reduced-range sin / cos that
 * stand in for the reference's libm calls (src/trace.rs:291-293, :401-403; <= 2 ulp for |x| < 2^45). */
rmd_status rmd_probe_elementary(rmd_context *ctx, size_t n, const double *x, double *sqrt_out, double *sin_out, double *cos_out,
                                double *root_out, double *inv_root_out); /* root/inv_root: normalize()'s sqrt and 1.0 / sqrt, both bit-exact */
/* src/trace.rs:322-333 with the two jitter uniforms given explicitly (u2[2i], u2[2i+1]) */
rmd_status rmd_probe_primary_ray(rmd_context *ctx, size_t n, const rmd_camera *cam, const uint32_t *xy2, const double *u2, double *ray6);
/* core/src/scene.rs:54-74: obj[i] = object index or -1, sub[i] = triangle index for grid objects */
rmd_status rmd_probe_scene_intersect(rmd_context *ctx, const rmd_scene *scene, size_t n, const double *ray6, int32_t *obj, double *t,
                                     uint32_t *sub);
/* core/src/geometry/acc_grid.rs:89-185 on grid `g` of the scene */
rmd_status rmd_probe_grid_intersect(rmd_context *ctx, const rmd_scene *scene, uint32_t g, size_t n, const double *ray6, int32_t *hit,
                                    double *t, uint32_t *tri);
/* One sample per entry (src/trace.rs:199-200) through the render kernel's own code path:
 * rgb_out[3i..] = radiance of (xy2[2i], xy2[2i+1], sample[i]).  path_obj/path_sub (optional, n*17 each):
 * object index (-1 = miss) and triangle index per trace() depth, -2 beyond the path's end. */
rmd_status rmd_probe_trace_samples(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *cam, const rmd_settings *settings,
                                   size_t n, const uint32_t *xy2, const uint32_t *sample, double *rgb_out, int32_t *path_obj,
                                   uint32_t *path_sub);

/* Host only (no device needed): the sphere rmd_scene_create puts around a triangle for the grid walk's pre-test — out5 = centre (3), inflated
 * squared radius r2a, the triangle's distance-proportional allowance kb (a grid uses the largest of its triangles').  A (ray, triangle) pair whose
 * line passes the centre at more than sqrt(r2a + kb * |centre - origin|^2) is not run through triangle.rs:11-44; tests/test_pretest_allowance.py
 * checks, with the reference's test evaluated in binary64 on adversarial pairs, that no such pair would have passed it. */
rmd_status rmd_probe_triangle_sphere(size_t n, const double *pos9, double *out5);
/* The walk's pre-test on explicit (triangle, ray) pairs IN THE DEVICE'S OWN ARITHMETIC — the function the chunk loop calls (grid_walk.hpp:
 * sphere_pretest, fused multiply-adds included) — beside the device's triangle.rs:11-44 on the same pair.  sphere5 = centre (3), r2a, kb (the caller
 * chooses the allowance: the triangle's own, or a grid's largest); pass[i] = the pre-test lets the pair through, hit[i] / t[i] = the reference's test.
 * A pair with hit = 1 and pass = 0 would be a silently missed hit: tests/test_gpu_reference_pins.py asserts there is none among the adversarial
 * pairs of tests/test_pretest_allowance.py. */
rmd_status rmd_probe_pretest_pairs(rmd_context *ctx, size_t n, const double *sphere5, const double *pos9, const double *ray6, int32_t *pass,
                                   int32_t *hit, double *t);

#ifdef __cplusplus
}
#endif
#endif
