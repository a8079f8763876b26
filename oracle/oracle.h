/*
 * oracle.h — C entry points of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's
 * per-pixel radiance loop (Nyrox/raymond, src/trace.rs + core/src/{scene,geometry}),
 * used as the parity checker and as the timed "port" CPU baseline.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under raymond_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED by reference TESTS: the reference has no tests, fixtures or
 * golden vectors for this path and cannot be compiled here (no Rust toolchain).
 * What the reference does hold pins the restatement as far as it reaches
 * (tests/test_oracle_golden.py, tests/test_ref_meshes.py):
 *   (a) its one render of the spheres scene, examples/ReflectiveSpheres.png (500 spp):
 *       the oracle's 500-spp frame differs from it by Monte-Carlo noise and nothing
 *       else — block means 0.44 of 255 apart where two oracle half-frames predict
 *       0.44, no region biased by more than 0.1 — which covers scene, camera, the
 *       BRDF and sampling code, and the output stage;
 *   (b) its mesh assets (assets/meshes, five PLY files): loader, bake_transform, bounds and
 *       the grid build agree bit for bit with a third, plain-Python implementation
 *       run over those files (tools/gen_ref_fixtures.py), including the reference's
 *       out-of-bounds panic on suzanne.ply;
 *   (c) the Random123 known-answer vectors for the RNG, and hand-derived known
 *       answers per primitive.
 * Nothing reference-held constrains the DDA walk's cell sequence, the Heron normals
 * or the thin lens beyond their agreement with the source text.
 *
 * The scene POD types are shared with the product header (inputs only).
 */
#ifndef RMD_ORACLE_H
#define RMD_ORACLE_H

#include "../include/raymond_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* RNG */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* out = { first 53-bit uniform, second 53-bit uniform, 22-bit uniform } of Philox block `block` of (pixel, sample): include/raymond_hip.h "RNG" */
void orc_block_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t block, double out[3]);

/* Batched device-function KATs.  Rays are 6 doubles (origin xyz, direction xyz).
 * hit[i] = 1/0; t[i] valid when hit. */
void orc_sphere_intersect(size_t n, const double *sphere4, const double *ray6, int32_t *hit, double *t);
void orc_sphere_normal(size_t n, const double *sphere4, const double *ray6, const double *t, double *n3);
void orc_plane_intersect(size_t n, const double *plane6, const double *ray6, int32_t *hit, double *t);
void orc_aabb_intersect(size_t n, const double *aabb6, const double *ray6, int32_t *hit, double *t);
void orc_triangle_intersect(size_t n, const double *pos9, const double *ray6, int32_t *hit, double *t);
void orc_triangle_normal(size_t n, const double *pos9, const double *nrm9, const double *ray6, const double *t,
                         double *n3);
void orc_onb(size_t n, const double *n3, double *t3, double *b3);
void orc_cosine_hemisphere(size_t n, const double *r1, const double *r2, double *dir3, double *pdf);
void orc_importance_sample_ggx(size_t n, const double *reflect3, const double *rough, const double *r1,
                               const double *r2, double *dir3);
void orc_ggx_distribution(size_t n, const double *n3, const double *h3, const double *rough, double *out);
void orc_geometry_smith(size_t n, const double *n3, const double *v3, const double *l3, const double *rough,
                        double *out);
void orc_fresnel_schlick(size_t n, const double *cos_theta, const double *f0_3, double *out3);
/* jitter uniforms given explicitly: u2[2*i], u2[2*i+1] */
void orc_primary_ray(size_t n, const rmd_camera *cam, const uint32_t *xy2, const double *u2, double *ray6);

/* grid build (acc_grid.rs:6-83).  Two-call protocol: sizes first, then fill. */
typedef struct orc_grid orc_grid;
int32_t orc_grid_build(const double *tri_pos, const double *tri_nrm, uint64_t n_tris, orc_grid **out); /* 0 ok, 5 = index panic */
void orc_grid_describe(const orc_grid *g, rmd_grid_desc *desc);
void orc_grid_destroy(orc_grid *g);

/* Mesh (core/src/geometry/mesh.rs): ASCII-PLY loader :58-121, bake_transform :48-56, find_mesh_bounds :123-140.
 * load returns 0, or 1 where the reference panics (missing file, malformed line, vertex index out of range). */
typedef struct orc_mesh orc_mesh;
int32_t orc_mesh_load_ply(const char *path, orc_mesh **out);
int32_t orc_mesh_load_ply_text(const char *text, size_t n_bytes, orc_mesh **out);
void orc_mesh_bake_transform(orc_mesh *m, const double translate[3]);
uint64_t orc_mesh_size(const orc_mesh *m);
void orc_mesh_arrays(const orc_mesh *m, const double **tri_pos, const double **tri_nrm); /* n*9 doubles each, owned by m */
void orc_mesh_bounds(const orc_mesh *m, double bbox_min[3], double bbox_max[3]);
void orc_mesh_destroy(orc_mesh *m);

/* Output stage: await's division by the sample count (src/trace.rs:95), then tone-map / gamma / u8 cast (cli_old/src/main.rs:161-181) */
void orc_resolve_tonemap(const double *accum, size_t n_pixels, double sample_count, double exposure, double gamma, uint8_t *rgb8);

/* scene */
orc_scene *orc_scene_create(const rmd_object *objects, uint32_t n_objects, const rmd_grid_desc *grids,
                            uint32_t n_grids);
void orc_scene_destroy(orc_scene *s);
/* Scene::intersect: obj[i] = object index or -1; sub[i] = triangle index for grids */
void orc_scene_intersect(const orc_scene *s, size_t n, const double *ray6, int32_t *obj, double *t, uint32_t *sub);
/* AccGrid::intersects on grid g of the scene */
void orc_grid_intersect(const orc_scene *s, uint32_t g, size_t n, const double *ray6, int32_t *hit, double *t,
                        uint32_t *tri);

/* One sample (generate_primary_ray[_with_dof] + trace(…,1)).  path_obj/path_sub: up to
 * RMD_MAX_BOUNCE_LIMIT+1 entries of (object index or -1, triangle index) per depth reached; returns depth count. */
int32_t orc_trace_sample(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, uint32_t x, uint32_t y,
                         uint32_t sample, double rgb[3], int32_t *path_obj, uint32_t *path_sub);
/* Batched: rgb_out[3*i] for (xy[2i], xy[2i+1], sample[i]). */
void orc_trace_samples(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, size_t n,
                       const uint32_t *xy2, const uint32_t *sample, double *rgb_out);

/* The tile worker pool (render_tiled, src/trace.rs:137-230): `n_threads` workers, shared FIFO
 * queue, one sample pass per pop, accum[(x+y*W)*3+c] += sample.  n_threads = 0 -> hardware_concurrency. */
void orc_render_tiles(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, const rmd_tile_rect *tiles,
                      uint32_t n_tiles, double *accum, uint32_t n_threads);

/* Work counters summed over all threads since the last reset (feed the algorithmic-bytes figure):
 * [0] samples, [1] path segments (Scene::intersect calls), [2] grid cells visited, [3] triangle tests,
 * [4] mesh hits shaded (Triangle::get_surface_properties calls), [5] shaded bounces, [6] rng draws, [7] grid walks */
/* Mutation switch of tools/mutation_pins.py (oracle.cpp: MUT_*): 0 = the faithful restatement.  No test, smoke() or bench.py sets it. */
int32_t orc_set_mutation(int32_t k);
/* Whether the work counters include the path segments behind a bounce weight of exactly zero (0 = no, the default: the work the product's
 * default does; 1 = yes: everything the reference executes).  Results never depend on it. */
void orc_count_black_paths(int32_t on);
void orc_counters_reset(void);
/* per-walk histograms, buckets 0..63 and 64+ : cells visited, non-empty cells visited, triangle tests, max triangles per cell; [260] = walks that hit */
void orc_walk_hist(uint64_t out[4 * 65 + 1]);
void orc_counters_get(uint64_t out[12]); /* samples, segments, cells, tri_tests, mesh_hits, bounces, draws, walks, occupied_cells, zero_weight_diffuse, zero_weight_specular, reserved */

#ifdef __cplusplus
}
#endif
#endif
