/*
 * raymond_hip.h — C-ABI of the MI355X-native radiance integrator.
 *
 * This library replaces ONE seam of Nyrox/raymond: the per-tile body of the
 * worker loop in `render_tiled`
 *
 *     reference  src/trace.rs:197-205   for y in tile rows { for x in tile cols {
 *                                           primary = generate_primary_ray(x, y, cam)   (:322-333 / :335-360)
 *                                           sample  = trace(primary, &context, 1)       (:232-320)
 *                                           tile.data[..] += sample } }
 *
 * Everything below that loop (Scene::intersect core/src/scene.rs:54-74, the
 * primitives core/src/geometry/primitives/{sphere,plane,triangle,aabb}.rs, the DDA grid walk
 * core/src/geometry/acc_grid.rs:89-185, the BRDF and samplers
 * src/trace.rs:362-416) runs inside one HIP kernel for gfx950.  Everything
 * above it (tile queue, progressive passes, TaskHandle, tone-map, file IO)
 * stays with the host.  There is no reference FFI for this path (the reference
 * has no `extern "C"` anywhere); the entry points below are what a cgo-style
 * Rust `extern "C"` block for that seam would bind — INTEGRATION.md shows the
 * Rust side.
 *
 * Conventions: plain C, POD structs, caller owns every buffer it passes, the
 * library owns the opaque handles.  Every function returns rmd_status
 * (0 = OK); nothing throws or aborts across the boundary — every entry point
 * that allocates on the host runs behind a catch: std::bad_alloc comes back as
 * RMD_ERR_OUT_OF_MEMORY (tests/test_abi.py forces it) — (the reference's
 * failure mode is a Rust panic — and a hang: `TaskHandle::await` polls a counter
 * that a panicked worker never decrements, src/trace.rs:82-92; here it is a
 * status + rmd_last_error()).  That holds inside the kernel too: every loop of
 * the render kernel has a bound that no input reaches, and a wave that runs
 * into one sets a device fault word, stops the launch handing out work and
 * leaves; the host then returns RMD_ERR_DEVICE_FAULT instead of a frame.
 * A context is thread-compatible (one calling thread at a time per handle).
 * All arithmetic is IEEE binary64, as in the reference (core/src/math.rs:10).
 */
#ifndef RAYMOND_HIP_H
#define RAYMOND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMD_ABI_VERSION 5u

typedef int32_t rmd_status;
enum {
	RMD_OK = 0,
	RMD_ERR_INVALID_ARGUMENT = 1, /* null pointer, zero size, out-of-range index            */
	RMD_ERR_NO_DEVICE = 2,        /* HIP runtime present but no usable gfx950 device         */
	RMD_ERR_HIP = 3,              /* a HIP call failed; text in rmd_last_error               */
	RMD_ERR_OUT_OF_MEMORY = 4,
	RMD_ERR_GRID_INDEX = 5,       /* grid build hit the reference's out-of-bounds panic (Q5) */
	RMD_ERR_UNSUPPORTED = 6,      /* e.g. bounce_limit above RMD_MAX_BOUNCE_LIMIT            */
	RMD_ERR_RCCL = 7,
	RMD_ERR_DEVICE_FAULT = 8      /* a loop of the render kernel ran past its bound (it cannot, for any input: an internal fault);
	                                 the launch was cut short, the frame it wrote to is NOT valid; text in rmd_last_error.
	                                 Reported by the first call that waits for the launch (rmd_render_tiles,
	                                 rmd_context_synchronize, rmd_last_kernel_ms, rmd_framebuffer_download[_tiles],
	                                 rmd_framebuffer_upload_tiles, rmd_context_wait_transfers, rmd_resolve_tonemap,
	                                 rmd_reduce_framebuffer) */
};

/* ---- scene description (mirrors core/src/scene.rs:8-45, core/src/lib.rs:21-26) ---- */

/* enum Geometry { Plane, Sphere, Grid }  — core/src/scene.rs:9-13 (same order) */
enum { RMD_GEOM_PLANE = 0, RMD_GEOM_SPHERE = 1, RMD_GEOM_GRID = 2 };
/* enum Material { Diffuse, Metal, Emission } — core/src/lib.rs:21-26 (same order) */
enum { RMD_MAT_DIFFUSE = 0, RMD_MAT_METAL = 1, RMD_MAT_EMISSION = 2 };

/* Material::Diffuse(color, roughness) | Metal(color, roughness) | Emission(e, v2, f1, f2).
 * For Emission `color` carries the first vector (the only field trace() reads,
 * src/trace.rs:250-252); v2/f1/f2 ride along untouched in `emission_aux`.
 * |roughness| must be <= 512 for Diffuse/Metal (rmd_scene_create returns RMD_ERR_UNSUPPORTED
 * otherwise): the GGX sampling angle roughness^2 * sqrt(u / (1 - u)) is reduced on the device
 * by a method that is accurate below 2^45. */
typedef struct rmd_material {
	uint32_t kind;
	uint32_t _pad;
	double color[3];
	double roughness;
	double emission_aux[5];
} rmd_material;

/* One scene::Object (core/src/scene.rs:33-37).  Object order is significant:
 * Scene::intersect keeps the FIRST object on distance ties (strict '<', :61). */
typedef struct rmd_object {
	uint32_t geometry_kind;
	uint32_t grid_index; /* RMD_GEOM_GRID: index into the grids array           */
	double origin[3];    /* Plane.origin (plane.rs:6) / Sphere.origin (sphere.rs:6) */
	double normal[3];    /* Plane.normal (plane.rs:7)                            */
	double radius;       /* Sphere.radius (sphere.rs:7)                          */
	rmd_material material;
} rmd_object;

/* One AccGrid (core/src/geometry/acc_grid.rs:27-33) in the compact layout:
 * `cells[c]` = offset into `mapping_table`; `mapping_table[off]` = count,
 * followed by `count` triangle indices (acc_grid.rs:67-74), both as u32
 * instead of usize.  Triangles are split SoA-by-role: positions (read by every
 * Möller-Trumbore test, triangle.rs:11-44) and vertex normals (read only on a
 * shaded hit, triangle.rs:47-68) instead of the 264-byte AoS Triangle.
 * All pointers are HOST pointers; rmd_scene_create copies them to HBM. */
typedef struct rmd_grid_desc {
	double bbox_min[3]; /* mesh.bounding_box.min (mesh.rs:123-140)      */
	double bbox_max[3];
	uint32_t resolution[3]; /* acc_grid.rs:6-17                          */
	uint32_t _pad;
	double cell_size[3]; /* acc_grid.rs:38                               */
	const uint32_t *cells;
	uint64_t n_cells; /* = res.x*res.y*res.z                            */
	const uint32_t *mapping_table;
	uint64_t n_mapping;
	const double *tri_pos; /* n_tris * 9: v0.xyz v1.xyz v2.xyz          */
	const double *tri_nrm; /* n_tris * 9: n0.xyz n1.xyz n2.xyz          */
	uint64_t n_tris;
	const struct rmd_grid_build *built; /* NULL, or the rmd_grid_build these pointers belong to (set by rmd_grid_build_describe): rmd_scene_create
	                                       then derives its device tables ONCE per build (about 50 ms of host time for 100k triangles) and
	                                       every later upload of the same grid — one per render_tiled call and GPU — reuses them.
	                                       The pointer is only as good as the build: a caller that COPIES a description, or hands its arrays
	                                       to another owner, and lets the copy outlive rmd_grid_build_destroy must set `built` to NULL in
	                                       the copy (the arrays then only have to stay valid for the duration of rmd_scene_create, as in
	                                       ABI 3).  A struct that was zero-initialised and filled by hand has NULL here.                  */
} rmd_grid_desc;

/* CameraSettings (src/trace.rs:32-40) + Transform (src/transform.rs:4-7, position only). */
typedef struct rmd_camera {
	uint32_t backbuffer_width;
	uint32_t backbuffer_height;
	double fov_vert; /* degrees */
	double position[3];
	double focal_length;
	double aperture_radius; /* CameraSettings.aperture_radius.  The reference's worker always calls the pinhole
	                           generate_primary_ray (:199, :322-333) whatever this field holds — its thin-lens
	                           generate_primary_ray_with_dof (:335-360) is never called — and so does this library
	                           unless rmd_settings.flags carries RMD_RENDER_DOF (and the radius is > 0: with
	                           radius 0 the reference's rejection loop never terminates). */
} rmd_camera;

#define RMD_MAX_BOUNCE_LIMIT 16u

/* The part of Settings (src/trace.rs:42-55) the per-tile body reads, plus the
 * RNG definition the reference lacks (it uses the unseedable thread_rng). */
#define RMD_RENDER_DOF 1u /* rmd_settings.flags: primary rays through generate_primary_ray_with_dof (an extension: the
                             reference defines that function but its render loop never calls it) */
/* Paths whose throughput has become exactly (0, 0, 0) — a diffuse bounce off a black surface ((1 - F)(1 - metal) (.) (0,0,0), src/trace.rs:279-281),
 * a GGX sample below the surface (geometry_smith's max(n.l, 0), :373).  trace() multiplies whatever the rest of such a path finds by that zero
 * (:281-282, :315-318), so its sample is exactly (0, 0, 0) in the reference too — unless a LATER vertex of the path produces a non-finite
 * radiance, because 0 x NaN = NaN.  The one source of such a vertex is the interpolated normal of a mesh hit (triangle.rs:47-68: Heron's
 * radicand rounding below zero for a hit on an edge, or vertex normals that sum to zero); a scene of planes and spheres has none.
 *
 *   flags = 0 (default)               REFERENCE-IDENTICAL on every scene.  Such paths are ended early only where that is PROVED not to
 *                                     change a sample: in scenes WITHOUT grid objects whose parameters are all REGULAR — every coordinate,
 *                                     colour and radiance finite and at most 1e150 in magnitude, no sphere of radius 0, no material of
 *                                     roughness 0 (rmd_scene_create decides this once per scene).  Everywhere else — a scene with a grid,
 *                                     or a grid-less scene outside that class: an Emission((inf, 0, 0)) behind a black bounce is
 *                                     0 x inf = NaN, roughness 0 makes geometry_schlick_ggx 0 / 0 (:372-378) — every path is traced to its
 *                                     end, the last depth included, as the reference does, so a sample that is NaN in the reference is NaN
 *                                     here (tests/test_gpu_parity.py::test_flags_0_is_reference_identical_for_non_finite_scene_parameters).
 *                                     What is NOT covered: events of probability ~2^-53 per path inside the regular class (r1 = 0 exactly
 *                                     in a diffuse pdf, a bounce ray exactly opposite to the view vector).
 *                                     This is what a drop-in caller gets (integration/gpu.rs).
 *   RMD_RENDER_END_BLACK_PATHS        opt-in, scenes with grids: end such paths there too.  Every sample that is finite in the reference
 *                                     keeps its value bit for bit; a sample the reference makes NaN behind a zero weight comes out (0, 0, 0)
 *                                     (how many pixels of the benchmark frames that is: DESIGN.md section 3, counted on the GPU).  A
 *                                     quarter to a third fewer path segments.
 *   RMD_RENDER_TRACE_BLACK_PATHS      never end a path early, grid or not (measurement and tests: the reference's full segment count).
 * END and TRACE together are refused (RMD_ERR_INVALID_ARGUMENT).  The rule itself: tests/test_gpu_parity.py::test_black_path_modes_*. */
#define RMD_RENDER_TRACE_BLACK_PATHS 2u
#define RMD_RENDER_END_BLACK_PATHS 4u
typedef struct rmd_settings {
	uint32_t bounce_limit; /* Settings.bounce_limit; trace() starts at depth 1 (:200,:235) */
	uint32_t sample_begin; /* first sample index s of this pass                             */
	uint32_t sample_count; /* number of consecutive samples to add per pixel               */
	uint32_t flags;        /* 0 or RMD_RENDER_DOF | one of RMD_RENDER_{TRACE,END}_BLACK_PATHS   */
	uint64_t seed; /* Philox key; see "RNG" below                                   */
} rmd_settings;

/* core::tile::Tile geometry (core/src/tile.rs:7-14) without the sample buffer. */
typedef struct rmd_tile_rect {
	uint32_t left, top, width, height;
} rmd_tile_rect;

/*
 * RNG (replaces rand::random::<f64>() at src/trace.rs:260,287,288,326,327,340,341,397,398).
 * Counter-based Philox4x32-10 (Salmon et al., SC'11), key = (seed lo32, seed hi32).  A sample's random numbers come
 * in BLOCKS: block b of sample s of pixel p is the Philox output for counter (p = y*W + x, s, b, 0), four words w0..w3:
 *     u_first  = (((uint64)w1 << 32 | w0) >> 11) * 2^-53        53-bit uniforms in [0,1), as rand 0.6 makes an f64
 *     u_second = (((uint64)w3 << 32 | w2) >> 11) * 2^-53
 *     u_22     = ((w0 & 0x7FF) << 11 | (w2 & 0x7FF)) * 2^-22     the 22 bits those two conversions discard
 * Every consumer takes exactly one block, in the order the reference makes its calls within a sample:
 *     block 0               pixel jitter: x <- u_first, y <- u_second (:326-327)
 *     then, thin lens only  one block per round of the rejection loop: r1 <- u_first, r2 <- u_second (:340-341)
 *     then                  one block per shaded depth: r1 <- u_first, r2 <- u_second (:397-398 or :287-288), and
 *                           r (:260) <- u_22 of the sample's PREVIOUS block (the jitter block or the last lens round for the first
 *                           depth, the preceding depth's block afterwards).  r only decides diffuse against specular: it is compared
 *                           with prob_d, which is 0.5 for Diffuse and 0.0 for Metal (:263-264), and for those two values a 22-bit
 *                           uniform gives exactly the probabilities a 53-bit one does.  Taking it from the previous block makes a
 *                           hit's lobe known when the hit is: a diffuse bounce off a black surface can end its path (see
 *                           RMD_RENDER_END_BLACK_PATHS) without the depth's block ever being drawn.  (Since ABI 2; ABI 1 took r from
 *                           the depth's own block: same distribution, different samples.  ABI 3 changed no sample, only which
 *                           rmd_settings.flags value ends black paths in scenes with grids; ABI 4 changed no sample either: it added
 *                           RMD_ERR_DEVICE_FAULT, rmd_reduce_framebuffer_async, rmd_launch_info.waves_per_workgroup and the rule for
 *                           non-finite scene parameters below; ABI 5 changed no sample: rmd_launch_info.queued, RMD_TUNE_PATH_QUEUES,
 *                           and every allocating entry point behind a catch.)
 * (One Philox evaluation per path segment, and no RNG state beyond a block counter and those 22 bits.)
 */

typedef struct rmd_context rmd_context;
typedef struct rmd_scene rmd_scene;
typedef struct rmd_comm rmd_comm;

/* ---- lifetime ---- */
/* RMD_ABI_VERSION of the library that was LOADED.  The structs of this header have grown from version to version (rmd_grid_desc::built in 4,
 * rmd_launch_info::queued in 5): a caller compiled against another version must compare this with its own RMD_ABI_VERSION before its first other
 * call and refuse to go on when they differ (raymond_amd/lib.py and integration/gpu.rs do). */
uint32_t rmd_abi_version(void);
/* One context per GPU (device_ordinal = HIP device index).  Creates its own stream. */
rmd_status rmd_context_create(int32_t device_ordinal, rmd_context **out);
/* Same, but launches on a caller-owned hipStream_t (passed as void*, 0 = null stream). */
rmd_status rmd_context_create_on_stream(int32_t device_ordinal, void *hip_stream, rmd_context **out);
void rmd_context_destroy(rmd_context *ctx);
/* Text of the last failure on this context (or of the last failed context-less call when ctx = NULL). */
const char *rmd_last_error(const rmd_context *ctx);

/* Scheduling tunables of a context.  NONE of them changes a result — every setting renders the same frame bit for bit
 * (tests/test_gpu_parity.py checks that) — they only move work between waves.  Defaults are read from the environment
 * variable named beside each key ONCE, when the context is created; 0 / unset = the library's own choice. */
enum {
	RMD_TUNE_SAMPLE_SPLIT = 0, /* RMD_SAMPLE_SPLIT: waves a wave tile's sample range is split over (0 = automatic)  */
	RMD_TUNE_WALK_BATCH = 1,   /* RMD_WALK_BATCH: lanes of a wave that wait for a grid walk before one is run        */
	RMD_TUNE_MASK_BUDGET = 2,  /* RMD_MASK_BUDGET: LDS bytes for the grids' occupancy masks (read by rmd_scene_create) */
	RMD_TUNE_LAUNCH_FORM = 3,  /* RMD_LAUNCH_FORM: 0 = the library's choice: persistent workgroups, one per CU, whose waves draw work
	                              items from a counter — launches with fewer items than the device has wave slots as one wave per
	                              item; 1 / "per-item" = always one wave per item; 2 / "persistent" = always persistent workgroups */
	RMD_TUNE_SCRATCH_CAP_MB = 4, /* RMD_SCRATCH_CAP_MB: cap of the per-sample scratch of split launches, MiB (0 = an eighth of the
	                              device memory that is free when the buffer is (re)allocated); a launch whose samples do not fit —
	                              or whose buffer the device cannot provide — runs as several passes                       */
	RMD_TUNE_WALK_CUT = 5,     /* RMD_WALK_CUT: K + 1, where a grid-walk call of a wave stops stepping under its last K rays and leaves their
	                              walks to the wave's next call (0 = the library's choice, K = 7; 1 = every call finishes every walk)      */
	RMD_TUNE_SPLIT_MIN_SAMPLES = 6, /* RMD_SPLIT_MIN_SAMPLES: fewest samples per pixel a work item of a split launch may hold (0 = the library's
	                              choice: 4 in scenes with grids — two items per wave tile from 4 samples per pixel on — 64 without)      */
	RMD_TUNE_CHAIN_ITEMS = 7,  /* RMD_CHAIN_ITEMS: split launches of scenes with grids whose persistent waves draw their next work item while the last
	                              paths of the current one finish: 0 = the library's choice (every such launch), 1 = never,                     
	                              2 = always                                                                                                  */
	RMD_TUNE_AXIS_PAIRS = 8,   /* RMD_AXIS_PAIRS: read by rmd_scene_create — pairs of opposite planes whose normals are exactly +e_k / -e_k (the walls of an
	                              axis-aligned room) tested with one component of the ray: 0 = the library's choice (yes, in scenes of regular
	                              parameters), 1 = never (every pair takes the general test: same samples, bit for bit)                      */
	RMD_TUNE_PATH_QUEUES = 9,  /* RMD_PATH_QUEUES: persistent split launches of scenes with grids keep their paths in queues in device memory — ray
	                              compaction between bounces: a wave's trips are 64 new samples, 64 parked hits or one grid walk for 64 parked rays
	                              (rmd_launch_info.queued): 0 = the library's choice (yes), 1 = never (a lane keeps its path: same samples, bit for bit) */
	RMD_TUNE_COUNT = 10
};
/* Free and total memory of the context's device, bytes (hipMemGetInfo): what a host that shares the GPU sizes its launches by. */
rmd_status rmd_context_memory_info(rmd_context *ctx, uint64_t *out_free_bytes, uint64_t *out_total_bytes);
rmd_status rmd_context_set_tunable(rmd_context *ctx, uint32_t key, int64_t value);
rmd_status rmd_context_get_tunable(const rmd_context *ctx, uint32_t key, int64_t *out_value);

/* Uploads the object table and the grids to HBM.  Replaces the per-worker
 * `scene.clone()` (src/trace.rs:182-185): one resident copy per GPU. */
rmd_status rmd_scene_create(rmd_context *ctx, const rmd_object *objects, uint32_t n_objects,
                            const rmd_grid_desc *grids, uint32_t n_grids, rmd_scene **out);
void rmd_scene_destroy(rmd_scene *scene);

/* ---- device buffers for hosts without their own HIP binding ---- */
rmd_status rmd_framebuffer_alloc(rmd_context *ctx, uint32_t width, uint32_t height, double **out_dev); /* zeroed W*H*3 */
rmd_status rmd_framebuffer_free(rmd_context *ctx, double *dev);
rmd_status rmd_framebuffer_zero(rmd_context *ctx, double *dev, size_t n_doubles);
rmd_status rmd_framebuffer_download(rmd_context *ctx, const double *dev, double *host, size_t n_doubles);
rmd_status rmd_framebuffer_upload(rmd_context *ctx, const double *host, double *dev, size_t n_doubles);

/* Tile rectangles of a device framebuffer <-> a PACKED host buffer: rect i's pixels row-major, width * height * 3 doubles, one rect after the
 * other in the order of `rects` — the layout of core::tile::Tile.data (core/src/tile.rs:13).  What a host scheduler needs to keep the tile sums
 * resident on the device between progressive passes and move only the tiles a message needs (TileProgressed / TileFinished, src/trace.rs:211-219)
 * instead of the whole W * H * 3 frame up and down around every call.  Rects must lie inside the W x H frame.
 * _async: the tiles are packed on the context's stream (behind the renders enqueued before), copied on the context's COPY stream — renders
 * enqueued afterwards overlap the copy — and are in `host_packed` once rmd_context_wait_transfers has returned; `host_packed` should be pinned
 * memory (rmd_host_alloc) for the copy to be asynchronous.  At most two such downloads are in flight per context: a third waits for the first. */
rmd_status rmd_framebuffer_download_tiles(rmd_context *ctx, const double *dev, uint32_t width, uint32_t height, const rmd_tile_rect *rects,
                                          uint32_t n_rects, double *host_packed);
rmd_status rmd_framebuffer_download_tiles_async(rmd_context *ctx, const double *dev, uint32_t width, uint32_t height, const rmd_tile_rect *rects,
                                                uint32_t n_rects, double *host_packed);
rmd_status rmd_context_wait_transfers(rmd_context *ctx);
rmd_status rmd_framebuffer_upload_tiles(rmd_context *ctx, const double *host_packed, double *dev, uint32_t width, uint32_t height,
                                        const rmd_tile_rect *rects, uint32_t n_rects);
/* Page-locked host memory (hipHostMalloc): copies to and from it run at the link's rate and asynchronously. */
rmd_status rmd_host_alloc(rmd_context *ctx, size_t bytes, void **out_host);
rmd_status rmd_host_free(rmd_context *ctx, void *host); /* ctx may be NULL (a block may outlive the context it was allocated through) */

/*
 * The hot path.  For every pixel (x,y) of every rect in `tiles`:
 *     accum[(x + y*W)*3 + c] += sample(x,y,s).c   for s = sample_begin .. sample_begin+sample_count-1, in that order
 * i.e. `sample_count` consecutive executions of src/trace.rs:197-205 over those
 * tiles.  Dividing by the sample count stays with the caller (TaskHandle::await,
 * src/trace.rs:95).  `accum_dev` is a DEVICE pointer to W*H*3 doubles, row-major
 * RGB (x + y*W, the layout await() assembles, :97).  Rects must lie inside the
 * backbuffer and must not overlap each other within one call.
 * Synchronous: returns after the kernel has completed.
 */
rmd_status rmd_render_tiles(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera,
                            const rmd_settings *settings, const rmd_tile_rect *tiles, uint32_t n_tiles,
                            double *accum_dev);
/* Same, enqueue only; pair with rmd_context_synchronize (lets one host thread drive several GPUs). */
rmd_status rmd_render_tiles_async(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera,
                                  const rmd_settings *settings, const rmd_tile_rect *tiles, uint32_t n_tiles,
                                  double *accum_dev);
rmd_status rmd_context_synchronize(rmd_context *ctx);
/* Host-buffer convenience for a caller that keeps Tile.data in RAM, as the
 * reference does: upload accum, render, download (PCIe-inclusive). */
rmd_status rmd_render_tiles_host(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera,
                                 const rmd_settings *settings, const rmd_tile_rect *tiles, uint32_t n_tiles,
                                 double *accum_host);
/* Duration of the most recent render kernel on this context, from HIP events
 * recorded on the context's stream around the launch (valid after a sync). */
rmd_status rmd_last_kernel_ms(rmd_context *ctx, float *out_ms);
/* How the most recent render on this context was launched (tests assert that a comparison exercised the instantiation they mean). */
typedef struct rmd_launch_info {
	uint32_t passes;          /* launches of the render kernel (the per-sample scratch may force several)                           */
	uint32_t split_k;         /* work items per wave tile of the last pass; > 1 = pooled (pixel, sample) hand-out + ordered sum    */
	uint32_t persistent;      /* 1 = persistent workgroups drawing work items from a counter, 0 = one wave per work item — the form
	                             the last pass was actually LAUNCHED in (a scene whose object table and masks leave a persistent
	                             workgroup too little LDS runs as one wave per item whatever was asked for)                       */
	uint32_t end_black_paths; /* 1 = zero-throughput paths were ended (see RMD_RENDER_END_BLACK_PATHS)                             */
	uint32_t has_grid;        /* 1 = the grid instantiation (wave-cooperative DDA walk) ran                                         */
	uint32_t waves_per_workgroup; /* waves of a workgroup of the last pass (persistent form: 16 unless the LDS left room for fewer) */
	uint32_t buffered;        /* 1 = the pooled (pixel, sample) hand-out + per-sample scratch + ordered sum ran (split_k > 1, or one item per
	                             wave tile: short launches of scenes with grids), 0 = direct mode (lane = pixel, no scratch)               */
	uint32_t chained;         /* 1 = persistent waves drew their next work item while the last paths of the current one finished (short
	                             split launches of scenes with grids; RMD_TUNE_CHAIN_ITEMS)                                                */
	uint32_t queued;          /* 1 = the paths lived in per-wave queues in device memory and every trip of a wave served lanes that all needed
	                             the same thing (persistent split launches of scenes with grids; RMD_TUNE_PATH_QUEUES; ABI 5)            */
} rmd_launch_info;
rmd_status rmd_last_launch_info(const rmd_context *ctx, rmd_launch_info *out);

/* ---- output stage (TaskHandle::await src/trace.rs:93-99, cli_old/src/main.rs:155-181) ---- */
/* out_rgb8[i] = trunc(255 * (1 - exp(-(accum[i]/sample_count) * exposure))^(1/gamma)); device in, host out.  A pixel with a channel that is
 * NaN or outside (-1, 256) stays (0, 0, 0), as cast::<u8>() returning None leaves it (:176-181).  Byte for byte what the host's libm gives:
 * the device evaluates every pixel and the few whose value lies within 1e-7 of a truncation boundary are recomputed on the host. */
rmd_status rmd_resolve_tonemap(rmd_context *ctx, const double *accum_dev, uint32_t width, uint32_t height,
                               uint32_t sample_count, double exposure, double gamma, uint8_t *out_rgb8_host);

/* ---- multi-GPU (no reference counterpart; the reference has no collectives) ---- */
#define RMD_COMM_ID_BYTES 128
/* One process per GPU.  rank 0 calls rmd_comm_unique_id and ships the 128 bytes to the other ranks by any means.
 * Environment: RCCL shares buffers between the ranks through HIP IPC; on hosts whose driver supports dmabuf IPC only, every rank needs
 * HSA_ENABLE_IPC_MODE_LEGACY=0 in its environment BEFORE its first HIP call (the HSA runtime reads its environment once).  The library never
 * changes the environment on its own: a multi-GPU caller either exports the variable or calls rmd_comm_prepare_process() — which sets it to 0
 * unless the environment already holds a value — before rmd_context_create and any other HIP call of the process; single-GPU callers are not
 * affected.  rmd_comm_create names the variable in its error text when RCCL's initialisation fails without it. */
rmd_status rmd_comm_prepare_process(void);
rmd_status rmd_comm_unique_id(uint8_t id_out[RMD_COMM_ID_BYTES]);
rmd_status rmd_comm_create(rmd_context *ctx, const uint8_t id[RMD_COMM_ID_BYTES], int32_t rank, int32_t world_size,
                           rmd_comm **out);
void rmd_comm_destroy(rmd_comm *comm);
/* In-place ncclReduce(sum, f64) of the accumulated framebuffer to `root` over xGMI.
 * Every pixel is non-zero on exactly one rank, so the sum is bit-identical to a 1-GPU render. */
rmd_status rmd_reduce_framebuffer(rmd_comm *comm, double *accum_dev, size_t n_doubles, int32_t root);
/* Same, enqueue only (on the context's stream, behind the renders enqueued before it); pair with rmd_context_synchronize.  Lets a rank queue
 * zeroing, rmd_render_tiles_async and the reduce of several frames back to back. */
rmd_status rmd_reduce_framebuffer_async(rmd_comm *comm, double *accum_dev, size_t n_doubles, int32_t root);

/* ---- host-side grid build (AccGrid::build_from_mesh, core/src/geometry/acc_grid.rs:6-83) ---- */
typedef struct rmd_grid_build rmd_grid_build; /* owns the arrays a rmd_grid_desc points at */
/* tri_pos/tri_nrm: n_tris*9 doubles each (host).  Computes mesh bounds (mesh.rs:123-140), resolution,
 * cell_size, cells, mapping_table.  RMD_ERR_GRID_INDEX where the reference would panic on the
 * `x + res.x*(y + z*res.z)` index (acc_grid.rs:61) running past the cell array. */
rmd_status rmd_grid_build_from_mesh(const double *tri_pos, const double *tri_nrm, uint64_t n_tris,
                                    rmd_grid_build **out);
/* The same build on the GPU of `ctx` (bounds reduction, per-cell atomic counts, device scan, fill, per-cell sort);
 * the resulting tables are byte-identical to rmd_grid_build_from_mesh's. */
rmd_status rmd_grid_build_from_mesh_gpu(rmd_context *ctx, const double *tri_pos, const double *tri_nrm, uint64_t n_tris,
                                        rmd_grid_build **out);
/* Fills `desc` with pointers into `build` (valid until rmd_grid_build_destroy). */
rmd_status rmd_grid_build_describe(const rmd_grid_build *build, rmd_grid_desc *desc);
void rmd_grid_build_destroy(rmd_grid_build *build);

#ifdef __cplusplus
}
#endif
#endif /* RAYMOND_HIP_H */
