// Host-callable launchers of kernels.hip (internal to the library).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/raymond_hip.h"
#include "device_types.hpp"

// entries per lane in the list-mode path record: one per trace() depth, bounce_limit <= 16
#define RMD_PATH_STRIDE 17
// include/raymond_hip.h: RMD_MAX_BOUNCE_LIMIT (api.cpp asserts the two agree) — a path has at most this many segments, which bounds the render loops
#define RMD_MAX_BOUNCE_LIMIT_DEV 16u

namespace rmd {

enum ProbeOp {
	PROBE_PHILOX = 0,
	PROBE_UNIFORM,
	PROBE_SPHERE_INTERSECT,
	PROBE_SPHERE_NORMAL,
	PROBE_PLANE_INTERSECT,
	PROBE_AABB_INTERSECT,
	PROBE_TRIANGLE_INTERSECT,
	PROBE_TRIANGLE_NORMAL,
	PROBE_ONB,
	PROBE_COSINE_HEMISPHERE,
	PROBE_SAMPLE_GGX,
	PROBE_GGX_DISTRIBUTION,
	PROBE_GEOMETRY_SMITH,
	PROBE_FRESNEL_SCHLICK,
	PROBE_PRIMARY_RAY,
	PROBE_ELEMENTARY,
	PROBE_PRETEST_PAIR,
	PROBE_OP_COUNT
};

// LDS a workgroup may use: the CU's 160 KiB less a margin for alignment
constexpr size_t kLdsBudgetBytes = 160u * 1024u;
// LDS bytes reserved for the grids' occupancy masks (shared by the waves of a workgroup)
constexpr size_t kMaskBudgetBytes = 48u * 1024u;
// doubles per entry of the per-sample scratch of split launches: r, g, b and one of padding = one 32-byte sector
constexpr uint32_t kSampleStride = 4;
// walk batching (kernels.hip): lanes of a wave that must be waiting for a grid walk before one is run
constexpr uint32_t kWalkBatchDefault = 32;
// fewest samples per pixel a work item of a split launch of a grid scene may hold (api.cpp: choose_split; RMD_TUNE_SPLIT_MIN_SAMPLES overrides)
#ifndef RMD_SPLIT_MIN_SAMPLES_GRID
#define RMD_SPLIT_MIN_SAMPLES_GRID 4
#endif
constexpr uint32_t kSplitMinSamplesGrid = RMD_SPLIT_MIN_SAMPLES_GRID;
// scenes without grids: fewest samples per pixel a launch that is too short to split runs as ONE buffered item per wave tile (role-sorted trips)
// instead of in direct mode (api.cpp: choose_split)
#ifndef RMD_SORTED_MIN_SAMPLES
#define RMD_SORTED_MIN_SAMPLES 128
#endif
constexpr uint32_t kSortedMinSamples = RMD_SORTED_MIN_SAMPLES;
// split launches of scenes with grids of at most this many samples per pixel run the instantiation whose waves chain their work items: all of
// them since the round's second half (the chained instantiation used to spill 27 registers against 15 and lost 2.6 % at 500 samples per pixel —
// hence a limit of 96 —; at 11 against 9 it wins at every size: C3 at 128 / 200 / 500 spp 101.7 / 157.2 / 387.8 -> 98.9 / 154.1 / 384.5 ms)
#ifndef RMD_CHAIN_MAX_SAMPLES
#define RMD_CHAIN_MAX_SAMPLES 0x7FFFFFFF
#endif
constexpr uint32_t kChainMaxSamples = RMD_CHAIN_MAX_SAMPLES;
// walks put aside (grid_walk.hpp: cut_lanes): a walk call leaves its last K walkers to the wave's next call
constexpr uint32_t kWalkCutDefault = 7; // (round 6: 4 -> 7 — the queued form's walks have 60 rays: C3 at 200 spp 120.1 -> 118.9 ms; the lane-per-path form times within 1 % for 2 .. 12)
// largest |roughness| a material may have: keeps the GGX sampling angle below 2^45 (device_core.hpp, sincos_cw)
constexpr double kMaxRoughness = 512.0;
// smallest non-zero |roughness|: the specular weight's denominator holds (roughness^2 / 8)^2 (device_core.hpp: next_ray), which must not underflow
constexpr double kMinRoughness = 1e-12;
// waves per workgroup of the grid instantiation (they share the LDS occupancy masks)
// 4-wave workgroups: 4 of them (16 waves) fit a CU's LDS beside their staged masks and retire at a finer grain than 8-wave ones
#ifndef RMD_GRID_WAVES
#define RMD_GRID_WAVES 4
#endif
constexpr uint32_t kGridWavesPerWg = RMD_GRID_WAVES;
// waves of a persistent workgroup (one per CU: all 16 wave slots that 128 registers per lane leave)
constexpr uint32_t kPersistWavesPerWg = 16;
#ifndef RMD_GRID_PERSIST_WAVES
#define RMD_GRID_PERSIST_WAVES 16
#endif
constexpr uint32_t kGridPersistWavesPerWg = RMD_GRID_PERSIST_WAVES; // ... of the grid instantiation
// the spheres kernel's split launches (render_kernel.hpp: render_wave_sorted): path slots of a wave's pool and waves of a persistent workgroup —
// 16 pools of 112 slots (86 bytes each) and the object table fit the CU's 160 KB
#ifndef RMD_SORT_SLOTS
#define RMD_SORT_SLOTS 120
#endif
#ifndef RMD_SORT_WAVES
#define RMD_SORT_WAVES 16 // waves of one persistent workgroup
#endif
#ifndef RMD_SORT_WGS_PER_CU
#define RMD_SORT_WGS_PER_CU 1 // persistent workgroups per CU (a workgroup holds at most 16 waves)
#endif
constexpr size_t kSortPoolBytes = 84u * RMD_SORT_SLOTS; // per-wave LDS (sizeof(HitStack): 9 doubles + 3 words an entry)
// every wave's LDS area ends with 16 bytes of bookkeeping (render_kernel.hpp: word 0 = 1 + the work item a persistent wave drew last)
constexpr size_t kWaveHeadBytes = 16u;
// the form a render launch was made in (render_kernel.hpp: launch_render) -> rmd_launch_info
struct LaunchShape {
	uint32_t persistent = 0, waves_per_wg = 0, queued = 0, resident_waves = 0;
};
// paths a wave of the queued form may have in flight (render_kernel.hpp: render_wave_queued; at least 192, a multiple of 64)
#ifndef RMD_QUEUE_PATHS
#define RMD_QUEUE_PATHS 256
#endif
constexpr uint32_t kQueuePaths = RMD_QUEUE_PATHS;
inline size_t path_queue_bytes_host(uint32_t cap) { return (size_t)cap * (9u * 8u + 13u * 8u + 4u * 4u + 9u * 4u); } // (render_kernel.hpp: path_queue_bytes)
static_assert(kQueuePaths >= 192u && kQueuePaths % 64u == 0u, "two stacks short of a full trip + the 64 paths of a generation trip");
size_t render_lds_bytes(uint32_t n_objects, uint32_t mask_words_total, uint32_t waves_per_wg);
uint32_t render_waves_per_wg(uint32_t n_objects, uint32_t mask_words_total);
// n_cus > 0 and P.work_counter set: grid scenes run as persistent workgroups (render_kernel.hpp)
hipError_t launch_render_tiles(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids,
                               const WaveTile *wave_tiles, double *accum, uint32_t n_cus = 0, LaunchShape *shape = nullptr);
hipError_t launch_render_list(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids,
                              const ListWork *list, double *rgb_out, int32_t *path_obj, uint32_t *path_sub);
// tile rectangles of a frame <-> a packed buffer (kernels.hip: tile_copy_kernel); `first[i]` = pixels in front of rect i
hipError_t launch_tile_copy(hipStream_t stream, bool to_packed, double *frame, double *packed, const rmd_tile_rect *rects, const uint64_t *first,
                            uint32_t n_rects, uint32_t W);
// pixel += the per-sample radiance of a split launch, in sample order (kernels.hip: sum_kernel)
hipError_t launch_sum(hipStream_t stream, const RenderParams &P, const WaveTile *wave_tiles, double *accum);
// |255 * tm - k| below this flags a pixel for the host's libm (kernels.hip: tonemap_kernel); device exp / pow are good to ~1e-12 there
constexpr double kTonemapGuard = 1e-7;
hipError_t launch_tonemap(hipStream_t stream, const double *accum, uint8_t *rgb8, size_t n_pixels, double sample_count,
                          double exposure, double inv_gamma, uint32_t *flagged, uint32_t *n_flagged);
hipError_t launch_probe(hipStream_t stream, int op, uint32_t n, const double *in, int in_stride, double *out, int out_stride,
                        const RenderParams &P);
hipError_t launch_probe_scene(hipStream_t stream, int mode, uint32_t g, uint32_t n, const DevObject *objs, uint32_t n_objects,
                              const DevGrid *grids, uint32_t n_grids, uint32_t mask_words_total, uint32_t axis_pairs, const double *rays, double *out);

} // namespace rmd
