// Device-resident scene layout shared by the kernels (kernels.hip) and the C-ABI host code (api.cpp).
// HBM layout is described in DESIGN.md §"Data layout in HBM".
#pragma once
#include <stdint.h>

namespace rmd {

// One scene::Object (reference core/src/scene.rs:33-37), flattened.  128 bytes, 16-byte aligned so the
// uniform object loop can fetch it with wide scalar loads and the per-lane post-hit lookup from the LDS
// copy is conflict-light.
struct alignas(16) DevObject {
	uint32_t geometry_kind; // RMD_GEOM_*
	uint32_t grid_index;
	uint32_t material_kind; // RMD_MAT_*
	uint32_t pair_info;     // planes only (rmd_scene_create: pair_opposite_planes): 0 = tested on its own; kPairTestedAtPartner | j = tested together
	                        // with plane j > this index, at j's turn; i + 1 = tested here together with plane i < this index
	double origin[3]; // plane origin / sphere centre
	double radius;
	double normal[3]; // plane normal
	double roughness;
	double color[3]; // Diffuse/Metal colour, Emission radiance
	double metalness; // 0.0 Diffuse, 1.0 Metal (src/trace.rs:248-249)
	uint32_t flags;   // kObj*
	uint32_t _pad2;
	double partner_origin_k; // kObjAxisPair, on the later plane of the pair: origin[k] of its partner
};
constexpr uint32_t kObjBlackDiffuse = 1u; // Diffuse with colour (0, 0, 0): its diffuse bounce has weight exactly zero (src/trace.rs:279-281)
// A pair of opposite planes whose normals are exactly +e_k and -e_k (the walls of an axis-aligned room), in a scene of regular parameters: tested
// by axis_pairs_visit (scene_split.hpp) ahead of the object loop, which skips both planes.  Set on the LATER plane of the pair (the one whose
// pair_info names its partner): kObjAxisPair | k << 8 | (the EARLIER plane is the one with normal +e_k) << 10.
constexpr uint32_t kObjAxisPair = 2u, kObjAxisShift = 8u, kObjAxisEarlierIsPlus = 1u << 10;
static_assert(sizeof(DevObject) == 128, "DevObject layout");
constexpr uint32_t kPairTestedAtPartner = 0x80000000u;
constexpr uint32_t kPairAxis = 0x40000000u; // with kPairTestedAtPartner, on BOTH planes of an axis pair (kObjAxisPair): pair_info & 0x3FFFFFFF = the partner's index

// The two planes of an axis pair (RenderParams::axis_pairs) as the axis rule uses them (scene_split.hpp: axis_pair_test): the coordinate of the plane
// with normal +e_k and of the one with normal -e_k, and the two planes' object indices.  Three of them (k = x, y, z) sit behind the object table's last
// record: scalar loads at ONE address the trip computes, where reading them out of the planes' records is an index computation, a load and six selects
// per pair and trip.
struct alignas(32) AxisWalls {
	double o_plus, o_minus;
	uint32_t idx_plus, idx_minus, _pad[2];
};
static_assert(sizeof(AxisWalls) == 32 && 3 * sizeof(AxisWalls) <= sizeof(DevObject), "the block behind the object table");

// One AccGrid (reference core/src/geometry/acc_grid.rs:27-33), re-laid out at upload for the wave-cooperative walk
// (grid_walk.hpp).  `cells[c] -> mapping_table[off] = count, idx...` (acc_grid.rs:67-74) becomes
//   tri_recs[t]                      ONE record per triangle: v0, edge1, edge2 (9 f64; kTriRecStride bytes apart) — 8 MB for the 99k-triangle
//                                    benchmark mesh, where per-cell copies of the records were 60 MB (measured and without effect on the
//                                    frame time: records renumbered in the order a scan over the cells meets them, and a 72-byte stride;
//                                    a 128-byte stride is 13 % slower: every lane of a load then reads the same 16 bytes of its line)
//   tri_ids[..]                      lists of triangle indices, in mapping_table order
//   cell_entries[c * 8 + s] = {first id, count}     s = 0: every triangle of cell c (a walk's first cell);
//                                    s = 1 .. 6: the triangles of c that cell c - delta_s does NOT list, delta_s = +1, -1, +res.x, -res.x,
//                                    +res.x*res.z, -res.x*res.z — the index step by which the DDA entered c (kEntrySlot*).
// A triangle that the previous cell of a walk listed has been tested against this very ray already and missed (a hit there would have ended
// the walk, acc_grid.rs:151-153), so it misses again: leaving it out changes no result and removes 22 % of the reference's triangle tests on
// the benchmark mesh (oracle counter `retests`).  One 8-byte gather per candidate cell, as before; a list equal to the full one shares it.
struct CellEntry {
	uint32_t first, count;
};
constexpr uint32_t kEntrySlots = 8; // entries per cell (slot 7 unused: a cell's row is one 64-byte line)
#ifndef RMD_TRI_REC_STRIDE
#define RMD_TRI_REC_STRIDE 80
#endif
constexpr uint32_t kTriRecStride = RMD_TRI_REC_STRIDE; // bytes between triangle records (72 used)

struct alignas(16) DevGrid {
	double bbox_min[3];
	double bbox_max[3];
	double cell_size[3];
	double inv_cell_size[3]; // 1.0 / cell_size, correctly rounded, or NaN (internal.hpp: exact_reciprocal): the divisors of the walk's first-cell quotients (div_by)
	uint64_t res[3];
	uint64_t n_cells;
	const CellEntry *cell_entries; // n_cells x kEntrySlots x {first id, count}
	const uint32_t *tri_ids;    // the entries' lists of triangle indices
	const void *tri_recs;       // n_tris x kTriRecStride bytes: v0, edge1, edge2
	const double *tri_pos;      // n_tris * 9: v0 v1 v2 (Heron normal, triangle.rs:47-68)
	const double *tri_nrm;      // n_tris * 9: n0 n1 n2
	const double *tri_aux;      // n_tris * 4: |v0v1|, |v0v2|, Heron area of the triangle, its exact reciprocal or NaN (internal.hpp: triangle_aux)
	const double *tri_sph;      // n_tris * 4: centre and inflated squared radius r2a of a sphere around the triangle (internal.hpp: triangle_sphere): the
	double sph_kb;              //   walk's pre-test drops a pair whose line passes the centre at more than sqrt(r2a + sph_kb * |centre - origin|^2)
	const uint32_t *mask_words; // occupancy bitmask (global copy, staged into LDS by every workgroup)
	uint64_t n_tris;
	uint32_t mask_bits;         // number of valid bits; bit i covers cells [i << mask_shift, (i+1) << mask_shift)
	uint32_t mask_shift;
	uint32_t mask_n_words;
	uint32_t mask_lds_word;     // word offset of this grid's mask inside the LDS mask area, 0xFFFFFFFF = not staged
};

// 8x8-pixel wave tile: one wavefront, lane = pixel (lane & 7, lane >> 3).
struct WaveTile {
	uint16_t x0, y0;
	uint8_t w, h; // 1..8
	uint16_t _pad;
};
static_assert(sizeof(WaveTile) == 8, "WaveTile layout");

// Loop-invariant camera terms of generate_primary_ray (src/trace.rs:322-333), evaluated once on the host.
struct RenderParams {
	double cam_pos[3];
	double width, height;  // backbuffer size as f64 (:323-324)
	double inv_width, inv_height; // 1.0 / width, 1.0 / height, correctly rounded (host): divisors of div_by() in primary_ray
	double aspect;         // width / height (:325)
	double tan_half_fov;   // tan(fov_vert / 2 * PI / 180) (:329-330)
	double focal_length;
	double aperture_radius;
	uint32_t W, H;
	uint32_t bounce_limit;
	uint32_t sample_begin;
	uint32_t sample_count;
	uint32_t n_objects;
	uint32_t key0, key1; // Philox key = seed lo/hi
	uint32_t n_work;     // wave tiles (tile mode) or list entries (list mode)
	uint32_t use_dof;
	uint32_t n_grids;
	uint32_t walk_batch;       // lanes of a wave that must wait for a grid walk before one is run (launch.hpp: kWalkBatchDefault; RMD_WALK_BATCH overrides)
	uint32_t mask_words_total; // LDS words reserved for the grids' occupancy masks
	uint32_t split_k;          // >1: each wave tile's sample range is split over split_k waves writing to sample_buf
	double *sample_buf;        // [wave tile][sample - sample_begin][lane][3] f64, only when split_k > 1
	uint32_t debug_flags;      // diagnostics (RMD_DEBUG env): 1 = skip triangle tests, 2 = skip grid walks (timing only, wrong results), 8 = count walk events
	unsigned long long *debug_counters; // 16 counters, only touched when debug_flags & 8
	uint32_t *work_counter;             // persistent launches: the next work item (zeroed by the host before the launch)
	uint32_t *tile_done;                // split launches: finished waves per wave tile (zeroed by the host); the last one adds the tile's samples
	uint32_t end_black_paths;           // 1: a path whose throughput is exactly (0, 0, 0) is ended — scenes without grids unless RMD_RENDER_TRACE_BLACK_PATHS, scenes with grids only with RMD_RENDER_END_BLACK_PATHS (api.cpp: make_params)
	uint32_t walk_cut;                  // K: a walk call puts the walks of its last K lanes aside for the wave's next call (grid_walk.hpp); 0 = never.  Any value gives the same image
	uint32_t shade_last_depth;          // 1: the hit at the bounce limit is shaded although its result is its weight times the zero the recursive call returns
	                                    // (src/trace.rs:235-237) — a scene outside the regular parameter class (api.cpp: rmd_scene::regular), where that weight can be
	                                    // NaN with finite inputs (roughness 0: 0 / 0 in geometry_schlick_ggx) and the reference's sample NaN x 0 = NaN
	uint32_t chain_items;               // 1: persistent waves of a split launch of a scene with grids draw their next work item while the last paths of the current one
	                                    // finish (render_kernel.hpp: render_wave, CHAIN) — launches of few samples per pixel, where an item's drain is a fifth of it
	uint32_t _pad1;
	uint32_t buffered;                  // 1: the tiles-buffered instantiation (pooled (pixel, sample) hand-out, per-sample scratch, ordered sum) — also with split_k = 1
	uint32_t axis_pairs;                // three 10-bit fields, one per axis k: index + 1 of the later plane of THE pair of opposite planes with normals +-e_k that is
	uint32_t _pad2;                     //   tested ahead of the object loop (kObjAxisPair), 0 = none
	uint32_t *fault;                    // the context's fault words (host memory mapped into the device's address space; kFault*): a wave whose loop runs past
	                                    // its bound reports here, poisons work_counter so that the launch drains, and leaves (render_kernel.hpp: report_fault)
	unsigned char *queue_buf;           // split launches of grid scenes in the persistent form (render_kernel.hpp: render_wave_queued): the waves' path queues in
	                                    // device memory, queue_wave_bytes per resident wave (wave = blockIdx.x * waves per workgroup + wave of the workgroup); null = the
	uint32_t queue_wave_bytes;          // lane-per-path form
	uint32_t queue_paths;               // paths a wave may have in flight (capacity of each of its queues)
	uint32_t walk_steps_bound;          // most steps a ray's walk can take in any grid of the scene (res.x + res.y + res.z + 3: grid_walk.hpp) — a walk that is put
	uint32_t _pad3;                     //   aside takes part in one WALK trip per step at worst: part of render_wave_queued's trip bound
	unsigned long long visit_mask;      // objects 0 .. 63 whose turn is in the object loops: not the planes tested at their partner's turn or by the axis rule —
	                                    //   passing one of those by costs a loop a scalar round trip each time (6 of the 8 objects of the reference's scenes).
	unsigned long long grid_mask;       //   grid_mask: objects 0 .. 63 that are grids (intersect_grids' turns).  Objects from 64 on are visited one by one.
	unsigned long long sample_magic;    // floor(2^64 / sample_count) + 1, or 0 when sample_count is 1: the multiplier by which a queued path gets its sample's
	                                    //   number back from its scratch sector (render_kernel.hpp: sample_of_sector)
};
// Fault words: [0] OR of the kFault* codes, [1] the work item (or list entry) of the wave that reported last, [2] number of reports.
constexpr uint32_t kFaultTripLoop = 1u;       // render_wave's trip loop (lane-per-path form: direct mode, the mesh kernel, the list probes)
constexpr uint32_t kFaultSortedTripLoop = 2u; // render_wave_sorted's trip loop (role-sorted spheres kernel)
constexpr uint32_t kFaultWorkLoop = 4u;       // a persistent wave drew more work items than the launch has
constexpr uint32_t kFaultWalkRounds = 8u;     // a grid walk call ran more rounds than a ray can take steps (grid_walk.hpp)
constexpr uint32_t kFaultQueuedTripLoop = 16u; // render_wave_queued's trip loop (the mesh kernel with its paths in queues)
constexpr uint32_t kFaultWords = 4u;
constexpr uint32_t kWorkCounterPoison = 0x80000000u; // OR-ed into the work counter by a faulting wave: every later draw is past the last item (items are < 2^31: api.cpp)

// List mode (probe): one lane per explicit (x, y, sample).
struct ListWork {
	uint32_t x, y, sample, _pad;
};

} // namespace rmd
