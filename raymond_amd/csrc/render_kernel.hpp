// The render kernel (all instantiations) and its launcher template: included by kernels.hip (tile modes, the product's hot path)
// and by probe_kernels.hip (list mode: the same code driven by an explicit (x, y, sample) list, for the parity probes).
#pragma once
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "grid_walk.hpp"
#include "launch.hpp"
#include "scene_split.hpp"

namespace rmd {

// LDS per wave: the cooperative-walk scratch, only when the scene has grids.
// per-wave LDS of the grid kernel: the walk scratch and the 64 paths' throughput (3 doubles per lane)
// ... and the DDA states of walks put aside for the wave's next walk call (grid_walk.hpp: WalkCarry)
// ... and the wave's 16 bytes of bookkeeping behind them (launch.hpp: kWaveHeadBytes)
// ... and, per lane, which (pixel, sample) its path belongs to: PathId (a wave's lanes may hold paths of two work items: render_wave, CHAIN)
struct alignas(16) PathId {
	uint32_t px[64];     // x | y << 16
	uint32_t smp[64];    // the sample's index (sample_begin included)
	uint32_t sector[64]; // the sample's 32-byte sector in the per-sample scratch
};
__host__ __device__ inline size_t wave_lds_bytes(uint32_t n_grids) { return kWaveHeadBytes + (n_grids ? sizeof(WalkScratch) + 64u * 3u * sizeof(double) + sizeof(WalkCarry) + sizeof(PathId) : 0); }

// Block -> work item mapping.  Workgroups are dealt round-robin over the 8 XCDs, and host tiles arrive in the
// reference's column-major order, so consecutive work items are vertical neighbours.  Plain order (block b -> item b)
// spreads every image region over all 8 XCDs: the mesh region costs ~10x a wall region per tile, and a mapping that
// hands each XCD one contiguous band of the image leaves most XCDs idle while two or three grind through the mesh
// (measured: 2x slower).  Balance beats L2 locality here — the scene (tens of MB) lives in L2 + Infinity Cache anyway.
RMD_DEV uint32_t work_item_of_block(uint32_t b, uint32_t nb) {
	(void)nb;
	return b;
}

// Every loop of the render kernel has a bound that no input reaches (each one's comment says why).  A wave that nevertheless runs into one —
// a fault of this library, or memory it relies on overwritten — must not spin (the reference's own failure mode: `TaskHandle::await` polls for
// ever after a worker's panic, src/trace.rs:82-92) and must not end quietly with a frame that looks valid either: it ORs the loop's code into
// the context's fault word (host memory the device writes through: the host reads it after the launch and returns RMD_ERR_DEVICE_FAULT,
// api.cpp: check_fault), poisons the launch's work counter so that no wave draws another item, and leaves.
RMD_DEV void report_fault(const RenderParams &P, uint32_t code, uint32_t detail) {
	if ((threadIdx.x & 63u) == 0u) {
		if (P.fault != nullptr) {
			__hip_atomic_fetch_or(P.fault, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(P.fault + 1, detail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_fetch_add(P.fault + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		if (P.work_counter != nullptr) __hip_atomic_fetch_or(P.work_counter, kWorkCounterPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
// (A/B switches for tools/ab_multi.sh: what the bounds cost — 0 compiles a check out)
#ifndef RMD_BOUND_TRIPS
#define RMD_BOUND_TRIPS 1
#endif
#ifndef RMD_FLAT_OBJECT_TESTS
#define RMD_FLAT_OBJECT_TESTS 1
#endif
#ifndef RMD_BOUND_DRAWS
#define RMD_BOUND_DRAWS 1
#endif
// render_wave_sorted: trips a wave may take per (pixel, sample) pair of its work item, as if ONE lane ran them all one after the other (a path has
// at most RMD_MAX_BOUNCE_LIMIT segments and one more trip ends it; a wave runs its pairs 64 at a time, so a real item takes a fraction of this).
constexpr uint32_t kTripBoundPerPair = RMD_MAX_BOUNCE_LIMIT_DEV + 4u;
constexpr uint32_t kStallBound = 1u << 18; // render_wave: trips a wave may take without one of its lanes finishing a sample or being handed one

// core/src/scene.rs:54-74: linear closest hit over the objects; strict '<' keeps the first object on ties.
// Wave-level: called by all 64 lanes in uniform control flow, `want` marks the lanes that carry a ray.  The object
// table is indexed uniformly (scalar loads); planes and spheres are tested per lane, a grid object runs the
// wave-cooperative walk.
template <bool GRID>
RMD_DEV int scene_intersect_wave(const DevObject *__restrict__ objs, uint32_t n_objects, const DevGrid *__restrict__ grids,
                                 const uint32_t *lds_masks, WalkScratch &scr, bool want, V3 ro, V3 rd, double &t_best, uint32_t &sub_best,
                                 uint32_t axis_pairs, uint32_t debug_flags = 0, unsigned long long *dbg = nullptr, bool arbitrary_rays = false,
                                 unsigned long long turns = ~0ull) {
	double closest = kFMax;
	int best = -1;
	uint32_t sub = 0;
	axis_pairs_visit(objs, n_objects, axis_pairs, want, ro, rd, closest, best, arbitrary_rays); // (the room's walls: scene_split.hpp; `sub` stays 0 for a plane)
#if RMD_FLAT_OBJECT_TESTS
	if constexpr (!GRID) {
		// without grid objects: tests without control flow, the running minimum updated by selects (device_core.hpp: *_test_flat)
		for (uint32_t i = next_turn(~0u, turns); i < n_objects; i = next_turn(i, turns)) {
			const DevObject &o = objs[i];
			double t;
			bool ok;
			int idx = (int)i;
			if (o.geometry_kind == 0u) {
				if (o.pair_info != 0u) {
					if (o.pair_info & kPairTestedAtPartner) continue; // at its partner's turn
					const uint32_t e = o.pair_info - 1u;
					bool first;
					ok = plane_pair_test_flat(ld3(objs[e].origin), ld3(objs[e].normal), ld3(o.origin), ld3(o.normal), ro, rd, t, first);
					idx = first ? (int)e : (int)i;
					ok = ok && want && lex_less(t, idx, closest, best);
				} else {
					ok = plane_test_flat(ld3(o.origin), ld3(o.normal), ro, rd, t) && want && lex_less(t, idx, closest, best);
				}
			} else {
				ok = sphere_test_flat(ld3(o.origin), o.radius, ro, rd, t) && want && lex_less(t, idx, closest, best);
			}
			closest = ok ? t : closest, best = ok ? idx : best;
		}
		t_best = closest, sub_best = 0u;
		return best;
	}
#endif
	for (uint32_t i = next_turn(~0u, turns); i < n_objects; i = next_turn(i, turns)) {
		const DevObject &o = objs[i];
		// planes and spheres: the hit is consumed where it is found (device_core.hpp: *_visit)
		if (o.geometry_kind == 0u) {
			if (o.pair_info != 0u) { // a plane with an exactly opposite partner (device_core.hpp: plane_pair_visit)
				const uint32_t e = o.pair_info - 1u; // the earlier partner, e < i (a plane marked kPairTestedAtPartner is tested at its partner's turn)
				// Scene::intersect's scan keeps the lexicographic minimum of (distance, index); objects between e and i have had their turn
				if (want && (o.pair_info & kPairTestedAtPartner) == 0u)
					plane_pair_visit(ld3(objs[e].origin), ld3(objs[e].normal), ld3(o.origin), ld3(o.normal), ro, rd, [&](double t, bool first) {
						const int idx = first ? (int)e : (int)i;
						if (lex_less(t, idx, closest, best)) closest = t, best = idx, sub = 0u;
					});
			} else if (want)
				plane_visit(ld3(o.origin), ld3(o.normal), ro, rd, [&](double t) {
					if (lex_less(t, (int)i, closest, best)) closest = t, best = (int)i, sub = 0u;
				});
		} else if (o.geometry_kind == 1u) {
			if (want)
				sphere_visit(ld3(o.origin), o.radius, ro, rd, [&](double t) {
					if (lex_less(t, (int)i, closest, best)) closest = t, best = (int)i, sub = 0u;
				});
		} else if constexpr (GRID) {
			const DevGrid &g = grids[o.grid_index];
			const uint32_t *mask = (lds_masks && g.mask_lds_word != 0xFFFFFFFFu) ? lds_masks + g.mask_lds_word : nullptr;
			double t;
			uint32_t tri = 0;
			bool hit = false;
			grid_intersect_wave(g, mask, scr, want, ro, rd, hit, t, tri, debug_flags, dbg);
			if (want && hit && lex_less(t, (int)i, closest, best)) closest = t, best = (int)i, sub = tri;
		}
	}
	t_best = closest;
	sub_best = sub;
	return best;
}

// A walk is run when RenderParams::walk_batch lanes of the wave wait for one (launch.hpp: kWalkBatchDefault = 32; with the persistent
// workgroups 40 / 24 / 6 / look-ahead 16 measured 1.7 % faster on the benchmark mesh than round 1's 32 / 16 / 4 / 12),
// ... or fewer than this many lanes could do anything else on this trip (a trip costs the same for 5 lanes as for 50)
#ifndef RMD_WALK_MIN_RUNNABLE
#define RMD_WALK_MIN_RUNNABLE 24
#endif
constexpr uint32_t kWalkMinRunnable = RMD_WALK_MIN_RUNNABLE;
// ... or this many trips have passed since the wave's last walk (scenes where few rays reach a grid: bounds the wait)
#ifndef RMD_WALK_MAX_WAIT
#define RMD_WALK_MAX_WAIT 6
#endif
constexpr uint32_t kWalkMaxWait = RMD_WALK_MAX_WAIT;
// walks put aside (grid_walk.hpp: cut_lanes = RenderParams::walk_cut): only in calls with at least this many walkers ...
#ifndef RMD_WALK_CUT_MIN_WALKERS
#define RMD_WALK_CUT_MIN_WALKERS 16
#endif
constexpr uint32_t kWalkCutMinWalkers = RMD_WALK_CUT_MIN_WALKERS;
// ... while at least this many lanes of the wave have something else to do
#ifndef RMD_WALK_CUT_MIN_RUNNABLE
#define RMD_WALK_CUT_MIN_RUNNABLE 0
#endif
constexpr uint32_t kWalkCutMinRunnable = RMD_WALK_CUT_MIN_RUNNABLE;

// Occupancy targets (waves per SIMD), measured on MI355X: the grid walk is latency-bound and gains 1.6x from 4 waves/SIMD
// (128 VGPRs, a few dozen spills) over 2; the grid-less kernel is VALU-bound and is fastest at 3 (168 VGPRs).
#ifndef RMD_GRID_MINW
#define RMD_GRID_MINW 4
#endif
#ifndef RMD_NOGRID_MINW
#define RMD_NOGRID_MINW 3
#endif
// MODE: 0 = wave tiles, the lane keeps its pixel's sum (one wave per tile); 1 = wave tiles with the samples of a tile split
// over several waves, every sample's radiance stored to the sample buffer for sum_kernel; 2 = explicit (x, y, sample) list.
// (A template parameter rather than a launch parameter: the buffer mode then carries no accumulator and the direct mode no
// buffer addressing — the grid kernel runs at its register limit.)
enum { kModeTiles = 0, kModeTilesBuffered = 1, kModeList = 2 };
#ifndef RMD_TRIP_RELOAD
#define RMD_TRIP_RELOAD 1
#endif
// A finished sample into its 32-byte sector of the per-sample buffer (spheres kernel).  Plain stores; the end of a work item releases them at
// agent scope (finish_sample_range) — on this part that writes back every dirty line of the XCD's L2 (buffer_wbl2) and is what a work item
// costs at its end: 66.4 ms per C2 frame at 16 items per wave tile against 53.5 at 4, 53.4 / 53.3 with the fence taken out (timing only).
// RMD_SAMPLE_STORE_WT = 1 is the measured alternative: two WRITE-THROUGH stores (sc1: the data goes to memory at once and leaves no dirty line),
// the wave then only waits for its own stores (vmcnt(0)) before it bumps the tile's counter.  Bit-identical (all tests, tools/stress_sum.py), the
// frame time no longer depends on the item size (54.2 ms at 7 .. 16 items per wave tile) — but the full frame is 1.3 % slower than plain stores at
// their best item size (same box: 54.2 vs 53.5 ms; an N = 8 tile share 7.16 vs 7.42 ms): the default stays plain stores + release.
#ifndef RMD_SAMPLE_STORE_WT
#define RMD_SAMPLE_STORE_WT 0
#endif
RMD_DEV void store_sample(RMD_GLOBAL double *dst, V3 L) {
#if RMD_SAMPLE_STORE_WT == 2 // non-temporal stores (what the queued mesh kernel uses: its samples are read by another kernel) — here the samples are read back by a wave of THIS kernel
	__builtin_nontemporal_store(L.x, dst), __builtin_nontemporal_store(L.y, dst + 1), __builtin_nontemporal_store(L.z, dst + 2);
#elif RMD_SAMPLE_STORE_WT
	typedef double d2 __attribute__((ext_vector_type(2)));
	const d2 xy = {L.x, L.y};
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx2 %0, %2, off offset:16 sc1" : : "v"(dst), "v"(xy), "v"(L.z) : "memory");
#else
	dst[0] = L.x, dst[1] = L.y, dst[2] = L.z;
#endif
}
// The end of a (wave tile, sample range) work item of the spheres kernel: the wave that finishes a wave tile's LAST sample range adds the tile's
// samples to the pixels, strictly in sample order (src/trace.rs:203: the reference's sequential sum, bit for bit) — inside the render kernel, where
// the reads (bandwidth) overlap the other waves' arithmetic; as a kernel of its own the sum cost 5.5 ms per 1080p / 500 spp frame.  Release: an
// agent-scope fence writes this wave's sample stores back before its count; acquire: the last wave invalidates its caches before it reads.
RMD_DEV void finish_sample_range(const RenderParams &P, const WaveTile &tile, uint32_t wt, uint32_t lane, double *__restrict__ out) {
	if (P.tile_done == nullptr || wt >= P.n_work) return;
#if RMD_SAMPLE_STORE_WT == 1
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through sample stores have reached memory before its count is seen
#else
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // this wave's samples leave the XCD's L2 before its count is seen
#endif
	uint32_t before = 0;
	if (lane == 0u) before = __hip_atomic_fetch_add(P.tile_done + wt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
	if (before + 1u != P.split_k) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	const uint32_t lx = lane & 7u, ly = lane >> 3;
	if (lx < tile.w && ly < tile.h) {
		const size_t pix = ((size_t)(tile.x0 + lx) + (size_t)(tile.y0 + ly) * P.W) * 3;
		V3 sum = ld3(out + pix);
		const RMD_GLOBAL double *src = (const RMD_GLOBAL double *)P.sample_buf + ((size_t)wt * P.sample_count * 64u + lane) * kSampleStride;
		// 16 samples' loads in flight at a time (the additions stay in sample order)
		uint32_t k = 0;
		for (; k + 16u <= P.sample_count; k += 16u) {
			V3 v[16];
#pragma unroll
			for (uint32_t j = 0; j < 16u; j++) v[j] = ld3(src + (size_t)j * 64u * kSampleStride);
#pragma unroll
			for (uint32_t j = 0; j < 16u; j++) sum = sum + v[j];
			src += 16u * 64u * kSampleStride;
		}
		for (; k < P.sample_count; k++) {
			sum = sum + ld3(src);
			src += 64u * kSampleStride;
		}
		out[pix + 0] = sum.x, out[pix + 1] = sum.y, out[pix + 2] = sum.z;
	}
}

// One wave's share of a launch: list mode — the 64 entries from `first`; tile modes — work item `first` = (wave tile, sample
// sub-range).  Called by all 64 lanes of a wave in uniform control flow; lobjs / lds_masks / wave_lds are the workgroup's staged
// object table and occupancy masks and this wave's scratch in LDS.
typedef const __attribute__((address_space(4))) unsigned long long *KernargWords; // the kernel-argument segment, as 8-byte words
// CHAIN (split launches of scenes with grids, persistent form): the wave does not end with its work item.  When the item's pool of (pixel,
// sample) pairs has run dry it DRAWS THE NEXT ITEM itself and goes on handing out pairs while the last paths of the old item finish — an item
// used to end with a drain of ~7 trips of ever fewer lanes (2 % of a 55-sample item's trips, 20 % of a 4-sample item's: a progressive pass).
// A path's identity — pixel, sample, scratch sector — therefore lives per lane (PathId, in LDS), not in wave-uniform item state.  Which lane
// and which trip compute a sample changes nothing: same bits.
template <int MODE, bool GRID, bool CHAIN = false>
RMD_DEV void render_wave(const RenderParams &P, KernargWords kernarg_params, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                         const void *__restrict__ work, double *__restrict__ out, int32_t *__restrict__ path_obj, uint32_t *__restrict__ path_sub,
                         const DevObject *lobjs, const uint32_t *lds_masks, unsigned char *wave_lds, uint32_t first) {
	static_assert(!CHAIN || (MODE == kModeTilesBuffered && GRID), "items are chained in split launches of scenes with grids");
	constexpr bool LIST = MODE == kModeList;
	const uint32_t lane = threadIdx.x & 63u;
	WalkScratch &scr = *reinterpret_cast<WalkScratch *>(wave_lds); // unused (and not allocated) when the scene has no grid

	constexpr bool to_buffer = MODE == kModeTilesBuffered;
	// Per-lane work.  LIST: one (x, y, sample) entry.  Tiles, direct mode: lane = pixel, samples s .. s_end-1 one after the
	// other, the lane keeps the sum.  Tiles, buffered mode: the wave owns a sample range of its tile and its (pixel, sample)
	// pairs form a pool — item k is pixel slot k % 64 of sample k / 64 — from which a lane whose path has ended takes the
	// next item, so no lane sits out while the longest pixel of the tile finishes (which lane computes a sample has no
	// influence on its value: the RNG is keyed by pixel and sample, and sum_kernel adds the samples in order).
	uint32_t x = 0, y = 0, s = 0, s_end = 0;
	bool alive = false; // the lane has samples left (direct / list) — buffered mode: the lane holds a pool item
	size_t out_index = 0;
	uint32_t list_idx = 0;
	WaveTile tile = {};
	uint32_t wt = 0, pool_first = 0, pool_items = 0, next_item = 0; // wave-uniform (buffered mode)
	uint32_t item = 0;                                               // this lane's pool item (buffered mode without a grid)
	constexpr bool id_in_lds = to_buffer && GRID;                    // ... with a grid: the path's identity in LDS (PathId)
	[[maybe_unused]] PathId *pid = GRID ? reinterpret_cast<PathId *>(wave_lds + sizeof(WalkScratch) + 64u * 3u * sizeof(double) + sizeof(WalkCarry)) : nullptr;
	if (LIST) {
		list_idx = first + lane;
		alive = list_idx < P.n_work;
		ListWork w = reinterpret_cast<const ListWork *>(work)[alive ? list_idx : 0];
		x = w.x, y = w.y, s = w.sample, s_end = w.sample + 1u;
		out_index = (size_t)list_idx * 3;
	} else {
		// work item = (wave tile, sample sub-range): with split_k > 1 the samples of a tile are spread over split_k
		// waves (neighbouring waves, same tile) that store every sample's radiance to the sample buffer; sum_kernel then
		// adds them to the pixel in sample order, so the result is the same sequential sum as with one wave per tile
		const uint32_t work_item = first;
		const uint32_t split = to_buffer ? P.split_k : 1u;
		wt = work_item / split;
		const uint32_t part = work_item % split;
		const bool have = wt < P.n_work;
		tile = reinterpret_cast<const WaveTile *>(work)[have ? wt : 0];
		const uint32_t per_part = (P.sample_count + split - 1u) / split;
		const uint32_t s_lo = part * per_part < P.sample_count ? part * per_part : P.sample_count;
		const uint32_t s_hi = s_lo + per_part < P.sample_count ? s_lo + per_part : P.sample_count;
		if constexpr (to_buffer) {
			pool_first = s_lo;
			pool_items = have ? (s_hi - s_lo) * 64u : 0u;
		} else {
			const uint32_t lx = lane & 7u, ly = lane >> 3;
			alive = have && lx < tile.w && ly < tile.h && s_hi > s_lo;
			x = tile.x0 + lx, y = tile.y0 + ly;
			s = P.sample_begin + s_lo, s_end = P.sample_begin + s_hi;
			out_index = ((size_t)x + (size_t)y * P.W) * 3;
		}
	}
	const bool writes = alive;

	// direct mode of the grid kernel (launches of a few samples per pixel: progressive passes): the pixel's running sum stays in memory and
	// every finished sample is added to it there — the same additions in the same order as a sum kept in registers, which at this kernel's
	// register limit was six spilled registers (a lane owns its pixel for the whole launch: no other lane touches it)
	constexpr bool acc_in_memory = GRID && MODE == kModeTiles;
	V3 acc = mk(0.0, 0.0, 0.0);
	if (!LIST && alive && !to_buffer && !acc_in_memory) acc = ld3(out + out_index);
	if (P.bounce_limit == 0u) { // trace(.., 1) with depth 1 > bounce_limit returns 0 unintersected (:235-237): every sample is (0, 0, 0)
		if (LIST && writes) out[out_index + 0] = 0.0, out[out_index + 1] = 0.0, out[out_index + 2] = 0.0;
		return; // tile launches with bounce_limit 0 are not made at all (api.cpp): the frame is unchanged
	}

	const V3 cam_pos = ld3(P.cam_pos);
	uint32_t rng_block = 0; // index of the sample's next Philox block — with lobe_bits all the RNG state a path carries (the key is its pixel and sample)
	uint32_t lobe_bits = 0; // the 22 spare bits of the path's last block: they decide diffuse against specular at its next shaded depth
	V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
	uint32_t depth = 1; // depth argument of the trace() call being evaluated
	// throughput: product of the bounce weights of the path so far.  It is read and written once per bounce and read when the path ends;
	// the grid kernel (at its register limit: one component was living in scratch) keeps it in LDS behind the wave's walk scratch
	V3 T_reg = mk(1.0, 1.0, 1.0);
	[[maybe_unused]] double *T_lds = GRID ? reinterpret_cast<double *>(wave_lds + sizeof(WalkScratch)) + lane : nullptr;
	auto load_T = [&]() -> V3 {
		if constexpr (GRID) return mk(T_lds[0], T_lds[64], T_lds[128]);
		else return T_reg;
	};
	auto store_T = [&](V3 v) {
		if constexpr (GRID) T_lds[0] = v.x, T_lds[64] = v.y, T_lds[128] = v.z;
		else T_reg = v;
	};
	uint32_t path_len = 0;
	// Every trip of the loop has two halves.  (B) each lane that needs a ray gets one — the bounce ray of the hit its last
	// intersection found (`to_shade`), or the primary ray of the next sample when its path has ended (`need_sample`) — in ONE
	// merged instruction stream (next_ray).  (A) every lane that has a ray intersects it with the scene and classifies the hit.
	bool need_sample = alive || to_buffer; // no path yet
	bool has_ray = false, to_shade = false;
	// the hit a lane will shade on its next trip: object and distance (the hit point and the material are re-derived from them),
	// and the surface normal — in registers, or, in the grid kernel (which runs at its register limit), parked in the wave's
	// walk scratch: that LDS is only in use during a walk, i.e. never between the end of (A) and the next (B)
	int hit_obj = -1;
	double hit_t = 0.0;
	V3 hit_normal = mk(0.0, 0.0, 1.0);
	double *parked = GRID ? reinterpret_cast<double *>(wave_lds) + lane : nullptr; // normal at [0], [64], [128]; distance at [192]
	int32_t *parked_obj = GRID ? reinterpret_cast<int32_t *>(wave_lds + 256u * sizeof(double)) + lane : nullptr;
	static_assert(!GRID || sizeof(WalkScratch) >= 256u * sizeof(double) + 64u * sizeof(int32_t), "the walk scratch holds a wave's parked hits");
	// grid scenes: a ray's closest plane/sphere hit while the lane waits for the walk that settles the grids (see intersect_simple)
	bool new_ray = false, waiting = false;
	bool carried = false; // ... and its walk has begun: put aside by the previous walk call, to be taken up by the next (grid_walk.hpp)
	[[maybe_unused]] WalkCarry *carry = GRID ? reinterpret_cast<WalkCarry *>(wave_lds + sizeof(WalkScratch) + 64u * 3u * sizeof(double)) : nullptr;
	// wave-uniform, ONE scalar register for two counters (the mesh kernel runs at its register limit: a 64-bit trip count of its own cost 48
	// more spilled registers): low byte — trips since the wave's last walk call; the rest — the STALL WATCH, trips since a lane of the wave last
	// finished a sample or was handed a pair.  A path ends after at most RMD_MAX_BOUNCE_LIMIT segments; a segment waits at most kWalkMaxWait
	// trips for its walk and a walk is put aside at most once per step it still has to take (only calls with >= 16 walkers put walks aside and
	// each advances every walker: < 256 times): 16 x 256 x 7 < 2^15 trips is the longest any wave can go without one of its lanes finishing a
	// sample.  kStallBound trips without one (2^18: a few seconds) is a fault (report_fault), whatever the cause.
	uint32_t trips_since_walk = 0;
	uint32_t stall_bound = kStallBound << 8;
#if RMD_DIAG
	if (P.debug_flags & 32u) stall_bound = 2u << 8; // tests/test_gpu_faults.py: forces the bound
#endif
	double part_t = kFMax;
	int part_obj = -1;
	uint32_t part_sub = 0;

	// carried from a trip's intersection phase (A) and ray phase (B) to the next trip's classification (C)
	bool complete = false, lens_failed = false, cut = false;
	double t = 0.0;
	uint32_t sub = 0;
	int oi = -1;
	// Wave-uniform main loop: all 64 lanes stay in it until every lane has finished its samples, so that finished
	// lanes still lend their ALUs to the cooperative grid walk.  Per-lane work is predicated.
#if RMD_DIAG
	const bool tstamp = (P.debug_flags & 16u) && P.debug_counters; // where a wave's time goes, trip by trip (perturbs the overlap of loads)
	unsigned long long tt_prev = tstamp ? __builtin_amdgcn_s_memtime() : 0ull, tt_b = 0, tt_simple = 0, tt_walk = 0, tt_class = 0;
	const unsigned long long tt_begin = tt_prev;
#define RMD_TSTAMP(acc)                                         \
	if (tstamp) {                                               \
		__builtin_amdgcn_s_waitcnt(0);                          \
		unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		acc += now_ - tt_prev;                                  \
		tt_prev = now_;                                         \
	}
#else
#define RMD_TSTAMP(acc)
#endif
	for (;;) {
		// The launch parameters a trip needs are read again from the kernel arguments (a few scalar loads per trip) instead of being
		// carried in scalar registers across trips — the grid kernel was spilling scalars into vector lanes all through the trip (168
		// v_readlane / v_writelane at trip level, 41 now; 496.4 -> 487.3 ms on C3), the spheres kernel less so (112.2 -> 110.6 ms on
		// C2).  The empty asm keeps the compiler from hoisting the loads out of the loop.
		auto trip_params = [&]() -> decltype(auto) {
			if constexpr (RMD_TRIP_RELOAD) {
				// `kernarg_params`: where the calling kernel's RenderParams argument lies in its kernel-argument segment (the kernel, which knows its
				// own signature, passes it: KernargWords below)
				KernargWords src = kernarg_params;
				asm volatile("" : "+s"(src));
				static_assert(sizeof(RenderParams) % 8 == 0, "copied in 8-byte words");
				unsigned long long w[sizeof(RenderParams) / 8];
#pragma unroll
				for (unsigned i = 0; i < sizeof(RenderParams) / 8; i++) w[i] = src[i];
				RenderParams copy;
				__builtin_memcpy(&copy, w, sizeof(copy));
				return copy; // by value; only the fields a trip uses are loaded
			} else {
				return (P);
			}
		};
		decltype(auto) Pt = trip_params();
		if constexpr (CHAIN) {
			// the item's pool has run dry: the wave draws its next work item (the persistent work loop's draw, render_kernel below, with the same
			// bound: a draw must be larger than the wave's last).  pool_items = 0xFFFFFFFF marks "the launch has no item left".
			if (next_item >= pool_items && pool_items != 0xFFFFFFFFu) {
				volatile uint32_t *last_draw = reinterpret_cast<volatile uint32_t *>(wave_lds + wave_lds_bytes(1u) - kWaveHeadBytes);
				uint32_t drawn = 0, floor = 0;
				if (lane == 0u) {
					drawn = atomicAdd(Pt.work_counter, 1u);
					floor = *last_draw;
					*last_draw = drawn + 1u;
				}
				drawn = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
				floor = (uint32_t)__builtin_amdgcn_readfirstlane((int)floor);
#if RMD_DIAG
				if ((Pt.debug_flags & 128u) && floor != 0u) floor = 0xFFFFFFFFu; // tests/test_gpu_faults.py: forces the bound at a wave's second draw
#endif
				if (drawn >= Pt.n_work * Pt.split_k) {
					pool_items = 0xFFFFFFFFu, next_item = 0xFFFFFFFFu;
				} else {
#if RMD_BOUND_DRAWS
					if (RMD_UNLIKELY(drawn < floor)) report_fault(Pt, kFaultWorkLoop, drawn); // (never reached; the poisoned counter ends the launch)
#endif
					wt = drawn / Pt.split_k;
					const uint32_t part = drawn - wt * Pt.split_k;
					tile = reinterpret_cast<const WaveTile *>(work)[wt];
					const uint32_t per_part = (Pt.sample_count + Pt.split_k - 1u) / Pt.split_k;
					const uint32_t s_lo = part * per_part < Pt.sample_count ? part * per_part : Pt.sample_count;
					const uint32_t s_hi = s_lo + per_part < Pt.sample_count ? s_lo + per_part : Pt.sample_count;
					pool_first = s_lo, pool_items = (s_hi - s_lo) * 64u, next_item = 0u;
				}
			}
		}
#if RMD_BOUND_TRIPS
		// (no `break` here: a second way out of this loop cost the mesh kernel 40 spilled registers.  A stalled wave reports, drops every path
		// and pair it holds and leaves through the loop's own exit below.)
		trips_since_walk += 0x100u;
		const bool stalled = trips_since_walk >= stall_bound; // (never true: see trips_since_walk)
		if (RMD_UNLIKELY(stalled)) {
			report_fault(Pt, kFaultTripLoop, first);
			complete = false, lens_failed = false, cut = false, has_ray = false, to_shade = false, need_sample = false, alive = false;
			if constexpr (CHAIN) pool_items = 0xFFFFFFFFu, next_item = 0xFFFFFFFFu;
			else if constexpr (to_buffer) next_item = pool_items;
			else s = s_end;
		}
#endif
		// ---------------- (C) the hits the previous trip found (`complete`): miss, emission, or a surface to shade.  The loop is entered here: a
		// trip is (C) classification -> (B) hand-out and next rays -> (A) intersection; a surface classified here is shaded a few lines
		// further down, with nothing but the hand-out in between (the mesh kernel: two spilled registers instead of four).
		bool terminal = lens_failed || cut;
		bool emitted = false; // the path has reached a light (:250-252): its radiance is fetched below, outside the nest of branches (a value that is
		                      // assigned three branches deep costs a copy per level and component on every trip)
		if (complete) {
			if (LIST && path_obj) {
				size_t pi = (size_t)list_idx * (RMD_PATH_STRIDE) + path_len;
				path_obj[pi] = oi;
				path_sub[pi] = oi >= 0 ? sub : 0u;
				path_len++;
			}
			if (oi < 0) {
				terminal = true; // :242 miss -> radiance 0
			} else {
				const DevObject &o = lobjs[oi];
				const V3 frag = ro + rd * t; // :246
				if (o.material_kind == 2u) {
					emitted = true; // :250-252 Emission
					terminal = true;
				} else {
					V3 normal;
					if (o.geometry_kind == 0u) normal = ld3(o.normal);                        // plane.rs:28-32
					else if (o.geometry_kind == 1u) normal = normalize(frag - ld3(o.origin)); // sphere.rs:31-35
					else if constexpr (GRID) {
						const DevGrid &g = grids[o.grid_index];
						normal = triangle_normal(as_global(g.tri_pos) + (size_t)sub * 9, as_global(g.tri_nrm) + (size_t)sub * 9, as_global(g.tri_aux) + (size_t)sub * 4, frag); // acc_grid.rs:85-87
					} else {
						normal = mk(0.0, 0.0, 0.0); // unreachable: a scene with grid objects runs the GRID instantiation
					}
					// At the bounce limit the recursive call returns 0 at once (:235-237) and this depth's result is its weight times
					// that zero (:281-282 / :315-318): exactly zero whenever the weight is finite, so the shading is not evaluated.
					// In a scene of regular parameters (api.cpp: rmd_scene::regular) a weight is non-finite only through a non-finite input — the
					// Heron normal of a degenerate hit, a hit point at infinity — (then the reference's sample is NaN, and so is this one: the
					// lane shades and multiplies by zero on its next trip), or through r1 = 0 exactly in the diffuse pdf (probability 2^-53 per
					// path; there the reference returns NaN and this kernel 0).  Outside that class — a NaN colour, roughness 0 (0 / 0 in
					// geometry_schlick_ggx for a surface seen from behind) — the last depth is shaded like any other (shade_last_depth).
					const double probe_sum = ((normal.x + normal.y) + normal.z) + ((frag.x + frag.y) + frag.z);
					const bool finite_inputs = __builtin_fabs(probe_sum) < __builtin_inf();
					// ... and so is a diffuse bounce off a black surface (o.flags: Diffuse, colour (0, 0, 0)): its weight (1 - F)(1 - metal) (.) colour
					// (:279-281) is exactly zero for finite inputs, so the path ends here with the sample the reference computes — zero — see the
					// throughput rule below.  Whether the bounce is the diffuse one (`r < prob_d`, :263-264, prob_d = 0.5 for Diffuse) is known
					// now: r is the 22-bit uniform of the path's PREVIOUS block.  The lane takes its next sample on this very trip instead of
					// shading, finding its throughput zero and sitting out the intersection phase.
					const bool black_bounce = !LIST && Pt.end_black_paths != 0u && (o.flags & kObjBlackDiffuse) != 0u && lobe_bits < (1u << 21);
					if (((depth == Pt.bounce_limit && Pt.shade_last_depth == 0u) || black_bounce) && finite_inputs) {
						terminal = true; // L = 0
					} else {
						if constexpr (GRID) parked[0] = normal.x, parked[64] = normal.y, parked[128] = normal.z, parked[192] = t, parked_obj[0] = oi;
						else hit_normal = normal, hit_obj = oi, hit_t = t;
						to_shade = true;
					}
				}
			}
			has_ray = false;
		}
		if (terminal) {
			V3 L = mk(0.0, 0.0, 0.0); // :242 miss, bounce limit, a black path ended: radiance 0 (times the throughput: a non-finite throughput makes NaN of it, as in the reference)
			if (emitted) L = ld3(lobjs[oi].color);
			L = hadamard(load_T(), L);
			if constexpr (to_buffer) {
				// one aligned 32-byte sector per sample (kSampleStride doubles): lanes finish their samples on different trips, so a
				// sample's store travels alone, and a 24-byte store that straddles sectors was costing 2.7x its size in L2 write-backs
				RMD_GLOBAL double *dst;
				if constexpr (id_in_lds) dst = (RMD_GLOBAL double *)Pt.sample_buf + (size_t)pid->sector[lane] * kSampleStride;
				else dst = (RMD_GLOBAL double *)Pt.sample_buf + (((size_t)wt * Pt.sample_count + pool_first + (item >> 6)) * 64u + (item & 63u)) * kSampleStride;
				// plain stores: when a wave of this kernel adds the tile's samples (below) — it may run on another XCD, whose L2 does not see this
				// one's dirty lines — the release in front of the tile's counter writes them back, once per work item (round 2 wrote every
				// sample through with three 8-byte agent-scope stores: 96 bytes at the memory side per 24-byte sample)
				if constexpr (GRID) dst[0] = L.x, dst[1] = L.y, dst[2] = L.z; // (added by sum_kernel, behind the kernel boundary)
				else store_sample(dst, L);                                  // (added by a wave of this kernel: finish_sample_range)
			} else {
				if constexpr (acc_in_memory) {
					RMD_GLOBAL double *px = (RMD_GLOBAL double *)out + ((size_t)(tile.x0 + (lane & 7u)) + (size_t)(tile.y0 + (lane >> 3)) * Pt.W) * 3;
					px[0] += L.x, px[1] += L.y, px[2] += L.z; // src/trace.rs:203
				} else {
					acc = acc + L; // src/trace.rs:203
				}
				s++;
			}
			has_ray = false;
			need_sample = true;
		}
		if (__ballot(terminal) != 0ull) trips_since_walk &= 0xFFu; // a lane has finished a sample: the stall watch starts again
		RMD_TSTAMP(tt_class)
		// ---------------- (B) hand out samples, then rays
		bool prim = false;
		if constexpr (to_buffer) {
			// the next pool items go to the lanes whose path has ended
			const unsigned long long idle = __ballot(need_sample);
			if (idle != 0ull && next_item < pool_items) {
				const uint32_t k = next_item + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
				next_item += (uint32_t)__popcll(idle);
				trips_since_walk &= 0xFFu; // pairs handed out: the stall watch starts again
				if (need_sample && k < pool_items && (k & 7u) < tile.w && ((k >> 3) & 7u) < tile.h) { // slots outside a ragged tile are skipped
					prim = true;
					if constexpr (id_in_lds) { // the path's identity: pixel, sample, scratch sector of the pair it was handed
						pid->px[lane] = (uint32_t)(tile.x0 + (k & 7u)) | (uint32_t)(tile.y0 + ((k >> 3) & 7u)) << 16;
						pid->smp[lane] = Pt.sample_begin + pool_first + (k >> 6);
						pid->sector[lane] = (wt * Pt.sample_count + pool_first + (k >> 6)) * 64u + (k & 63u); // (< 2^32: api.cpp caps a pass)
					} else {
						item = k;
					}
				}
			}
			alive = prim || has_ray || to_shade;
			if (__ballot(alive) == 0ull) {
				if (next_item >= pool_items && (!CHAIN || pool_items == 0xFFFFFFFFu)) break;
				complete = false, cut = false, lens_failed = false; // (C) has consumed them: the next pass through it must not classify a hit twice
				continue; // a handout that fell entirely on slots outside the tile
			}
		} else {
			if (need_sample) {
				if (s != s_end) prim = true; // src/trace.rs:199 — primary ray of sample s
				else alive = false;
			}
			if (__ballot(alive) == 0ull) break;
		}
		if (prim) {
			need_sample = false;
			rng_block = 0;
			depth = 1;
			has_ray = true, new_ray = true;
		}
		// pixel and sample of the lane's path: in a split launch they are functions of the pool item (one integer of state per
		// lane; kept as x, y, s they were spilled to scratch at every hand-out), otherwise the lane's own
		if constexpr (id_in_lds) {
			const uint32_t pxv = pid->px[lane];
			x = pxv & 0xFFFFu, y = pxv >> 16, s = pid->smp[lane];
		} else if constexpr (to_buffer) {
			x = tile.x0 + (item & 7u), y = tile.y0 + ((item >> 3) & 7u);
			s = Pt.sample_begin + pool_first + (item >> 6);
		} else if constexpr (acc_in_memory) {
			x = tile.x0 + (lane & 7u), y = tile.y0 + (lane >> 3); // recomputed per trip instead of carried (registers)
		}
		Rng rng;
		rng.pixel = y * Pt.W + x, rng.sample = s, rng.block = rng_block, rng.lobe_bits = lobe_bits;
		lens_failed = false;
		// the shading inputs of the hit a lane carries — evaluated for every lane, used by next_ray for the lanes that shade (a lane
		// without a hit reads object 0 and whatever its slots hold: cheaper than nine register moves of stand-in values per trip)
		NextRayShadeIn hit;
		if constexpr (GRID) hit_normal = mk(parked[0], parked[64], parked[128]), hit_t = parked[192], hit_obj = parked_obj[0];
		{
			const DevObject &o = lobjs[to_shade ? hit_obj : 0];
			hit.normal = hit_normal;
			hit.frag = ro + rd * hit_t; // :246, the same operations as at classification
			hit.color = ld3(o.color), hit.roughness = o.roughness, hit.metal = o.metalness;
		}
		[[maybe_unused]] bool black = false; // the path's throughput has become exactly (0, 0, 0)
		if constexpr (GRID) {
			V3 T = mk(1.0, 1.0, 1.0);
			if (to_shade) T = load_T();
			next_ray(Pt, to_shade, prim, hit, cam_pos, x, y, rng, ro, rd, T);
			if (to_shade || prim) store_T(T);
			black = Pt.end_black_paths != 0u && T.x == 0.0 && T.y == 0.0 && T.z == 0.0;
		} else {
			next_ray(Pt, to_shade, prim, hit, cam_pos, x, y, rng, ro, rd, T_reg);
			black = Pt.end_black_paths != 0u && T_reg.x == 0.0 && T_reg.y == 0.0 && T_reg.z == 0.0;
		}
		if (Pt.use_dof) { // thin lens (:335-360): the pinhole ray of :336 has just come out of the merged stream (jitter block included); the rejection
			// loop's blocks, the focal plane and the new direction follow for the lanes that start a sample
			if (prim) lens_failed = !thin_lens_from_pinhole(Pt, ro, rd, rng, ro, rd); // the reference panics there; the sample contributes zero
		}
		rng_block = rng.block, lobe_bits = rng.lobe_bits;
		cut = false; // shaded at the bounce limit (non-finite inputs, see below): the recursive call returns 0 unintersected (:235-237)
		if (to_shade) {
			depth++;
			to_shade = false;
			// A path whose throughput is exactly zero in every channel — a diffuse bounce off a black surface ((1 - F)(1 - metal) (.) (0, 0, 0),
			// :279-281), a GGX sample below the surface (geometry_smith's max(n.l, 0), :373) — is ended here: whatever its remaining segments
			// find is multiplied by that zero on the way back up trace()'s recursion (:281-282, :315-318), so the sample is exactly zero in
			// the reference as well — unless a later vertex produces a non-finite radiance (0 x NaN = NaN).  A scene of planes and spheres
			// cannot (its one source, r1 = 0 in a diffuse pdf, is the 2^-53 case of DESIGN.md section 3); on a mesh the Heron normal of a
			// degenerate hit can (~1e-9 per sample on the benchmark mesh): there the reference's sample is NaN and this one 0, unless the
			// caller sets RMD_RENDER_TRACE_BLACK_PATHS, which keeps tracing such paths.  Three of the reference scenes' six walls are
			// black: a quarter of all path segments (a third with the mesh) belong to paths that can no longer contribute.  The explicit
			// (x, y, sample) probes always trace them — they record the reference's hit sequence.
			if (depth > Pt.bounce_limit || (!LIST && black)) cut = true, has_ray = false;
			else has_ray = true, new_ray = true;
		}
#if RMD_DIAG
		if ((Pt.debug_flags & 8u) && Pt.debug_counters) { // main-loop occupancy: trips, live lanes, lanes with a ray
			const unsigned long long am = __ballot(alive), wm = __ballot(has_ray && !lens_failed);
			if (lane == 0) atomicAdd(&Pt.debug_counters[10], 1ull), atomicAdd(&Pt.debug_counters[11], (unsigned long long)__popcll(am)), atomicAdd(&Pt.debug_counters[12], (unsigned long long)__popcll(wm));
		}
#endif
		RMD_TSTAMP(tt_b)
		// ---------------- (A) src/trace.rs:239 — closest hit of every lane that has a ray
		const bool want = has_ray && !lens_failed;
		if constexpr (GRID) {
			if (want && new_ray) {
				waiting = intersect_simple(objs, Pt.n_objects, grids, true, ro, rd, part_t, part_obj, Pt.axis_pairs, Pt.visit_mask);
				part_sub = 0u, new_ray = false;
			}
			RMD_TSTAMP(tt_simple)
			// run the grid walks when enough lanes wait for one, or when no lane of the wave could do anything else
			const unsigned long long wm = __ballot(want && waiting), rm = __ballot(alive && !(want && waiting));
			if ((trips_since_walk & 0xFFu) != 0xFFu) trips_since_walk++; // (saturating: the byte's neighbours are the stall watch)
			if (wm != 0ull && ((uint32_t)__popcll(wm) >= Pt.walk_batch || (uint32_t)__popcll(rm) < kWalkMinRunnable || (trips_since_walk & 0xFFu) >= kWalkMaxWait)) {
				trips_since_walk &= ~0xFFu;
				// walks are put aside only by a call with many walkers in a wave that has other lanes to run: otherwise every walk is finished
				if constexpr (MODE != kModeTiles) {
					const uint32_t n_walkers = (uint32_t)__popcll(wm);
					const bool cut = n_walkers >= kWalkCutMinWalkers && (uint32_t)__popcll(rm) >= kWalkCutMinRunnable;
					const uint32_t cut_lanes = cut ? Pt.walk_cut & 0xffu : 0u, cut_round = cut ? (Pt.walk_cut >> 8) & 0xffu : 0u;
					intersect_grids<true>(objs, Pt.n_objects, grids, lds_masks, scr, want && waiting, ro, rd, part_t, part_obj, part_sub, Pt.debug_flags, Pt.debug_counters,
					                      cut_lanes, carry, &carried, cut_round, Pt.grid_mask);
					waiting = carried;
				} else { // direct mode (launches of a few samples per pixel): every call finishes its walks — the carry's registers are not worth it there
					intersect_grids<false>(objs, Pt.n_objects, grids, lds_masks, scr, want && waiting, ro, rd, part_t, part_obj, part_sub, Pt.debug_flags, Pt.debug_counters, 0u, nullptr, nullptr, 0u,
					                       Pt.grid_mask);
					waiting = false;
				}
			}
			RMD_TSTAMP(tt_walk)
			complete = want && !waiting;
			t = part_t, oi = part_obj, sub = part_sub;
		} else {
			oi = scene_intersect_wave<false>(objs, Pt.n_objects, grids, lds_masks, scr, want, ro, rd, t, sub, Pt.axis_pairs, Pt.debug_flags, Pt.debug_counters, false, Pt.visit_mask);
			complete = want;
		}
	}

#if RMD_DIAG
	if (tstamp && lane == 0) {
		atomicAdd(&P.debug_counters[0], __builtin_amdgcn_s_memtime() - tt_begin), atomicAdd(&P.debug_counters[1], tt_b);
		atomicAdd(&P.debug_counters[2], tt_simple), atomicAdd(&P.debug_counters[3], tt_walk), atomicAdd(&P.debug_counters[4], tt_class);
	}
#endif
#undef RMD_TSTAMP
	if constexpr (to_buffer && !GRID) finish_sample_range(P, tile, wt, lane, out); // (mesh scenes keep sum_kernel: api.cpp)
	if (writes && !to_buffer && !acc_in_memory) {
		out[out_index + 0] = acc.x;
		out[out_index + 1] = acc.y;
		out[out_index + 2] = acc.z;
	}
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Split launches of scenes WITHOUT grids (the spheres kernel): trips sorted by role.
//
// In render_wave() a lane keeps its path and every trip runs ONE merged instruction stream for the two things a lane may need — the
// shading of the hit it carries, or the primary ray of its next sample.  With zero-throughput paths ended a sample is 2.13 path segments,
// so 47 % of a trip's lanes start a sample and sit out the shading stream (331 of the trip's ~940 vector instructions) while the other
// 53 % sit out ray generation: 68 % lane utilisation, measured.  Here the wave owns a STACK of up to kSortSlots parked hits in LDS and a trip is
// one of two kinds, each with (up to) 64 lanes that all need the same thing:
//   GI  the work item's next 64 (pixel, sample) pairs, one per lane: primary ray (src/trace.rs:322-333, thin lens :335-360), intersection,
//       classification;
//   SI  the top 64 hits of the stack: shading (:256-319) -> bounce ray, intersection, classification.
// Classification (:242-252 and the rules of DESIGN.md section 3) ends a path — its sample goes to the per-sample buffer — or parks the hit
// (hit point, normal, throughput, RNG state: 9 doubles + 3 words) on the stack.  A path's arithmetic is the same functions in the same order
// as in render_wave(): every sample has the same bits; which lane and which trip compute it changes nothing (the RNG is keyed by pixel and
// sample, the samples are added in order afterwards).  A generation trip runs while the stack has room for the 64 hits it may park, so a
// shading trip finds more than kSortSlots - 64 hits (57 .. 64 lanes at 120 entries) until the item runs out of pairs; the remaining hits are
// then shaded in ever smaller trips.
#ifndef RMD_SORT_OBJ_PRIO
#define RMD_SORT_OBJ_PRIO 1 // s_setprio level of the closest-hit loop over the objects in the role-sorted spheres kernel (0 = not raised)
#endif
constexpr int kSortObjPrio = RMD_SORT_OBJ_PRIO;
#ifndef RMD_SORTED_TRIPS
#define RMD_SORTED_TRIPS 1
#endif
constexpr uint32_t kSortSlots = RMD_SORT_SLOTS; // (launch.hpp)
// The wave's parked hits: a dense STACK in LDS (round 5; round 4 kept a path in a fixed slot of a pool and two lists of slot numbers).  A hit is
// pushed where the stack ends — the lanes of a trip that park write to consecutive entries — and a shading trip pops the top 64, lane i entry
// n_hit - 64 + i: every access of a trip is 64 consecutive 8-byte (or 4-byte) words, the one pattern the LDS serves without a bank conflict
// (53 % of the LDS-active cycles of round 4's pool were conflicts: a trip's slot numbers were arbitrary), there are no slot lists to read and
// write, and a lane that starts a sample needs no slot at all.
struct alignas(16) HitStack {
	double frag[3][kSortSlots], normal[3][kSortSlots], T[3][kSortSlots]; // the parked hit: point, surface normal; the path's throughput
	uint32_t state[kSortSlots];     // object (16 bits) | next RNG block (16)
	uint32_t lobe_bits[kSortSlots]; // the 22 spare bits of the path's last block | depth << 24
	uint32_t item[kSortSlots];      // the path's (pixel, sample) pair: its number in the work item's pool
};
static_assert(kSortSlots >= 72u && kSortSlots <= 512u && kSortSlots % 8u == 0u, "a generation trip needs room for 64 more hits");
static_assert(sizeof(HitStack) == kSortPoolBytes, "launch.hpp: kSortPoolBytes");
using SortPool = HitStack;
// waves of a persistent workgroup of this kernel: 16 stacks + the object table fit the CU's 160 KB
constexpr uint32_t kSortedWavesPerWg = RMD_SORT_WAVES;

// A register pair the compiler may fill with anything: the value of a variable in the lanes that never use it.
#define RMD_UNDEF3(v) RMD_UNDEF(v.x) RMD_UNDEF(v.y) RMD_UNDEF(v.z)
RMD_DEV void render_wave_sorted(const RenderParams &P, KernargWords kernarg_params, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                                const void *__restrict__ work, double *__restrict__ out, const DevObject *lobjs, unsigned char *wave_lds, uint32_t work_item) {
	const uint32_t lane = threadIdx.x & 63u;
	HitStack &stack = *reinterpret_cast<HitStack *>(wave_lds);
	WalkScratch *no_scratch = nullptr; // (scene_intersect_wave<false> never touches it)
	const uint32_t wt = work_item / P.split_k, part = work_item % P.split_k;
	const bool have = wt < P.n_work;
	const WaveTile tile = reinterpret_cast<const WaveTile *>(work)[have ? wt : 0];
	const uint32_t per_part = (P.sample_count + P.split_k - 1u) / P.split_k;
	const uint32_t s_lo = part * per_part < P.sample_count ? part * per_part : P.sample_count;
	const uint32_t s_hi = s_lo + per_part < P.sample_count ? s_lo + per_part : P.sample_count;
	const uint32_t pool_first = s_lo, pool_items = have ? (s_hi - s_lo) * 64u : 0u;
	if (P.bounce_limit == 0u) return; // (such launches are not made: api.cpp)
	const V3 cam_pos = ld3(P.cam_pos);

	uint32_t n_hit = 0, next_item = 0; // wave-uniform: entries on the stack, pairs handed out
	// the trip loop's bound (report_fault): every trip hands out 64 pairs or advances at least one path by a segment
	unsigned long long trips_left = (unsigned long long)pool_items * kTripBoundPerPair + 64ull;
#if RMD_DIAG
	if (P.debug_flags & 32u) trips_left = 1ull; // tests/test_gpu_faults.py: forces the bound
#endif
	for (;;) {
		// the launch parameters a trip needs, re-read from the kernel arguments (see render_wave)
#ifndef RMD_SORT_RELOAD
#define RMD_SORT_RELOAD 1
#endif
#if RMD_SORT_RELOAD
		KernargWords src = kernarg_params;
		asm volatile("" : "+s"(src));
		unsigned long long w[sizeof(RenderParams) / 8];
#pragma unroll
		for (unsigned i = 0; i < sizeof(RenderParams) / 8; i++) w[i] = src[i];
		RenderParams Pt;
		__builtin_memcpy(&Pt, w, sizeof(Pt));
#else
		const RenderParams &Pt = P;
#endif

		// Which kind of trip.  A generation trip (64 new samples) needs room for the 64 hits it may park: it runs while the item has pairs and
		// the stack holds at most kSortSlots - 64 hits; else the top min(n_hit, 64) hits are shaded — more than kSortSlots - 64 of them (57 .. 64
		// lanes at 120 entries) unless the item has run out of pairs, when what is left is shaded in ever smaller trips.  (RMD_SORT_FULL_SI = 1
		// is the other way round — shading trips only when 64 hits wait, generation trips of min(64, kSortSlots - n_hit) pairs: measured 0.4 %
		// slower on C2, 0.8 % with every path traced.)
		// Every trip makes progress, for every stack size the static_assert admits: a generation trip hands out 64 pairs (kSortSlots - n_hit >= 9
		// with RMD_SORT_FULL_SI) — next_item grows; a shading trip runs when no pair is left (n_hit > 0 then, else the loop has ended) or n_hit >
		// kSortSlots - 64 >= 8 (>= 64), so at least one path advances by a segment, and a path has at most bounce_limit of them.  There is no
		// state in which a trip runs with no lane —
		// unlike a pool with THREE lists, where all three can be short of a full trip while the empty list holds nothing
		// (tools/experiments/README.md: the run that was killed for silence in round 4).
		if (RMD_UNLIKELY(trips_left-- == 0ull)) { // (never reached: see above) — the wave reports, drops what it holds and leaves through the loop's own exit
			report_fault(Pt, kFaultSortedTripLoop, work_item);
			n_hit = 0u, next_item = pool_items;
		}
		const bool items_left = next_item < pool_items;
		if (n_hit == 0u && !items_left) break;
#ifndef RMD_SORT_FULL_SI
#define RMD_SORT_FULL_SI 0
#endif
		const bool shade_trip = !items_left || (RMD_SORT_FULL_SI ? n_hit >= 64u : n_hit > kSortSlots - 64u);
		bool active;
		uint32_t item = 0, depth = 1;
		Rng rng;
		rng.block = 0u, rng.lobe_bits = 0u;
		// (no stand-in values for the lanes that sit a trip out: what such a lane holds is never stored, and a stand-in is a move per register and
		// a select where the branches meet — 18 doubles a trip)
		V3 ro, rd, T;
		RMD_UNDEF3(ro) RMD_UNDEF3(rd) RMD_UNDEF3(T)
		bool failed = false; // lens_failed / cut: the path ends with the sample T (.) 0
		if (shade_trip) {
			// ---------------- SI: the top (up to) 64 parked hits, lane i the entry n_hit - n + i
			const uint32_t n = n_hit < 64u ? n_hit : 64u;
			active = lane < n;
			const uint32_t e = active ? n_hit - n + lane : 0u;
			n_hit -= n;
			const uint32_t st = stack.state[e];
			item = stack.item[e];
			const uint32_t lb = stack.lobe_bits[e];
			rng.block = st >> 16, rng.lobe_bits = lb & 0x3FFFFFu;
			depth = lb >> 24;
			const DevObject &o = lobjs[active ? (st & 0xFFFFu) : 0u];
			T = mk(stack.T[0][e], stack.T[1][e], stack.T[2][e]);
			const V3 normal = mk(stack.normal[0][e], stack.normal[1][e], stack.normal[2][e]);
			const V3 frag = mk(stack.frag[0][e], stack.frag[1][e], stack.frag[2][e]);
			const uint32_t x = tile.x0 + (item & 7u), y = tile.y0 + ((item >> 3) & 7u);
			rng.pixel = y * Pt.W + x, rng.sample = Pt.sample_begin + pool_first + (item >> 6);
#if RMD_SORT_PREDICATED_ARMS
			if (active) {
#else
			{ // every lane shades — a lane beyond the trip's n the stack's entry 0, a hit of this work item like any other; what it computes is never
			  // stored (`active` gates everything below).  Shading under `if (active)` made ro, rd and T values that are assigned inside a divergent
			  // branch: nine 64-bit copies of the other lanes' undefined values per trip.
#endif
				shade(Pt, normal, frag, ld3(o.color), o.roughness, o.metalness, cam_pos, rng, ro, rd, T);
				depth++;
				// (see render_wave: a path whose throughput is exactly zero is ended unless the caller traces such paths on)
				const bool black = Pt.end_black_paths != 0u && T.x == 0.0 && T.y == 0.0 && T.z == 0.0;
				failed = active && (depth > Pt.bounce_limit || black);
			}
		} else {
			// ---------------- GI: the work item's next 64 (pixel, sample) pairs, one per lane (slots outside a ragged tile are skipped)
			const uint32_t room = kSortSlots - n_hit, n = RMD_SORT_FULL_SI && room < 64u ? room : 64u;
			item = next_item + lane;
			next_item += n;
			active = lane < n && item < pool_items && (item & 7u) < tile.w && ((item >> 3) & 7u) < tile.h;
			const uint32_t x = tile.x0 + (item & 7u), y = tile.y0 + ((item >> 3) & 7u);
			rng.pixel = y * Pt.W + x, rng.sample = Pt.sample_begin + pool_first + (item >> 6);
			T = mk(1.0, 1.0, 1.0);
#if RMD_SORT_PREDICATED_ARMS
			if (active) {
#else
			{ // (likewise: a lane without a pair computes the ray of a pixel position outside the tile, which nobody reads)
#endif
				double u0, u1;
				rng.next2(Pt.key0, Pt.key1, u0, u1); // block 0: the pixel jitter (:326-327)
				primary_ray(Pt, x, y, u0, u1, ro, rd);
				if (Pt.use_dof) failed = active && !thin_lens_from_pinhole(Pt, ro, rd, rng, ro, rd); // the reference panics there; the sample contributes zero
			}
		}
		// ---------------- src/trace.rs:239 — closest hit of every lane that has a ray
		const bool want = active && !failed;
		double t = 0.0;
		uint32_t sub = 0;
		// (the object loop is a chain of scalar loads with a few vector instructions behind each: at a raised priority it is through sooner and the SIMD's
		// other waves fill what it leaves with their shading — RMD_SORT_OBJ_PRIO, measured −1.7 %)
		if constexpr (kSortObjPrio != 0) __builtin_amdgcn_s_setprio(kSortObjPrio);
		const int oi = scene_intersect_wave<false>(objs, Pt.n_objects, grids, nullptr, *no_scratch, want, ro, rd, t, sub, Pt.axis_pairs, 0u, nullptr, false, Pt.visit_mask);
		if constexpr (kSortObjPrio != 0) __builtin_amdgcn_s_setprio(0);
		// ---------------- classification (the rules of render_wave's phase C)
		bool terminal = failed, park = false, emitted = false;
		V3 frag, normal;
		RMD_UNDEF3(frag) RMD_UNDEF3(normal)
		if (want) {
			if (oi < 0) {
				terminal = true; // :242 miss -> radiance 0
			} else {
				const DevObject &o = lobjs[oi];
				frag = ro + rd * t; // :246
				if (o.material_kind == 2u) {
					emitted = true; // :250-252 Emission
					terminal = true;
				} else {
					if (o.geometry_kind == 0u) normal = ld3(o.normal);       // plane.rs:28-32
					else normal = normalize(frag - ld3(o.origin));            // sphere.rs:31-35
					const double probe_sum = ((normal.x + normal.y) + normal.z) + ((frag.x + frag.y) + frag.z);
					const bool finite_inputs = __builtin_fabs(probe_sum) < __builtin_inf();
					const bool black_bounce = Pt.end_black_paths != 0u && (o.flags & kObjBlackDiffuse) != 0u && rng.lobe_bits < (1u << 21);
					if (((depth == Pt.bounce_limit && Pt.shade_last_depth == 0u) || black_bounce) && finite_inputs) terminal = true; // L = 0
					else park = true;
				}
			}
		}
		if (active && terminal) { // the finished sample: T (.) L into its 32-byte sector of the per-sample buffer
			V3 L = mk(0.0, 0.0, 0.0);
			if (emitted) L = ld3(lobjs[oi].color); // (fetched here, outside the nest of branches that found the light)
			L = hadamard(T, L);
			RMD_GLOBAL double *dst = (RMD_GLOBAL double *)Pt.sample_buf + (((size_t)wt * Pt.sample_count + pool_first + (item >> 6)) * 64u + (item & 63u)) * kSampleStride;
			store_sample(dst, L);
		}
		// the hits that go on: pushed onto the stack, consecutive entries for the lanes that park
		{
			const unsigned long long pm = __ballot(park);
			const uint32_t e = n_hit + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
			if (park) {
				stack.frag[0][e] = frag.x, stack.frag[1][e] = frag.y, stack.frag[2][e] = frag.z;
				stack.normal[0][e] = normal.x, stack.normal[1][e] = normal.y, stack.normal[2][e] = normal.z;
				stack.T[0][e] = T.x, stack.T[1][e] = T.y, stack.T[2][e] = T.z;
				stack.state[e] = (uint32_t)oi | (rng.block << 16); // (fewer than 2^16 objects fit the LDS; a lens loop runs at most 4096 rounds)
				stack.lobe_bits[e] = rng.lobe_bits | (depth << 24), stack.item[e] = item;
			}
			n_hit += (uint32_t)__popcll(pm);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}
	finish_sample_range(P, tile, wt, lane, out);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Split launches of scenes WITH grids, persistent form (the mesh kernel): PATH QUEUES — ray compaction between bounces.
//
// In render_wave() a lane keeps its path from its first ray to its last vertex.  On a mesh scene a third of a trip's lanes then sit on a ray
// that waits for the wave's next grid walk, the others run a merged shade / primary-ray stream of which each needs one half, and a walk serves
// the ~36 lanes that happen to wait: the trips — half of the kernel's vector instructions — issue at 40 % lanes.
// Round 4's repair (tools/experiments/mesh_sorted_trips.patch) kept the wave's paths in a pool in LDS and lost more in resident waves (12 for 16)
// than the fuller trips gained: the LDS is spoken for by the occupancy masks and the walk scratch.  Here the paths live in DEVICE MEMORY — two
// dense stacks per resident wave, private to it (no atomics, nothing shared; the top of a stack is what the wave wrote last, so it is read back
// from the XCD's L2) — and a lane holds a path only for the length of one trip.  A trip is one of three kinds, each with up to 64 lanes that all
// need the same thing:
//   GEN    the work item's next 64 (pixel, sample) pairs: primary ray (src/trace.rs:322-333, thin lens :335-360), planes / spheres / grid boxes;
//   SHADE  the top 64 parked hits: shading (:256-319) -> bounce ray, planes / spheres / grid boxes;
//   WALK   the top 64 parked rays — rays that enter a grid's box: ONE cooperative walk (grid_walk.hpp), merged with the ray's closest plane /
//          sphere hit (core/src/scene.rs:54-74: the lexicographic minimum of (distance, object index)).
// A ray of a GEN / SHADE trip that enters no box is classified on the spot; one that does is pushed onto the ray stack (origin, direction, its
// closest plane / sphere hit so far, the path's throughput and RNG state).  Classification (:242-252 and the rules of DESIGN.md section 3) ends a
// path — its sample goes to its 32-byte sector of the per-sample buffer — or pushes the hit (point, normal, throughput, RNG state) onto the hit
// stack.  A path's arithmetic is the same functions on the same values in the same order as in render_wave(): every sample has the same bits;
// which lane and which trip compute it changes nothing (the RNG is keyed by pixel and sample, sum_kernel adds the samples in order).
// A path carries its pixel, sample and scratch sector with it, so the wave draws its next work item as soon as the current one has no pair left
// (as render_wave<.., CHAIN> does) and only the launch's last paths are finished in trips that are not full.
struct PathQueues { // one resident wave's queues: SoA, `cap` entries each (a multiple of 64: every field is 512-byte aligned)
	RMD_GLOBAL double *hit_d;   // [9][cap]: hit point, surface normal, throughput
	RMD_GLOBAL double *ray_d;   // [13][cap]: origin, direction, throughput, distance of the closest plane / sphere hit so far (kFMax: none); t_max of a walk put aside
	RMD_GLOBAL uint32_t *hit_w; // [4][cap]: object | next RNG block << 16; lobe bits | depth << 24; x | y << 16; scratch sector (the sample's number is
	                            //   a function of its sector: sample_of_sector)
	RMD_GLOBAL uint32_t *ray_w; // [9][cap]: (closest plane / sphere + 1, 0 = none) | next RNG block << 16; then as above, kRayCarried in the second word:
	uint32_t cap;               //   the ray's walk was put aside (grid_walk.hpp: WalkCarry) and its cell index, previous cell and exit counters follow
};
constexpr uint32_t kRayCarried = 1u << 23; // (the lobe bits are 22, the depth sits in the top byte)
__host__ __device__ inline size_t path_queue_bytes(uint32_t cap) { return (size_t)cap * (9u * 8u + 13u * 8u + 4u * 4u + 9u * 4u); }
// The sample a scratch sector belongs to.  sector = (wave tile * sample_count + s) * 64 + pixel slot with s < sample_count the sample's number within
// the pass (render_wave_queued: GEN), so s = (sector >> 6) mod sample_count: by the host's multiplier M = floor(2^64 / sample_count) + 1
// (RenderParams::sample_magic; exact for every 32-bit dividend: a * sample_count < 2^64), 0 for a pass of ONE sample, where s = 0.
RMD_DEV uint32_t sample_of_sector(const RenderParams &P, uint32_t sector) {
	const uint32_t a = sector >> 6, q = (uint32_t)__umul64hi((unsigned long long)a, P.sample_magic);
	return P.sample_begin + (P.sample_magic != 0ull ? a - q * P.sample_count : 0u);
}
constexpr uint32_t kQueuedTripBoundPerPath = 2u * RMD_MAX_BOUNCE_LIMIT_DEV + 4u; // trips per path, as if ONE lane ran them all: a segment is at most a SHADE and a WALK trip
// LDS of a wave of the queued form: the walk scratch, the walk's carry area (grid_walk.hpp: WalkCarry — during a call the ring of its pre-test; around a
// call the DDA states of the walks it takes up / puts aside, on their way from / to the ray stack), per lane what a walking path does not need
// during its walk (throughput, RNG state: 32 bytes a lane), then the head — 7,184 bytes: 16 waves beside the benchmark mesh's 36.7 KB of masks
#ifndef RMD_QUEUE_SIDE_ALL
#define RMD_QUEUE_SIDE_ALL 1 // 1: pixel and sector wait in LDS too (40 bytes a lane: 7,696 bytes a wave, 16 waves beside the benchmark mesh's masks) instead of being fetched again behind the walk
#endif
constexpr size_t kQueuedSideBytes = 64u * (3u * sizeof(double) + (RMD_QUEUE_SIDE_ALL ? 4u : 2u) * sizeof(uint32_t));
constexpr size_t kQueuedDiagBytes = RMD_DIAG ? 24u * sizeof(unsigned long long) : 0u; // DIAG builds: the wave's phase clocks (RMD_DEBUG = 16)
__host__ __device__ inline size_t queued_wave_lds_bytes() { return sizeof(WalkScratch) + sizeof(WalkCarry) + kQueuedSideBytes + kQueuedDiagBytes + kWaveHeadBytes; }

// Queue traffic is non-temporal (RMD_QUEUE_NT): an entry is written once and read once, the waves' working set is several times the L2, and what it
// displaces there are the scene's tables, which every walk gathers from (measured with plain accesses: L2 hit rate 73 -> 59 %, mean L1 -> L2 read
// latency 227 -> 373 cycles against the lane-per-path form).
#ifndef RMD_QUEUE_NT
#define RMD_QUEUE_NT 0
#endif
#ifndef RMD_SAMPLE_NT
#define RMD_SAMPLE_NT 1
#endif
template <class T>
RMD_DEV T qld(const RMD_GLOBAL T *p) {
#if RMD_QUEUE_NT
	return __builtin_nontemporal_load(p);
#else
	return *p;
#endif
}
template <class T, class U>
RMD_DEV void qst(RMD_GLOBAL T *p, U v) {
#if RMD_QUEUE_NT
	__builtin_nontemporal_store((T)v, p);
#else
	*p = (T)v;
#endif
}
RMD_DEV void render_wave_queued(const RenderParams &P, KernargWords kernarg_params, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                                const void *__restrict__ work, const DevObject *lobjs, const uint32_t *lds_masks, unsigned char *wave_lds, uint32_t first_item) {
	const uint32_t lane = threadIdx.x & 63u;
	WalkScratch &scr = *reinterpret_cast<WalkScratch *>(wave_lds);
	WalkCarry *carry = reinterpret_cast<WalkCarry *>(wave_lds + sizeof(WalkScratch));
	// what a walking path does not need during its walk waits in LDS, one column per lane (the walk is where the kernel's register pressure peaks)
	double *side_d = reinterpret_cast<double *>(wave_lds + sizeof(WalkScratch) + sizeof(WalkCarry)) + lane;                             // [0], [64], [128]: throughput
	uint32_t *side_w = reinterpret_cast<uint32_t *>(wave_lds + sizeof(WalkScratch) + sizeof(WalkCarry) + 192u * sizeof(double)) + lane; // [0], [64]: the entry's first two words
	if (P.bounce_limit == 0u) return; // (such launches are not made: api.cpp)
#if RMD_DIAG
	volatile unsigned long long *qacc = reinterpret_cast<volatile unsigned long long *>(wave_lds + sizeof(WalkScratch) + sizeof(WalkCarry) + kQueuedSideBytes);
	if (lane < 24u) qacc[lane] = 0ull;
#endif
	PathQueues q; // (the capacity is a constant of the build — api.cpp sizes the buffer by the same one — so the four arrays are ONE base address and constant offsets)
	{
		constexpr uint32_t cap = kQueuePaths;
		const uint32_t wave_index = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
		RMD_GLOBAL unsigned char *base = (RMD_GLOBAL unsigned char *)P.queue_buf + (size_t)wave_index * path_queue_bytes(cap);
		q.cap = cap;
		q.hit_d = (RMD_GLOBAL double *)base;
		q.ray_d = q.hit_d + (size_t)9u * cap;
		q.hit_w = (RMD_GLOBAL uint32_t *)(q.ray_d + (size_t)13u * cap);
		q.ray_w = q.hit_w + (size_t)4u * cap;
	}
	enum { kGen = 0, kShade = 1, kWalk = 2 };

	// the work item the wave hands pairs out of (wave-uniform); pool_items = 0xFFFFFFFF: the launch has no item left
	uint32_t wt = 0, pool_first = 0, pool_items = 0, next_item = 0;
	WaveTile tile = {};
	auto take_item = [&](const RenderParams &Q, uint32_t drawn) {
		wt = drawn / Q.split_k;
		const uint32_t part = drawn - wt * Q.split_k;
		tile = reinterpret_cast<const WaveTile *>(work)[wt];
		const uint32_t per_part = (Q.sample_count + Q.split_k - 1u) / Q.split_k;
		const uint32_t s_lo = part * per_part < Q.sample_count ? part * per_part : Q.sample_count;
		const uint32_t s_hi = s_lo + per_part < Q.sample_count ? s_lo + per_part : Q.sample_count;
		pool_first = s_lo, pool_items = (s_hi - s_lo) * 64u, next_item = 0u;
	};
	if (first_item / P.split_k < P.n_work) take_item(P, first_item);
	uint32_t n_hit = 0, n_ray = 0; // wave-uniform: entries on the two stacks
	// The trip loop's bound (report_fault).  Every trip hands out 64 pairs or takes at least one path off a stack and either ends it or moves it on
	// by half a segment (WALK: ray -> hit or end; SHADE: hit -> ray, hit or end), a path has at most RMD_MAX_BOUNCE_LIMIT segments, and at most
	// cap paths are in flight when an item is drawn: an item's pairs + cap paths, times kQueuedTripBoundPerPath, is more trips than the wave can
	// take before its next draw even if every trip served ONE lane.  A walk that is put aside (below) goes back onto the ray stack and takes part in
	// another WALK trip: every such trip moves it on by at least one step, of which it has at most walk_steps_bound.  The count starts again at every draw.
	const unsigned long long trips_per_path = kQueuedTripBoundPerPath + (unsigned long long)RMD_MAX_BOUNCE_LIMIT_DEV * P.walk_steps_bound;
	unsigned long long trips_left = ((unsigned long long)pool_items + q.cap) * trips_per_path + 64ull;
#if RMD_DIAG
	if (P.debug_flags & 32u) trips_left = 1ull; // tests/test_gpu_faults.py: forces the bound
#endif
	// Hits HELD in their lanes (RMD_QUEUE_HOLD_HITS).  A hit that a trip has classified goes onto the hit stack and comes back off it for the SHADE trip
	// that takes it — 92 bytes written and 92 read through an L2 that the queues' working set overflows several times (measured: L2 hit rate
	// 73 -> 59 %, L1 -> L2 read latency 227 -> 373 cycles against the lane-per-path form).  When the very next trip is a SHADE trip anyway — no full
	// walk waits, and the stack's hits and this trip's together fill a trip — this trip's hits stay where they are, in their lanes' registers, and
	// the SHADE trip pops only what it needs to fill the other lanes.  Same trips, same lanes per trip; a hit's values are the ones it would have
	// read back.
#ifndef RMD_QUEUE_HOLD_HITS
#define RMD_QUEUE_HOLD_HITS 1
#endif
	// Rays held likewise (RMD_QUEUE_HOLD_RAYS): when the rays a GEN / SHADE trip sends to the grids fill a WALK trip together with the stack's, they
	// do not travel through the stack: each waits in its lane's columns of the wave's LDS — origin, direction, closest plane / sphere hit in the walk
	// scratch and the carry area (free between walks), the rest where a walking path keeps it anyway (the side area) — and the WALK trip pops only
	// what fills the other lanes.
#ifndef RMD_QUEUE_HOLD_RAYS
#define RMD_QUEUE_HOLD_RAYS 1
#endif
	bool rheld = false;
	double *stash_a = reinterpret_cast<double *>(wave_lds) + lane;                        // [0], [64], [128], [192]: origin, direction.x
	int32_t *stash_oi = reinterpret_cast<int32_t *>(wave_lds + 256u * sizeof(double)) + lane; // the closest plane / sphere so far
	double *stash_b = reinterpret_cast<double *>(wave_lds + sizeof(WalkScratch)) + lane;  // [0], [64], [128]: direction.y, .z, the distance of that hit
	static_assert(sizeof(WalkScratch) >= 256u * sizeof(double) + 64u * sizeof(int32_t) && sizeof(WalkCarry) >= 192u * sizeof(double), "a held ray's columns");
	bool held = false;
	V3 h_frag, h_normal, h_T;
	RMD_UNDEF3(h_frag) RMD_UNDEF3(h_normal) RMD_UNDEF3(h_T)
	uint32_t h_st = 0, h_lb = 0, h_px = 0, h_sector = 0;
	for (;;) {
		// the launch parameters a trip needs, re-read from the kernel arguments (see render_wave)
		KernargWords src = kernarg_params;
		asm volatile("" : "+s"(src));
		unsigned long long w[sizeof(RenderParams) / 8];
#pragma unroll
		for (unsigned i = 0; i < sizeof(RenderParams) / 8; i++) w[i] = src[i];
		RenderParams Pt;
		__builtin_memcpy(&Pt, w, sizeof(Pt));

		// the item's pool has run dry: the wave draws its next work item (the persistent work loop's draw, render_kernel below, with the same bound)
		if (next_item >= pool_items && pool_items != 0xFFFFFFFFu) {
			volatile uint32_t *last_draw = reinterpret_cast<volatile uint32_t *>(wave_lds + queued_wave_lds_bytes() - kWaveHeadBytes);
			uint32_t drawn = 0, floor = 0;
			if (lane == 0u) {
				drawn = atomicAdd(Pt.work_counter, 1u);
				floor = *last_draw;
				*last_draw = drawn + 1u;
			}
			drawn = (uint32_t)__builtin_amdgcn_readfirstlane((int)drawn);
			floor = (uint32_t)__builtin_amdgcn_readfirstlane((int)floor);
#if RMD_DIAG
			if ((Pt.debug_flags & 128u) && floor != 0u) floor = 0xFFFFFFFFu; // tests/test_gpu_faults.py: forces the bound at a wave's second draw
#endif
			if (drawn >= Pt.n_work * Pt.split_k) {
				pool_items = 0xFFFFFFFFu, next_item = 0xFFFFFFFFu;
			} else {
				if (RMD_UNLIKELY(drawn < floor)) report_fault(Pt, kFaultWorkLoop, drawn); // (never reached; the poisoned counter ends the launch)
				take_item(Pt, drawn);
				trips_left = ((unsigned long long)pool_items + q.cap) * trips_per_path + 64ull;
#if RMD_DIAG
				if (Pt.debug_flags & 32u) trips_left = 1ull;
#endif
			}
		}
		// (no `break` here: the wave that runs into the bound reports, drops every path and pair it holds and leaves through the loop's own exit below —
		// a second way out of this loop, behind the lane-0 branch of the report, came out of the compiler as a loop that some lanes left and others
		// did not: the forced-bound test hung)
		if (RMD_UNLIKELY(trips_left-- == 0ull)) { // (never reached: see below)
			report_fault(Pt, kFaultQueuedTripLoop, wt);
			n_hit = 0u, n_ray = 0u, pool_items = 0xFFFFFFFFu, next_item = 0xFFFFFFFFu, held = false, rheld = false;
		}
		const bool pairs_left = next_item < pool_items; // (false once the launch has no item left: next_item = pool_items = 0xFFFFFFFF)
		// Which kind of trip.  A FULL trip whenever a stack holds 64 entries — walks first: they are what the other kinds wait for —, else the item's
		// next 64 pairs, and only when the launch has no pair left to hand out the fuller of the two stacks in a trip that is not full.
		// Every reachable state takes a trip that makes progress: (1) n_ray >= 64 or n_hit >= 64: a full trip, 64 paths move on.  (2) both below 64
		// and pairs left: fewer than 128 paths are in flight, the 64 a GEN trip may add fit (cap >= 192: api.cpp), next_item grows by 64 (all of
		// them may fall outside a ragged tile: a trip without a lane, but the pool has shrunk).  (3) both below 64, no pair left, a stack
		// non-empty: a trip of max(n_ray, n_hit) >= 1 lanes.  (4) both empty and no pair left: the loop ends.  A WALK trip always serves whatever
		// rays wait (no ray ever waits for others to join it for ever), and no state selects a trip of a kind whose source is empty — the hang of
		// round 4's three-list pool (tools/experiments/README.md) was such a state: a generation trip chosen with no free slot.
#if RMD_DIAG
		// where a wave's time goes, by kind of trip and phase (RMD_DEBUG = 16) — debug_counters[kind * 8 + phase]: 0 = the trip's entries fetched (until
		// the first of them is used), 1 = shading / ray generation, 2 = planes, spheres and boxes, 3 = the walk, 4 = ray pushes, 5 = classification,
		// 6 = sample stores and hit pushes, 7 = trips.  The clock is read where a phase ends, nothing is drained: a phase is charged what it WAITS for,
		// not what it requests.  Summed per wave in LDS, added to the launch's counters when the wave leaves.
		const bool qstamp = (Pt.debug_flags & 16u) && Pt.debug_counters;
		unsigned long long qt_prev = qstamp ? __builtin_amdgcn_s_memtime() : 0ull;
#define RMD_QSTAMP(phase)                                                              \
		if (qstamp) {                                                                      \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
			if (lane == 0u) qacc[kind * 8u + (phase)] += now_ - qt_prev;                   \
			qt_prev = now_;                                                                \
		}
#define RMD_QSTAMP_FETCH                      \
		if (qstamp) __builtin_amdgcn_s_waitcnt(0); \
		RMD_QSTAMP(0u) /* (the fetch is charged its own wait) */
#else
#define RMD_QSTAMP(phase)
#define RMD_QSTAMP_FETCH
#endif
		uint32_t kind;
		// (the stacks' counters are wave-uniform by construction; said once per trip, because with the held hits in the loop the compiler's uniformity
		// analysis gives up on them and keeps them — and every address and branch made from them — in vector registers)
		n_hit = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_hit), n_ray = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_ray);
		const unsigned long long held_mask = RMD_QUEUE_HOLD_HITS ? __ballot(held) : 0ull;
		const unsigned long long rheld_mask = RMD_QUEUE_HOLD_RAYS ? __ballot(rheld) : 0ull;
		if (rheld_mask != 0ull) kind = kWalk;      // (decided when the rays were held: the stack's rays and the held ones fill the trip)
		else if (held_mask != 0ull) kind = kShade; // (likewise the hits)
		else if (n_ray >= 64u) kind = kWalk;
		else if (n_hit >= 64u) kind = kShade;
		else if (pairs_left) kind = kGen;
		else if (n_ray != 0u && n_ray >= n_hit) kind = kWalk;
		else if (n_hit != 0u) kind = kShade;
		else break;

		bool active, failed = false, classify, to_ray = false;
		uint32_t depth = 1, lobe_bits = 0, rng_block = 0, pxw = 0, sector = 0, sub = 0;
		int oi = -1;
		double t;
		V3 ro, rd, T;
		RMD_UNDEF(t) RMD_UNDEF3(ro) RMD_UNDEF3(rd) RMD_UNDEF3(T)
		if (kind == kWalk) {
			// ---------------- WALK: the rays held in their lanes, and in the other lanes the top entries of the ray stack (as many as there are: up to 64 in
			// all); one cooperative walk per grid object
			const uint32_t n_rheld = (uint32_t)__popcll(rheld_mask), n_pop = n_ray < 64u - n_rheld ? n_ray : 64u - n_rheld, base = n_ray - n_pop, n = n_rheld + n_pop;
			const unsigned long long free_mask = ~rheld_mask;
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u)); // free lanes before this one
			const bool take = !rheld && rank < n_pop;
			active = rheld || take;
			const uint32_t e = take ? base + rank : base; // (a lane without a ray reads the trip's first entry: sane values that nobody uses)
			n_ray = base;
			constexpr uint32_t cap = kQueuePaths;
			uint32_t lb = 0u;
			if (!rheld) {
				ro = mk(qld(&q.ray_d[e]), qld(&q.ray_d[cap + e]), qld(&q.ray_d[2u * cap + e]));
				rd = mk(qld(&q.ray_d[3u * cap + e]), qld(&q.ray_d[4u * cap + e]), qld(&q.ray_d[5u * cap + e]));
				t = qld(&q.ray_d[9u * cap + e]);
				const uint32_t st = qld(&q.ray_w[e]);
				lb = qld(&q.ray_w[cap + e]);
				oi = (int)(st & 0xFFFFu) - 1;
				// the rest of the path's state goes from its entry to the lane's column of the side area, and comes back behind the walk
				side_d[0] = qld(&q.ray_d[6u * cap + e]), side_d[64] = qld(&q.ray_d[7u * cap + e]), side_d[128] = qld(&q.ray_d[8u * cap + e]);
				side_w[0] = st, side_w[64] = lb;
#if RMD_QUEUE_SIDE_ALL
				side_w[128] = qld(&q.ray_w[2u * cap + e]), side_w[192] = qld(&q.ray_w[3u * cap + e]);
#endif
			} else { // a held ray: out of its lane's columns (its side-area column was filled when it was held)
				ro = mk(stash_a[0], stash_a[64], stash_a[128]);
				rd = mk(stash_a[192], stash_b[0], stash_b[64]);
				t = stash_b[128], oi = stash_oi[0];
			}
			rheld = false;
			RMD_QSTAMP_FETCH
			// Walks put aside (grid_walk.hpp: cut_lanes): a call with many walkers stops stepping under its last K rays and ends under its last 2K
			// walkers; what is left of such a walk — its DDA state — goes back onto the ray stack with the ray (kRayCarried) and the walk goes on in
			// the trip that pops it, beside that trip's new rays.  The state travels through the wave's WalkCarry in LDS, lane by lane, the way
			// render_wave's walks hand it from one call to the next.
			bool carried = active && (lb & kRayCarried) != 0u;
			if (__ballot(carried) != 0ull) {
				if (carried) {
					carry->tm[0][lane] = qld(&q.ray_d[10u * cap + e]), carry->tm[1][lane] = qld(&q.ray_d[11u * cap + e]), carry->tm[2][lane] = qld(&q.ray_d[12u * cap + e]);
					carry->idx[lane] = qld(&q.ray_w[4u * cap + e]), carry->prev[lane] = qld(&q.ray_w[5u * cap + e]);
					carry->rem[0][lane] = qld(&q.ray_w[6u * cap + e]), carry->rem[1][lane] = qld(&q.ray_w[7u * cap + e]), carry->rem[2][lane] = qld(&q.ray_w[8u * cap + e]);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier(); // (every held ray is out of the walk scratch and the carry area before the walk writes to them)
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const bool cut = n >= kWalkCutMinWalkers; // (every walker of such a call takes at least one step: walks always finish)
			const uint32_t cut_lanes = cut ? Pt.walk_cut & 0xffu : 0u, cut_round = cut ? (Pt.walk_cut >> 8) & 0xffu : 0u;
			intersect_grids<true>(objs, Pt.n_objects, grids, lds_masks, scr, active, ro, rd, t, oi, sub, Pt.debug_flags, (Pt.debug_flags & 16u) && Pt.debug_counters ? Pt.debug_counters + 16 : Pt.debug_counters /* (the walk's own phase clocks land behind this body's: counters 24 .. 31) */, cut_lanes, carry, &carried, cut_round, Pt.grid_mask);
			RMD_QSTAMP(3u)
			T = mk(side_d[0], side_d[64], side_d[128]);
			rng_block = side_w[0] >> 16;
			const uint32_t lb2 = side_w[64];
			lobe_bits = lb2 & 0x3FFFFFu, depth = lb2 >> 24;
			// (pixel, sample and sector are not needed before the trip's stores: fetched from the entry here — it stays as it is until this trip's own pushes —
			// with the classification to arrive under)
#if RMD_QUEUE_SIDE_ALL
			pxw = side_w[128], sector = side_w[192];
#else
			const uint32_t e2 = n_ray + (active ? lane : 0u); // (= e, made again: one register fewer across the walk)
			pxw = qld(&q.ray_w[2u * cap + e2]), sector = qld(&q.ray_w[3u * cap + e2]);
#endif
			to_ray = carried; // an unfinished walk: back onto the stack (its closest plane / sphere hit is unchanged: a walk that has found nothing yet merges nothing)
			classify = active && !carried;
		} else {
			Rng rng;
			if (kind == kShade) {
				// ---------------- SHADE: the hits held in their lanes, and in the other lanes the top entries of the hit stack (as many as there are: up to 64 in all)
				const uint32_t n_held = (uint32_t)__popcll(held_mask), n_pop = n_hit < 64u - n_held ? n_hit : 64u - n_held, base = n_hit - n_pop;
				const unsigned long long free_mask = ~held_mask;
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_mask, 0u)); // free lanes before this one
				const bool take = !held && rank < n_pop;
				active = held || take;
				n_hit = base;
				constexpr uint32_t cap = kQueuePaths;
				if (__builtin_expect(n_pop != 0u, 1)) {
					const uint32_t e = take ? base + rank : base; // (a lane without a hit reads the trip's first entry: sane values that nobody uses)
					if (!held) {
						h_st = qld(&q.hit_w[e]), h_lb = qld(&q.hit_w[cap + e]);
						h_px = qld(&q.hit_w[2u * cap + e]), h_sector = qld(&q.hit_w[3u * cap + e]);
						h_frag = mk(qld(&q.hit_d[e]), qld(&q.hit_d[cap + e]), qld(&q.hit_d[2u * cap + e]));
						h_normal = mk(qld(&q.hit_d[3u * cap + e]), qld(&q.hit_d[4u * cap + e]), qld(&q.hit_d[5u * cap + e]));
						h_T = mk(qld(&q.hit_d[6u * cap + e]), qld(&q.hit_d[7u * cap + e]), qld(&q.hit_d[8u * cap + e]));
					}
				}
				const uint32_t st = h_st, lb = h_lb;
				pxw = h_px, sector = h_sector;
				const V3 frag = h_frag, normal = h_normal;
				T = h_T;
				RMD_QSTAMP_FETCH
				rng.pixel = (pxw >> 16) * Pt.W + (pxw & 0xFFFFu), rng.sample = sample_of_sector(Pt, sector);
				rng.block = st >> 16, rng.lobe_bits = lb & 0x3FFFFFu;
				depth = lb >> 24;
				const DevObject &o = lobjs[st & 0xFFFFu];
				// (every lane shades — see render_wave_sorted)
				shade(Pt, normal, frag, ld3(o.color), o.roughness, o.metalness, ld3(Pt.cam_pos), rng, ro, rd, T);
				depth++;
				// (see render_wave: a path whose throughput is exactly zero is ended where the caller asked for that)
				const bool black = Pt.end_black_paths != 0u && T.x == 0.0 && T.y == 0.0 && T.z == 0.0;
				failed = active && (depth > Pt.bounce_limit || black);
			} else {
				// ---------------- GEN: the work item's next 64 (pixel, sample) pairs, one per lane (slots outside a ragged tile are skipped)
				const uint32_t k = next_item + lane;
				next_item += 64u;
				active = k < pool_items && (k & 7u) < tile.w && ((k >> 3) & 7u) < tile.h;
				const uint32_t x = tile.x0 + (k & 7u), y = tile.y0 + ((k >> 3) & 7u);
				pxw = x | y << 16;
				sector = (wt * Pt.sample_count + pool_first + (k >> 6)) * 64u + (k & 63u); // (< 2^32: api.cpp caps a pass)
				rng.pixel = y * Pt.W + x, rng.sample = Pt.sample_begin + pool_first + (k >> 6), rng.block = 0u, rng.lobe_bits = 0u;
				T = mk(1.0, 1.0, 1.0);
				double u0, u1;
				rng.next2(Pt.key0, Pt.key1, u0, u1); // block 0: the pixel jitter (:326-327)
				primary_ray(Pt, x, y, u0, u1, ro, rd);
				if (Pt.use_dof) failed = active && !thin_lens_from_pinhole(Pt, ro, rd, rng, ro, rd); // the reference panics there; the sample contributes zero
			}
			rng_block = rng.block, lobe_bits = rng.lobe_bits;
			RMD_QSTAMP(1u)
			// ---------------- src/trace.rs:239, first part — planes, spheres and the grids' boxes (scene_split.hpp)
			const bool want = active && !failed;
			const bool enters = intersect_simple(objs, Pt.n_objects, grids, want, ro, rd, t, oi, Pt.axis_pairs, Pt.visit_mask);
			to_ray = want && enters;
			classify = want && !enters;
			RMD_QSTAMP(2u)
		}
#if RMD_DIAG
		if ((Pt.debug_flags & 8u) && Pt.debug_counters) { // trips by kind and the lanes they serve
			const unsigned long long am = __ballot(active);
			if (lane == 0) atomicAdd(&Pt.debug_counters[10], 1ull), atomicAdd(&Pt.debug_counters[11], (unsigned long long)__popcll(am)), atomicAdd(&Pt.debug_counters[12], kind == kWalk ? (unsigned long long)__popcll(am) : 0ull);
		}
#endif
		// ---------------- the rays that have to walk (and the walks that were put aside): pushed onto the ray stack, consecutive entries for the lanes that push
		// (a WALK trip's lanes have read everything they need of their own entries — by loads that precede these stores in program order)
		bool rhold = false;
		{
			const unsigned long long pm = __ballot(to_ray);
			// held where they are when the next trip walks them anyway: this trip's rays and the stack's fill a WALK trip (a WALK trip's own rays — the
			// walks it puts aside — go back onto the stack: their DDA state sits in the carry area's columns)
			rhold = RMD_QUEUE_HOLD_RAYS && kind != kWalk && pm != 0ull && n_ray + (uint32_t)__popcll(pm) >= 64u;
			if (rhold) {
				if (to_ray) {
					stash_a[0] = ro.x, stash_a[64] = ro.y, stash_a[128] = ro.z, stash_a[192] = rd.x, stash_b[0] = rd.y, stash_b[64] = rd.z, stash_b[128] = t;
					stash_oi[0] = oi;
					side_d[0] = T.x, side_d[64] = T.y, side_d[128] = T.z;
					side_w[0] = (uint32_t)(oi + 1) | (rng_block << 16), side_w[64] = lobe_bits | (depth << 24);
#if RMD_QUEUE_SIDE_ALL
					side_w[128] = pxw, side_w[192] = sector;
#endif
				}
				rheld = to_ray;
#if RMD_DIAG
				if ((Pt.debug_flags & 8u) && Pt.debug_counters && lane == 0) atomicAdd(&Pt.debug_counters[23], (unsigned long long)__popcll(pm));
#endif
			}
			if (pm != 0ull && !rhold) {
				constexpr uint32_t cap = kQueuePaths;
				const uint32_t e = n_ray + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
				if (to_ray) {
					qst(&q.ray_d[e], ro.x), qst(&q.ray_d[cap + e], ro.y), qst(&q.ray_d[2u * cap + e], ro.z);
					qst(&q.ray_d[3u * cap + e], rd.x), qst(&q.ray_d[4u * cap + e], rd.y), qst(&q.ray_d[5u * cap + e], rd.z);
					qst(&q.ray_d[6u * cap + e], T.x), qst(&q.ray_d[7u * cap + e], T.y), qst(&q.ray_d[8u * cap + e], T.z);
					qst(&q.ray_d[9u * cap + e], t);
					qst(&q.ray_w[e], (uint32_t)(oi + 1) | (rng_block << 16)); // (fewer than 2^16 objects fit the LDS; a lens loop runs at most 4096 rounds)
					qst(&q.ray_w[cap + e], lobe_bits | (depth << 24) | (kind == kWalk ? kRayCarried : 0u));
					qst(&q.ray_w[2u * cap + e], pxw), qst(&q.ray_w[3u * cap + e], sector);
					if (kind == kWalk) { // the DDA state grid_intersect_wave has left in this lane's column of the carry
						qst(&q.ray_d[10u * cap + e], carry->tm[0][lane]), qst(&q.ray_d[11u * cap + e], carry->tm[1][lane]), qst(&q.ray_d[12u * cap + e], carry->tm[2][lane]);
						qst(&q.ray_w[4u * cap + e], carry->idx[lane]), qst(&q.ray_w[5u * cap + e], carry->prev[lane]);
						qst(&q.ray_w[6u * cap + e], carry->rem[0][lane]), qst(&q.ray_w[7u * cap + e], carry->rem[1][lane]), qst(&q.ray_w[8u * cap + e], carry->rem[2][lane]);
					}
				}
				n_ray += (uint32_t)__popcll(pm);
#if RMD_DIAG
				if ((Pt.debug_flags & 8u) && Pt.debug_counters && lane == 0) { // rays pushed; of them walks that were put aside
					atomicAdd(&Pt.debug_counters[19], (unsigned long long)__popcll(pm));
					if (kind == kWalk) atomicAdd(&Pt.debug_counters[22], (unsigned long long)__popcll(pm));
				}
#endif
			}
		}
		RMD_QSTAMP(4u)
		// ---------------- classification (the rules of render_wave's phase C)
		bool terminal = failed, park = false, emitted = false;
		V3 frag, normal;
		RMD_UNDEF3(frag) RMD_UNDEF3(normal)
		if (classify) {
			if (oi < 0) {
				terminal = true; // :242 miss -> radiance 0
			} else {
				const DevObject &o = lobjs[oi];
				frag = ro + rd * t; // :246
				if (o.material_kind == 2u) {
					emitted = true; // :250-252 Emission
					terminal = true;
				} else {
					if (o.geometry_kind == 0u) normal = ld3(o.normal);                        // plane.rs:28-32
					else if (o.geometry_kind == 1u) normal = normalize(frag - ld3(o.origin)); // sphere.rs:31-35
					else {
						const DevGrid &g = grids[o.grid_index];
						normal = triangle_normal(as_global(g.tri_pos) + (size_t)sub * 9, as_global(g.tri_nrm) + (size_t)sub * 9, as_global(g.tri_aux) + (size_t)sub * 4, frag); // acc_grid.rs:85-87
					}
					const double probe_sum = ((normal.x + normal.y) + normal.z) + ((frag.x + frag.y) + frag.z);
					const bool finite_inputs = __builtin_fabs(probe_sum) < __builtin_inf();
					const bool black_bounce = Pt.end_black_paths != 0u && (o.flags & kObjBlackDiffuse) != 0u && lobe_bits < (1u << 21);
					if (((depth == Pt.bounce_limit && Pt.shade_last_depth == 0u) || black_bounce) && finite_inputs) terminal = true; // L = 0
					else park = true;
				}
			}
		}
		RMD_QSTAMP(5u)
		if (terminal) { // the finished sample: T (.) L into its 32-byte sector of the per-sample buffer (added by sum_kernel, behind the kernel boundary)
			V3 L = mk(0.0, 0.0, 0.0);
			if (emitted) L = ld3(lobjs[oi].color); // (fetched here, outside the nest of branches that found the light)
			L = hadamard(T, L);
			RMD_GLOBAL double *dst = (RMD_GLOBAL double *)Pt.sample_buf + (size_t)sector * kSampleStride;
#if RMD_SAMPLE_NT
			__builtin_nontemporal_store(L.x, dst), __builtin_nontemporal_store(L.y, dst + 1), __builtin_nontemporal_store(L.z, dst + 2); // (written once, read by sum_kernel)
#else
			dst[0] = L.x, dst[1] = L.y, dst[2] = L.z;
#endif
		}
		// ---------------- the hits that go on: held where they are when the next trip shades them anyway, else pushed onto the hit stack
		{
			const unsigned long long pm = __ballot(park);
			const uint32_t n_park = (uint32_t)__popcll(pm);
			const bool hold = RMD_QUEUE_HOLD_HITS && n_park != 0u && !rhold && n_ray < 64u && n_hit + n_park >= 64u; // = the rule above would select a full SHADE trip next (no rays are held: the next trip would be theirs)
			// (unconditional copies: the held values are made here for every lane, so that nothing of them is live across the trip's other phases)
			h_frag = frag, h_normal = normal, h_T = T;
			h_st = (uint32_t)oi | (rng_block << 16), h_lb = lobe_bits | (depth << 24), h_px = pxw, h_sector = sector;
			held = hold && park;
#if RMD_DIAG
			if ((Pt.debug_flags & 8u) && Pt.debug_counters && lane == 0 && n_park != 0u) atomicAdd(&Pt.debug_counters[hold ? 21 : 20], (unsigned long long)n_park);
#endif
			if (pm != 0ull && !hold) {
				constexpr uint32_t cap = kQueuePaths;
				const uint32_t e = n_hit + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
				if (park) {
					qst(&q.hit_d[e], frag.x), qst(&q.hit_d[cap + e], frag.y), qst(&q.hit_d[2u * cap + e], frag.z);
					qst(&q.hit_d[3u * cap + e], normal.x), qst(&q.hit_d[4u * cap + e], normal.y), qst(&q.hit_d[5u * cap + e], normal.z);
					qst(&q.hit_d[6u * cap + e], T.x), qst(&q.hit_d[7u * cap + e], T.y), qst(&q.hit_d[8u * cap + e], T.z);
					qst(&q.hit_w[e], h_st);
					qst(&q.hit_w[cap + e], h_lb);
					qst(&q.hit_w[2u * cap + e], pxw), qst(&q.hit_w[3u * cap + e], sector);
				}
				n_hit += n_park;
			}
		}
		RMD_QSTAMP(6u)
#if RMD_DIAG
		if (qstamp && lane == 0u) qacc[kind * 8u + 7u] += 1ull;
#endif
	}
#undef RMD_QSTAMP
#undef RMD_QSTAMP_FETCH
#if RMD_DIAG
	if ((P.debug_flags & 16u) && P.debug_counters && lane < 24u) atomicAdd(&P.debug_counters[lane], qacc[lane]);
#endif
}

// PERSIST = false: one wave per work item, block b's waves take items b * waves .. ; PERSIST = true (tile modes of grid scenes):
// as many 16-wave workgroups as the device has CUs, each staging the masks ONCE, whose waves draw work items from a launch-wide
// counter (P.work_counter, zeroed by the host) until none is left — the masks cost one copy per CU instead of one per
// 4-wave workgroup, which leaves each wave 8 KB of LDS, and no wave slot idles while the rest of a workgroup finishes.
template <int MODE, bool GRID>
constexpr bool kSortedTrips = RMD_SORTED_TRIPS && MODE == kModeTilesBuffered && !GRID;
#ifndef RMD_CHAIN_ITEMS
#define RMD_CHAIN_ITEMS 1
#endif
template <int MODE, bool GRID>
constexpr bool kChainItems = RMD_CHAIN_ITEMS && MODE == kModeTilesBuffered && GRID; // (persistent form only: render_wave, CHAIN)
// LDS of one wave of an instantiation
// QUEUED (persistent split launches of scenes with grids): the wave body with the paths in queues in device memory (render_wave_queued)
#ifndef RMD_PATH_QUEUES
#define RMD_PATH_QUEUES 1
#endif
template <int MODE, bool GRID>
constexpr bool kPathQueues = RMD_PATH_QUEUES && MODE == kModeTilesBuffered && GRID;
template <int MODE, bool GRID, bool QUEUED = false>
__host__ __device__ inline size_t wave_lds_of(uint32_t n_grids) { return QUEUED ? queued_wave_lds_bytes() : kSortedTrips<MODE, GRID> ? kWaveHeadBytes + sizeof(SortPool) : wave_lds_bytes(n_grids); }
template <int MODE, bool GRID>
constexpr uint32_t kPersistWaves = kSortedTrips<MODE, GRID> ? kSortedWavesPerWg : GRID ? kGridPersistWavesPerWg : kPersistWavesPerWg;
template <int MODE, bool GRID, bool PERSIST, bool CHAIN = false, bool QUEUED = false>
__global__ __launch_bounds__(PERSIST ? 64 * (kPersistWaves<MODE, GRID>) : GRID ? 64 * kGridWavesPerWg : 64,
                             GRID ? RMD_GRID_MINW : (kSortedTrips<MODE, GRID>) ? (RMD_SORT_WAVES * RMD_SORT_WGS_PER_CU / 4) : (PERSIST ? 4 : RMD_NOGRID_MINW)) void render_kernel(
    RenderParams P, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids, const void *__restrict__ work, double *__restrict__ out,
    int32_t *__restrict__ path_obj, uint32_t *__restrict__ path_sub) {
	extern __shared__ __align__(16) unsigned char smem[];
	// P is this kernel's FIRST by-value argument: it starts at offset 0 of the kernel-argument segment, from where render_wave re-reads
	// the launch parameters on every trip instead of carrying them in registers
	const KernargWords kernarg_params = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr();
	// LDS: [object table][grid occupancy masks][one walk scratch per wave]
	DevObject *lobjs = reinterpret_cast<DevObject *>(smem);
	uint32_t *lmasks = reinterpret_cast<uint32_t *>(smem + (size_t)P.n_objects * sizeof(DevObject));
	const uint32_t tid = threadIdx.x, wave = tid >> 6, waves_per_wg = blockDim.x >> 6;
	// a wave's area: the instantiation's own data, then 16 bytes more — word 0: 1 + the work item the wave drew last (persistent form).  (Behind
	// the data, not in front: the mesh kernel's register allocation is at its limit, and with the data 16 bytes further on it came out with 22
	// spilled registers instead of 13.)
	unsigned char *wave_lds = smem + (size_t)P.n_objects * sizeof(DevObject) + (size_t)((P.mask_words_total + 3u) & ~3u) * 4u +
	                          (size_t)wave * wave_lds_of<MODE, GRID, QUEUED>(P.n_grids);
	[[maybe_unused]] unsigned char *wave_head = wave_lds + wave_lds_of<MODE, GRID, QUEUED>(P.n_grids) - kWaveHeadBytes;
	// stage the object table and the occupancy masks: coalesced, once per workgroup
	{
		const double *src = reinterpret_cast<const double *>(objs);
		double *dst = reinterpret_cast<double *>(lobjs);
		for (uint32_t i = tid; i < P.n_objects * 16u; i += blockDim.x) dst[i] = src[i];
		for (uint32_t gi = 0; gi < P.n_grids; gi++) {
			const DevGrid &g = grids[gi];
			if (g.mask_lds_word == 0xFFFFFFFFu) continue;
			for (uint32_t i = tid; i < g.mask_n_words; i += blockDim.x) lmasks[g.mask_lds_word + i] = as_global(g.mask_words)[i];
		}
	}
	__syncthreads(); // the only workgroup barrier: from here on every wave runs on its own
	const uint32_t *lds_masks = P.mask_words_total ? lmasks : nullptr;
	if constexpr (PERSIST) {
		static_assert(MODE != kModeList, "list launches are not persistent");
		const uint32_t n_items = P.n_work * (MODE == kModeTilesBuffered ? P.split_k : 1u);
		// The work loop's bound.  The counter only grows, so every draw of a wave is larger than its last one and the draws that pass the test
		// below are fewer than n_items: the loop ends by itself.  What is checked is the premise — a draw that is NOT larger than the wave's last
		// (the counter was overwritten: items would be rendered twice, for ever) is a fault.  The last draw + 1 lives in the wave's LDS head,
		// not in a register (the mesh kernel runs at its register limit); a faulting wave ORs kWorkCounterPoison into the counter, which ends
		// every other wave's loop at its next draw.
		volatile uint32_t *last_draw = reinterpret_cast<volatile uint32_t *>(wave_head);
		if ((tid & 63u) == 0u) *last_draw = 0u;
		for (;;) {
			uint32_t item = 0, floor = 0;
			if ((tid & 63u) == 0u) {
				item = atomicAdd(P.work_counter, 1u);
				floor = *last_draw;
				*last_draw = item + 1u;
#if RMD_DIAG
				if ((P.debug_flags & 128u) && floor != 0u) floor = 0xFFFFFFFFu; // tests/test_gpu_faults.py: forces the bound at a wave's second draw
#endif
			}
			item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
			if (item >= n_items) break;
#if RMD_BOUND_DRAWS
			floor = (uint32_t)__builtin_amdgcn_readfirstlane((int)floor);
			if (RMD_UNLIKELY(item < floor)) { // (never reached)
				report_fault(P, kFaultWorkLoop, item);
				break;
			}
#endif
			if constexpr (QUEUED) render_wave_queued(P, kernarg_params, objs, grids, work, lobjs, lds_masks, wave_lds, item);
			else if constexpr (kSortedTrips<MODE, GRID>) render_wave_sorted(P, kernarg_params, objs, grids, work, out, lobjs, wave_lds, item);
			else render_wave<MODE, GRID, CHAIN>(P, kernarg_params, objs, grids, work, out, path_obj, path_sub, lobjs, lds_masks, wave_lds, item);
		}
	} else {
		const uint32_t unit = work_item_of_block(blockIdx.x, gridDim.x) * waves_per_wg + wave;
		if constexpr (kSortedTrips<MODE, GRID>) render_wave_sorted(P, kernarg_params, objs, grids, work, out, lobjs, wave_lds, unit);
		else render_wave<MODE, GRID>(P, kernarg_params, objs, grids, work, out, path_obj, path_sub, lobjs, lds_masks, wave_lds, MODE == kModeList ? unit * 64u : unit);
	}
}

// n_cus > 0 (tile modes of grid scenes): the persistent form, one 16-wave workgroup per CU (fewer when there are fewer work items);
// P.work_counter must point at a zeroed device word.
// `shape` (optional) receives the form the kernel was actually launched in.
template <int MODE, bool GRID>
inline hipError_t launch_render(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids, const void *work,
                                uint32_t n_waves, double *out, int32_t *path_obj, uint32_t *path_sub, uint32_t n_cus = 0, LaunchShape *shape = nullptr) {
	// LDS of a workgroup of `waves` waves of this instantiation: [object table][grid occupancy masks][per-wave area]
	[[maybe_unused]] const bool queued = kPathQueues<MODE, GRID> && P.queue_buf != nullptr; // (api.cpp provides the queues for the launches that take this form)
	auto lds_for = [&](uint32_t waves) {
		return (size_t)P.n_objects * sizeof(DevObject) + (size_t)((P.mask_words_total + 3u) & ~3u) * 4u +
		       (size_t)waves * (queued ? wave_lds_of<MODE, GRID, true>(1u) : wave_lds_of<MODE, GRID>(P.mask_words_total ? 1u : 0u));
	};
	if constexpr (MODE != kModeList) {
		uint32_t pw = kPersistWaves<MODE, GRID>;
		// a large object table (128 bytes an object) or several grids' masks leave room for fewer per-wave areas — pools of the spheres kernel,
		// walk scratch + throughputs + carried walks (6,656 bytes) of the mesh kernel: fewer waves per workgroup (the kernel takes its wave count
		// from blockDim); below 4 waves the launch runs as one wave per item
		while (pw > 4u && lds_for(pw) > kLdsBudgetBytes) pw--;
		if (n_cus != 0u && P.work_counter != nullptr && lds_for(pw) <= kLdsBudgetBytes) {
			const size_t lds = lds_for(pw);
			// short launches of scenes with grids: the instantiation whose waves chain their work items (render_wave, CHAIN)
			const bool chain = kChainItems<MODE, GRID> && P.chain_items != 0u && !queued;
			hipError_t e = hipFuncSetAttribute(queued  ? reinterpret_cast<const void *>(&render_kernel<MODE, GRID, true, false, (kPathQueues<MODE, GRID>)>)
			                                   : chain ? reinterpret_cast<const void *>(&render_kernel<MODE, GRID, true, (kChainItems<MODE, GRID>)>)
			                                           : reinterpret_cast<const void *>(&render_kernel<MODE, GRID, true>),
			                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes);
			if (e != hipSuccess) return e;
			const uint32_t wgs = (n_waves + pw - 1u) / pw, resident = n_cus * (kSortedTrips<MODE, GRID> ? RMD_SORT_WGS_PER_CU : 1u);
			if (queued)
				hipLaunchKernelGGL((render_kernel<MODE, GRID, true, false, (kPathQueues<MODE, GRID>)>), dim3(wgs < resident ? wgs : resident), dim3(64u * pw), lds, stream, P, objs,
				                   grids, work, out, path_obj, path_sub);
			else if (chain)
				hipLaunchKernelGGL((render_kernel<MODE, GRID, true, (kChainItems<MODE, GRID>)>), dim3(wgs < resident ? wgs : resident), dim3(64u * pw), lds, stream, P, objs,
				                   grids, work, out, path_obj, path_sub);
			else
				hipLaunchKernelGGL((render_kernel<MODE, GRID, true>), dim3(wgs < resident ? wgs : resident), dim3(64u * pw), lds, stream, P, objs, grids, work, out,
				                   path_obj, path_sub);
			if (shape) shape->persistent = 1u, shape->waves_per_wg = pw, shape->queued = queued ? 1u : 0u, shape->resident_waves = (wgs < resident ? wgs : resident) * pw;
			return hipGetLastError();
		}
	}
	const uint32_t wpw = render_waves_per_wg(P.n_objects, P.mask_words_total);
	const size_t lds = lds_for(wpw);
	if (lds > 64u * 1024u) { // above the default dynamic-LDS limit: opt in on the current device (cheap, and correct per device)
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<MODE, GRID, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                                   (int)kLdsBudgetBytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL((render_kernel<MODE, GRID, false>), dim3((n_waves + wpw - 1u) / wpw), dim3(64u * wpw), lds, stream, P, objs, grids, work, out,
	                   path_obj, path_sub);
	if (shape) shape->persistent = 0u, shape->waves_per_wg = wpw;
	return hipGetLastError();
}


} // namespace rmd
