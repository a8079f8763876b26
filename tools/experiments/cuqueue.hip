// Grid scenes, second schedule: one persistent workgroup per CU whose 16 waves split into TRACER waves and WALKER waves
// that talk through a ray queue in the CU's LDS.  Same arithmetic as the megakernel (kernels.hip) — the device functions are
// shared — and the same per-sample output (sample buffer + sum_kernel), hence the same frame bit for bit
// (tests/test_gpu_parity.py::test_cu_queue_mode_is_bit_identical); only who computes what, and when, differs.
//
// Why.  In the megakernel a wave that holds 64 paths walks the grid for the ~32 of them that need it and then waits for the
// longest of those walks: counters of the benchmark mesh show 7 of 64 lanes active per DDA step, and paths that wait for
// their walk inflate the number of trips by a fifth.  A walk only needs a ray (48 bytes) and returns (distance, triangle);
// nothing ties it to the lane that owns the path.  So here
//   * a TRACER wave runs the path loop of kernels.hip — next_ray (shade | primary ray), planes and spheres, classification —
//     and when a ray enters a grid's box it PARKS the path in LDS (ray, throughput, best plane/sphere hit, identity: 108
//     bytes), pushes a ticket on the CU's queue and gives the lane another sample; it picks the path up again when a walker
//     has marked it done.  A lane keeps one path in registers and one parked;
//   * a WALKER wave holds no paths: every lane pops a ticket, walks that ray with the cooperative triangle tests of
//     grid_walk.hpp, merges the hit into the parked record by Scene::intersect's rule (lexicographic minimum of distance and
//     object index), marks it done and pops the next ticket at once — the wave never waits for its longest walk, and steps in
//     rounds of at most kStepCap cells so that a long empty stretch in one lane does not hold up the tests of the others.
// Everything stays inside one CU: no HBM round trip for path state, no inter-workgroup synchronisation (workgroups only
// share an atomic work-item counter), LDS operations of one wave are performed in order, and a ticket is published by the
// last LDS write of its producer, so no fences beyond s_waitcnt are involved.
#define RMD_WITH_HIP 1
#include <hip/hip_runtime.h>

#include <cstdio>

#include "device_core.hpp"
#include "grid_walk.hpp"
#include "internal.hpp"
#include "launch.hpp"
#include "scene_split.hpp"

namespace rmd {

constexpr uint32_t kCuqWaves = 16;      // waves per workgroup = per CU (4 per SIMD at <= 128 VGPRs)
constexpr uint32_t kParkSlots = 2;     // parked paths per tracer lane
constexpr uint32_t kQueueSize = 2048;   // tickets; at most kParkSlots per tracer lane are ever queued
#ifndef RMD_CUQ_STEPCAP
#define RMD_CUQ_STEPCAP 16
#endif
constexpr uint32_t kStepCap = RMD_CUQ_STEPCAP; // cells a walker lane steps per round before the round's tests are run
constexpr uint32_t kTicketValid = 0x80000000u;
// A tracer trip costs the same for 5 runnable lanes as for 60: the wave waits (up to kTracerMaxWait short sleeps) until this many
// lanes have something to do.  Measured on the benchmark mesh: tracer trips 28.1 M -> 17.3 M per 100-spp frame (15.5 M is the minimum).
constexpr uint32_t kTracerMinRunnable = 48, kTracerMaxWait = 200;
constexpr uint32_t kWatchdogTrips = 1u << 26; // a wave that loops this often without finishing gives up (error flag) instead of hanging

// One parked path per tracer lane, structure of arrays over the 64 lanes of the wave.
struct ParkedWave {
	double ray[6][64];   // ro.xyz rd.xyz
	double thr[3][64];   // throughput T
	double best_t[64];   // closest plane/sphere hit so far (kFMax = none); the walker merges the grids' hits into it
	int32_t best_obj[64];
	uint32_t best_sub[64];
	uint32_t pixel[64], sample[64], out_idx[64], depth_block[64]; // identity: RNG key, sample-buffer slot, depth | block << 8
	uint32_t status[64]; // 0 empty, 1 waiting for its walk, 2 walk done
};
static_assert(sizeof(ParkedWave) == 64 * (10 * 8 + 7 * 4), "ParkedWave layout");

struct CuqControl {
	uint32_t head, tail;     // tickets taken / tickets pushed (free-running)
	uint32_t tracers_alive;  // tracer waves still running; walkers leave when it reaches 0 and the queue is empty
	uint32_t error;          // watchdog
};

struct CuqLayout { // byte offsets into the dynamic LDS
	uint32_t objs, masks, control, queue, parked, scratch, total;
};
__host__ __device__ inline CuqLayout cuq_layout(uint32_t n_objects, uint32_t mask_words_total, uint32_t n_tracers) {
	CuqLayout L;
	uint32_t o = 0;
	L.objs = o, o += n_objects * (uint32_t)sizeof(DevObject);
	L.masks = o, o += ((mask_words_total + 3u) & ~3u) * 4u;
	L.control = o, o += 64u;
	L.queue = o, o += kQueueSize * 4u;
	o = (o + 15u) & ~15u;
	L.parked = o, o += n_tracers * kParkSlots * (uint32_t)sizeof(ParkedWave);
	o = (o + 15u) & ~15u;
	L.scratch = o, o += (kCuqWaves - n_tracers) * (uint32_t)sizeof(WalkScratch);
	L.total = o;
	return L;
}

// LDS words shared between waves: accessed through the LDS address space (ds_ instructions) with relaxed workgroup-scope atomics
typedef __attribute__((address_space(3))) uint32_t lds_u32;
RMD_DEV uint32_t lds_add(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RMD_DEV uint32_t lds_load(const uint32_t *p) { return __hip_atomic_load((const lds_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RMD_DEV void lds_store(uint32_t *p, uint32_t v) { __hip_atomic_store((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RMD_DEV bool lds_cas(uint32_t *p, uint32_t expected, uint32_t desired) {
	return __hip_atomic_compare_exchange_strong((lds_u32 *)p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
RMD_DEV uint32_t bcast_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
RMD_DEV uint32_t lane_rank(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }

// ---------------------------------------------------------------- walker
// Per-lane DDA state of AccGrid::intersects (acc_grid.rs:89-185), as in grid_walk.hpp.
struct Dda {
	uint32_t idx, remx, remy, remz;
	int32_t dix, diy, diz;
	double tmx, tmy, tmz, tdx, tdy, tdz;
};
// acc_grid.rs:90-131: bounding-box test, first cell, DDA increments.  Returns false where the walk returns None at once.
RMD_DEV bool dda_setup(const DevGrid &g, V3 ro, V3 rd, Dda &s) {
	const int32_t rx = (int32_t)g.res[0], ry = (int32_t)g.res[1], rz = (int32_t)g.res[2];
	const uint64_t resx = g.res[0], resz = g.res[2], n_cells = g.n_cells;
	V3 bmin = ld3(g.bbox_min);
	double t_outer;
	if (!aabb_intersect(bmin, ld3(g.bbox_max), ro, rd, t_outer)) return false;
	int32_t cx = 0, cy = 0, cz = 0;
	V3 cs = ld3(g.cell_size);
	V3 start = ro - bmin;
	bool ok = cast_i32(start.x / cs.x, cx) && cast_i32(start.y / cs.y, cy) && cast_i32(start.z / cs.z, cz);
	if (ok && (cx < 0 || cy < 0 || cz < 0)) {
		V3 outer_pos = ro + rd * t_outer;
		start = outer_pos - bmin;
		ok = cast_i32(start.x / cs.x, cx) && cast_i32(start.y / cs.y, cy) && cast_i32(start.z / cs.z, cz);
	}
	ok = ok && !(rd.x != rd.x || rd.y != rd.y || rd.z != rd.z); // signum(NaN).cast::<i32>() panics: miss
	if (!ok) return false;
	// first cell (:128-131): `as usize` sign-extends and the index arithmetic wraps (release build)
	const uint64_t idx0 = (uint64_t)(int64_t)cx + resx * ((uint64_t)(int64_t)cy + (uint64_t)(int64_t)cz * resz);
	if (!(idx0 < n_cells)) return false; // else None
	s.idx = (uint32_t)idx0;
	const int32_t sx = signbit(rd.x) ? -1 : 1, sy = signbit(rd.y) ? -1 : 1, sz = signbit(rd.z) ? -1 : 1;
	s.dix = sx, s.diy = sy * (int32_t)resx, s.diz = sz * (int32_t)(resx * resz);
	s.remx = steps_to_exit(cx, sx, rx), s.remy = steps_to_exit(cy, sy, ry), s.remz = steps_to_exit(cz, sz, rz);
	s.tdx = (rd.x < 0.0 ? -cs.x : cs.x) / rd.x;
	s.tdy = (rd.y < 0.0 ? -cs.y : cs.y) / rd.y;
	s.tdz = (rd.z < 0.0 ? -cs.z : cs.z) / rd.z;
	s.tmx = (((double)(cx + (rd.x < 0.0 ? 0 : 1)) * cs.x) - start.x) / rd.x;
	s.tmy = (((double)(cy + (rd.y < 0.0 ? 0 : 1)) * cs.y) - start.y) / rd.y;
	s.tmz = (((double)(cz + (rd.z < 0.0 ? 0 : 1)) * cs.z) - start.z) / rd.z;
	return true;
}

// One round for the lanes in `walkers` (they all walk grid g): up to kStepCap cells each, collecting candidate cells, then the
// cooperative triangle tests of the round.  walking is cleared for lanes whose walk has ended; found/found_t/found_tri are set
// for those that ended with a hit.  Must be called by all 64 lanes.
RMD_DEV void walker_round(const DevGrid &g, const uint32_t *lds_mask, WalkScratch &scr, bool walkers, V3 ro, V3 rd, Dda &s, bool &walking, bool &found,
                          double &found_t, uint32_t &found_tri) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t mask_shift = g.mask_shift;
	const uint32_t mask_pad_bit = g.mask_n_words * 32u - 32u;
	const uint32_t idx_limit = (uint32_t)g.n_cells;
	const RMD_GLOBAL CellEntry *entries = as_global(g.cell_entries);
	const RMD_GLOBAL unsigned char *runs = as_global(reinterpret_cast<const unsigned char *>(g.tri_runs));
	// 1. stepping (grid_walk.hpp step 1, with a cap on the cells per round)
	uint32_t n_cand = 0, budget = kStepCap;
	bool go = walkers && walking;
	while (go && budget != 0u) {
		uint32_t bit = s.idx >> mask_shift;
		bit = bit < mask_pad_bit ? bit : mask_pad_bit;
		const uint32_t word = lds_mask[bit >> 5];
		const uint32_t here = s.idx;
		{
			unsigned long long m_xy, m_xz, saved;
			asm volatile("v_cmp_lt_f64 %[mxy], %[tmx], %[tmy]\n\t"
			             "v_cmp_lt_f64 %[mxz], %[tmx], %[tmz]\n\t"
			             "v_cmp_lt_f64 vcc, %[tmy], %[tmz]\n\t"
			             "s_mov_b64 %[sv], exec\n\t"
			             "s_and_b64 %[mxz], %[mxy], %[mxz]\n\t"
			             "s_andn2_b64 vcc, vcc, %[mxy]\n\t"
			             "s_mov_b64 exec, %[mxz]\n\t"
			             "v_add_f64 %[tmx], %[tmx], %[tdx]\n\t"
			             "v_add_u32 %[rx], -1, %[rx]\n\t"
			             "v_add_u32 %[idx], %[idx], %[dix]\n\t"
			             "s_mov_b64 exec, vcc\n\t"
			             "v_add_f64 %[tmy], %[tmy], %[tdy]\n\t"
			             "v_add_u32 %[ry], -1, %[ry]\n\t"
			             "v_add_u32 %[idx], %[idx], %[diy]\n\t"
			             "s_or_b64 vcc, vcc, %[mxz]\n\t"
			             "s_andn2_b64 exec, %[sv], vcc\n\t"
			             "v_add_f64 %[tmz], %[tmz], %[tdz]\n\t"
			             "v_add_u32 %[rz], -1, %[rz]\n\t"
			             "v_add_u32 %[idx], %[idx], %[diz]\n\t"
			             "s_mov_b64 exec, %[sv]"
			             : [tmx] "+v"(s.tmx), [tmy] "+v"(s.tmy), [tmz] "+v"(s.tmz), [rx] "+v"(s.remx), [ry] "+v"(s.remy), [rz] "+v"(s.remz), [idx] "+v"(s.idx),
			               [mxy] "=&s"(m_xy), [mxz] "=&s"(m_xz), [sv] "=&s"(saved)
			             : [tdx] "v"(s.tdx), [tdy] "v"(s.tdy), [tdz] "v"(s.tdz), [dix] "v"(s.dix), [diy] "v"(s.diy), [diz] "v"(s.diz)
			             : "vcc", "scc");
		}
		const uint32_t rem_min = s.remx < s.remy ? (s.remx < s.remz ? s.remx : s.remz) : (s.remy < s.remz ? s.remy : s.remz);
		go = rem_min != 0u && s.idx < idx_limit; // left the grid, or the next cell is past the cell array: the walk returns None
		budget--;
		if ((word >> (bit & 31u)) & 1u) {
			scr.first[n_cand * 64u + lane] = here;
			n_cand++;
			budget = n_cand == kWalkCand ? 0u : (budget < kWalkLookahead ? budget : kWalkLookahead);
		}
	}
	if (walkers && walking) walking = go;
	if (__ballot(n_cand != 0u) == 0ull) return;

	// 2. the candidates' cell entries
	uint32_t c_first[kWalkCand], c_count[kWalkCand];
	{
		uint32_t ci[kWalkCand];
#pragma unroll
		for (uint32_t m = 0; m < kWalkCand; m++) ci[m] = scr.first[m * 64u + lane];
#pragma unroll
		for (uint32_t m = 0; m < kWalkCand; m++) ci[m] = m < n_cand ? ci[m] : 0u;
#pragma unroll
		for (uint32_t m = 0; m < kWalkCand; m++) {
			const unsigned long long e = reinterpret_cast<const RMD_GLOBAL unsigned long long *>(entries)[ci[m]];
			c_first[m] = (uint32_t)e, c_count[m] = (uint32_t)(e >> 32);
		}
#pragma unroll
		for (uint32_t m = 0; m < kWalkCand; m++) c_count[m] = m < n_cand ? c_count[m] : 0u;
	}
	// 3. triangle tests, distributed over the whole wave (grid_walk.hpp step 3)
	uint32_t my_tests = 0;
#pragma unroll
	for (uint32_t m = 0; m < kWalkCand; m++) my_tests += c_count[m];
	const uint32_t incl_t = wave_scan_add(my_tests);
	const uint32_t total = readlane_u32(incl_t, 63);
	const uint32_t my_begin = incl_t - my_tests, my_end = incl_t;
	bool any = false;
	uint32_t any_slot = 0, closest_tri = 0;
	double closest = 5712515.0;
	if (total != 0u) {
		{
			uint32_t run = my_begin;
#pragma unroll
			for (uint32_t m = 0; m < kWalkCand; m++) {
				scr.start[lane * kWalkCand + m] = run, scr.first[lane * kWalkCand + m] = c_first[m];
				run += c_count[m];
			}
		}
		for (uint32_t base = 0; base < total; base += 64u) {
			const uint32_t w = base + lane;
			scr.marker[lane] = 0u;
			if (my_tests != 0u && my_begin < base + 64u && my_end > base) scr.marker[umax(my_begin, base) - base] = lane + 1u;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const uint32_t owner_lane = wave_scan_max(scr.marker[lane]) - 1u;
			bool h = false;
			double t = 0.0;
			uint32_t tri = 0, own = 0, rec_index = 0;
			if (w < total) {
				uint32_t slot = 0, slot_start = scr.start[owner_lane * kWalkCand];
#pragma unroll
				for (uint32_t m = 1; m < kWalkCand; m++) {
					const uint32_t sm = scr.start[owner_lane * kWalkCand + m];
					if (sm <= w) slot = m, slot_start = sm;
				}
				own = owner_lane | (slot << 8);
				rec_index = scr.first[owner_lane * kWalkCand + slot] + (w - slot_start);
			}
			const int src = (int)((own & 63u) << 2);
			const V3 pro = mk(bperm_f64(src, ro.x), bperm_f64(src, ro.y), bperm_f64(src, ro.z));
			const V3 prd = mk(bperm_f64(src, rd.x), bperm_f64(src, rd.y), bperm_f64(src, rd.z));
			if (w < total) {
				const RMD_GLOBAL unsigned char *rec = runs + (size_t)rec_index * 80u;
				const TriRecord r = load_record(rec);
				tri = r.tri;
				h = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, t);
			}
			unsigned long long hits = __ballot(h);
			while (hits) {
				const int l = (int)__builtin_ctzll(hits);
				hits &= hits - 1ull;
				const uint32_t ol = readlane_u32(own, l);
				const double tl = readlane_f64(t, l);
				const uint32_t tril = readlane_u32(tri, l);
				const uint32_t slot = ol >> 8;
				if (lane == (ol & 63u) && (!any || slot == any_slot)) {
					if (tl < closest) closest = tl, closest_tri = tril, any_slot = slot, any = true;
				}
			}
		}
		__builtin_amdgcn_wave_barrier(); // the scratch is rewritten next round
	}
	if (any) found = true, found_t = closest, found_tri = closest_tri, walking = false; // :151-153 first cell with any hit wins
}

// ---------------------------------------------------------------- the kernel
__global__ __launch_bounds__(64 * kCuqWaves) void render_kernel_cuq(RenderParams P, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                                                                     const WaveTile *__restrict__ wave_tiles, uint32_t *__restrict__ work_counter,
                                                                     uint32_t n_tracers, uint32_t *__restrict__ error_flag, unsigned long long *__restrict__ dbg) {
	constexpr uint32_t min_runnable = kTracerMinRunnable, max_wait = kTracerMaxWait;
	extern __shared__ __align__(16) unsigned char smem[];
	const CuqLayout L = cuq_layout(P.n_objects, P.mask_words_total, n_tracers);
	DevObject *lobjs = reinterpret_cast<DevObject *>(smem + L.objs);
	uint32_t *lmasks = reinterpret_cast<uint32_t *>(smem + L.masks);
	CuqControl *ctl = reinterpret_cast<CuqControl *>(smem + L.control);
	uint32_t *queue = reinterpret_cast<uint32_t *>(smem + L.queue);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	{
		const double *src = reinterpret_cast<const double *>(objs);
		double *dst = reinterpret_cast<double *>(lobjs);
		for (uint32_t i = tid; i < P.n_objects * 16u; i += blockDim.x) dst[i] = src[i];
		for (uint32_t gi = 0; gi < P.n_grids; gi++) {
			const DevGrid &g = grids[gi];
			if (g.mask_lds_word == 0xFFFFFFFFu) continue;
			for (uint32_t i = tid; i < g.mask_n_words; i += blockDim.x) lmasks[g.mask_lds_word + i] = as_global(g.mask_words)[i];
		}
		for (uint32_t i = tid; i < kQueueSize; i += blockDim.x) queue[i] = 0u;
		if (tid == 0) ctl->head = 0u, ctl->tail = 0u, ctl->tracers_alive = n_tracers, ctl->error = 0u;
		for (uint32_t w = 0; w < n_tracers * kParkSlots; w++) {
			ParkedWave *pw = reinterpret_cast<ParkedWave *>(smem + L.parked) + w;
			if (tid < 64u) pw->status[tid] = 0u;
		}
	}
	__syncthreads(); // the only workgroup barrier
	// waves 0, 1, 2, ... alternate between the two roles so that every SIMD hosts both kinds (wave w runs on SIMD w % 4)
	// tracer k = the k-th wave with role tracer
	bool is_tracer;
	uint32_t role_index;
	{
		// spread: tracer iff floor((wave + 1) * n_tracers / 16) > floor(wave * n_tracers / 16)
		const uint32_t a = (wave * n_tracers) / kCuqWaves, b = ((wave + 1u) * n_tracers) / kCuqWaves;
		is_tracer = b > a;
		role_index = is_tracer ? a : wave - a;
	}

	if (!is_tracer) {
		// ================================================================ WALKER
		WalkScratch &scr = *(reinterpret_cast<WalkScratch *>(smem + L.scratch) + role_index);
		bool has_ray = false, walking = false, found = false, pending_setup = false;
		uint32_t owner = 0;  // parked record (tracer * kParkSlots + slot) << 6 | lane
		uint32_t gobj = 0;   // object index of the grid being walked
		V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
		Dda dda = {};
		double best_t = kFMax, found_t = 0.0;
		int best_obj = -1;
		uint32_t best_sub = 0, found_tri = 0;
		unsigned long long c_trips = 0, c_idle = 0, c_rays = 0, c_rounds = 0, c_round_lanes = 0, c_have = 0, c_fin = 0, c_invalid = 0, c_taken = 0;
		for (uint32_t trip = 0;; trip++) {
			if (trip == kWatchdogTrips) {
				if (lane == 0) lds_store(&ctl->error, 1u), atomicExch(error_flag, 1u);
				break;
			}
			c_trips++;
			// ---- refill: lanes without a ray take tickets
			const unsigned long long need = __ballot(!has_ray);
			if (need != 0ull) {
				uint32_t base = 0, n = 0;
				if (lane == 0) {
					const uint32_t want_n = (uint32_t)__popcll(need);
					for (;;) {
						const uint32_t h = lds_load(&ctl->head), t = lds_load(&ctl->tail);
						const uint32_t avail = t - h;
						n = avail < want_n ? avail : want_n;
						if (n == 0u) break;
						if (lds_cas(&ctl->head, h, h + n)) {
							base = h;
							break;
						}
					}
				}
				base = bcast_first(base), n = bcast_first(n);
				c_rays += n;
				const uint32_t r = lane_rank(need);
				if (!has_ray && r < n) {
					uint32_t *slot = &queue[(base + r) & (kQueueSize - 1u)];
					uint32_t e = 0;
					for (uint32_t spin = 0; spin < (1u << 24); spin++) { // the producer reserved this slot before it advanced the tail's count
						e = lds_load(slot);
						if (e & kTicketValid) break;
					}
					// (The compare below is made on a fresh copy of e: hipcc 7.2 otherwise reuses the VCC of the spin loop's last
					// iteration for it, which holds zeros for the lanes that left the loop in an earlier iteration.)
					asm volatile("" : "+v"(e));
					lds_store(slot, 0u);
					owner = e & 0xFFFFu;
					const ParkedWave *pw = reinterpret_cast<const ParkedWave *>(smem + L.parked) + (owner >> 6);
					const uint32_t ol = owner & 63u;
					ro = mk(pw->ray[0][ol], pw->ray[1][ol], pw->ray[2][ol]);
					rd = mk(pw->ray[3][ol], pw->ray[4][ol], pw->ray[5][ol]);
					best_t = pw->best_t[ol], best_obj = pw->best_obj[ol], best_sub = 0u;
					has_ray = (e & kTicketValid) != 0u;
					gobj = 0u, pending_setup = true, walking = false, found = false;
				}
				c_taken += (unsigned long long)__popcll(__ballot(!(need & (1ull << lane)) ? false : (r < n)));
				c_invalid += (unsigned long long)__popcll(__ballot(((need >> lane) & 1ull) && r < n && !has_ray));
			}
			if (__ballot(has_ray) == 0ull) {
				if (lds_load(&ctl->tracers_alive) == 0u && lds_load(&ctl->head) == lds_load(&ctl->tail)) break;
				__builtin_amdgcn_s_sleep(8);
				c_idle++;
				continue;
			}
			// ---- set-up: a lane's next grid object (uniform loop over the objects; scenes have one or two grids)
			if (__ballot(pending_setup) != 0ull) {
				for (uint32_t i = 0; i < P.n_objects; i++) {
					const DevObject &o = objs[i];
					if (o.geometry_kind != 2u) continue; // uniform
					const bool mine = has_ray && pending_setup && gobj <= i;
					if (__ballot(mine) == 0ull) continue;
					if (mine) {
						gobj = i;
						walking = dda_setup(grids[o.grid_index], ro, rd, dda);
						found = false;
						pending_setup = !walking; // a ray that misses this grid's box goes on to the next grid object
						if (!walking) gobj = i + 1u;
					}
				}
				// lanes still pending have no grid object left: their ticket is finished below
			}
			// ---- one round on the lowest grid object any lane is walking
			uint32_t g_cur = 0xFFFFFFFFu;
			{
				unsigned long long wm = __ballot(has_ray && walking);
				// wave-uniform minimum of gobj over the walking lanes
				while (wm) {
					const int l = (int)__builtin_ctzll(wm);
					wm &= wm - 1ull;
					const uint32_t v = readlane_u32(gobj, l);
					g_cur = v < g_cur ? v : g_cur;
				}
			}
			c_have += (unsigned long long)__popcll(__ballot(has_ray));
			if (g_cur != 0xFFFFFFFFu) {
				c_rounds++, c_round_lanes += (unsigned long long)__popcll(__ballot(has_ray && walking && gobj == g_cur));
				const DevGrid &g = grids[objs[g_cur].grid_index];
				walker_round(g, lmasks + g.mask_lds_word, scr, has_ray && walking && gobj == g_cur, ro, rd, dda, walking, found, found_t, found_tri);
			}
			// ---- lanes whose walk of the current grid object has ended: merge, then the next grid object or the finished ticket
			if (has_ray && !walking && !pending_setup) {
				if (found && lex_less(found_t, (int)gobj, best_t, best_obj)) best_t = found_t, best_obj = (int)gobj, best_sub = found_tri;
				found = false;
				gobj++;
				pending_setup = true; // looks for a further grid object on the next trip
			}
			bool fin_now = false;
			if (has_ray && pending_setup) {
				// any grid object at index >= gobj left?  (uniform scan, cheap: a handful of objects)
				bool more = false;
				for (uint32_t i = 0; i < P.n_objects; i++)
					if (objs[i].geometry_kind == 2u && i >= gobj) more = true;
				if (!more) {
					ParkedWave *pw = reinterpret_cast<ParkedWave *>(smem + L.parked) + (owner >> 6);
					const uint32_t ol = owner & 63u;
					pw->best_t[ol] = best_t, pw->best_obj[ol] = best_obj, pw->best_sub[ol] = best_sub;
					__builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the record is written before it is marked done
					lds_store(&pw->status[ol], 2u);
					has_ray = false, pending_setup = false;
					fin_now = true;
				}
			}
			c_fin += (unsigned long long)__popcll(__ballot(fin_now));
		}
		if (dbg && lane == 0) atomicAdd(&dbg[0], c_trips), atomicAdd(&dbg[1], c_idle), atomicAdd(&dbg[2], c_rays), atomicAdd(&dbg[3], c_rounds), atomicAdd(&dbg[4], c_round_lanes), atomicAdd(&dbg[5], c_have), atomicAdd(&dbg[18], c_fin), atomicAdd(&dbg[19], c_invalid), atomicAdd(&dbg[20], c_taken);
		return;
	}

	// ================================================================ TRACER
	ParkedWave *const parks = reinterpret_cast<ParkedWave *>(smem + L.parked) + role_index * kParkSlots; // this wave's kParkSlots records per lane
	const V3 cam_pos = ld3(P.cam_pos);
	// this wave's work item: a wave tile and a sample sub-range (api.cpp: choose_split), taken from the launch-wide counter
	const uint32_t split = P.split_k > 1u ? P.split_k : 1u;
	const uint32_t n_items = P.n_work * split;
	WaveTile tile = {};
	uint32_t wt = 0, pool_first = 0, pool_items = 0, next_item = 0; // wave-uniform
	bool work_left = true;                                          // wave-uniform
	// The path a lane holds in registers is in exactly one of these states:
	//   need_sample   none (the lane takes the next sample, or its parked path once that is done)
	//   to_shade      a classified hit; (B) turns it into the bounce ray
	//   new_ray       a ray made in (B) of this trip, intersected in (A) of this trip
	//   blocked       a ray that needs a walk (planes and spheres done) while the lane's parked record is still taken
	bool need_sample = true, to_shade = false, new_ray = false, blocked = false;
	bool parked[kParkSlots] = {}; // this lane's record k is in use (its path waits for a walk, or the walk is done)
	Rng rng;
	rng.init(0u, 0u);
	V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1), T = mk(1.0, 1.0, 1.0);
	uint32_t depth = 1, out_idx = 0;
	double part_t = kFMax; // closest hit of the register path: planes and spheres, then merged with the grids' by the walker
	int part_obj = -1;
	uint32_t part_sub = 0;
	int hit_obj = -1; // the hit a to_shade lane will shade
	double hit_t = 0.0;
	V3 hit_normal = mk(0.0, 0.0, 1.0);

	auto store_record = [&](ParkedWave &park) { // the register path -> one of this lane's parked records, marked as waiting for its walk
		park.ray[0][lane] = ro.x, park.ray[1][lane] = ro.y, park.ray[2][lane] = ro.z;
		park.ray[3][lane] = rd.x, park.ray[4][lane] = rd.y, park.ray[5][lane] = rd.z;
		park.thr[0][lane] = T.x, park.thr[1][lane] = T.y, park.thr[2][lane] = T.z;
		park.best_t[lane] = part_t, park.best_obj[lane] = part_obj, park.best_sub[lane] = 0u;
		park.pixel[lane] = rng.pixel, park.sample[lane] = rng.sample, park.out_idx[lane] = out_idx, park.depth_block[lane] = depth | (rng.block << 8);
		lds_store(&park.status[lane], 1u);
	};

	unsigned long long t_trips = 0, t_idle = 0, t_blocked = 0, t_progress = 0, t_shade = 0, t_push = 0, t_unpark = 0;
	for (uint32_t trip = 0;; trip++) {
		if (trip == kWatchdogTrips) {
			if (lane == 0) lds_store(&ctl->error, 1u), atomicExch(error_flag, 1u);
			if (dbg) {
				const uint32_t st = lds_load(&parks[0].status[lane]);
				const unsigned long long s1 = __ballot(parked[0] && st == 1u), s2 = __ballot(parked[0] && st == 2u), s0 = __ballot(parked[0] && st == 0u), nb = __ballot(blocked), ns = __ballot(need_sample);
				if (lane == 0) atomicAdd(&dbg[6], (unsigned long long)__popcll(s1)), atomicAdd(&dbg[7], (unsigned long long)__popcll(s2)), atomicAdd(&dbg[15], (unsigned long long)__popcll(s0)),
					atomicAdd(&dbg[13], (unsigned long long)__popcll(nb)), atomicAdd(&dbg[14], (unsigned long long)__popcll(ns));
			}
			break;
		}
		// A trip costs the same for 5 runnable lanes as for 60, and what it does not spend the walkers on the same SIMD can use:
		// wait (briefly) until enough lanes have something to do — a hit to shade, a finished walk to pick up, a sample to start.
		for (uint32_t waited = 0; waited < max_wait; waited++) {
			bool done_record = false;
			if (need_sample || blocked) {
#pragma unroll
				for (uint32_t k = 0; k < kParkSlots; k++) done_record = done_record || (parked[k] && lds_load(&parks[k].status[lane]) == 2u);
			}
			const bool runnable = to_shade || done_record || (need_sample && (next_item < pool_items || work_left));
			if ((uint32_t)__popcll(__ballot(runnable)) >= min_runnable) break;
			__builtin_amdgcn_s_sleep(2);
		}
		t_trips++, t_blocked += (unsigned long long)__popcll(__ballot(blocked)), t_shade += (unsigned long long)__popcll(__ballot(to_shade));
		bool progressed = false;
		bool complete = false; // lanes whose closest hit is known on this trip
		bool push = false;     // lanes that put a path into their record on this trip: a ticket each
		// ---------------- (U) a lane without a runnable register path takes a parked path back once its walk is done
		uint32_t push_slot = 0; // the record a pushing lane has just filled
		if (need_sample || blocked) {
			int take = -1;
#pragma unroll
			for (uint32_t k = 0; k < kParkSlots; k++)
				if (take < 0 && parked[k] && lds_load(&parks[k].status[lane]) == 2u) take = (int)k;
			if (take >= 0) {
				ParkedWave &park = parks[take];
				const V3 pro = mk(park.ray[0][lane], park.ray[1][lane], park.ray[2][lane]), prd = mk(park.ray[3][lane], park.ray[4][lane], park.ray[5][lane]);
				const V3 pT = mk(park.thr[0][lane], park.thr[1][lane], park.thr[2][lane]);
				const double pt = park.best_t[lane];
				const int pobj = park.best_obj[lane];
				const uint32_t psub = park.best_sub[lane], ppix = park.pixel[lane], psam = park.sample[lane], pout = park.out_idx[lane], pdb = park.depth_block[lane];
				if (blocked) { // swap: the blocked path takes the record's place
					store_record(park);
					push = true, push_slot = (uint32_t)take, blocked = false;
				} else {
					lds_store(&park.status[lane], 0u);
#pragma unroll
					for (uint32_t k = 0; k < kParkSlots; k++)
						if ((int)k == take) parked[k] = false;
				}
				ro = pro, rd = prd, T = pT;
				part_t = pt, part_obj = pobj, part_sub = psub;
				rng.pixel = ppix, rng.sample = psam, out_idx = pout, depth = pdb & 0xFFu, rng.block = pdb >> 8;
				complete = true, need_sample = false;
			}
		}
		t_unpark += (unsigned long long)__popcll(__ballot(complete));
		// ---------------- (B) hand out samples, then rays
		bool prim = false;
		uint32_t px = 0, py = 0;
		{
			const unsigned long long idle = __ballot(need_sample);
			if (idle != 0ull) {
				if (next_item >= pool_items && work_left) { // uniform: this wave's work item is used up, take the next one
					uint32_t item_id = 0;
					if (lane == 0) item_id = atomicAdd(work_counter, 1u);
					item_id = bcast_first(item_id);
					if (item_id < n_items) {
						wt = item_id / split;
						const uint32_t part = item_id % split;
						tile = wave_tiles[wt];
						const uint32_t per_part = (P.sample_count + split - 1u) / split;
						const uint32_t s_lo = part * per_part < P.sample_count ? part * per_part : P.sample_count;
						const uint32_t s_hi = s_lo + per_part < P.sample_count ? s_lo + per_part : P.sample_count;
						pool_first = s_lo, pool_items = (s_hi - s_lo) * 64u, next_item = 0u;
					} else {
						work_left = false, pool_items = 0u, next_item = 0u;
					}
				}
				if (next_item < pool_items) {
					const uint32_t k = next_item + lane_rank(idle);
					next_item += (uint32_t)__popcll(idle);
					if (need_sample && k < pool_items && (k & 7u) < tile.w && ((k >> 3) & 7u) < tile.h) { // slots outside a ragged tile are skipped
						prim = true;
						px = tile.x0 + (k & 7u), py = tile.y0 + ((k >> 3) & 7u);
						rng.init(py * P.W + px, P.sample_begin + pool_first + (k >> 6)); // src/trace.rs:199 — primary ray of this sample
						out_idx = (wt * P.sample_count + pool_first + (k >> 6)) * 64u + (k & 63u);
						depth = 1;
						need_sample = false, new_ray = true;
					}
				}
			}
		}
		bool lens_failed = false;
		if (P.use_dof) { // thin lens (:335-360): a variable number of blocks; not merged with the shading stream
			if (prim) {
				T = mk(1.0, 1.0, 1.0);
				lens_failed = !primary_ray_dof(P, px, py, rng, ro, rd); // the reference panics there; the sample contributes zero
				if (lens_failed) new_ray = false;
			}
			prim = false;
		}
		NextRayShadeIn hit;
		hit.normal = hit_normal, hit.frag = ro, hit.color = ro, hit.roughness = 0.0, hit.metal = 0.0;
		if (to_shade) {
			const DevObject &o = lobjs[hit_obj];
			hit.frag = ro + rd * hit_t; // :246, the same operations as at classification
			hit.color = ld3(o.color), hit.roughness = o.roughness, hit.metal = o.metalness;
		}
		next_ray(P, to_shade, prim, hit, cam_pos, px, py, rng, ro, rd, T);
		bool cut = false; // shaded at the bounce limit (non-finite inputs, kernels.hip): the recursive call returns 0 unintersected
		if (to_shade) {
			depth++;
			to_shade = false;
			if (depth > P.bounce_limit) cut = true;
			else new_ray = true;
			progressed = true;
		}
		// ---------------- (A) planes and spheres for the new rays; a ray that enters a grid's box is parked for the walkers
		if (new_ray) {
			const bool enters = intersect_simple(objs, P.n_objects, grids, true, ro, rd, part_t, part_obj);
			part_sub = 0u, new_ray = false;
			progressed = true;
			if (!enters) complete = true;
			else {
				int free_slot = -1;
#pragma unroll
				for (uint32_t k = 0; k < kParkSlots; k++)
					if (free_slot < 0 && !parked[k]) free_slot = (int)k;
				if (free_slot >= 0) {
					store_record(parks[free_slot]);
#pragma unroll
					for (uint32_t k = 0; k < kParkSlots; k++)
						if ((int)k == free_slot) parked[k] = true;
					push = true, push_slot = (uint32_t)free_slot, need_sample = true; // the lane goes on with another sample
				} else {
					blocked = true; // all its records are taken: the lane waits for one of those walks, then swaps (U)
				}
			}
		}
		// tickets: the record is in LDS before its ticket is (LDS operations of a wave are performed in order)
		{
			const unsigned long long pushers = __ballot(push);
			t_push += (unsigned long long)__popcll(pushers);
			if (pushers != 0ull) {
				uint32_t base = 0;
				if (lane == 0) base = lds_add(&ctl->tail, (uint32_t)__popcll(pushers));
				base = bcast_first(base);
				if (push) {
					__builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
					lds_store(&queue[(base + lane_rank(pushers)) & (kQueueSize - 1u)], kTicketValid | ((role_index * kParkSlots + push_slot) << 6) | lane);
				}
			}
		}
		// ---------------- classification (src/trace.rs:239-252) of the lanes whose closest hit is known
		bool terminal = lens_failed || cut;
		V3 Lr = mk(0.0, 0.0, 0.0);
		if (complete) {
			progressed = true;
			const int oi = part_obj;
			const double t = part_t;
			if (oi < 0) {
				terminal = true; // :242 miss -> radiance 0
			} else {
				const DevObject &o = lobjs[oi];
				const V3 frag = ro + rd * t; // :246
				if (o.material_kind == 2u) {
					Lr = ld3(o.color); // :250-252 Emission
					terminal = true;
				} else {
					V3 normal;
					if (o.geometry_kind == 0u) normal = ld3(o.normal);                        // plane.rs:28-32
					else if (o.geometry_kind == 1u) normal = normalize(frag - ld3(o.origin)); // sphere.rs:31-35
					else {
						const DevGrid &g = grids[o.grid_index];
						normal = triangle_normal(as_global(g.tri_pos) + (size_t)part_sub * 9, as_global(g.tri_nrm) + (size_t)part_sub * 9, as_global(g.tri_aux) + (size_t)part_sub * 4, frag); // acc_grid.rs:85-87
					}
					// the last depth's weight multiplies the zero of the cut-off recursion (kernels.hip): not evaluated when its inputs are finite
					const double probe_sum = ((normal.x + normal.y) + normal.z) + ((frag.x + frag.y) + frag.z);
					const bool finite_inputs = __builtin_fabs(probe_sum) < __builtin_inf();
					if (depth == P.bounce_limit && finite_inputs) terminal = true;
					else hit_normal = normal, hit_obj = oi, hit_t = t, to_shade = true;
				}
			}
		}
		if (terminal) {
			progressed = true;
			Lr = hadamard(T, Lr);
			double *dst = P.sample_buf + (size_t)out_idx * kSampleStride;
			dst[0] = Lr.x, dst[1] = Lr.y, dst[2] = Lr.z;
			need_sample = true;
		}
		// ---------------- done?  nothing in registers, nothing parked, no work left
		bool any_parked = false;
#pragma unroll
		for (uint32_t k = 0; k < kParkSlots; k++) any_parked = any_parked || parked[k];
		if (__ballot(!need_sample || any_parked) == 0ull && !work_left && next_item >= pool_items) break;
		t_progress += (unsigned long long)__popcll(__ballot(progressed));
		if (__ballot(progressed) == 0ull) {
			__builtin_amdgcn_s_sleep(4); // every lane of the wave waits for a walker
			t_idle++;
		}
	}
	if (dbg && lane == 0) atomicAdd(&dbg[8], t_trips), atomicAdd(&dbg[9], t_idle), atomicAdd(&dbg[10], t_blocked), atomicAdd(&dbg[11], t_progress), atomicAdd(&dbg[12], t_shade), atomicAdd((unsigned long long *)&dbg[16], t_push), atomicAdd((unsigned long long *)&dbg[17], t_unpark);
	if (lane == 0) __hip_atomic_fetch_sub((lds_u32 *)&ctl->tracers_alive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------- launcher
size_t cuq_workspace_bytes() { return 64 + 64 * sizeof(unsigned long long); }

hipError_t launch_render_cuq(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids, const WaveTile *wave_tiles,
                             double *accum, void *workspace, uint32_t n_cus, uint32_t n_tracers) {
	if (P.n_work == 0 || P.sample_count == 0) return hipSuccess;
	if (n_tracers == 0 || n_tracers >= kCuqWaves) n_tracers = 7u; // measured best of 4..13 on the benchmark mesh
	while (n_tracers > 1u && cuq_layout(P.n_objects, P.mask_words_total, n_tracers).total > kLdsBudgetBytes) n_tracers--; // parked records are the big item
	const CuqLayout L = cuq_layout(P.n_objects, P.mask_words_total, n_tracers);
	if (L.total > kLdsBudgetBytes) return hipErrorInvalidValue;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel_cuq), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes);
	if (e != hipSuccess) return e;
	uint32_t *counter = static_cast<uint32_t *>(workspace);
	e = hipMemsetAsync(counter, 0, cuq_workspace_bytes(), stream);
	if (e != hipSuccess) return e;
	const uint32_t n_items = P.n_work * (P.split_k > 1u ? P.split_k : 1u);
	const uint32_t wgs = n_items < n_cus * n_tracers ? (n_items + n_tracers - 1u) / n_tracers : n_cus;
	hipLaunchKernelGGL(render_kernel_cuq, dim3(wgs), dim3(64u * kCuqWaves), L.total, stream, P, objs, grids, wave_tiles, counter, n_tracers, counter + 1,
	                   (RMD_DIAG && (P.debug_flags & 8u)) ? reinterpret_cast<unsigned long long *>(counter + 16) : nullptr);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	if (RMD_DIAG && (P.debug_flags & 8u)) { // DIAG builds with RMD_DEBUG=8: event counters of the two roles
		unsigned long long h[64];
		(void)hipStreamSynchronize(stream);
		(void)hipMemcpy(h, counter + 16, sizeof(h), hipMemcpyDeviceToHost);
		std::fprintf(stderr, "[cuq] walker: trips=%llu idle=%llu rays=%llu rounds=%llu round_lanes=%llu lanes_with_ray=%llu | tracer: trips=%llu idle=%llu blocked_lane_trips=%llu progressed_lane_trips=%llu shade_lanes=%llu\n",
		             h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10], h[11], h[12]);
		std::fprintf(stderr, "[cuq] pushes=%llu unparks=%llu | at watchdog: parked&status1=%llu status2=%llu status0=%llu blocked=%llu need_sample=%llu\n", h[16], h[17], h[6], h[7], h[15], h[13], h[14]);
		std::fprintf(stderr, "[cuq] walker finishes=%llu invalid_tickets=%llu taken=%llu\n", h[18], h[19], h[20]);
	}
	return launch_sum(stream, P, wave_tiles, accum);
}

} // namespace rmd
