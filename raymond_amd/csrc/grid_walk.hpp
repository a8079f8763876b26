// Wave-cooperative AccGrid::intersects (reference core/src/geometry/acc_grid.rs:89-185) for gfx950.
//
// The reference walks one ray through the grid cell by cell and, in every cell, tests that cell's
// triangles one after the other.  Measured on the 99k-triangle benchmark mesh a walk visits ~20 cells of
// which ~2.7 are non-empty, and a non-empty cell holds ~9-13 triangles (up to 50+).  A lane-per-ray port of
// that loop leaves ~90 % of the lanes idle and chains 2-4 dependent memory latencies per cell.  Here:
//
//   1. every lane steps ITS OWN ray's DDA (bit-identical arithmetic, :100-125 and :155-183) but skips empty
//      cells with an occupancy bitmask held in LDS — no global memory access until a candidate cell;
//   2. lanes standing on candidate cells fetch their 8-byte cell entry {first record, count} together;
//   3. for each such (lane, cell) pair in turn the WHOLE WAVE tests that cell's triangles, one triangle per
//      lane: the ray is broadcast through SGPRs (v_readlane), the cell's triangle records are contiguous
//      80-byte rows (coalesced), and the winner is picked by a scalar loop over the hit ballot in ascending
//      lane order with a strict '<' — the reference's `if distance < closest` scan (:137-149) exactly,
//      including the 5712515.0 start value and first-wins ties.
// Results are therefore identical to the sequential walk for every ray.
#pragma once
#include "device_core.hpp"

#ifndef RMD_DIAG
#define RMD_DIAG 0 // 1: compile the walk's event counters / s_memtime stamps (RMD_DEBUG=8 / 16); never in the shipped build
#endif

namespace rmd {

RMD_DEV double readlane_f64(double v, int lane) {
	unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
	unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
RMD_DEV uint32_t readlane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// One 80-byte triangle record of a cell run: v0, edge1, edge2, original triangle index.
struct TriRecord {
	V3 v0, e1, e2;
	uint32_t tri;
};
RMD_DEV TriRecord load_record(const unsigned char *p) {
	const double *d = reinterpret_cast<const double *>(p);
	TriRecord r;
	r.v0 = ld3(d), r.e1 = ld3(d + 3), r.e2 = ld3(d + 6);
	r.tri = reinterpret_cast<const uint32_t *>(p)[18];
	return r;
}

// One DDA step (acc_grid.rs:155-183), branch-free: the same three comparisons select the axis
// (x if tmx<tmy && tmx<tmz; y if !(tmx<tmy) && tmy<tmz; else z), the cell moves, and that axis' t_max advances
// (its value is irrelevant once the ray has left the grid).  `idx` is the reference's linear cell index
// x + res.x*(y + z*res.z) (Q5: res.z where res.y is meant) kept incrementally: a step along x/y/z adds dix/diy/diz.
// 32-bit arithmetic is exact because the upload rejects grids whose largest reachable index does not fit 31 bits.
// Outputs the stepped state in n*; returns false when the ray leaves the grid (:158,:164,:172,:178).
RMD_DEV bool dda_step(int32_t cx, int32_t cy, int32_t cz, uint32_t idx, double tmx, double tmy, double tmz, int32_t sx, int32_t sy,
                      int32_t sz, int32_t dix, int32_t diy, int32_t diz, double tdx, double tdy, double tdz, int32_t rx, int32_t ry,
                      int32_t rz, int32_t &ncx, int32_t &ncy, int32_t &ncz, uint32_t &nidx, double &ntmx, double &ntmy, double &ntmz) {
	const bool lt_xy = tmx < tmy, lt_xz = tmx < tmz, lt_yz = tmy < tmz;
	const bool ax = lt_xy && lt_xz;
	const bool ay = !lt_xy && lt_yz;
	const bool az = !ax && !ay;
	const int32_t c_new = ax ? cx + sx : (ay ? cy + sy : cz + sz);
	const int32_t r_sel = ax ? rx : (ay ? ry : rz);
	ncx = ax ? c_new : cx;
	ncy = ay ? c_new : cy;
	ncz = az ? c_new : cz;
	nidx = idx + (uint32_t)(ax ? dix : (ay ? diy : diz));
	const double nx = tmx + tdx, ny = tmy + tdy, nz = tmz + tdz;
	ntmx = ax ? nx : tmx;
	ntmy = ay ? ny : tmy;
	ntmz = az ? nz : tmz;
	return (uint32_t)c_new < (uint32_t)r_sel; // 0 <= c_new < res
}

// Per-wave LDS scratch of the cooperative triangle tests (3.75 KiB).
struct WalkScratch {
	double ray[6][64];  // ro.xyz, rd.xyz of the lanes with a pending cell, indexed by lane
	uint32_t start[64]; // exclusive prefix sum of the pending cells' triangle counts, compacted by rank
	uint32_t first[64]; // first record of the cell's run, by rank
	uint32_t owner[64]; // lane that owns the pair, by rank
};

// Must be called by all 64 lanes of the wave in uniform control flow; `want` selects the lanes that have a ray.
// lds_mask: occupancy bits of this grid in LDS (bit i covers cells [i << shift, (i+1) << shift)), or nullptr.
// scr: this wave's scratch in LDS.
RMD_DEV void grid_intersect_wave(const DevGrid &g, const uint32_t *lds_mask, WalkScratch &scr, bool want, V3 ro, V3 rd, bool &hit_out,
                                 double &t_out, uint32_t &tri_out, uint32_t debug_flags = 0, unsigned long long *dbg = nullptr) {
	const uint32_t lane = threadIdx.x & 63u;
	const int32_t rx = (int32_t)g.res[0], ry = (int32_t)g.res[1], rz = (int32_t)g.res[2];
	const uint64_t resx = g.res[0], resz = g.res[2], n_cells = g.n_cells;
	const uint32_t mask_bits = g.mask_bits, mask_shift = g.mask_shift;
	const CellEntry *__restrict__ entries = g.cell_entries;
	const unsigned char *__restrict__ runs = reinterpret_cast<const unsigned char *>(g.tri_runs);
#if RMD_DIAG
	const bool stamp = (debug_flags & 16u) && dbg;
	unsigned long long t_prev = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
	unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph5 = 0, ph6 = 0, ph7 = 0;
#define RMD_STAMP(n)                                         \
	if (stamp) {                                             \
		__builtin_amdgcn_s_waitcnt(0);                       \
		unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		ph##n += now_ - t_prev;                               \
		t_prev = now_;                                       \
	}
#else
#define RMD_STAMP(n)
#endif

	bool walking = false;
	int32_t cx = 0, cy = 0, cz = 0, sx = 1, sy = 1, sz = 1, dix = 0, diy = 0, diz = 0;
	uint32_t idx = 0;
	double tmx = 0.0, tmy = 0.0, tmz = 0.0, tdx = 0.0, tdy = 0.0, tdz = 0.0;
	if (want && !(debug_flags & 2u)) {
		// acc_grid.rs:90-125
		V3 bmin = ld3(g.bbox_min);
		double t_outer;
		if (aabb_intersect(bmin, ld3(g.bbox_max), ro, rd, t_outer)) {
			V3 cs = ld3(g.cell_size);
			V3 start = ro - bmin;
			bool ok = cast_i32(start.x / cs.x, cx) && cast_i32(start.y / cs.y, cy) && cast_i32(start.z / cs.z, cz);
			if (ok && (cx < 0 || cy < 0 || cz < 0)) {
				V3 outer_pos = ro + rd * t_outer;
				start = outer_pos - bmin;
				ok = cast_i32(start.x / cs.x, cx) && cast_i32(start.y / cs.y, cy) && cast_i32(start.z / cs.z, cz);
			}
			ok = ok && !(rd.x != rd.x || rd.y != rd.y || rd.z != rd.z); // signum(NaN).cast::<i32>() panics: miss
			if (ok) {
				// first cell (:128-131): `as usize` sign-extends and the index arithmetic wraps (release build)
				const uint64_t idx0 = (uint64_t)(int64_t)cx + resx * ((uint64_t)(int64_t)cy + (uint64_t)(int64_t)cz * resz);
				ok = idx0 < n_cells; // else None
				idx = (uint32_t)idx0;
			}
			if (ok) {
				sx = signbit(rd.x) ? -1 : 1, sy = signbit(rd.y) ? -1 : 1, sz = signbit(rd.z) ? -1 : 1;
				dix = sx, diy = sy * (int32_t)resx, diz = sz * (int32_t)(resx * resz);
				tdx = (rd.x < 0.0 ? -cs.x : cs.x) / rd.x;
				tdy = (rd.y < 0.0 ? -cs.y : cs.y) / rd.y;
				tdz = (rd.z < 0.0 ? -cs.z : cs.z) / rd.z;
				tmx = (((double)(cx + (rd.x < 0.0 ? 0 : 1)) * cs.x) - start.x) / rd.x;
				tmy = (((double)(cy + (rd.y < 0.0 ? 0 : 1)) * cs.y) - start.y) / rd.y;
				tmz = (((double)(cz + (rd.z < 0.0 ? 0 : 1)) * cs.z) - start.z) / rd.z;
				walking = true;
			}
		}
	}

	bool found = false;
	double found_t = 0.0;
	uint32_t found_tri = 0;
	RMD_STAMP(0)

#if RMD_DIAG
	const bool count_events = (debug_flags & 8u) && dbg;
#else
	constexpr bool count_events = false;
#endif
	if (count_events && lane == 0) atomicAdd(&dbg[0], 1ull), atomicAdd(&dbg[1], (unsigned long long)__popcll(__ballot(walking)) * 0ull);
	if (count_events) { unsigned long long wm = __ballot(walking); if (lane == 0) { atomicAdd(&dbg[1], (unsigned long long)__popcll(wm)); if (wm) atomicAdd(&dbg[2], 1ull); } }
	for (;;) {
		// 1. per lane: advance to the next cell that may hold triangles (ALU + LDS only).  Every cell the lane stands on
		//    has a valid index (< n_cells, checked for the first cell above and after every step below, :129-131).
		//    The next step is computed while the mask word of the current cell is still in flight.
		if (count_events && lane == 0) atomicAdd(&dbg[3], 1ull);
		while (walking) {
			if (count_events) { unsigned long long am = __ballot(true); if (lane == (uint32_t)__builtin_ctzll(am)) { atomicAdd(&dbg[4], 1ull); atomicAdd(&dbg[5], (unsigned long long)__popcll(am)); } }
			bool candidate = true;
			uint32_t word = 0xFFFFFFFFu;
			const uint32_t bit = idx >> mask_shift;
			if (lds_mask) {
				candidate = bit < mask_bits;
				word = lds_mask[candidate ? (bit >> 5) : 0u];
			}
			int32_t ncx, ncy, ncz;
			uint32_t nidx;
			double ntmx, ntmy, ntmz;
			bool inside = dda_step(cx, cy, cz, idx, tmx, tmy, tmz, sx, sy, sz, dix, diy, diz, tdx, tdy, tdz, rx, ry, rz, ncx, ncy, ncz, nidx, ntmx, ntmy, ntmz);
			inside = inside && nidx < (uint32_t)n_cells; // next cell past the cell array: the walk returns None there
			candidate = candidate && ((word >> (bit & 31u)) & 1u);
			if (candidate) break;
			cx = ncx, cy = ncy, cz = ncz, idx = nidx, tmx = ntmx, tmy = ntmy, tmz = ntmz;
			walking = inside;
		}
		RMD_STAMP(1)
		if (__ballot(walking) == 0ull) break;

		// 2. candidate cells: {first record, count} in one 8-byte gather per lane
		uint32_t first = 0, count = 0;
		if (walking) {
			CellEntry e = entries[idx];
			first = e.first, count = e.count;
		}

		// 3. triangle tests, distributed over the whole wave.  The (lane, cell) pairs of this round own `count` tests each;
		//    an exclusive prefix sum over the counts numbers all tests of the round 0..T-1 in (lane, triangle) order and
		//    the wave takes them 64 at a time: lane l of a chunk finds its pair by binary search in the scanned counts
		//    (LDS), fetches that pair's ray from LDS and the triangle record from the cell's contiguous run (neighbouring
		//    lanes read neighbouring 80-byte records: coalesced).  Hits are rare; they are applied by a scalar loop over
		//    the hit ballot in ascending lane order = ascending (pair, triangle) order with a strict '<', which is the
		//    reference's sequential scan of the cell (acc_grid.rs:135-149: closest starts at 5712515.0, first wins ties).
		RMD_STAMP(2)
		const bool pending = walking && count > 0u && !(debug_flags & 1u);
		const unsigned long long pmask = __ballot(pending);
		if (pmask != 0ull) {
			const uint32_t n_pairs = (uint32_t)__popcll(pmask);
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask, 0u));
			// inclusive scan of the counts over the lanes (Hillis-Steele through the LDS crossbar)
			uint32_t incl = pending ? count : 0u;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
				if ((int)lane >= d) incl += up;
			}
			const uint32_t total = readlane_u32(incl, 63);
			if (count_events && lane == 0) atomicAdd(&dbg[6], 1ull), atomicAdd(&dbg[7], (unsigned long long)n_pairs), atomicAdd(&dbg[8], (unsigned long long)total), atomicAdd(&dbg[9], (unsigned long long)((total + 63u) / 64u));
			if (pending) {
				scr.start[rank] = incl - count;
				scr.first[rank] = first;
				scr.owner[rank] = lane;
				scr.ray[0][lane] = ro.x, scr.ray[1][lane] = ro.y, scr.ray[2][lane] = ro.z;
				scr.ray[3][lane] = rd.x, scr.ray[4][lane] = rd.y, scr.ray[5][lane] = rd.z;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			RMD_STAMP(3)
			double closest = 5712515.0;
			uint32_t closest_tri = 0;
			bool any = false;
			for (uint32_t base = 0; base < total; base += 64u) {
				const uint32_t w = base + lane;
				bool h = false;
				double t = 0.0;
				uint32_t tri = 0, own = 0;
				if (w < total) {
					uint32_t k = 0;
#pragma unroll
					for (uint32_t step = 32u; step > 0u; step >>= 1) {
						const uint32_t mid = k + step;
						if (mid < n_pairs && scr.start[mid] <= w) k = mid;
					}
					own = scr.owner[k];
					const unsigned char *rec = runs + (size_t)(scr.first[k] + (w - scr.start[k])) * 80u;
					const TriRecord r = load_record(rec);
					const V3 pro = mk(scr.ray[0][own], scr.ray[1][own], scr.ray[2][own]);
					const V3 prd = mk(scr.ray[3][own], scr.ray[4][own], scr.ray[5][own]);
					tri = r.tri;
					h = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, t);
				}
				RMD_STAMP(5)
				unsigned long long hits = __ballot(h);
				while (hits) {
					const int l = (int)__builtin_ctzll(hits);
					hits &= hits - 1ull;
					const uint32_t pl = readlane_u32(own, l);
					const double tl = readlane_f64(t, l);
					const uint32_t tril = readlane_u32(tri, l);
					if (lane == pl && tl < closest) {
						closest = tl;
						closest_tri = tril;
						any = true;
					}
				}
			}
			RMD_STAMP(6)
			__builtin_amdgcn_wave_barrier(); // the scratch is rewritten next round
			if (any) {                       // :151-153 first cell with any hit wins
				found = true;
				found_t = closest;
				found_tri = closest_tri;
				walking = false;
			}
		}

		// 4. lanes whose cell yielded nothing move on
		if (walking) {
			const bool inside = dda_step(cx, cy, cz, idx, tmx, tmy, tmz, sx, sy, sz, dix, diy, diz, tdx, tdy, tdz, rx, ry, rz, cx, cy, cz, idx, tmx, tmy, tmz);
			walking = inside && idx < (uint32_t)n_cells;
		}
	}
	RMD_STAMP(7)
#if RMD_DIAG
	if (stamp && lane == 0) {
		atomicAdd(&dbg[8], ph0), atomicAdd(&dbg[9], ph1), atomicAdd(&dbg[10], ph2), atomicAdd(&dbg[11], ph3);
		atomicAdd(&dbg[13], ph5), atomicAdd(&dbg[14], ph6), atomicAdd(&dbg[15], ph7);
	}
#endif
#undef RMD_STAMP
	hit_out = found;
	t_out = found_t;
	tri_out = found_tri;
}

} // namespace rmd
