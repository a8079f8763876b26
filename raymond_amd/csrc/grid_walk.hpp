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

// One DDA step (acc_grid.rs:155-183).  Returns false when the ray leaves the grid.
RMD_DEV bool dda_step(int32_t &cx, int32_t &cy, int32_t &cz, double &tmx, double &tmy, double &tmz, int32_t sx, int32_t sy, int32_t sz,
                      double tdx, double tdy, double tdz, int32_t rx, int32_t ry, int32_t rz) {
	if (tmx < tmy) {
		if (tmx < tmz) {
			cx += sx;
			if (cx >= rx || cx < 0) return false;
			tmx += tdx;
		} else {
			cz += sz;
			if (cz >= rz || cz < 0) return false;
			tmz += tdz;
		}
	} else {
		if (tmy < tmz) {
			cy += sy;
			if (cy >= ry || cy < 0) return false;
			tmy += tdy;
		} else {
			cz += sz;
			if (cz >= rz || cz < 0) return false;
			tmz += tdz;
		}
	}
	return true;
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
                                 double &t_out, uint32_t &tri_out) {
	const uint32_t lane = threadIdx.x & 63u;
	const int32_t rx = (int32_t)g.res[0], ry = (int32_t)g.res[1], rz = (int32_t)g.res[2];
	const uint64_t resx = g.res[0], resz = g.res[2], n_cells = g.n_cells;
	const uint32_t mask_bits = g.mask_bits, mask_shift = g.mask_shift;
	const CellEntry *__restrict__ entries = g.cell_entries;
	const unsigned char *__restrict__ runs = reinterpret_cast<const unsigned char *>(g.tri_runs);

	bool walking = false;
	int32_t cx = 0, cy = 0, cz = 0, sx = 1, sy = 1, sz = 1;
	double tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0;
	if (want) {
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
				sx = signbit(rd.x) ? -1 : 1, sy = signbit(rd.y) ? -1 : 1, sz = signbit(rd.z) ? -1 : 1;
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

	for (;;) {
		// 1. per lane: advance to the next cell that may hold triangles (ALU + LDS only)
		uint64_t idx = 0;
		while (walking) {
			// `as usize` sign-extends and the index arithmetic wraps (release build); Q5: res.z where res.y is meant
			idx = (uint64_t)(int64_t)cx + resx * ((uint64_t)(int64_t)cy + (uint64_t)(int64_t)cz * resz);
			if (idx >= n_cells) { // :129-131 -> None
				walking = false;
				break;
			}
			bool candidate = true;
			if (lds_mask) {
				uint32_t bit = (uint32_t)(idx >> mask_shift);
				candidate = bit < mask_bits && ((lds_mask[bit >> 5] >> (bit & 31u)) & 1u);
			}
			if (candidate) break;
			if (!dda_step(cx, cy, cz, tmx, tmy, tmz, sx, sy, sz, tdx, tdy, tdz, rx, ry, rz)) walking = false;
		}
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
		const bool pending = walking && count > 0u;
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
			if (!dda_step(cx, cy, cz, tmx, tmy, tmz, sx, sy, sz, tdx, tdy, tdz, rx, ry, rz)) walking = false;
		}
	}
	hit_out = found;
	t_out = found_t;
	tri_out = found_tri;
}

} // namespace rmd
