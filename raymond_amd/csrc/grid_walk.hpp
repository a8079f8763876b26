// Wave-cooperative AccGrid::intersects (reference core/src/geometry/acc_grid.rs:89-185) for gfx950.
//
// The reference walks one ray through the grid cell by cell and, in every cell, tests that cell's
// triangles one after the other.  Measured on the 99k-triangle benchmark mesh a walk visits ~20 cells of
// which ~2.7 are non-empty, and a non-empty cell holds ~9-13 triangles (up to 50+).  A lane-per-ray port of
// that loop leaves ~90 % of the lanes idle and chains 2-4 dependent memory latencies per cell.  Here:
//
//   1. every lane steps ITS OWN ray's DDA (bit-identical arithmetic, :100-125 and :155-183) but skips empty
//      cells with an occupancy bitmask held in LDS — no global memory access until a candidate cell — and
//      collects up to kWalkCand candidate cells per round (it steps past a candidate speculatively);
//   2. the lanes fetch the 8-byte entries {first id, count} of their candidates together — of the list that leaves out the triangles
//      the cell the ray came from lists too (device_types.hpp: they were tested against this ray one cell earlier and missed);
//   3. all triangle tests of the round — every (lane, candidate, triangle) — are numbered by a prefix sum and
//      taken by the WHOLE WAVE 64 at a time, one test per lane: the test's ray comes from its owner lane's
//      registers (ds_bpermute), its triangle index from the cell's list (neighbouring tests read neighbouring indices), the
//      triangle's record — one per triangle, 8 MB for the benchmark mesh — by that index, and the
//      winner is picked by a scalar loop over the hit ballot in ascending (lane, candidate, triangle) order with
//      a strict '<' — the reference's `if distance < closest` scan (:137-149) exactly, including the 5712515.0
//      start value and first-wins ties; the earliest candidate cell with an accepted hit wins (:151-153).
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
// value of `v` in the lane whose byte address (lane * 4) is `addr`; must be executed by every lane that is read from
RMD_DEV double bperm_f64(int addr, double v) {
	unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)b);
	unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)(unsigned)(b >> 32));
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Inclusive scans over the 64 lanes of the wave in 6 DPP steps (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3) — VALU only, no LDS round trips.  Lanes for which a
// step has no source keep the identity 0.  Must be executed by all 64 lanes.
template <int CTRL, int ROW_MASK>
RMD_DEV uint32_t dpp_from(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
RMD_DEV uint32_t wave_scan_add(uint32_t v) {
	v += dpp_from<0x111, 0xf>(v);
	v += dpp_from<0x112, 0xf>(v);
	v += dpp_from<0x114, 0xf>(v);
	v += dpp_from<0x118, 0xf>(v);
	v += dpp_from<0x142, 0xa>(v);
	v += dpp_from<0x143, 0xc>(v);
	return v;
}
RMD_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
RMD_DEV uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
RMD_DEV uint32_t wave_scan_max(uint32_t v) {
	v = umax(v, dpp_from<0x111, 0xf>(v));
	v = umax(v, dpp_from<0x112, 0xf>(v));
	v = umax(v, dpp_from<0x114, 0xf>(v));
	v = umax(v, dpp_from<0x118, 0xf>(v));
	v = umax(v, dpp_from<0x142, 0xa>(v));
	v = umax(v, dpp_from<0x143, 0xc>(v));
	return v;
}

// One triangle record (kTriRecStride bytes apart): v0, edge1, edge2.
struct TriRecord {
	V3 v0, e1, e2;
};
RMD_DEV TriRecord load_record(const RMD_GLOBAL unsigned char *p) {
	const RMD_GLOBAL double *d = reinterpret_cast<const RMD_GLOBAL double *>(p);
	TriRecord r;
	r.v0 = ld3(d), r.e1 = ld3(d + 3), r.e2 = ld3(d + 6);
	return r;
}
constexpr uint32_t kNoCell = 0xFFFFFFFFu; // "the cell before this one" of a walk's first cell

// One DDA step (acc_grid.rs:155-183) is done inline in the walk loop below.  The same three comparisons select the axis
// (x if tmx<tmy && tmx<tmz; y if !(tmx<tmy) && tmy<tmz; else z); that axis' t_max advances by its t_delta (same
// additions in the same order as the reference) and the cell moves.  The cell itself is not kept: what the walk needs is
//   * `idx`, the reference's linear cell index x + res.x*(y + z*res.z) (Q5: res.z where res.y is meant), updated
//     incrementally — a step along x/y/z adds dix/diy/diz (32-bit arithmetic is exact: the upload rejects grids whose
//     largest reachable index does not fit 31 bits);
//   * `rem[a]`, the number of steps along axis a after which the reference's range test (:158,:164,:172,:178:
//     `cell.a < 0 || cell.a >= res.a` → None) fails; a step decrements its axis' counter and the ray has left the
//     grid when a counter reaches 0.
// steps_to_exit() derives the counter from the start cell exactly as those tests would fire, including start cells
// that are themselves out of range on an axis (the reference only tests an axis when it steps along it).
RMD_DEV uint32_t steps_to_exit(int32_t c, int32_t s, int32_t r) {
	if (s > 0) {
		if (c <= -2) return 1u; // c+1 is still negative
		const int64_t k = (int64_t)r - (int64_t)c;
		return k < 1 ? 1u : (uint32_t)k;
	}
	if (c > r) return 1u; // c-1 is still >= r
	const int64_t k = (int64_t)c + 1;
	return k < 1 ? 1u : (uint32_t)k;
}

// Candidates a lane may collect per round (speculative look-ahead along its own DDA path).
#ifndef RMD_WALK_CANDIDATES
#define RMD_WALK_CANDIDATES 4
#endif
constexpr uint32_t kWalkCand = RMD_WALK_CANDIDATES;
static_assert(kWalkCand >= 1 && kWalkCand <= 8, "1..8 candidates per round");
// After its first candidate a lane keeps stepping at most this many cells looking for more (the non-empty cells of
// one surface crossing are adjacent); further cells wait for the next round.
#ifndef RMD_WALK_ASM_STEP
#define RMD_WALK_ASM_STEP 1
#endif
#ifndef RMD_WALK_LOOKAHEAD
#define RMD_WALK_LOOKAHEAD 16
#endif
constexpr uint32_t kWalkLookahead = RMD_WALK_LOOKAHEAD;
// chunks the owner search (and the load of the triangle indices) runs ahead of the tests: 1 or 2
#ifndef RMD_WALK_STEP_PRIO
#define RMD_WALK_STEP_PRIO 2 // s_setprio level of a round's stepping loop (0 = not raised)
#endif
constexpr int kWalkStepPrio = RMD_WALK_STEP_PRIO;
#ifndef RMD_WALK_SEARCH_AHEAD
#define RMD_WALK_SEARCH_AHEAD 2
#endif
constexpr uint32_t kWalkSearchAhead = RMD_WALK_SEARCH_AHEAD;
static_assert(kWalkSearchAhead == 1u || kWalkSearchAhead == 2u, "the search runs one or two chunks ahead");

// Per-wave LDS scratch of the cooperative triangle tests.  A (lane, candidate slot) pair has the key lane * kWalkCand + slot.
struct alignas(16) WalkScratch {
	uint32_t start[64 * kWalkCand]; // by key: number of the pair's first test in the round (exclusive prefix sum of the counts); during the stepping,
	                                // the cell BEFORE candidate m of lane l at [m * 64 + l]
	uint32_t first[64 * kWalkCand]; // by key: first entry of the pair's list in tri_ids minus the number of the pair's first test; during the stepping, candidate m of lane l at [m * 64 + l]
	uint32_t marker[64];            // per 64-test chunk: lane + 1 of the lane whose tests begin at that position
};

// A walk that is put aside in the middle and taken up again by the wave's NEXT walk call (grid_intersect_wave: `cut_lanes`): the state of the
// ray's DDA — everything the stepping changes; the per-ray constants (t_delta, index strides) are derived from the ray again.  One column per lane.
struct alignas(16) WalkCarry {
	double tm[3][64];
	uint32_t idx[64], prev[64];
	uint32_t rem[3][64];
};

// One DDA step (see above): three compares, the three axis masks on the scalar unit, and each axis' {t_max += t_delta;
// counter -= 1; index += stride} under its mask — 12 vector instructions, no branches (the compiler's rendering of the same C++
// re-evaluates a compare, routes the stride through a select and branches around two blocks).  exec is saved in %[sv] and
// restored by the last instruction; the scalar mask arithmetic overwrites SCC and VCC (both clobbered), and the block is
// volatile so that it is neither duplicated nor moved across the exec-dependent code around it.
#define RMD_DDA_STEP_ASM()                                                                                                              \
	{                                                                                                                                   \
		unsigned long long m_xy, m_xz, saved;                                                                                           \
		asm volatile("v_cmp_lt_f64 %[mxy], %[tmx], %[tmy]\n\t"                                                                           \
		             "v_cmp_lt_f64 %[mxz], %[tmx], %[tmz]\n\t"                                                                           \
		             "v_cmp_lt_f64 vcc, %[tmy], %[tmz]\n\t"                                                                              \
		             "s_mov_b64 %[sv], exec\n\t"                                                                                         \
		             "s_and_b64 %[mxz], %[mxy], %[mxz]\n\t" /* x:  tmx < tmy && tmx < tmz */                                             \
		             "s_andn2_b64 vcc, vcc, %[mxy]\n\t"     /* y: !(tmx < tmy) && tmy < tmz   (masks only hold active lanes) */          \
		             "s_mov_b64 exec, %[mxz]\n\t"                                                                                        \
		             "v_add_f64 %[tmx], %[tmx], %[tdx]\n\t"                                                                              \
		             "v_add_u32 %[rx], -1, %[rx]\n\t"                                                                                    \
		             "v_add_u32 %[idx], %[idx], %[dix]\n\t"                                                                              \
		             "s_mov_b64 exec, vcc\n\t"                                                                                           \
		             "v_add_f64 %[tmy], %[tmy], %[tdy]\n\t"                                                                              \
		             "v_add_u32 %[ry], -1, %[ry]\n\t"                                                                                    \
		             "v_add_u32 %[idx], %[idx], %[diy]\n\t"                                                                              \
		             "s_or_b64 vcc, vcc, %[mxz]\n\t"                                                                                     \
		             "s_andn2_b64 exec, %[sv], vcc\n\t" /* z: the rest */                                                                \
		             "v_add_f64 %[tmz], %[tmz], %[tdz]\n\t"                                                                              \
		             "v_add_u32 %[rz], -1, %[rz]\n\t"                                                                                    \
		             "v_add_u32 %[idx], %[idx], %[diz]\n\t"                                                                              \
		             "s_mov_b64 exec, %[sv]"                                                                                             \
		             : [tmx] "+v"(tmx), [tmy] "+v"(tmy), [tmz] "+v"(tmz), [rx] "+v"(remx), [ry] "+v"(remy), [rz] "+v"(remz), [idx] "+v"(idx), \
		               [mxy] "=&s"(m_xy), [mxz] "=&s"(m_xz), [sv] "=&s"(saved)                                                           \
		             : [tdx] "v"(tdx), [tdy] "v"(tdy), [tdz] "v"(tdz), [dix] "v"(dix), [diy] "v"(diy), [diz] "v"(diz)                      \
		             : "vcc", "scc");                                                                                                    \
	}
#if !RMD_WALK_ASM_STEP
#undef RMD_DDA_STEP_ASM
#define RMD_DDA_STEP_ASM()                                                \
	{                                                                     \
		const bool lt_xy = tmx < tmy, lt_xz = tmx < tmz, lt_yz = tmy < tmz; \
		if (lt_xy && lt_xz) {                                             \
			tmx += tdx, remx--, idx += (uint32_t)dix;                     \
		} else if (!lt_xy && lt_yz) {                                     \
			tmy += tdy, remy--, idx += (uint32_t)diy;                     \
		} else {                                                          \
			tmz += tdz, remz--, idx += (uint32_t)diz;                     \
		}                                                                 \
	}
#endif
// The occupancy bit of the cell a lane stands on, then the step, then the exit test — the body of both stepping loops.
// LEAN: one mask bit per cell (no shift) and no test of the index against the cell array (see grid_intersect_wave).
#define RMD_DDA_ITERATION(LEAN)                                                                                          \
	uint32_t bit = LEAN ? idx : idx >> mask_shift;                                                                       \
	bit = bit < mask_pad_bit ? bit : mask_pad_bit; /* indices past the mask read the zero word that pads it */          \
	const uint32_t word = lds_mask[bit >> 5];                                                                            \
	const uint32_t here = idx;                                                                                           \
	RMD_DDA_STEP_ASM() /* computed while the mask word is in flight */                                                    \
	const uint32_t rem_min = remx < remy ? (remx < remz ? remx : remz) : (remy < remz ? remy : remz);                    \
	/* left the grid (a range test fired), or the next cell is past the cell array (:129-131): the walk returns None */ \
	walking = LEAN ? rem_min != 0u : (rem_min != 0u && idx < idx_limit);                                                 \
	const bool occupied = __builtin_amdgcn_ubfe(word, bit, 1u) != 0u; /* v_bfe_u32 takes the bit position modulo 32 */
// The round's stepping: up to kWalkCand candidate cells per lane.  A candidate is stepped over at once — speculating that it
// yields no hit — so that one round can gather the adjacent non-empty cells of a surface crossing; after its first candidate
// a lane steps at most kWalkLookahead further cells, the rest waits for the next round.  Candidates are parked in this lane's
// column of scr.first (slot m at [m * 64 + lane]).  `budget` = steps the lane may still take in this round: unlimited until its
// first candidate, kWalkLookahead after it, 0 once kWalkCand candidates are recorded.  (One loop for both parts: split in two,
// the look-ahead of the early lanes no longer overlaps the search of the late ones — 3 % fewer instructions, 3 % more time.)
template <bool LEAN>
RMD_DEV void dda_collect_candidates(const uint32_t *lds_mask, uint32_t mask_shift, uint32_t mask_pad_bit, uint32_t idx_limit, WalkScratch &scr,
                                    uint32_t lane, bool &walking, uint32_t &n_cand, uint32_t &idx, uint32_t &prev, uint32_t &remx, uint32_t &remy, uint32_t &remz,
                                    double &tmx, double &tmy, double &tmz, double tdx, double tdy, double tdz, int32_t dix, int32_t diy, int32_t diz,
                                    uint32_t cut_lanes, [[maybe_unused]] unsigned long long *dbg = nullptr) {
	uint32_t budget = 0x7FFFFFFFu;
#if RMD_DIAG
	if (dbg) scr.marker[0] = 0u; // DIAG: set once a lane of the wave has a candidate in this round
#endif
	while (walking && budget != 0u) {
#if RMD_DIAG
		if (dbg) { // occupancy of the stepping loop: iterations, stepping lanes, iterations with few lanes, and of those the ones with test work already waiting
			const unsigned long long act = __ballot(true);
			const uint32_t n = (uint32_t)__popcll(act), have = scr.marker[0];
			if (lane == (uint32_t)__builtin_ctzll(act)) {
				atomicAdd(&dbg[4], 1ull), atomicAdd(&dbg[5], (unsigned long long)n);
				if (n <= 4u) atomicAdd(&dbg[7], 1ull);
				if (n <= 8u) atomicAdd(&dbg[13], 1ull);
				if (n <= 16u) atomicAdd(&dbg[14], 1ull);
				if (n <= 8u && have) atomicAdd(&dbg[15], 1ull);
			}
		}
#endif
		RMD_DDA_ITERATION(LEAN)
		budget--;
		if (occupied) {
			scr.first[n_cand * 64u + lane] = here, scr.start[n_cand * 64u + lane] = prev; // the candidate and the cell the ray entered it from
			n_cand++;
			budget = n_cand == kWalkCand ? 0u : (budget < kWalkLookahead ? budget : kWalkLookahead);
#if RMD_DIAG
			if (dbg) scr.marker[0] = 1u;
#endif
		}
		prev = here;
		if ((uint32_t)__popcll(__ballot(walking && budget != 0u)) <= cut_lanes) break; // (0: the loop's own condition) the stragglers go on in a later round or call
	}
}

// The stepping loop of a round for grids with one mask bit per cell, written out as one block of gfx950 assembly
// (RMD_WALK_ASM_LOOP; dda_collect_candidates above is the same algorithm in C++ and serves coarser masks).  A SIMD issues at
// most one vector and one scalar instruction per 4 cycles, from different waves, so the loop is as slow as the LONGER of
// its two streams: the compiler's rendering of the C++ loop has 26 vector + 20 scalar instructions per step (32 + 22 when
// a candidate is recorded), this one 19 + 17 (22 + 18):
//   * the lane's cell index — and the index of the cell it came from, which selects the candidate's triangle list — is stored to its
//     next candidate slot on EVERY step (LDS stores, neither stream) and the slot only advances when the cell turns out occupied;
//   * the exit counters are kept minus one and decremented with v_sub_co: the borrow IS "this axis left the grid"
//     (acc_grid.rs:158,164,172,178) — no minimum of three, no compare; carry-outs of lanes outside exec are written as 0,
//     like a compare's, so the three axis blocks' borrows are simply OR-ed;
//   * the step budget and the candidates left count down the same way, their borrows join the round's stop mask;
//   * the set of lanes still stepping lives in one scalar pair from which exec is re-made, instead of save / restore pairs.
// Executed by the lanes that are walking (divergent call); exec is saved on entry and restored on exit; the scalar temporaries
// are fixed registers named in the clobber list (an asm statement takes at most 30 operands).  Every step uses up one of the
// lane's exit counters, so a lane leaves the loop after at most res.x + res.y + res.z steps.  Arithmetic on t_max, the axis
// choice, the index, the end-of-array test and the order of the recorded candidates are those of RMD_DDA_ITERATION, bit for bit.
#ifndef RMD_SPHERE_PREFILTER
#define RMD_SPHERE_PREFILTER 1 // grid_intersect_wave: the triangle tests of a round behind a bounding-sphere pre-test
#endif
#ifndef RMD_FLAT_TRIANGLE_TEST
#define RMD_FLAT_TRIANGLE_TEST 0 // device_core.hpp: triangle_test_flat — measured: seven more spilled registers, C3 426.5 vs 421.0 ms: not used
#endif
#ifndef RMD_WALK_ASM_LOOP
#define RMD_WALK_ASM_LOOP 1
#endif
// cand_base: LDS address of this lane's column of WalkScratch::start (the cells before the candidates); the candidates go to the same column
// of WalkScratch::first, sizeof(start) bytes further on.
#define RMD_DDA_LOOP_ASM(CLAMP_AND_WORD, BIT_POSITION, LIMIT_TEST, LIMIT_JOIN) \
	asm volatile( \
	    "s_mov_b64 s[86:87], exec\n\t"                      /* entry exec */ \
	    "s_mov_b64 s[94:95], exec\n\t"                      /* lanes still stepping in this round */ \
	    "s_mov_b64 s[88:89], 0\n\t"                         /* lanes whose walk has ended */ \
	    "v_mov_b32 %[budget], 0x7ffffffe\n\t"               /* unlimited until the first candidate */ \
	    "v_mov_b32 %[cleft], %[ncand1]\n\t" \
	    "v_add_u32 %[rx], -1, %[rx]\n\t" \
	    "v_add_u32 %[ry], -1, %[ry]\n\t" \
	    "v_add_u32 %[rz], -1, %[rz]\n\t" \
	    "s_waitcnt lgkmcnt(0)\n" \
	    "Lrmd_dda_loop%=:\n\t" \
	    CLAMP_AND_WORD \
	    "v_lshl_add_u32 %[word], %[word], 2, %[mbase]\n\t" \
	    "ds_read_b32 %[word], %[word]\n\t" \
	    "ds_write_b32 %[caddr], %[idx] offset:%[firstoff]\n\t" /* the cell the lane stands on, into its next candidate slot */ \
	    "ds_write_b32 %[caddr], %[prev]\n\t"                /* ... and the cell it came from */ \
	    "v_mov_b32 %[prev], %[idx]\n\t" \
	    "v_cmp_lt_f64 s[90:91], %[tmx], %[tmy]\n\t" \
	    "v_cmp_lt_f64 s[92:93], %[tmx], %[tmz]\n\t" \
	    "v_cmp_lt_f64 vcc, %[tmy], %[tmz]\n\t" \
	    "s_and_b64 s[92:93], s[90:91], s[92:93]\n\t"        /* x:  tmx < tmy && tmx < tmz */ \
	    "s_andn2_b64 vcc, vcc, s[90:91]\n\t"                /* y: !(tmx < tmy) && tmy < tmz */ \
	    "s_mov_b64 exec, s[92:93]\n\t" \
	    "v_add_f64 %[tmx], %[tmx], %[tdx]\n\t" \
	    "v_sub_co_u32_e64 %[rx], s[90:91], %[rx], 1\n\t"    /* borrow: the x counter was 0 = the ray leaves the grid (0 for lanes outside exec) */ \
	    "v_add_u32 %[idx], %[idx], %[dix]\n\t" \
	    "s_mov_b64 exec, vcc\n\t" \
	    "v_add_f64 %[tmy], %[tmy], %[tdy]\n\t" \
	    "v_sub_co_u32_e64 %[ry], s[96:97], %[ry], 1\n\t" \
	    "v_add_u32 %[idx], %[idx], %[diy]\n\t" \
	    "s_or_b64 vcc, vcc, s[92:93]\n\t" \
	    "s_andn2_b64 exec, s[94:95], vcc\n\t"               /* z: the rest */ \
	    "s_or_b64 s[90:91], s[90:91], s[96:97]\n\t" \
	    "v_add_f64 %[tmz], %[tmz], %[tdz]\n\t" \
	    "v_sub_co_u32_e64 %[rz], s[96:97], %[rz], 1\n\t" \
	    "v_add_u32 %[idx], %[idx], %[diz]\n\t" \
	    "s_mov_b64 exec, s[94:95]\n\t" \
	    "s_or_b64 s[90:91], s[90:91], s[96:97]\n\t" \
	    LIMIT_TEST                                          /* the next cell is past the cell array (:129-131): None */ \
	    "v_subrev_co_u32_e32 %[budget], vcc, 1, %[budget]\n\t" /* borrow: the look-ahead budget is used up */ \
	    LIMIT_JOIN                                          /* lanes whose walk ended on this step */ \
	    "s_or_b64 s[88:89], s[88:89], s[90:91]\n\t" \
	    "s_or_b64 s[90:91], s[90:91], vcc\n\t"              /* the round's stop mask */ \
	    "s_waitcnt lgkmcnt(2)\n\t"                          /* the mask word (the two stores behind it may still be in flight) */ \
	    "v_bfe_u32 %[bit], %[word], " BIT_POSITION ", 1\n\t"  /* bit position modulo 32 */ \
	    "v_cmpx_ne_u32_e32 vcc, 0, %[bit]\n\t"              /* exec (the lanes still stepping) &= occupied */ \
	    "s_cbranch_execz Lrmd_dda_nocand%=\n\t" \
	    "v_add_u32 %[caddr], 0x100, %[caddr]\n\t"           /* the stored index stays: next slot */ \
	    "v_min_u32 %[budget], %[look1], %[budget]\n\t" \
	    "v_subrev_co_u32_e32 %[cleft], vcc, 1, %[cleft]\n\t" /* borrow: that was the lane's last slot */ \
	    "s_or_b64 s[90:91], s[90:91], vcc\n" \
	    "Lrmd_dda_nocand%=:\n\t" \
	    "s_andn2_b64 s[94:95], s[94:95], s[90:91]\n\t" \
	    "s_bcnt1_i32_b64 s98, s[94:95]\n\t" \
	    "s_mov_b64 exec, s[94:95]\n\t" \
	    "s_cmp_gt_u32 s98, %[cut]\n\t"                     /* go on while more than `cut` lanes are stepping (0: until none is) */ \
	    "s_cbranch_scc1 Lrmd_dda_loop%=\n\t" \
	    "s_mov_b64 exec, s[86:87]\n\t" \
	    "v_mov_b32 %[bit], 0\n\t" \
	    "s_and_b64 exec, s[88:89], s[86:87]\n\t" \
	    "v_mov_b32 %[bit], 1\n\t"                           /* the walk of these lanes is over */ \
	    "s_mov_b64 exec, s[86:87]\n\t" \
	    "v_add_u32 %[rx], 1, %[rx]\n\t" \
	    "v_add_u32 %[ry], 1, %[ry]\n\t" \
	    "v_add_u32 %[rz], 1, %[rz]\n\t" \
	    "s_waitcnt lgkmcnt(0)" \
	    : [tmx] "+v"(tmx), [tmy] "+v"(tmy), [tmz] "+v"(tmz), [idx] "+v"(idx), [prev] "+v"(prev), [rx] "+v"(remx), [ry] "+v"(remy), [rz] "+v"(remz), \
	      [caddr] "+v"(caddr), [bit] "=&v"(bit), [word] "=&v"(word), [budget] "=&v"(budget), [cleft] "=&v"(cleft) \
	    : [tdx] "v"(tdx), [tdy] "v"(tdy), [tdz] "v"(tdz), [dix] "v"(dix), [diy] "v"(diy), [diz] "v"(diz), [pad] "s"(mask_pad_bit), [mbase] "s"(mask_base), \
	      [limit] "s"(idx_limit), [cut] "s"(cut_lanes), [ncand1] "n"(RMD_WALK_CANDIDATES - 1), [look1] "n"(RMD_WALK_LOOKAHEAD - 1), \
	      [firstoff] "n"(sizeof(WalkScratch::start)) \
	    : "vcc", "scc", "memory", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98");
// LEAN (uniform per round): the mask holds a bit for every cell of the array, every cell inside the grid has an index inside the array
// (res.z <= res.y) and no lane of the wave started from a cell outside the grid (Q6): the index then needs neither the clamp to the mask's zero
// word nor the test against the end of the array — 19 + 15 instructions per step for 20 + 17 (two of the scalar ones count the lanes still stepping: cut_lanes).
template <bool LEAN>
RMD_DEV void dda_collect_candidates_asm(uint32_t mask_base, uint32_t mask_pad_bit, uint32_t idx_limit, uint32_t cand_base, bool &walking, uint32_t &n_cand,
                                             uint32_t &idx, uint32_t &prev, uint32_t &remx, uint32_t &remy, uint32_t &remz, double &tmx, double &tmy, double &tmz,
                                             double tdx, double tdy, double tdz, int32_t dix, int32_t diy, int32_t diz, uint32_t cut_lanes) {
	uint32_t caddr = cand_base, bit, word, budget, cleft;
	if constexpr (LEAN) {
		RMD_DDA_LOOP_ASM("v_lshrrev_b32 %[word], 5, %[idx]\n\t", "%[prev]" /* the index before the step */, "", "")
	} else {
		RMD_DDA_LOOP_ASM("v_min_u32 %[bit], %[pad], %[idx]\n\tv_lshrrev_b32 %[word], 5, %[bit]\n\t", "%[bit]", "v_cmp_le_u32_e64 s[96:97], %[limit], %[idx]\n\t",
		                 "s_or_b64 s[90:91], s[90:91], s[96:97]\n\t")
	}
	n_cand = (caddr - cand_base) >> 8;
	walking = bit == 0u;
}

// The walk's sphere pre-test (RMD_SPHERE_PREFILTER): does the LINE of the ray (origin pro, direction prd) pass the centre c of a triangle's sphere
// within sqrt(r2a + kb * |c - pro|^2)?  c, r2a: DevGrid::tri_sph (internal.hpp: triangle_sphere has the error argument); kb: DevGrid::sph_kb, the
// largest of the grid's triangles'.  This is the one piece of arithmetic on the hot path that is not the reference's: explicit fused multiply-adds
// (fewer instructions, smaller errors than the allowance assumes).  Precondition: |prd| = 1 to rounding — every ray of the render loop is a
// normalised vector (generate_primary_ray :332, :266 / :295; the thin lens :359) —: |d|^2 - (d.prd)^2 is the squared distance of the line from c
// only then (a longer direction would shrink the left-hand side and pass MORE pairs, a shorter one fewer: tests/test_gpu_reference_pins.py feeds
// the device form the adversarial pairs of tests/test_pretest_allowance.py through rmd_probe_pretest_pairs).  A NaN anywhere: the pair is dropped
// (the reference's test fails on a NaN too).
RMD_DEV bool sphere_pretest(V3 c, double r2a, double kb, V3 pro, V3 prd) {
	const V3 d = c - pro;
	const double along = __builtin_fma(d.x, prd.x, __builtin_fma(d.y, prd.y, d.z * prd.z)), dd = __builtin_fma(d.x, d.x, __builtin_fma(d.y, d.y, d.z * d.z));
	return __builtin_fma(-along, along, dd) <= __builtin_fma(kb, dd, r2a);
}

// Must be called by all 64 lanes of the wave in uniform control flow; `want` selects the lanes that have a ray.
// lds_mask: occupancy bits of this grid in LDS (bit i covers cells [i << shift, (i+1) << shift)).
// scr: this wave's scratch in LDS.
//
// Walks put aside (cut_lanes > 0, with `carry` = this wave's WalkCarry): a walk call lasts as long as its longest ray — ~85 steps where the
// average ray needs 26 — and most of a call's stepping iterations serve a handful of lanes.  With cut_lanes = K a round's stepping ends once at
// most K lanes are still stepping, and a call ends once at most K lanes are still walking; a lane whose walk is unfinished then stores its DDA
// state (`carried` comes back true, no result) and the caller presents the SAME ray again at the wave's next walk call with `carried` set, where the
// walk goes on from that state beside the new rays.  The steps, candidates and tests of every ray are those of an uninterrupted walk, in the
// same order: the result is the same.  The caller passes K > 0 only for calls with many more than K walkers (every call then advances every
// walker by at least a step) and K = 0 when the wave has nothing else to do, which finishes every walk.
// DEEP: the form of the split launches' kernel — walks may be put aside, and the chunk loop's owner search runs two chunks ahead; the plain form
// (direct mode: launches of a few samples per pixel; the probes) finishes every walk and searches one chunk ahead, in 10 fewer registers.
template <bool DEEP = false>
RMD_DEV void grid_intersect_wave(const DevGrid &g, const uint32_t *lds_mask, WalkScratch &scr, bool want, V3 ro, V3 rd, bool &hit_out,
                                 double &t_out, uint32_t &tri_out, uint32_t debug_flags = 0, unsigned long long *dbg = nullptr, uint32_t cut_lanes = 0u,
                                 WalkCarry *carry = nullptr, bool *carried = nullptr, uint32_t cut_round = 0u) {
	const uint32_t lane = threadIdx.x & 63u;
	const int32_t rx = (int32_t)g.res[0], ry = (int32_t)g.res[1], rz = (int32_t)g.res[2];
	const uint64_t resx = g.res[0], resz = g.res[2], n_cells = g.n_cells;
	const uint32_t mask_shift = g.mask_shift;
	const uint32_t mask_pad_bit = g.mask_n_words * 32u - 32u; // first bit of the all-zero word that ends every mask (api.cpp)
	// Every cell index is tested against the cell array (:129-131) — with Q5, or a start cell beyond the grid on an axis
	// that is never stepped, an index can pass the array while all range tests hold.  n_cells <= 2^31 and one step moves
	// the index by less than 2^31, so the wrapped 32-bit index is out of range exactly when the reference's usize one is.
	const uint32_t idx_limit = (uint32_t)n_cells;
	const RMD_GLOBAL CellEntry *entries = as_global(g.cell_entries);
	const RMD_GLOBAL uint32_t *ids = as_global(g.tri_ids);
	const RMD_GLOBAL unsigned char *recs = as_global(reinterpret_cast<const unsigned char *>(g.tri_recs));
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

	// every cell inside the grid has an index inside the cell array when res.z <= res.y (the Q5 index x + res.x*(y + z*res.z) of a
	// cell with y < res.y, z < res.z is then below res.x*res.y*res.z); with one mask bit per cell as well, the stepping loops need
	// neither the mask shift nor the index test — unless a lane starts from a cell outside the grid (Q6)
	[[maybe_unused]] const bool lean_grid = mask_shift == 0u && g.res[2] <= g.res[1];
	bool walking = false;
	[[maybe_unused]] bool start_outside = false;
	int32_t dix = 0, diy = 0, diz = 0;
	uint32_t idx = 0, remx = 1, remy = 1, remz = 1;
	uint32_t prev = kNoCell; // the cell the ray stood on before `idx`: selects the triangle list of a candidate (device_types.hpp: cell_entries)
#if RMD_WALK_STATE_STAND_IN
	double tmx = 0.0, tmy = 0.0, tmz = 0.0, tdx = 0.0, tdy = 0.0, tdz = 0.0;
#else
	double tmx, tmy, tmz, tdx, tdy, tdz; // set where a walk starts (or goes on) and only read by lanes that walk
#endif
#if RMD_DIAG
	const bool skip_walk = (debug_flags & 2u) != 0u, skip_tests = (debug_flags & 1u) != 0u; // timing ablations: wrong results, DIAG builds only
#else
	constexpr bool skip_walk = false, skip_tests = false;
#endif
	if (want && !skip_walk) {
		// acc_grid.rs:90-125
		V3 bmin = ld3(g.bbox_min);
		double t_outer;
		if (aabb_intersect(bmin, ld3(g.bbox_max), ro, rd, t_outer)) {
			int32_t cx = 0, cy = 0, cz = 0;
			V3 cs = ld3(g.cell_size);
			V3 start = ro - bmin;
			// (position / cell_size).cast::<i32>() (:94, :98): the cell size is the grid's, so the quotient comes from its exact reciprocal where the host
			// could provide one (device_core.hpp: div_by — three instructions for an IEEE division's fourteen; a non-finite position gives NaN
			// where the division gives an infinity: the cast refuses both)
			const V3 ics = ld3(g.inv_cell_size);
#if RMD_NO_CELL_RECIPROCAL
			const bool by_reciprocal = false; // (A/B)
#else
			const bool by_reciprocal = ics.x == ics.x && ics.y == ics.y && ics.z == ics.z; // uniform
#endif
			auto cell_of = [&](V3 p, int32_t &ix, int32_t &iy, int32_t &iz) {
				if (RMD_UNLIKELY(!by_reciprocal)) return cast_i32(p.x / cs.x, ix) && cast_i32(p.y / cs.y, iy) && cast_i32(p.z / cs.z, iz);
				return cast_i32(div_by(p.x, cs.x, ics.x), ix) && cast_i32(div_by(p.y, cs.y, ics.y), iy) && cast_i32(div_by(p.z, cs.z, ics.z), iz);
			};
			bool ok = cell_of(start, cx, cy, cz);
			if (ok && (cx < 0 || cy < 0 || cz < 0)) {
				V3 outer_pos = ro + rd * t_outer;
				start = outer_pos - bmin;
				ok = cell_of(start, cx, cy, cz);
			}
			ok = ok && !(rd.x != rd.x || rd.y != rd.y || rd.z != rd.z); // signum(NaN).cast::<i32>() panics: miss
			if (ok) {
				// first cell (:128-131): `as usize` sign-extends and the index arithmetic wraps (release build)
				const uint64_t idx0 = (uint64_t)(int64_t)cx + resx * ((uint64_t)(int64_t)cy + (uint64_t)(int64_t)cz * resz);
				ok = idx0 < n_cells; // else None
				idx = (uint32_t)idx0;
			}
			if (ok) {
				const int32_t sx = signbit(rd.x) ? -1 : 1, sy = signbit(rd.y) ? -1 : 1, sz = signbit(rd.z) ? -1 : 1;
				dix = sx, diy = sy * (int32_t)resx, diz = sz * (int32_t)(resx * resz);
				remx = steps_to_exit(cx, sx, rx), remy = steps_to_exit(cy, sy, ry), remz = steps_to_exit(cz, sz, rz);
				start_outside = cx < 0 || cy < 0 || cz < 0 || cx >= rx || cy >= ry || cz >= rz;
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

	if constexpr (!DEEP) cut_lanes = 0u, cut_round = 0u, carry = nullptr, carried = nullptr;
	if (carried) {
		if (*carried && walking) { // the walk goes on where the previous call left it
			tmx = carry->tm[0][lane], tmy = carry->tm[1][lane], tmz = carry->tm[2][lane];
			idx = carry->idx[lane], prev = carry->prev[lane];
			// (clamped to what a fresh walk could hold: the counters are this call's loop bound, and LDS is not to be trusted with that)
			remx = umin(carry->rem[0][lane], (uint32_t)rx + 1u), remy = umin(carry->rem[1][lane], (uint32_t)ry + 1u), remz = umin(carry->rem[2][lane], (uint32_t)rz + 1u);
		}
		*carried = false;
	}
	// A lane whose walk is put aside stops walking — the stepping runs under `walking`, so its DDA state stays in its registers as it is — and the
	// state is stored when the call ends (store_aside, below): during the call the carry's LDS serves the rounds' pre-test (the ring of pairs).
	bool aside = false;
	auto put_aside = [&]() { aside = true, walking = false; };
	auto store_aside = [&]() {
		if (aside) {
			carry->tm[0][lane] = tmx, carry->tm[1][lane] = tmy, carry->tm[2][lane] = tmz;
			carry->idx[lane] = idx, carry->prev[lane] = prev;
			carry->rem[0][lane] = remx, carry->rem[1][lane] = remy, carry->rem[2][lane] = remz;
			*carried = true;
		}
	};

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
	// The round loop needs no counter of its own: it is bounded by the rays' exit counters.  A round's stepping loop runs its body at least once
	// for every lane that is still walking (both forms test their condition at the END of an iteration), every step uses up one unit of one of the
	// lane's three counters, a counter that is used up ends the walk, and the counters start at no more than res + 1 each (steps_to_exit; a
	// carried walk's are clamped to that when they are loaded): after at most res.x + res.y + res.z + 3 rounds no lane walks, whatever the
	// candidates yield and whatever the memory holds.  (A counter was tried: 5 more spilled registers in the loop that runs at the kernel's
	// register limit.)  What a wave does BETWEEN its walk calls is watched by render_wave's stall watch.
	for (;;) {
		// 1. per lane: step along the ray (ALU + LDS only), recording up to kWalkCand candidate cells (mask bit set).
		//    A candidate is stepped over at once — speculating that it yields no hit — so that one round can gather the
		//    adjacent non-empty cells of a surface crossing; if an earlier candidate does hit, the later ones are simply
		//    ignored below.  Every cell the lane stands on has a valid index (< n_cells: checked for the first cell
		//    above and after every step, acc_grid.rs:129-131).  The step is computed while the mask word is in flight.
		// Candidates are parked in this lane's column of scr.first (slot m at [m * 64 + lane]) — the array is only
		// (re)written for the tests after every lane has read its candidates back, in program order within the wave.
		// `budget` = steps this lane may still take in this round: unlimited until its first candidate, kWalkLookahead
		// after it, 0 once kWalkCand candidates are recorded.
		uint32_t n_cand = 0;
		// the stepping loop is a chain of dependent LDS reads and branches run by a quarter of the wave's lanes: at a raised priority it is through sooner, and
		// what it yields the SIMD's other waves — their trips are dense vector code — take up (RMD_WALK_STEP_PRIO: measured −1.4 %)
		if constexpr (kWalkStepPrio != 0) __builtin_amdgcn_s_setprio(kWalkStepPrio);
		if (count_events && lane == 0) atomicAdd(&dbg[3], 1ull);
		// LEAN (uniform per round): the mask has one bit per cell and every cell inside the grid has an index inside the cell
		// array (res.z <= res.y), so neither the shift nor the index test is needed as long as no lane of the wave started
		// from a cell outside the grid (Q6).
#if RMD_WALK_ASM_LOOP
		if (mask_shift == 0u && !count_events) { // one mask bit per cell (uniform): the assembly loop (the event counters are in the C++ loop)
			// its lean form when the mask has a bit for every cell of the array, every cell inside the grid has an index inside the array and no
			// lane's walk started from a cell outside the grid (uniform per round)
			const bool lean = lean_grid && g.mask_bits >= idx_limit && __ballot(walking && start_outside) == 0ull;
			if (lean) {
				if (walking)
					dda_collect_candidates_asm<true>((uint32_t)(uintptr_t)lds_mask, mask_pad_bit, idx_limit, (uint32_t)(uintptr_t)&scr.start[lane], walking, n_cand, idx,
					                                 prev, remx, remy, remz, tmx, tmy, tmz, tdx, tdy, tdz, dix, diy, diz, cut_lanes);
			} else if (walking)
				dda_collect_candidates_asm<false>((uint32_t)(uintptr_t)lds_mask, mask_pad_bit, idx_limit, (uint32_t)(uintptr_t)&scr.start[lane], walking, n_cand, idx,
				                                  prev, remx, remy, remz, tmx, tmy, tmz, tdx, tdy, tdz, dix, diy, diz, cut_lanes);
		}
#else
		const bool lean = lean_grid && __ballot(walking && start_outside) == 0ull;
		if (lean) dda_collect_candidates<true>(lds_mask, mask_shift, mask_pad_bit, idx_limit, scr, lane, walking, n_cand, idx, prev, remx, remy, remz, tmx, tmy, tmz, tdx, tdy, tdz, dix, diy, diz, cut_lanes);
#endif
		else dda_collect_candidates<false>(lds_mask, mask_shift, mask_pad_bit, idx_limit, scr, lane, walking, n_cand, idx, prev, remx, remy, remz, tmx, tmy, tmz, tdx, tdy, tdz, dix, diy, diz, cut_lanes, count_events ? dbg : nullptr);
		if constexpr (kWalkStepPrio != 0) __builtin_amdgcn_s_setprio(0);
		RMD_STAMP(1)
		if (cut_lanes != 0u && walking && n_cand == 0u) put_aside(); // the stepping was cut short under a lane that has found nothing to test yet
		if (__ballot(n_cand != 0u) == 0ull) {
			if (__ballot(walking) == 0ull) break;
			continue; // look-ahead budget used up without a candidate: keep stepping
		}

		// 2. the candidates' cell entries {first id, count}: all gathers of the round in flight together.  Which of a cell's lists: the one
		//    without the triangles of the cell the ray came from, selected by the index step that led into the candidate (slot 0, the
		//    full list, for a walk's first cell — and for a step that is none of the ray's three, which cannot happen)
		uint32_t c_first[kWalkCand], c_count[kWalkCand];
		{
			// unconditional gathers (an unused slot reads cell 0) so that the kWalkCand loads overlap; masked afterwards
			uint32_t ci[kWalkCand], pv[kWalkCand];
			uint64_t ei[kWalkCand];
#pragma unroll
			for (uint32_t m = 0; m < kWalkCand; m++) ci[m] = scr.first[m * 64u + lane], pv[m] = scr.start[m * 64u + lane];
#pragma unroll
			for (uint32_t m = 0; m < kWalkCand; m++) {
				const uint32_t step = ci[m] - pv[m];
				const uint32_t slot = pv[m] == kNoCell ? 0u : step == (uint32_t)dix ? (dix > 0 ? 1u : 2u) : step == (uint32_t)diy ? (diy > 0 ? 3u : 4u) : step == (uint32_t)diz ? (diz > 0 ? 5u : 6u) : 0u;
				ei[m] = m < n_cand ? (uint64_t)ci[m] * kEntrySlots + slot : 0ull;
			}
#pragma unroll
			for (uint32_t m = 0; m < kWalkCand; m++) { // one 8-byte load per entry: {first, count}
				const unsigned long long e = reinterpret_cast<const RMD_GLOBAL unsigned long long *>(entries)[ei[m]];
				c_first[m] = (uint32_t)e, c_count[m] = (uint32_t)(e >> 32);
			}
#pragma unroll
			for (uint32_t m = 0; m < kWalkCand; m++) c_count[m] = (m < n_cand && !skip_tests) ? c_count[m] : 0u;
		}
		RMD_STAMP(2)

		// 3. triangle tests, distributed over the whole wave.  Lane l's candidates own c_count[0..] tests each; an exclusive
		//    prefix sum over the lanes numbers all tests of the round 0..T-1 in (lane, candidate, triangle) order and the wave
		//    takes them 64 at a time.  For a chunk, every lane whose tests overlap it drops lane+1 at the position where its
		//    tests begin inside the chunk (LDS), a max-scan carries that to the following positions, and each test then finds
		//    its candidate slot by comparing against the owner's pair starts (one 16-byte LDS read).  It fetches that pair's
		//    ray from its owner lane's registers (ds_bpermute), its triangle's index from the candidate's list (neighbouring
		//    tests read neighbouring entries of tri_ids) and the triangle's record by that index.  Hits are rare; they are applied
		//    by a scalar loop over the hit ballot in ascending lane order = ascending (lane, candidate, triangle) order with a
		//    strict '<': within a candidate cell that is the reference's sequential scan (acc_grid.rs:135-149: closest starts at
		//    5712515.0, first wins ties) over the triangles that can still hit (the list leaves out those the previous cell
		//    tested: they missed), and across candidates the earliest cell with an accepted hit wins (:151-153).
		uint32_t my_tests = 0;
#pragma unroll
		for (uint32_t m = 0; m < kWalkCand; m++) my_tests += c_count[m];
		const uint32_t incl_t = wave_scan_add(my_tests);
		const uint32_t total = readlane_u32(incl_t, 63);
		const uint32_t my_begin = incl_t - my_tests, my_end = incl_t;
		if (count_events && lane == 0) atomicAdd(&dbg[6], 1ull), atomicAdd(&dbg[8], (unsigned long long)total), atomicAdd(&dbg[9], (unsigned long long)((total + 63u) / 64u));
		bool any = false;
		uint32_t any_slot = 0, closest_tri = 0;
		double closest = 5712515.0;
		if (total != 0u) {
			{
				uint32_t run = my_begin;
#pragma unroll
				for (uint32_t m = 0; m < kWalkCand; m++) {
					scr.start[lane * kWalkCand + m] = run, scr.first[lane * kWalkCand + m] = c_first[m] - run; // (first entry of the pair's list MINUS the number of its first test: test w reads entry w + that)
					run += c_count[m];
				}
			}
			RMD_STAMP(3)
			// Owner search of one chunk: which (lane, candidate) pair test number base + lane belongs to, and its triangle record.
			// It runs ONE CHUNK AHEAD of the tests (RMD_WALK_SEARCH_AHEAD): a chunk's record loads are issued first, then the
			// rays are fetched and the next chunk is searched while the records are on their way.
			auto search = [&](uint32_t base, uint32_t &own, uint32_t &tri_id) {
				const uint32_t w = base + lane;
				scr.marker[lane] = 0u;
				if (my_tests != 0u && my_begin < base + 64u && my_end > base) scr.marker[umax(my_begin, base) - base] = lane + 1u;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const uint32_t owner_lane = wave_scan_max(scr.marker[lane]) - 1u; // position 0 is always marked
				own = 0;
				uint32_t id_index = 0;
				if (w < total) {
					// the pair's slot: the LAST one whose tests begin at or before w — the starts do not decrease from slot to slot (an empty slot shares its
					// start with the next one), so that is the number of slots 1 .. 3 whose start is <= w
					uint32_t slot = 0;
#pragma unroll
					for (uint32_t m = 1; m < kWalkCand; m++) slot += scr.start[owner_lane * kWalkCand + m] <= w ? 1u : 0u;
					own = owner_lane | (slot << 8);
					id_index = scr.first[owner_lane * kWalkCand + slot] + w;
				}
				tri_id = ids[id_index]; // (a lane without a test reads entry 0) — on its way while the current chunk is tested
				__builtin_amdgcn_wave_barrier(); // every lane has read the markers before the next search rewrites them
			};
			// One chunk of tests.  (own_x, tri_x): the chunk's owner pairs and triangle indices, searched TWO chunks ago (RMD_WALK_SEARCH_AHEAD = 2) —
			// the index load has had a whole chunk to arrive — and overwritten here by the search of the chunk two ahead.
			constexpr uint32_t ahead = DEEP ? kWalkSearchAhead : 1u;
			auto chunk = [&](uint32_t base, uint32_t &own_x, uint32_t &tri_x) {
				const uint32_t w = base + lane;
				// every lane loads a record (a lane without a test: some triangle's) and tests it — no zero-filled stand-in, no branch around the loads
				const uint32_t tri = tri_x;
				const TriRecord r = load_record(recs + (size_t)tri * kTriRecStride);
				// the owner lane's ray, straight from its registers (every lane takes part in the permute)
				const int src = (int)((own_x & 63u) << 2);
				const V3 pro = mk(bperm_f64(src, ro.x), bperm_f64(src, ro.y), bperm_f64(src, ro.z));
				const V3 prd = mk(bperm_f64(src, rd.x), bperm_f64(src, rd.y), bperm_f64(src, rd.z));
				const uint32_t own_now = own_x;
				if (base + 64u * ahead < total) search(base + 64u * ahead, own_x, tri_x);
#if RMD_FLAT_TRIANGLE_TEST
				double t;
				const bool h = triangle_test_flat(r.v0, r.e1, r.e2, pro, prd, t) && w < total;
#else
#if RMD_TRIANGLE_T_STAND_IN
				double t = 0.0; // (A/B: round 4's form — a stand-in costs a 64-bit copy at each of the test's four exits)
#else
				double t; // set by a hit and only read — below, by readlane — for the lanes of `h`
#endif
				const bool h = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, t) && w < total;
#endif
#if !defined(RMD_STAMP_OUTER_ONLY)
				RMD_STAMP(5)
#endif
				unsigned long long hits = __ballot(h);
				while (hits) {
					const int l = (int)__builtin_ctzll(hits);
					hits &= hits - 1ull;
					const uint32_t ol = readlane_u32(own_now, l);
					const double tl = readlane_f64(t, l);
					const uint32_t tril = readlane_u32(tri, l);
					const uint32_t slot = ol >> 8;
					if (lane == (ol & 63u) && (!any || slot == any_slot)) {
						if (tl < closest) { // `closest` is still 5712515.0 until the first accepted hit
							closest = tl;
							closest_tri = tril;
							any_slot = slot;
							any = true;
						}
					}
				}
			};
			// One chunk of tests behind the sphere pre-test (RMD_SPHERE_PREFILTER; the walks of the split launches, DEEP).  85 % of the (ray, triangle)
			// pairs a round numbers fail a test that costs a quarter of triangle.rs:11-44: the ray's line passes the triangle's sphere (DevGrid::tri_sph,
			// 32 bytes instead of the 72-byte record) by.  pretest() runs that on a chunk — owner search, ray fetch and index lookahead as in chunk() —
			// and appends the pairs that pass, in their order, to a ring in LDS (the wave's WalkCarry: its contents are loaded when a call begins and
			// stored when it ends; in between it is free); whenever 64 pairs wait — and for what is left at the round's end — full() runs the
			// reference's test on them and applies the hits exactly as chunk() does.  The pairs that pass keep their ascending (lane, candidate,
			// triangle) order, so the hit rule sees the same hits in the same order; a pair that is dropped is a pair whose test fails
			// (internal.hpp: triangle_sphere has the argument; tests/test_pretest_allowance.py checks it on adversarial pairs; tests/test_gpu_faults.py counts, in a DIAG build, that no dropped pair passes the test).
			[[maybe_unused]] unsigned long long *ring = reinterpret_cast<unsigned long long *>(carry); // 128 entries of {owner | slot << 8, triangle}
			[[maybe_unused]] uint32_t ring_head = 0u, ring_tail = 0u;
			[[maybe_unused]] const RMD_GLOBAL double *spheres = as_global(g.tri_sph);
			[[maybe_unused]] const double sph_kb = g.sph_kb;
			[[maybe_unused]] auto full = [&]() {
				const uint32_t n = umin(ring_tail - ring_head, 64u);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const unsigned long long e = ring[(ring_head + lane) & 127u]; // (a lane beyond n reads an older entry or what the carry left there: masked on the next lines)
				ring_head += n;
				const bool valid = lane < n;
				if (count_events && lane == 0) atomicAdd(&dbg[18], 1ull);
				const uint32_t own_now = valid ? (uint32_t)e : 0u, tri = valid ? (uint32_t)(e >> 32) : 0u;
				const TriRecord r = load_record(recs + (size_t)tri * kTriRecStride);
				const int src = (int)((own_now & 63u) << 2);
				const V3 pro = mk(bperm_f64(src, ro.x), bperm_f64(src, ro.y), bperm_f64(src, ro.z));
				const V3 prd = mk(bperm_f64(src, rd.x), bperm_f64(src, rd.y), bperm_f64(src, rd.z));
				double t; // set by a hit and only read — below, by readlane — for the lanes of `h`
				const bool h = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, t) && valid;
				unsigned long long hits = __ballot(h);
				while (hits) {
					const int l = (int)__builtin_ctzll(hits);
					hits &= hits - 1ull;
					const uint32_t ol = readlane_u32(own_now, l);
					const double tl = readlane_f64(t, l);
					const uint32_t tril = readlane_u32(tri, l);
					const uint32_t slot = ol >> 8;
					if (lane == (ol & 63u) && (!any || slot == any_slot)) {
						if (tl < closest) { // `closest` is still 5712515.0 until the first accepted hit
							closest = tl;
							closest_tri = tril;
							any_slot = slot;
							any = true;
						}
					}
				}
				__builtin_amdgcn_wave_barrier(); // every lane has read its entry before the ring is written again
			};
			[[maybe_unused]] auto pretest = [&](uint32_t base, uint32_t &own_x, uint32_t &tri_x) {
				const uint32_t w = base + lane;
				const uint32_t tri = tri_x, own_now = own_x;
				const RMD_GLOBAL double *sp = spheres + (size_t)tri * 4u;
				const V3 c = ld3(sp);
				const double r2a = sp[3];
				const int src = (int)((own_x & 63u) << 2);
				const V3 pro = mk(bperm_f64(src, ro.x), bperm_f64(src, ro.y), bperm_f64(src, ro.z));
				const V3 prd = mk(bperm_f64(src, rd.x), bperm_f64(src, rd.y), bperm_f64(src, rd.z));
				if (base + 64u * ahead < total) search(base + 64u * ahead, own_x, tri_x);
				const bool pass = w < total && sphere_pretest(c, r2a, sph_kb, pro, prd);
#if RMD_DIAG
				if (count_events && (debug_flags & 64u)) { // cross-check: a pair the pre-test drops must fail the reference's test (dbg[16] stays 0)
					const TriRecord r = load_record(recs + (size_t)tri * kTriRecStride);
					double tt;
					const bool hh = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, tt) && w < total;
					const unsigned long long bad = __ballot(hh && !pass);
					if (bad != 0ull && lane == 0) atomicAdd(&dbg[16], (unsigned long long)__popcll(bad));
				}
#endif
				const unsigned long long pm = __ballot(pass);
				if (pass) ring[(ring_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u))) & 127u] = (unsigned long long)own_now | ((unsigned long long)tri << 32);
				ring_tail += (uint32_t)__popcll(pm);
				if (count_events && lane == 0) atomicAdd(&dbg[17], (unsigned long long)__popcll(pm));
				if (ring_tail - ring_head >= 64u) full();
			};
			// RMD_SPHERE_AHEAD: the same with the NEXT chunk's spheres requested while this chunk is tested.  The chunk loop is what a walk's time is
			// (DIAG phase clocks of render_wave_queued: the chunks are ~85 % of a walk call, the walks ~60 % of the kernel) and what it waits for is the
			// 32-byte gather of each pair's sphere — scattered over 3.2 MB, behind an L2 that the scene's tables overflow.  pretest() requests a
			// chunk's spheres where it needs them (under the ray fetch and the search of the chunk after next: half a chunk of cover); here the
			// search comes first, then the request for the spheres of the chunk that is tested NEXT (its triangle indices were requested a chunk
			// ago), then this chunk's tests on the spheres the previous chunk requested: a whole chunk of cover for 8 more live registers.
			struct Sph {
				V3 c;
				double r2a;
			};
			[[maybe_unused]] auto load_sph = [&](uint32_t tri) {
				const RMD_GLOBAL double *sp = spheres + (size_t)tri * 4u;
				Sph sph;
				sph.c = ld3(sp), sph.r2a = sp[3];
				return sph;
			};
			[[maybe_unused]] auto pretest_ahead = [&](uint32_t base, uint32_t &own_x, uint32_t &tri_x, const Sph &cur, const uint32_t &tri_next, Sph &next, bool have_next) {
				const uint32_t w = base + lane;
				const uint32_t tri = tri_x, own_now = own_x;
				if (base + 64u * ahead < total) search(base + 64u * ahead, own_x, tri_x);
				if (have_next) next = load_sph(tri_next);
				const int src = (int)((own_now & 63u) << 2);
				const V3 pro = mk(bperm_f64(src, ro.x), bperm_f64(src, ro.y), bperm_f64(src, ro.z));
				const V3 prd = mk(bperm_f64(src, rd.x), bperm_f64(src, rd.y), bperm_f64(src, rd.z));
				const bool pass = w < total && sphere_pretest(cur.c, cur.r2a, sph_kb, pro, prd);
#if RMD_DIAG
				if (count_events && (debug_flags & 64u)) { // cross-check: a pair the pre-test drops must fail the reference's test (dbg[16] stays 0)
					const TriRecord r = load_record(recs + (size_t)tri * kTriRecStride);
					double tt;
					const bool hh = triangle_intersect(r.v0, r.e1, r.e2, pro, prd, tt) && w < total;
					const unsigned long long bad = __ballot(hh && !pass);
					if (bad != 0ull && lane == 0) atomicAdd(&dbg[16], (unsigned long long)__popcll(bad));
				}
#endif
				const unsigned long long pm = __ballot(pass);
				if (pass) ring[(ring_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u))) & 127u] = (unsigned long long)own_now | ((unsigned long long)tri << 32);
				ring_tail += (uint32_t)__popcll(pm);
				if (count_events && lane == 0) atomicAdd(&dbg[17], (unsigned long long)__popcll(pm));
				if (ring_tail - ring_head >= 64u) full();
			};
#ifndef RMD_SPHERE_AHEAD
#define RMD_SPHERE_AHEAD 0 // measured (round 6): 38 spilled registers for 7 in the queued mesh kernel (53 for 11 in the lane-per-path one), C3 at 200 spp 130.6 ms against 123.4: not used
#endif
			uint32_t own_a = 0, tri_a = 0;
			search(0u, own_a, tri_a);
			if constexpr (DEEP && RMD_SPHERE_PREFILTER) {
				static_assert(sizeof(WalkCarry) >= 128u * sizeof(unsigned long long), "the ring of pairs that passed the pre-test lives in the wave's WalkCarry");
				uint32_t own_b = 0, tri_b = 0;
				if (64u < total) search(64u, own_b, tri_b);
				if constexpr (RMD_SPHERE_AHEAD && ahead == 2u) {
					Sph sph_a = load_sph(tri_a), sph_b;
					RMD_UNDEF(sph_b.c.x) RMD_UNDEF(sph_b.c.y) RMD_UNDEF(sph_b.c.z) RMD_UNDEF(sph_b.r2a)
					for (uint32_t base = 0; base < total; base += 128u) {
						pretest_ahead(base, own_a, tri_a, sph_a, tri_b, sph_b, base + 64u < total);
						if (base + 64u >= total) break;
						pretest_ahead(base + 64u, own_b, tri_b, sph_b, tri_a, sph_a, base + 128u < total);
					}
				} else {
					for (uint32_t base = 0; base < total; base += 128u) {
						pretest(base, own_a, tri_a);
						if (base + 64u >= total) break;
						pretest(base + 64u, own_b, tri_b);
					}
				}
				if (ring_tail != ring_head) full(); // (fewer than 64 are left: full() is run as soon as 64 wait)
			} else if constexpr (ahead == 2u) { // two register pairs take turns (no copies: a copy would wait for the load it copies)
				uint32_t own_b = 0, tri_b = 0;
				if (64u < total) search(64u, own_b, tri_b);
				for (uint32_t base = 0; base < total; base += 128u) {
					chunk(base, own_a, tri_a);
					if (base + 64u >= total) break;
					chunk(base + 64u, own_b, tri_b);
				}
			} else {
				for (uint32_t base = 0; base < total; base += 64u) chunk(base, own_a, tri_a);
			}
			RMD_STAMP(6)
			__builtin_amdgcn_wave_barrier(); // the scratch is rewritten next round
		}
		if (any) { // :151-153 first cell with any hit wins
			found = true;
			found_t = closest;
			found_tri = closest_tri;
			walking = false;
		}
		const unsigned long long still = __ballot(walking);
		if (still == 0ull) break;
		if ((uint32_t)__popcll(still) <= cut_round) { // not worth another round: these walks go on in the wave's next call
			if (walking) put_aside();
			break;
		}
	}
	if constexpr (DEEP) store_aside();
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
