// Device functions of the radiance integrator (gfx950, binary64, no FMA contraction).
// Each function names the reference code whose arithmetic — operation order included — it performs.
// Paths are relative to the reference tree (Nyrox/raymond).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.hpp"

#define RMD_DEV __device__ __forceinline__
// measured micro-optimisations, each bit-exact; the switches exist for tools/ab_multi.sh
#ifndef RMD_OPT_BITOP3
#if defined(__gfx950__) || !defined(__HIP_DEVICE_COMPILE__)
#define RMD_OPT_BITOP3 1 // v_bitop3_b32 exists on gfx950 only (the host pass just parses the builtin)
#else
#define RMD_OPT_BITOP3 0 // `make ARCH=...` for another target: the two-XOR form, bit-identical
#endif
#endif
#ifndef RMD_OPT_SHARED_SQRT
#define RMD_OPT_SHARED_SQRT 1
#endif
// Pointers read out of structs in memory are generic-address-space to the compiler, which then emits flat_load +
// full waits; these casts state that they point to global memory (HBM), giving global_load and counted waits.
#define RMD_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const RMD_GLOBAL T *as_global(const T *p) {
	return (const RMD_GLOBAL T *)p;
}

namespace rmd {

// ---------------------------------------------------------------- cgmath::Vector3<f64> subset
struct V3 {
	double x, y, z;
};
RMD_DEV V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RMD_DEV V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
RMD_DEV V3 ld3(const RMD_GLOBAL double *p) { return V3{p[0], p[1], p[2]}; }
RMD_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RMD_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RMD_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
RMD_DEV V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
RMD_DEV V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
RMD_DEV V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
RMD_DEV V3 hadamard(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RMD_DEV double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } // cgmath: mul_element_wise().sum()
RMD_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
#if RMD_NO_BRANCH_HINTS
#define RMD_UNLIKELY(c) (c)
#else
#define RMD_UNLIKELY(c) __builtin_expect(!!(c), 0) // the fallback of a wave-level range test: laid out away from the path every trip takes
#endif
// IEEE square root.  The compiler's expansion of sqrt(double) pre-scales arguments below 2^-767 (compare, two selects, two ldexp around the
// refinement) and ends with a select that returns +-0 and +inf unchanged (class compare, two selects); no length, discriminant or area on this
// path is that small, zero or infinite in the ordinary course, so when every lane of the wave holds a finite argument >= 2^-767 — ONE unsigned
// range test on the high word: 0x10000000 <= hi < 0x7FF00000 excludes small, zero, negative, infinite and NaN arguments alike — the same
// refinement runs without the scaling and without the final select: identical operations on identical values, hence identical results
// (tools/microbench: 0 mismatches in 3.4e11 arguments).  Otherwise (one ballot) the whole wave takes the compiler's sequence.
// RMD_SQRT_RANGE_TEST: 1 = this form (two integer instructions of special-casing per root); 0 = round 4's (two f64 compares in front for the small
// arguments, the class select behind: five instructions, three of them at the price of an f64 addition each — tools/microbench/valu_rate.hip);
// 2 = one integer compare in front, the select behind.  Measured with the branch-free object tests (C2 / every path traced / C3 / C3 with black
// paths ended): 49.0 / 86.9 / 418.0 / 263.2 ms with 0, 49.0 / 86.0 / 410.5 / 260.2 with 1, 49.0 / 86.8 / 417.1 / 262.8 with 2.
#ifndef RMD_SQRT_RANGE_TEST
#define RMD_SQRT_RANGE_TEST 1
#endif
RMD_DEV double sqrt64(double x) {
#if RMD_SQRT_RANGE_TEST == 1
	const uint32_t hi = (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32);
	if (RMD_UNLIKELY(__ballot(hi - 0x10000000u >= 0x7FF00000u - 0x10000000u) != 0ull)) return __builtin_sqrt(x);
#elif RMD_SQRT_RANGE_TEST == 2
	// +0 and positive arguments below 2^-767 as ONE unsigned compare of the high word (an f64 compare costs as much as an f64 addition, an integer
	// one half: tools/microbench/valu_rate.hip); negative, infinite and NaN arguments take the refinement as before
	const uint32_t hi = (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32);
	if (__ballot(hi < 0x10000000u) != 0ull) return __builtin_sqrt(x);
#else
	if (__ballot(x < 0x1p-767 && x > 0.0) != 0ull) return __builtin_sqrt(x);
#endif
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y, h = y * 0.5;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g), h = __builtin_fma(h, r, h);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
#if RMD_SQRT_RANGE_TEST == 1
	return g;
#else
	return __builtin_amdgcn_class(x, 0x260) ? x : g; // +-0 and +inf return themselves (class mask: -0 | +0 | +inf)
#endif
}
RMD_DEV double length(V3 a) { return sqrt64(dot(a, a)); }
// a / b for a divisor whose correctly rounded reciprocal r = 1.0 / b was computed beforehand (host: exact_reciprocal(),
// internal.hpp): q = a*r, then Markstein's correction q + (a - b*q)*r with the residual exact in the FMA — three
// instructions for the 13 + v_rcp_f64 of a division.  The result is the IEEE quotient for every divisor whose significand is
// not all ones (the host stores NaN for those and for zero / extreme divisors, and the call sites then divide plainly);
// tools/microbench/div_by_reciprocal_check.hip: 0 mismatches in 3.4e11 quotients.  a must be finite.
// 1 / b to within an ulp or two: the hardware estimate and two Newton steps, without the scaling, special-case and final rounding
// steps of an IEEE division (5 instructions for 11).  For values that only scale a sample's weight.
RMD_DEV double fast_rcp(double b) {
	double y = __builtin_amdgcn_rcp(b);
	y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
	y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
	return y;
}
// a / b as the IEEE division computes it, WITHOUT the division's scaling and special-case instructions: the compiler expands an f64 division into
// v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal (4 FMAs), q = a * y, the residual FMA, v_div_fmas (the final FMA, times 2^+-64 when
// v_div_scale has scaled) and v_div_fixup (zeros, infinities, NaNs, denormal results) — and when nothing is scaled and nothing is special, the scale
// instructions pass their operands through, v_div_fmas IS the final FMA and v_div_fixup passes the quotient through: the eight instructions below are
// the division's own arithmetic, operation for operation, hence the same bits (tools/microbench/div_lean_check.hip: 0 mismatches against `a / b` in
// 6.9e10 quotients over the admitted range, divisors and numerators next to all-ones and to powers of two).  Admitted: b finite, normal, non-zero,
// and 2^-700 <= |a| <= 2^700 with |exponent(a) - exponent(b)| < 700 (v_div_scale acts on exponent differences >= 768, on denormals and on
// numerators below 2^-969), or a = +0 over a positive b; NOT a zero in general: the sequence loses the quotient's sign of zero (-0 / b comes out +0).
// NaNs give NaNs.  The CALLER guarantees the range (see the call sites): there is no test in here.
// Four to five instructions fewer per division, five divisions per path segment on a scene of planes and spheres.
#ifndef RMD_LEAN_DIVISION
#define RMD_LEAN_DIVISION 1
#endif
RMD_DEV double div_lean(double a, double b) {
#if RMD_LEAN_DIVISION
	double y = __builtin_amdgcn_rcp(b);
	y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
	y = __builtin_fma(y, __builtin_fma(-b, y, 1.0), y);
	const double q = a * y;
	return __builtin_fma(__builtin_fma(-b, q, a), y, q);
#else
	return a / b;
#endif
}
RMD_DEV double div_by(double a, double b, double r) {
	const double q = a * r;
	return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
// root = sqrt(x) and inv = 1.0 / root, both correctly rounded — the two operations of cgmath's normalize — for the price of
// the square root plus five FMAs.  The root's refinement already carries h ~ 1/(2*root); one Newton step on 2h and the
// IEEE division's own final correction (e = 1 - root*r; inv = r + e*r) give the quotient the 13-instruction division
// sequence (v_rcp_f64 at 1/3 rate, two Newton steps, scaling and fix-up) would.  The one divisor that correction cannot
// round is a root whose significand is all ones (Markstein): 1/root then lies 2^-106 above a rounding midpoint — and it is
// common here, because re-normalising a unit vector takes the root of 1 - 2^-53.  In that case the quotient is the power
// of two the estimate was rounded to plus one ulp, set directly.  Arguments outside [2^-700, 2^700) send the whole wave
// down the plain sqrt + division (one ballot).  tools/microbench/inv_length_check.hip compares the shortcut with
// sqrt + division bit for bit: 0 mismatches in 3.4e11 arguments, 4.8e9 of them with all-ones roots.
RMD_DEV void sqrt_and_inverse(double x, double &root, double &inv) {
#ifndef RMD_SQRT_INV_RANGE_TEST
#define RMD_SQRT_INV_RANGE_TEST 1 // an f64 compare costs as much as an f64 addition, an integer compare half (tools/microbench/valu_rate.hip): C2 51.3 -> 50.6 ms
#endif
#if RMD_SQRT_INV_RANGE_TEST
	const uint32_t hi = (uint32_t)(__builtin_bit_cast(unsigned long long, x) >> 32); // 2^-700 <= x < 2^700 as one unsigned range test on the high word
	if (RMD_UNLIKELY(__ballot(hi - 0x14300000u >= 0x6BB00000u - 0x14300000u) != 0ull)) {
#else
	if (__ballot(!(x >= 0x1p-700 && x <= 0x1p700)) != 0ull) {
#endif
		root = sqrt64(x);
		inv = 1.0 / root;
		return;
	}
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y, h = y * 0.5;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g), h = __builtin_fma(h, r, h);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	const double r0 = h + h;
	const double r1 = __builtin_fma(__builtin_fma(-g, r0, 1.0), r0, r0);
	root = g;
	inv = __builtin_fma(__builtin_fma(-g, r1, 1.0), r1, r1);
	const unsigned long long gb = __builtin_bit_cast(unsigned long long, g);
	if ((uint32_t)gb == 0xFFFFFFFFu && ((uint32_t)(gb >> 32) | 0xFFF00000u) == 0xFFFFFFFFu)
		inv = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, r1) | 1ull);
}
RMD_DEV V3 normalize(V3 a) { // cgmath normalize_to(1.0): a * (1.0 / magnitude)
	double len, inv;
	sqrt_and_inverse(dot(a, a), len, inv);
	return a * inv;
}
RMD_DEV double dist(V3 a, V3 b) { return length(b - a); }     // MetricSpace::distance
// a / |a| to within a few ulp: the hardware reciprocal square root and two Newton steps (16 instructions for normalize()'s 27).
// For vectors that only enter a sample's weight (the half vector of the BRDF terms).
// A dot product for quantities that only scale a sample's weight (RMD_WEIGHT_FMA: fused, three instructions for five; the reference's
// unfused sum otherwise)
#ifndef RMD_WEIGHT_FMA
#define RMD_WEIGHT_FMA 0
#endif
RMD_DEV double dot_w(V3 a, V3 b) {
#if RMD_WEIGHT_FMA
	return __builtin_fma(a.x, b.x, __builtin_fma(a.y, b.y, a.z * b.z));
#else
	return dot(a, b);
#endif
}
RMD_DEV V3 normalize_for_weight(V3 a) {
	const double x = dot_w(a, a), h = 0.5 * x;
	double y = __builtin_amdgcn_rsq(x);
	y = y * __builtin_fma(-(h * y), y, 1.5);
	y = y * __builtin_fma(-(h * y), y, 1.5);
	return a * y;
}

// A binary64 constant made on the spot in a scalar register pair.  Written as a literal the compiler materialises such a constant in a vector
// register pair, hoists that out of the render loop and — at the grid kernel's register limit — spills it: the trip then waits on a scratch
// load for the value of a constant (three of them, five reloads per trip, in round 3's grid kernel).  volatile: stays where it is used.
RMD_DEV double scalar_const(double c) {
	const unsigned long long b = __builtin_bit_cast(unsigned long long, c);
	uint32_t lo, hi;
	asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((uint32_t)b), "n"((uint32_t)(b >> 32)));
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

constexpr double kPi = 3.14159265358979323846; // core/src/math.rs:19
constexpr double kFMax = 1.7976931348623157e308; // core/src/math.rs:20

// ---------------------------------------------------------------- RNG (include/raymond_hip.h "RNG")
RMD_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t &o0,
                           uint32_t &o1, uint32_t &o2, uint32_t &o3) {
	constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
	for (int round = 0; round < 10; round++) {
		const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2; // one 32x32->64 multiply each (v_mad_u64_u32)
		const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
		const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
		// a ^ b ^ c in one instruction: gfx950's v_bitop3_b32 with the truth table of a three-way XOR (the compiler emits two v_xor_b32)
#if RMD_OPT_BITOP3
		uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#else
		uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
#endif
		c0 = n0, c1 = lo1, c2 = n2, c3 = lo0;
		k0 += W0, k1 += W1;
	}
	o0 = c0, o1 = c1, o2 = c2, o3 = c3;
}

// Per-lane counter RNG (include/raymond_hip.h "RNG").  A sample's random numbers come in BLOCKS — one Philox4x32-10
// evaluation, counter (pixel, sample, block, 0) — and every consumer takes exactly one block: block 0 is the pixel jitter,
// each round of the thin lens' rejection loop one block, each shaded depth one block.  The state is the next block's index.
struct Rng {
	uint32_t pixel, sample, block;
	uint32_t lobe_bits; // the 22 spare bits of the path's last block: `r` (:260) of the NEXT shaded depth, r = lobe_bits * 2^-22
	RMD_DEV void init(uint32_t pixel_, uint32_t sample_) { pixel = pixel_, sample = sample_, block = 0u, lobe_bits = 0u; }
	static RMD_DEV uint32_t spare22(uint32_t w0, uint32_t w2) { return ((w0 & 0x7FFu) << 11) | (w2 & 0x7FFu); }
	static RMD_DEV double to_unit(uint32_t lo, uint32_t hi) { // 53-bit uniform in [0, 1), as rand 0.6's f64
		uint64_t bits = (((uint64_t)hi << 32) | lo) >> 11;
		return (double)bits * (1.0 / 9007199254740992.0);
	}
	// the 22 bits the two 53-bit conversions of a block discard, as a uniform in [0, 1)
	static RMD_DEV double to_unit22(uint32_t w0, uint32_t w2) { return (double)(((w0 & 0x7FFu) << 11) | (w2 & 0x7FFu)) * (1.0 / 4194304.0); }
	// two uniforms: pixel jitter x, y (src/trace.rs:326-327); one round of the lens rejection loop (:340-341)
	RMD_DEV void next2(uint32_t k0, uint32_t k1, double &u0, double &u1) {
		uint32_t w0, w1, w2, w3;
		philox4x32_10(pixel, sample, block, 0u, k0, k1, w0, w1, w2, w3);
		block++;
		u0 = to_unit(w0, w1), u1 = to_unit(w2, w3);
		lobe_bits = spare22(w0, w2);
	}
	// the three draws of one shaded depth: r (:260) decides diffuse against specular — it is only ever compared with 0.5
	// (Diffuse) or 0.0 (Metal) (:263-264), for which a 22-bit uniform gives the same probabilities as a 53-bit one — and r1, r2
	// (:397-398 or :287-288) are the block's two 53-bit uniforms
	RMD_DEV void next3(uint32_t k0, uint32_t k1, double &r, double &r1, double &r2) {
		uint32_t w0, w1, w2, w3;
		r = (double)lobe_bits * (1.0 / 4194304.0); // of the path's previous block
		philox4x32_10(pixel, sample, block, 0u, k0, k1, w0, w1, w2, w3);
		block++;
		r1 = to_unit(w0, w1), r2 = to_unit(w2, w3);
		lobe_bits = spare22(w0, w2);
	}
};

// ---------------------------------------------------------------- primitives
// The primitives come in two forms: `*_visit(..., on_hit)` calls on_hit(t) where the hit is found — what the closest-hit loops use, so that the
// distance is consumed inside the branch that computed it and never has to exist, as a value, on the paths that miss (held in a variable that
// leaves the branch it costs a 64-bit copy per nesting level and object: tools/experiments/README.md, round 5) — and the reference's
// `intersects(ray) -> Option<distance>` shape built on top of it for everything else.
// core/src/geometry/primitives/sphere.rs:11-27
template <class F>
RMD_DEV void sphere_visit(V3 center, double radius, V3 ro, V3 rd, F &&on_hit) {
	V3 c = center - ro;
	double t = dot(c, rd);
	V3 q = c - t * rd;
	double p = dot(q, q);
	double r2 = radius * radius;
	if (p > r2) return;
	t -= sqrt64(r2 - p);
	if (t <= 0.0) return;
	on_hit(t);
}
RMD_DEV bool sphere_intersect(V3 center, double radius, V3 ro, V3 rd, double &t_out) {
	bool hit = false;
	sphere_visit(center, radius, ro, rd, [&](double t) { t_out = t, hit = true; });
	return hit;
}
// core/src/geometry/primitives/plane.rs:11-24
template <class F>
RMD_DEV void plane_visit(V3 origin, V3 normal, V3 ro, V3 rd, F &&on_hit) {
	double denom = dot(normal, -rd);
	if (denom > 1e-6) {
		V3 p0l0 = origin - ro;
		double t = dot(p0l0, -normal) / denom;
		if (t >= 0.0) on_hit(t);
	}
}
RMD_DEV bool plane_intersect(V3 origin, V3 normal, V3 ro, V3 rd, double &t_out) {
	bool hit = false;
	plane_visit(origin, normal, ro, rd, [&](double t) { t_out = t, hit = true; });
	return hit;
}
// Two planes whose normals are exact negations of each other (the opposite walls of a box), tested together.  Plane::intersects
// (plane.rs:11-24) is one-sided: a ray is tested further only against a plane it faces, `dot(normal, -rd) > 1e-6`.  With n_b = -n_a every
// product and sum of the second denominator is the exact negation of the first's (round-to-nearest is symmetric), so
// dot(n_b, -rd) = -dot(n_a, -rd) bit for bit (up to the sign of a zero, which no comparison here sees): the two facing conditions
// exclude each other, and a lane's one division is num / |denom_a| with ITS plane's numerator — the operands and the IEEE division
// plane.rs:17 performs for that plane.  One division sequence at full lane occupancy instead of two at about half each.
// `first` tells which plane a hit belongs to.
template <class F>
RMD_DEV void plane_pair_visit(V3 origin_a, V3 normal_a, V3 origin_b, V3 normal_b, V3 ro, V3 rd, F &&on_hit) { // on_hit(t, first)
	const double denom_a = dot(normal_a, -rd);
	const bool faces_a = denom_a > 1e-6, faces_b = -denom_a > 1e-6; // = dot(normal_b, -rd) > 1e-6
	const double num_a = dot(origin_a - ro, -normal_a), num_b = dot(origin_b - ro, -normal_b);
	if (faces_a || faces_b) {
		const double t = (faces_b ? num_b : num_a) / __builtin_fabs(denom_a);
		if (t >= 0.0) on_hit(t, faces_a);
	}
}
RMD_DEV bool plane_pair_intersect(V3 origin_a, V3 normal_a, V3 origin_b, V3 normal_b, V3 ro, V3 rd, double &t_out, bool &first) {
	bool hit = false;
	plane_pair_visit(origin_a, normal_a, origin_b, normal_b, ro, rd, [&](double t, bool f) { t_out = t, first = f, hit = true; });
	return hit;
}
// The same three tests without control flow (RMD_FLAT_OBJECT_TESTS, the closest-hit loops): every lane computes the whole test and ONE condition
// says whether its result counts.  The operations that produce a counted distance are those of the branching forms, in the same order, on the
// same values; a lane whose ray misses computes a quotient or root nobody reads (a division by a small or zero denominator, the root of a negative
// number: no traps on this hardware).  Why: a value that lives across a divergent branch and is assigned inside it — the closest distance and
// object so far — costs the compiler a 64-bit and a 32-bit copy per nesting level (three or four levels per object), more than the tests' early
// exits save (they only help when NO lane of the wave gets past them).
RMD_DEV bool sphere_test_flat(V3 center, double radius, V3 ro, V3 rd, double &t_out) {
	V3 c = center - ro;
	double t = dot(c, rd);
	V3 q = c - t * rd;
	double p = dot(q, q);
	double r2 = radius * radius;
#if RMD_SQRT_RANGE_TEST == 1
	t -= sqrt64(__builtin_fabs(r2 - p)); // (that form sends the wave down the compiler's sequence for a negative argument: a lane that misses takes the root of |r2 - p|, which nobody reads)
#else
	t -= sqrt64(r2 - p); // NaN where p > r2 (sqrt64's fast path takes negative arguments: only SMALL POSITIVE ones send the wave to the compiler's sequence)
#endif
	t_out = t;
	return !(p > r2) && !(t <= 0.0);
}
RMD_DEV bool plane_test_flat(V3 origin, V3 normal, V3 ro, V3 rd, double &t_out) {
	double denom = dot(normal, -rd);
	V3 p0l0 = origin - ro;
	double t = dot(p0l0, -normal) / denom;
	t_out = t;
	return denom > 1e-6 && t >= 0.0;
}
RMD_DEV bool plane_pair_test_flat(V3 origin_a, V3 normal_a, V3 origin_b, V3 normal_b, V3 ro, V3 rd, double &t_out, bool &first) {
	const double denom_a = dot(normal_a, -rd);
	const bool faces_a = denom_a > 1e-6, faces_b = -denom_a > 1e-6;
	const double num_a = dot(origin_a - ro, -normal_a), num_b = dot(origin_b - ro, -normal_b);
	const double t = (faces_b ? num_b : num_a) / __builtin_fabs(denom_a);
	t_out = t, first = faces_a;
	return (faces_a || faces_b) && t >= 0.0;
}
// core/src/geometry/primitives/aabb.rs:10-31 (fmin/fmax = Rust f64::min/max NaN rule)
RMD_DEV bool aabb_intersect(V3 bmin, V3 bmax, V3 ro, V3 rd, double &tmin_out) {
	double ix = 1.0 / rd.x, iy = 1.0 / rd.y, iz = 1.0 / rd.z;
	double t1 = (bmin.x - ro.x) * ix, t2 = (bmax.x - ro.x) * ix;
	double tmin = fmin(t1, t2), tmax = fmax(t1, t2);
	t1 = (bmin.y - ro.y) * iy, t2 = (bmax.y - ro.y) * iy;
	tmin = fmax(tmin, fmin(t1, t2)), tmax = fmin(tmax, fmax(t1, t2));
	t1 = (bmin.z - ro.z) * iz, t2 = (bmax.z - ro.z) * iz;
	tmin = fmax(tmin, fmin(t1, t2)), tmax = fmin(tmax, fmax(t1, t2));
	if (!(tmax > fmax(tmin, 0.0))) return false;
	tmin_out = tmin;
	return true;
}
// core/src/geometry/primitives/triangle.rs:11-44 with edge1/edge2 (:16-17) precomputed at upload
RMD_DEV bool triangle_intersect(V3 v0, V3 edge1, V3 edge2, V3 ro, V3 rd, double &t_out) {
	constexpr double EPSILON = 0.00000001;
	V3 h = cross(rd, edge2);
	double a = dot(edge1, h);
	// v0 is only needed behind the determinant test, and the compiler would sink its load there: a second dependent
	// memory round trip per test.  Nearly every test passes the determinant, so pin the load in front of the branch.
	asm volatile("" ::"v"(v0.x), "v"(v0.y), "v"(v0.z));
	if (a < EPSILON && a > -EPSILON) return false;
	double f = 1.0 / a;
	V3 s = ro - v0;
	double u = f * dot(s, h);
	if (u < 0.0 || u > 1.0) return false;
	V3 q = cross(s, edge1);
	double v = f * dot(rd, q);
	if (v < 0.0 || u + v > 1.0) return false;
	double t = f * dot(edge2, q);
	if (t > EPSILON) {
		t_out = t;
		return true;
	}
	return false;
}
// The same test without control flow (RMD_FLAT_TRIANGLE_TEST, the walk's chunk loop): a chunk tests 64 (ray, triangle) pairs of different rays and
// cells, so an early exit is only taken when all 64 fail the same test — practically never — while the four nested branches cost their scalar
// bookkeeping and a copy of `t` per level every time.  The conditions are the negations of the exits above, written so that a NaN takes the same
// way through them (every compare with a NaN is false there and here).
RMD_DEV bool triangle_test_flat(V3 v0, V3 edge1, V3 edge2, V3 ro, V3 rd, double &t_out) {
	constexpr double EPSILON = 0.00000001;
	V3 h = cross(rd, edge2);
	double a = dot(edge1, h);
	double f = 1.0 / a;
	V3 s = ro - v0;
	double u = f * dot(s, h);
	V3 q = cross(s, edge1);
	double v = f * dot(rd, q);
	double t = f * dot(edge2, q);
	t_out = t;
	return !(a < EPSILON && a > -EPSILON) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) && t > EPSILON;
}
// triangle.rs:47-68
RMD_DEV double heron_area_of_sides(double ab, double ac, double bc) {
	double s = (ab + ac + bc) / 2.0;
	return sqrt64(s * (s - ab) * (s - ac) * (s - bc));
}
RMD_DEV double heron_area(V3 a, V3 b, V3 c) { return heron_area_of_sides(dist(a, b), dist(a, c), dist(b, c)); }
// The reference evaluates three Heron areas per shaded mesh hit — (p0,p1,p2), (p0,p1,P), (p0,p2,P) — i.e. six distances and
// three more square roots.  |p0p1|, |p0p2| and the area of the triangle itself do not depend on the hit point: the scene
// upload precomputes them with the same operations (internal.hpp: triangle_aux; IEEE + - * / sqrt on both sides, so the
// values are the ones the kernel would compute), which leaves three distances and two areas here — about half the work of
// a block that runs on nearly every trip of a mesh scene for the few lanes that hit the mesh.
// aux = { |p0p1|, |p0p2|, area(p0,p1,p2), unused }.
template <class P>
RMD_DEV V3 triangle_normal_with(P pos9, P nrm9, double side_ab, double side_ac, double abc, double inv_abc, V3 position) {
	V3 p0 = ld3(pos9), p1 = ld3(pos9 + 3), p2 = ld3(pos9 + 6);
	const double d0 = dist(p0, position), d1 = dist(p1, position), d2 = dist(p2, position);
	double abp = heron_area_of_sides(side_ab, d0, d1); // heron_area(p0, p1, position)
	double bcp = heron_area_of_sides(side_ac, d0, d2); // heron_area(p0, p2, position)
	double ba, bb;
	const double fmax_ = scalar_const(kFMax);
	if (inv_abc == inv_abc && abp < fmax_ && bcp < fmax_) ba = div_by(abp, abc, inv_abc), bb = div_by(bcp, abc, inv_abc);
	else ba = abp / abc, bb = bcp / abc; // degenerate or all-ones area, or a non-finite numerator: the plain divisions
	double bc = 1.0 - (ba + bb);
	V3 n = (ld3(nrm9 + 6) * ba) + (ld3(nrm9 + 3) * bb) + (ld3(nrm9) * bc);
	return normalize(n);
}
template <class P>
RMD_DEV V3 triangle_normal(P pos9, P nrm9, P aux4, V3 position) {
	return triangle_normal_with(pos9, nrm9, aux4[0], aux4[1], aux4[2], aux4[3], position);
}
// the same with the triangle's own sides and area computed here (known-answer probe)
template <class P>
RMD_DEV V3 triangle_normal(P pos9, P nrm9, V3 position) {
	V3 p0 = ld3(pos9), p1 = ld3(pos9 + 3), p2 = ld3(pos9 + 6);
	return triangle_normal_with(pos9, nrm9, dist(p0, p1), dist(p0, p2), heron_area(p0, p1, p2), __builtin_nan(""), position);
}

// f64 -> i32 as num-traits NumCast does it (truncate; fail on NaN / out of range)
RMD_DEV bool cast_i32(double v, int32_t &out) {
	if (!(v > -2147483649.0 && v < 2147483648.0)) return false;
	out = (int32_t)v;
	return true;
}

// ---------------------------------------------------------------- BRDF + samplers (src/trace.rs:362-416)
RMD_DEV double lerp(double mn, double mx, double a) { return mn + a * (mx - mn); } // :392-394
// :362-370 — NdotH.powf(2.0) is x*x after LLVM's unconditional pow(x, 2.0) fold
RMD_DEV double ggx_distribution(V3 n, V3 h, double roughness) {
	double a2 = roughness * roughness;
	double ndh = dot(n, h);
	double den = (ndh * ndh) * (a2 - 1.0) + 1.0;
	den = fmax(kPi * den * den, 1e-7);
	return a2 / den;
}
// :372-378
RMD_DEV double geometry_schlick_ggx(V3 n, V3 v, double r) {
	double num = fmax(dot(n, v), 0.0);
	double k = (r * r) / 8.0;
	return num / (num * (1.0 - k) + k);
}
// :380-382
RMD_DEV double geometry_smith(V3 n, V3 v, V3 l, double r) { return geometry_schlick_ggx(n, v, r) * geometry_schlick_ggx(n, l, r); }
// :384-386
RMD_DEV double pow5(double x);
RMD_DEV V3 fresnel_schlick(double cos_theta, V3 f0) { return f0 + (mk(1.0, 1.0, 1.0) - f0) * pow5(1.0 - cos_theta); }
// :408-416
RMD_DEV void onb(V3 n, V3 &t, V3 &b) {
	double sign = n.z > 0.0 ? 1.0 : -1.0;
	double a = div_lean(-1.0, sign + n.z); // (|sign + n.z| in [1, 2] for a unit vector: inside div_lean's range; a NaN normal gives NaN either way)
	double bb = n.x * n.y * a;
	t = mk(1.0 + sign * n.x * n.x * a, sign * bb, -sign * n.x);
	b = mk(bb, sign + n.y * n.y * a, -n.y);
}
// cgmath Matrix3::from_cols(c0, c1, c2) * v
RMD_DEV V3 mat3_mul(V3 c0, V3 c1, V3 c2, V3 v) {
	return mk((c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z, (c0.z * v.x + c1.z * v.y) + c2.z * v.z);
}
// Fused multiply-adds whose constant operand is held in a scalar register pair.  Written as inline assembly because the
// compiler otherwise materialises the twelve polynomial coefficients of sincos_cw in vector registers, hoists them out of
// the render loop and — at the grid kernel's register limit — spills them, so that every Horner step waited on a scratch
// load (22 dependent scratch round trips per shading pass).  A VALU instruction may read one scalar operand.
RMD_DEV double fma_vvs(double a, double b, double c_uniform) { // a * b + C
	double d;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
	return d;
}
RMD_DEV double fma_vs(double a, double b_uniform, double c) { // a * B + c
	double d;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
	return d;
}
RMD_DEV double fma_ss(double a, double b_uniform, double c_uniform) { // a * B + C: C is copied to a vector register on the spot
	double c;
	asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "s"(c_uniform)); // volatile: not to be hoisted out of the loop (and spilled) again
	return fma_vs(a, b_uniform, c);
}

// sin and cos of one angle (the reference calls libm sin() and cos(), src/trace.rs:401-403 and :291-293).  The angles on
// this path are 2*pi*u and the GGX angle roughness^2 * sqrt(u/(1-u)) with u < 1 - 2^-53, so the general-purpose sincos
// (140 instructions, most of them for arguments this path never produces) is replaced by: quadrant count
// k = rint(x * 2/pi), a two-term Cody-Waite reduction r = x - k*pi/2 in fused multiply-adds (k*pi_hi is exact inside
// the FMA and the neglected third term is k * 1.5e-33) and the fdlibm kernel polynomials on [-pi/4, pi/4].
// Measured against 80-bit sinl/cosl over 6e7 arguments of those shapes up to 2^27: <= 1.6 ulp, absolute error <= 1.8e-16
// — the accuracy class of a device libm (the host libm the oracle calls is <= 0.53 ulp; a direction component differs
// from it by a few 1e-16 either way).  Valid for |x| < 2^45; rmd_scene_create rejects roughness values that could
// exceed that (kMaxRoughness).  Inf and NaN give NaN, as in libm.
RMD_DEV void sincos_cw(double x, double &s, double &c) {
	const double k = __builtin_rint(x * 6.36619772367581382433e-01);
	double r = fma_vs(-k, 1.57079632679489655800e+00, x);
	r = fma_vs(-k, 6.12323399573676603587e-17, r);
	const double z = r * r;
	double ps = fma_ss(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
	ps = fma_vvs(z, ps, 2.75573137070700676789e-06);
	ps = fma_vvs(z, ps, -1.98412698298579493134e-04);
	ps = fma_vvs(z, ps, 8.33333333332248946124e-03);
	ps = fma_vvs(z, ps, -1.66666666666666324348e-01);
	const double sr = __builtin_fma(r * z, ps, r);
	double pc = fma_ss(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
	pc = fma_vvs(z, pc, -2.75573143513906633035e-07);
	pc = fma_vvs(z, pc, 2.48015872894767294178e-05);
	pc = fma_vvs(z, pc, -1.38888888888741095749e-03);
	pc = fma_vvs(z, pc, 4.16666666666666019037e-02);
	const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
	const int n = (int)__builtin_fma(-4.0, __builtin_rint(k * 0.25), k); // k mod 4 in {-2..2}: k itself may not fit an int
	const double ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
	s = (n & 2) ? -ss : ss;
	c = ((n + 1) & 2) ? -cc : cc;
}
// sin and cos of theta = acos(sr), sr = sqrt(r1) in [0, 1] (src/trace.rs:399-403 goes through libm acos, sin, cos).
// cos(acos(sr)) is sr and sin(acos(sr)) is sqrt(1 - sr^2); (1 - sr) is exact for sr >= 1/2 and the product form has no
// cancellation, so the pair is within 2 ulp of the exact values — measured closer to them than the libm chain itself
// (1.9e-16 vs 2.2e-16 relative) and within 3.1e-16 relative of it.
RMD_DEV void hemisphere_sincos(double sr, double &st, double &ct) {
	ct = sr;
	st = sqrt64((1.0 - sr) * (1.0 + sr));
}

// :396-406
RMD_DEV void cosine_hemisphere(double r1, double r2, V3 &dir, double &pdf) {
	double sr = sqrt64(r1);
	double phi = 2.0 * kPi * r2;
	pdf = sr;
	double st, ct, sp, cp;
	hemisphere_sincos(sr, st, ct);
	sincos_cw(phi, sp, cp);
	dir = mk(st * cp, ct, st * sp);
}
// :286-296
RMD_DEV V3 importance_sample_ggx(V3 reflect, double roughness, double r1, double r2) {
	double a = roughness * roughness;
	double phi = 2.0 * kPi * r1;
	double theta = a * sqrt64(r2 / (1.0 - r2));
	double st, ct, sp, cp;
	sincos_cw(theta, st, ct);
	sincos_cw(phi, sp, cp);
	V3 h = mk(st * cp, ct, st * sp);
	V3 tg, bt;
	onb(reflect, tg, bt);
	return normalize(mat3_mul(tg, reflect, bt, h));
}

// x^5 for the Schlick term (src/trace.rs:385 `(1.0 - cos_theta).powf(5.0)`, a libm pow call in the reference).
// Three multiplications: <= 1.5 ulp from the exact power, the same accuracy class as a libm pow, at ~1/60 of its cost.
// It only scales bounce weights — never a direction or a branch — so it cannot change a hit sequence.
RMD_DEV double pow5(double x) {
	const double x2 = x * x;
	const double x4 = x2 * x2;
	return x4 * x;
}
RMD_DEV V3 sel(bool c, V3 a, V3 b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// The "next ray" of a path, for both ways a path gets one:
//   SHADE  the shading half of trace() (src/trace.rs:256-319) for a non-emissive hit at a depth below the bounce limit:
//          the bounce ray, and the bounce's weight multiplied into the throughput T.  Everything the reference computes
//          after its recursive call depends only on values known before it, so the weight is produced here:
//            diffuse  (:281-282): out = ((A (.) radiance) * cos) / d1           A = diffuse_part (.) color, d1 = prob_d * pdf
//            specular (:315-318): out = (((A (.) radiance) * cos) / d1) / d2    A = specular, d1 = 1 - prob_d, d2 = pdf
//   PRIM   generate_primary_ray (:322-333) for the first segment of a new sample (pinhole; the thin lens has its own routine).
// A wave's lanes are at different points of their paths, and both kinds need the same two expensive things — one Philox block
// and one normalisation of the new direction — so the two are ONE instruction stream here, with per-lane selects for the
// inputs: the Philox evaluation (shade: the depth's block; prim: block 0 of the new sample), its two 53-bit conversions and
// normalize() run once for every lane of the wave that needs a ray of either kind.  Within SHADE, diffuse and specular lanes
// likewise share what the two branches have in common — the azimuth's sin/cos, the frame transform, the half vector and the
// Fresnel term; only the polar angle's sin/cos, the reflection vector and the D/G terms stay divergent.  Every lane still
// performs exactly its own branch's operations in the reference's order.
// Must be called in wave-uniform control flow; a lane with neither flag set is left untouched.
// A register (pair) the compiler may fill with anything: the value of a variable in the lanes that never use it.
#ifndef RMD_NO_UNDEF
#define RMD_UNDEF(v) asm volatile("" : "=v"(v));
#else
#define RMD_UNDEF(v) v = 0.0; // (A/B: the stand-in values of rounds 1 - 4)
#endif
struct NextRayShadeIn {
	V3 normal, frag, color;
	double roughness, metal;
};
RMD_DEV void next_ray(const RenderParams &P, bool do_shade, bool do_prim, const NextRayShadeIn &in, V3 cam_pos, uint32_t prim_x, uint32_t prim_y, Rng &rng,
                      V3 &ro, V3 &rd, V3 &T) {
	const bool gen = do_shade || do_prim;
	// (values that only the lanes of one kind use carry no stand-in for the others: a stand-in is a move per register where it is made and a
	// select where the branches meet — RMD_UNDEF)
	double u_first, u_second;
	RMD_UNDEF(u_first) RMD_UNDEF(u_second)
	const double r = (double)rng.lobe_bits * (1.0 / 4194304.0); // shade only (:260): the 22-bit uniform of the path's previous block
	if (gen) {
		uint32_t w0, w1, w2, w3;
		philox4x32_10(rng.pixel, rng.sample, rng.block, 0u, P.key0, P.key1, w0, w1, w2, w3);
		rng.block++;
		u_first = Rng::to_unit(w0, w1), u_second = Rng::to_unit(w2, w3);
		rng.lobe_bits = Rng::spare22(w0, w2); // for the next shaded depth
	}
	// ---- SHADE, first half: the direction before its normalisation
	V3 pre, view, f0;
	RMD_UNDEF(pre.x) RMD_UNDEF(pre.y) RMD_UNDEF(pre.z) RMD_UNDEF(view.x) RMD_UNDEF(view.y) RMD_UNDEF(view.z) RMD_UNDEF(f0.x) RMD_UNDEF(f0.y) RMD_UNDEF(f0.z)
	bool diffuse = false;
	double prob_d, pdf_d;
	RMD_UNDEF(prob_d) RMD_UNDEF(pdf_d)
	if (do_shade) {
		const V3 normal = in.normal;
		view = normalize(cam_pos - in.frag); // :256
		f0 = mk(lerp(0.04, in.color.x, in.metal), lerp(0.04, in.color.y, in.metal), lerp(0.04, in.color.z, in.metal)); // :257-258
		const double r1 = u_first, r2 = u_second; // :397-398 or :287-288
		prob_d = lerp(0.5, 0.0, in.metal);         // :263
		diffuse = r < prob_d;                      // :264
		double phi, st, ct, sp, cp;
		V3 axis;
		// both samplers begin with a square root — sqrt(r1) (:399) or sqrt(r2 / (1 - r2)) (:289): one sequence for the lanes of either kind
#if RMD_OPT_SHARED_SQRT
		double root_arg = r1;
		if (!diffuse) root_arg = div_lean(r2, 1.0 - r2); // (r2 a multiple of 2^-53 in [0, 1): +0 or >= 2^-53 over a divisor in [2^-53, 1] — inside div_lean's range)
		const double root = sqrt64(root_arg);
#else
		const double root = diffuse ? sqrt64(r1) : sqrt64(r2 / (1.0 - r2));
#endif
		if (diffuse) {
			// uniform_sample_hemisphere (:396-406), frame around the normal (:261-262)
			hemisphere_sincos(root, st, ct);
			phi = 2.0 * kPi * r2;
			pdf_d = root;
			axis = normal;
		} else {
			// importance_sample_ggx (:286-296), frame around the mirror direction (:285)
			const double a = in.roughness * in.roughness;
			phi = 2.0 * kPi * r1;
			sincos_cw(a * root, st, ct);
			axis = normalize(-view - 2.0 * (-dot(view, normal) * normal));
		}
		sincos_cw(phi, sp, cp);
		const V3 local = mk(st * cp, ct, st * sp);
		V3 tg, bt;
		onb(axis, tg, bt);
		pre = mat3_mul(tg, axis, bt, local); // :266 / :295 before normalize()
	}
	// ---- PRIM: generate_primary_ray's direction before its normalisation (:326-331)
	if (do_prim) {
		const double x = (double)prim_x + (u_first - 0.5);
		const double y = (double)prim_y + (u_second - 0.5);
		// (x + 0.5) / width and (y + 0.5) / height (:326-327) through the exact reciprocals when the host provided them
		const double sx = P.inv_width == P.inv_width ? div_by(x + 0.5, P.width, P.inv_width) : (x + 0.5) / P.width;
		const double sy = P.inv_height == P.inv_height ? div_by(y + 0.5, P.height, P.inv_height) : (y + 0.5) / P.height;
		pre = mk((2.0 * sx - 1.0) * P.tan_half_fov * P.aspect, (1.0 - 2.0 * sy) * P.tan_half_fov, 1.0);
	}
	V3 sw = pre;
	if (gen) sw = normalize(pre); // :266 / :295 / :332
	if (do_prim) {
		ro = cam_pos, rd = sw;
		T = mk(1.0, 1.0, 1.0);
	}
	// ---- SHADE, second half: the bounce's weight (:275-282 / :301-318) and the bounce ray's origin (:269 / :300)
	if (do_shade) {
		const V3 normal = in.normal;
		const double n_dot_sw = dot_w(normal, sw);
		// Weight of the bounce.  trace() returns  diffuse (:281-282)  ((A (.) radiance) * cos) / (prob_d * pdf)
		//                                         specular (:315-318) (((A (.) radiance) * cos) / (1 - prob_d)) / pdf
		// i.e. radiance times a per-channel weight known before the recursive call; the kernel multiplies the weights forward
		// into the throughput instead of applying them on the way back up.  A weight never feeds a
		// direction or a branch — it only scales the sample — so it is evaluated as ONE quotient, vector x (N / Dn), with the
		// reference's factors but not its seven correctly rounded divisions (f64 division is the most expensive operation on
		// this path: v_rcp_f64 at quarter rate + 10 more instructions): <= a few ulp per bounce against the 1e-9 bar.
		//   diffuse   (1 - F)(1 - metal) (.) color x  max(n.l, 0) / (prob_d * pdf)
		//   specular  F x  D G (n.l) / ((4 (n.v)(n.l) + 0.001) (1 - prob_d) (D (n.h) / (4 (h.v)) + 0.0001))       with
		//             D = a2 / Dd (:362-370), G = g1n g2n / (g1d g2d) (:372-382); Dd cancels:
		//             N = a2 g1n g2n (n.l) 4(h.v),   Dn = g1d g2d (4 (n.v)(n.l) + 0.001) (1 - prob_d) (a2 (n.h) + 0.0001 Dd 4(h.v))
		// :276 halfway of (sample_world, view); :307-308 normalise sample_world once more first — it is a unit vector already, the
		// second normalisation moves it by at most an ulp, and only this weight would see that
		const V3 halfway = normalize_for_weight(sw + view);
		const double h_dot_v = dot_w(halfway, view);
		const double fc = diffuse ? fmax(h_dot_v, 0.0) : h_dot_v; // :277 clamps, :309 does not
		const V3 F = f0 + (mk(1.0, 1.0, 1.0) - f0) * pow5(1.0 - fc); // fresnel_schlick :384-386
		V3 vec;
		double N, Dn;
		if (diffuse) {
			const V3 diffuse_part = (mk(1.0, 1.0, 1.0) - F) * (1.0 - in.metal); // :279-280
			vec = hadamard(diffuse_part, in.color);
			N = fmax(n_dot_sw, 0.0); // :275
			Dn = prob_d * pdf_d;     // :282
		} else {
			const double a2 = in.roughness * in.roughness; // Q4
			const double ndh = dot_w(normal, halfway);
			const double den = (ndh * ndh) * (a2 - 1.0) + 1.0;
			const double Dd = fmax(kPi * den * den, 1e-7); // :367-368
			const double k = (in.roughness * in.roughness) / 8.0; // :374
			const double n_dot_v = dot_w(normal, view);
			const double g1n = fmax(n_dot_v, 0.0), g2n = fmax(n_dot_sw, 0.0); // :373
			const double g1d = g1n * (1.0 - k) + k, g2d = g2n * (1.0 - k) + k;         // :375-377
			const double denominator = 4.0 * n_dot_v * n_dot_sw + 0.001;     // :312
			const double hv4 = 4.0 * h_dot_v;
			vec = F;
			N = (((a2 * g1n) * g2n) * n_dot_sw) * hv4; // cos_theta = n.l unclamped (:306)
			Dn = (((g1d * g2d) * denominator) * (1.0 - prob_d)) * (a2 * ndh + 0.0001 * (Dd * hv4));
		}
		const V3 wgt = vec * (N * fast_rcp(Dn));
		T = hadamard(T, wgt);
		ro = in.frag + normal * (diffuse ? scalar_const(0.00001) : scalar_const(0.0001)); // :269 / :300
		rd = sw;
	}
}
// The shading half alone (the streaming pipeline's step kernel, and the reading order of the reference): same function.
RMD_DEV void shade(const RenderParams &P, V3 normal, V3 frag, V3 color, double roughness, double metal, V3 cam_pos, Rng &rng, V3 &ro, V3 &rd, V3 &T) {
	NextRayShadeIn in{normal, frag, color, roughness, metal};
	next_ray(P, true, false, in, cam_pos, 0u, 0u, rng, ro, rd, T);
}

// ---------------------------------------------------------------- ray generation (src/trace.rs:322-360)
// :322-333 with the two jitter uniforms given (known-answer probe, thin lens); the render loop's pinhole rays come out of next_ray()
RMD_DEV void primary_ray(const RenderParams &P, uint32_t xi, uint32_t yi, double u0, double u1, V3 &ro, V3 &rd) {
	double x = (double)xi + (u0 - 0.5);
	double y = (double)yi + (u1 - 0.5);
	const double sx = P.inv_width == P.inv_width ? div_by(x + 0.5, P.width, P.inv_width) : (x + 0.5) / P.width;
	const double sy = P.inv_height == P.inv_height ? div_by(y + 0.5, P.height, P.inv_height) : (y + 0.5) / P.height;
	double px = (2.0 * sx - 1.0) * P.tan_half_fov * P.aspect;
	double py = (1.0 - 2.0 * sy) * P.tan_half_fov;
	ro = ld3(P.cam_pos);
	rd = normalize(mk(px, py, 1.0));
}
// :335-360.  Returns false where the reference's unwrap() on the focal-plane hit would panic.
// The lens part of it (:337-359) for a pinhole ray (po, pd) that generate_primary_ray has made already (:336): the render loop takes that ray —
// jitter block included — from next_ray()'s merged stream and only the rejection loop, the focal plane and the new direction run here.
RMD_DEV bool thin_lens_from_pinhole(const RenderParams &P, V3 po, V3 pd, Rng &rng, V3 &ro, V3 &rd);
RMD_DEV bool primary_ray_dof(const RenderParams &P, uint32_t xi, uint32_t yi, Rng &rng, V3 &ro, V3 &rd) {
	double u0, u1;
	rng.next2(P.key0, P.key1, u0, u1);
	V3 po, pd;
	primary_ray(P, xi, yi, u0, u1, po, pd);
	return thin_lens_from_pinhole(P, po, pd, rng, ro, rd);
}
RMD_DEV bool thin_lens_from_pinhole(const RenderParams &P, V3 po, V3 pd, Rng &rng, V3 &ro, V3 &rd) {
	V3 pos = ld3(P.cam_pos);
	V3 start = pos;
	// unbounded rejection loop in the reference; 4096 rounds at acceptance pi/4 is never reached,
	// and gives every wave a guaranteed exit.
	for (int guard = 0; guard < 4096; guard++) {
		double r1, r2;
		rng.next2(P.key0, P.key1, r1, r2);
		r1 = r1 * 2.0 - 1.0, r2 = r2 * 2.0 - 1.0;
		start = mk(pos.x + r1 * P.aperture_radius, pos.y + r2 * P.aperture_radius, pos.z);
		if (dist(start, pos) < P.aperture_radius) break;
	}
	V3 fp_origin = pos + mk(0.0, 0.0, 1.0) * P.focal_length;
	double t;
	if (!plane_intersect(fp_origin, mk(0.0, 0.0, -1.0), po, pd, t)) return false;
	V3 end = pos + t * pd;
	ro = start;
	rd = normalize(end - start);
	return true;
}

} // namespace rmd
