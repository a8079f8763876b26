/*
 * oracle.cpp — CPU restatement (C++17, binary64) of Nyrox/raymond's per-pixel radiance loop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity is UNPINNED by reference tests: the
 * reference ships none and cannot be built here; oracle.h lists what pins it instead (the
 * reference's own render at its own 500 spp, its own mesh files).  Every function cites the reference
 * file:line it restates (paths relative to /root/reference).  Recursion, evaluation order
 * and every quirk are kept literal; the only additions are
 *   - the RNG: rand::random::<f64>() (unseedable thread_rng) is replaced by counter-based
 *     Philox4x32-10 keyed by seed with counter (pixel, sample, block), one block per consumer
 *     (include/raymond_hip.h "RNG"), so that a GPU kernel can consume the identical stream;
 *   - failure behaviour: where the reference panics (cast overflow, unwrap on None) the
 *     oracle returns a miss / zero sample and says so at the site.
 * cgmath 0.17 semantics relied on (crate source not vendored in the reference):
 *   dot = (x*x' + y*y') + z*z';  magnitude = sqrt(dot);  normalize(v) = v * (1.0/|v|);
 *   distance(a,b) = |b - a|;  Matrix3::from_cols(c0,c1,c2)*v = (c0*v.x + c1*v.y) + c2*v.z;
 *   Vector3/f64 and f64/Vector3 are element-wise divisions.
 * Built with -O2 -ffp-contract=off: no FMA contraction (rustc never contracts).
 */
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------ vectors (cgmath::Vector3<f64>) */
struct V3 {
	double x, y, z;
	double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(double s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline V3 mul_ew(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 div_ew(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double magnitude(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / magnitude(a)); }
inline double distance(V3 a, V3 b) { return magnitude(b - a); }
/* Rust f64::max / f64::min: the non-NaN operand wins */
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline double rmin(double a, double b) { return std::fmin(a, b); }

const double PI = 3.14159265358979323846; /* core/src/math.rs:19 */
const double F_MAX = std::numeric_limits<double>::max(); /* core/src/math.rs:20 */

/* ------------------------------------------------------------------ mutation switch (tools/mutation_pins.py)
 * What does the reference's own render (examples/ReflectiveSpheres.png) pin?  Each value > 0 "repairs" ONE of the reference's quirks the
 * way a well-meaning port would; the script renders the PNG's scene with each and reports which the statistical pin rejects
 * (DESIGN.md section 2).  0 = the faithful restatement, the only value any test, smoke() or bench.py uses; liboracle_fast.so has no switch. */
enum {
	MUT_NONE = 0,
	MUT_Q1_VIEW_FROM_RAY,        /* :256 view_dir = -ray.direction instead of normalize(camera - P) at every depth */
	MUT_Q2_PDF_WITH_PI,          /* :399-401 pdf = sqrt(r1) / PI (the 1/PI the BRDF omits as well) */
	MUT_Q2_UNIFORM_SAMPLER,      /* :396-406 cos(theta) = r1, pdf = 1/2: the uniform sampler the name promises (an unbiased alternative) */
	MUT_Q3_GGX_ATAN,             /* :289 theta = atan(a * sqrt(r2 / (1 - r2))) */
	MUT_Q4_A2_IS_ALPHA_SQUARED,  /* :363 a2 = roughness^4 */
	MUT_Q4_K_DIRECT_LIGHTING,    /* :374 k = (r + 1)^2 / 8 */
	MUT_Q4_NO_EPSILONS,          /* :312 without + 0.001, :316 without + 0.0001 */
	MUT_Q4_CLAMPED_SPEC_COS,     /* :306 cos_theta = max(n.l, 0) in the specular branch */
	MUT_PROB_D_ALL_DIFFUSE,      /* :263 prob_d = 1 - metalness: Diffuse materials get no specular lobe */
	MUT_F0_ZERO,                 /* :257 F0 = lerp(0.0, colour, metalness) */
	MUT_OFFSETS_1E3,             /* :269 / :300 ray offsets 1e-3 instead of 1e-5 / 1e-4 */
	MUT_Q10_SPHERE_FAR_ROOT,     /* sphere.rs:21-24 origin inside: the far root instead of a miss */
	MUT_Q11_TWO_SIDED_PLANES,    /* plane.rs:14 |denom| > 1e-6 */
	MUT_Q14_ONE_MORE_SEGMENT,    /* :200 trace(.., 0): bounce_limit + 1 segments */
	MUT_Q14_EMISSION_FRONT_ONLY, /* :250-252 emission only when the surface faces the ray */
	MUT_COUNT
};
#ifdef ORC_NO_COUNTERS
constexpr int g_mutation = MUT_NONE;
#else
int g_mutation = MUT_NONE;
#endif
inline bool mut(int k) { return g_mutation == k; }

/* ------------------------------------------------------------------ work counters */
struct Counters {
	uint64_t c[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
enum { K_SAMPLES, K_SEGMENTS, K_CELLS, K_TRI_TESTS, K_MESH_HITS, K_BOUNCES, K_DRAWS, K_WALKS, K_OCCUPIED_CELLS /* visited cells that hold a triangle */,
       K_ZERO_DIFFUSE /* bounces whose weight is exactly zero: diffuse lobe */, K_ZERO_SPECULAR /* ... GGX lobe */,
       K_RETESTS /* triangle tests of a triangle that the PREVIOUS cell of the same walk listed too (it missed there, so it misses here) */, K_COUNT = 12 };
std::mutex g_counter_mutex;
Counters g_counters;
thread_local Counters tl_counters;
/* -DORC_NO_COUNTERS (liboracle_fast.so, the build bench.py times as cpu_baseline): the work counters and walk histograms compile to
 * nothing, so the timed loop is the reference's loop and not the reference's loop plus its instrumentation.  Same results bit for bit. */
/* Work on a path whose bounce weights so far multiply to exactly zero can no longer reach the pixel (trace() multiplies whatever the rest
 * of the path finds by that zero) and the product ends such a path (render_kernel.hpp; rmd_settings.flags RMD_RENDER_TRACE_BLACK_PATHS keeps
 * it).  The oracle always traces it — it is the reference — but by default does not COUNT the work behind the zero, so that the counters are
 * the work the product's default does: orc_count_black_paths(1) counts everything, as the reference executes it. */
thread_local int tl_black = 0;
int g_count_black_paths = 0;
#ifdef ORC_NO_COUNTERS
#define ORC_COUNT(k, n) ((void)0)
#else
#define ORC_COUNT(k, n) ((tl_black == 0 || g_count_black_paths) ? (void)(tl_counters.c[k] += (n)) : (void)0)
#endif
struct BlackScope { /* around trace()'s recursive call from a vertex whose weight is exactly zero */
	bool on;
	explicit BlackScope(bool b) : on(b) { if (on) tl_black++; }
	~BlackScope() { if (on) tl_black--; }
};
/* per-walk histograms (design instrumentation): [kind][bucket], kind 0 = cells visited, 1 = non-empty cells visited, 2 = triangle tests, 3 = max triangles in one cell */
std::atomic<uint64_t> g_walk_hist[4][65];
std::atomic<uint64_t> g_walk_hits{0};
#ifdef ORC_NO_COUNTERS
inline void hist_add(int, uint64_t) {}
#else
inline void hist_add(int kind, uint64_t v) { g_walk_hist[kind][v > 64 ? 64 : v]++; }
#endif
void flush_counters() {
	std::lock_guard<std::mutex> lock(g_counter_mutex);
	for (int i = 0; i < K_COUNT; i++) {
		g_counters.c[i] += tl_counters.c[i];
		tl_counters.c[i] = 0;
	}
}

/* ------------------------------------------------------------------ RNG
 * Stands in for rand::random::<f64>() (src/trace.rs:260,287,288,326,327,340,341,397,398).
 * Philox4x32-10, Salmon/Moraes/Dror/Shaw SC'11; constants from the Random123 distribution. */
void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
	const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
	uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
	uint32_t k0 = key_in[0], k1 = key_in[1];
	for (int round = 0; round < 10; round++) {
		uint64_t p0 = (uint64_t)M0 * c0;
		uint64_t p1 = (uint64_t)M1 * c2;
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		uint32_t n1 = (uint32_t)p1;
		uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		uint32_t n3 = (uint32_t)p0;
		c0 = n0, c1 = n1, c2 = n2, c3 = n3;
		k0 += W0;
		k1 += W1;
	}
	out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* A sample's random numbers come in blocks — one Philox evaluation, counter (pixel, sample, block, 0) — and every consumer
 * takes one block: block 0 the pixel jitter (:326-327), each round of the lens rejection loop one block (:340-341), each shaded
 * depth one block (:260 and :397-398 / :287-288).  include/raymond_hip.h "RNG". */
struct Rng {
	uint32_t key[2];
	uint32_t pixel, sample;
	uint32_t block = 0;
	double lobe = 0.0; /* the 22-bit uniform of the path's last block: `r` (:260) of the NEXT shaded depth */
	Rng(uint64_t seed, uint32_t pixel_, uint32_t sample_) : pixel(pixel_), sample(sample_) {
		key[0] = (uint32_t)seed;
		key[1] = (uint32_t)(seed >> 32);
	}
	static double u53(uint32_t lo, uint32_t hi) {
		return (double)(((((uint64_t)hi) << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); /* 2^-53, [0,1) like rand 0.6 */
	}
	/* the block's two 53-bit uniforms and the 22-bit uniform made of the bits their conversion discards */
	static void at(const uint32_t key[2], uint32_t pixel, uint32_t sample, uint32_t block, double &u0, double &u1, double &u22) {
		uint32_t ctr[4] = {pixel, sample, block, 0u}, w[4];
		philox4x32_10(ctr, key, w);
		u0 = u53(w[0], w[1]), u1 = u53(w[2], w[3]);
		u22 = (double)(((w[0] & 0x7FFu) << 11) | (w[2] & 0x7FFu)) * (1.0 / 4194304.0);
	}
	void next2(double &u0, double &u1) { /* two rand::random::<f64>() calls */
		ORC_COUNT(K_DRAWS, 2);
		at(key, pixel, sample, block++, u0, u1, lobe);
	}
	/* the three calls of one shaded depth, in the reference's order r, r1, r2.  r is only ever compared with prob_d = 0.5
	 * (Diffuse) or 0.0 (Metal) (:263-264): a 22-bit uniform gives those comparisons the probabilities a 53-bit one does.  It is the
	 * 22-bit uniform of the path's PREVIOUS block (the jitter block for the first depth), so that the lobe of a hit is known when the hit
	 * is — before the depth's own block is drawn (include/raymond_hip.h "RNG") */
	void next3(double &r, double &r1, double &r2) {
		ORC_COUNT(K_DRAWS, 3);
		r = lobe;
		at(key, pixel, sample, block++, r1, r2, lobe);
	}
};

/* ------------------------------------------------------------------ geometry types (core/src/geometry/mod.rs:7-47) */
struct Ray {
	V3 origin, direction;
};
struct Hit {
	bool some = false;
	double distance = 0.0;
	size_t subobject_index = 0;
};
inline Hit hit_new(double distance) { return Hit{true, distance, 0}; }
inline Hit hit_none() { return Hit{}; }

struct Sphere {
	V3 origin;
	double radius;
};
struct Plane {
	V3 origin, normal;
};
struct AABB {
	V3 min, max;
};
struct Triangle {
	V3 p0, p1, p2; /* Vertex.position ×3 (vertex.rs:6)  */
	V3 n0, n1, n2; /* Vertex.normal   ×3 (vertex.rs:7)  */
};

/* core/src/geometry/primitives/sphere.rs:11-27 (Q10: near root only, inside => miss) */
Hit sphere_intersects(const Sphere &s, const Ray &ray) {
	V3 c = s.origin - ray.origin;
	double t = dot(c, ray.direction);
	V3 q = c - t * ray.direction;
	double p = dot(q, q);
	if (p > s.radius * s.radius) return hit_none();
	const double half_chord = std::sqrt(s.radius * s.radius - p);
	t -= half_chord;
	if (t <= 0.0) {
		if (mut(MUT_Q10_SPHERE_FAR_ROOT) && t + 2.0 * half_chord > 0.0) return hit_new(t + 2.0 * half_chord);
		return hit_none();
	}
	return hit_new(t);
}
/* sphere.rs:31-35 */
V3 sphere_normal(const Sphere &s, const Ray &ray, double distance) {
	return normalize((ray.origin + ray.direction * distance) - s.origin);
}

/* core/src/geometry/primitives/plane.rs:11-24 (Q11: one-sided) */
Hit plane_intersects(const Plane &pl, const Ray &ray) {
	V3 normal = pl.normal;
	double denom = dot(normal, -ray.direction);
	if (denom > 1e-6 || (mut(MUT_Q11_TWO_SIDED_PLANES) && denom < -1e-6)) {
		V3 p0l0 = pl.origin - ray.origin;
		double t = dot(p0l0, -normal) / denom;
		if (t >= 0.0) return hit_new(t);
	}
	return hit_none();
}

/* core/src/geometry/primitives/aabb.rs:10-31 */
Hit aabb_intersects(const AABB &b, const Ray &ray) {
	V3 inv = 1.0 / ray.direction;
	double t1 = (b.min[0] - ray.origin[0]) * inv[0];
	double t2 = (b.max[0] - ray.origin[0]) * inv[0];
	double tmin = rmin(t1, t2);
	double tmax = rmax(t1, t2);
	for (int i = 1; i < 3; i++) {
		t1 = (b.min[i] - ray.origin[i]) * inv[i];
		t2 = (b.max[i] - ray.origin[i]) * inv[i];
		tmin = rmax(tmin, rmin(t1, t2));
		tmax = rmin(tmax, rmax(t1, t2));
	}
	if (!(tmax > rmax(tmin, 0.0))) return hit_none();
	return hit_new(tmin);
}

/* core/src/geometry/primitives/triangle.rs:11-44 (Möller-Trumbore, two-sided, t > EPS) */
Hit triangle_intersects(const Triangle &tri, const Ray &ray) {
	const double EPSILON = 0.00000001;
	V3 edge1 = tri.p1 - tri.p0;
	V3 edge2 = tri.p2 - tri.p0;
	V3 h = cross(ray.direction, edge2);
	double a = dot(edge1, h);
	if (a < EPSILON && a > -EPSILON) return hit_none();
	double f = 1.0 / a;
	V3 s = ray.origin - tri.p0;
	double u = f * dot(s, h);
	if (u < 0.0 || u > 1.0) return hit_none();
	V3 q = cross(s, edge1);
	double v = f * dot(ray.direction, q);
	if (v < 0.0 || u + v > 1.0) return hit_none();
	double t = f * dot(edge2, q);
	if (t > EPSILON) return hit_new(t);
	return hit_none();
}

/* triangle.rs:47-68 (Q13: Heron-area barycentrics, n2*ba + n1*bb + n0*bc) */
double heron_area(V3 a, V3 b, V3 c) {
	double ab = distance(a, b);
	double ac = distance(a, c);
	double bc = distance(b, c);
	double s = (ab + ac + bc) / 2.0;
	return std::sqrt(s * (s - ab) * (s - ac) * (s - bc));
}
V3 triangle_normal(const Triangle &tri, const Ray &ray, double dist) {
	V3 position = ray.origin + ray.direction * dist;
	double abc = heron_area(tri.p0, tri.p1, tri.p2);
	double abp = heron_area(tri.p0, tri.p1, position);
	double bcp = heron_area(tri.p0, tri.p2, position);
	double ba = abp / abc;
	double bb = bcp / abc;
	double bc = 1.0 - (ba + bb);
	V3 normal = (tri.n2 * ba) + (tri.n1 * bb) + (tri.n0 * bc);
	return normalize(normal);
}

/* triangle.rs:70-84 and mesh.rs:123-140 share the odd seed constants (Q9) */
void bounds_accumulate(const Triangle &t, V3 &mn, V3 &mx) {
	const V3 *p[3] = {&t.p0, &t.p1, &t.p2};
	double *mnv[3] = {&mn.x, &mn.y, &mn.z};
	double *mxv[3] = {&mx.x, &mx.y, &mx.z};
	for (int i = 0; i < 3; i++) {
		for (int k = 0; k < 3; k++) *mnv[i] = rmin(*mnv[i], (*p[k])[i]);
		for (int k = 0; k < 3; k++) *mxv[i] = rmax(*mxv[i], (*p[k])[i]);
	}
}
AABB triangle_bounds(const Triangle &t) {
	V3 mn = v3(125125.0, 1251251.0, 12512512.0), mx = v3(-123125.0, -125123.0, -512123.0);
	bounds_accumulate(t, mn, mx);
	return {mn, mx};
}
AABB mesh_bounds(const std::vector<Triangle> &tris) {
	V3 mn = v3(125125.0, 1251251.0, 12512512.0), mx = v3(-123125.0, -125123.0, -512123.0);
	for (const Triangle &t : tris) bounds_accumulate(t, mn, mx);
	return {mn, mx};
}

/* ------------------------------------------------------------------ AccGrid (core/src/geometry/acc_grid.rs) */
struct AccGrid {
	std::vector<uint64_t> cells;         /* Vec<Cell(usize)>  :28 */
	std::vector<uint64_t> mapping_table; /* Vec<usize>        :30 */
	std::vector<Triangle> triangles;     /* mesh.triangles        */
	AABB bounding_box;
	uint64_t res[3];
	V3 cell_size;
	/* compact copies handed out through orc_grid_describe */
	std::vector<uint32_t> cells32, map32;
	std::vector<double> pos, nrm;
};

/* num-traits NumCast f64 -> integer: truncation toward zero, None on NaN / out of range */
inline bool cast_usize(double v, uint64_t &out) {
	if (!(v > -1.0 && v < 18446744073709551616.0)) return false;
	out = (uint64_t)v;
	return true;
}
inline bool cast_i32(double v, int32_t &out) {
	if (!(v > -2147483649.0 && v < 2147483648.0)) return false;
	out = (int32_t)v;
	return true;
}

/* acc_grid.rs:6-17 */
bool estimate_grid_resolution(const AABB &bounds, size_t triangle_count, uint64_t res[3]) {
	V3 size = bounds.max - bounds.min;
	double volume = std::fabs(size.x * size.y * size.z);
	double density = std::pow((3.0 * (double)triangle_count) / volume, 1.0 / 3.0);
	/* `as usize` (Rust >= 1.45 saturates; NaN -> 0) */
	auto as_usize = [](double v) -> uint64_t {
		if (!(v == v) || v <= 0.0) return 0;
		if (v >= 18446744073709551616.0) return UINT64_MAX;
		return (uint64_t)v;
	};
	res[0] = as_usize(std::fabs(size.x) * density);
	res[1] = as_usize(std::fabs(size.y) * density);
	res[2] = as_usize(std::fabs(size.z) * density);
	return true;
}

/* acc_grid.rs:36-83.  Returns 0, or 5 where the reference would panic (index past the cell
 * array at :61, a failed usize cast at :44-51, or `grid_res[i] - 1` underflow at :54). */
int build_from_mesh(std::vector<Triangle> tris, AccGrid &g) {
	g.bounding_box = mesh_bounds(tris);
	estimate_grid_resolution(g.bounding_box, tris.size(), g.res);
	if (g.res[0] == 0 || g.res[1] == 0 || g.res[2] == 0) return 5;
	g.cell_size = div_ew(g.bounding_box.max - g.bounding_box.min, v3((double)g.res[0], (double)g.res[1], (double)g.res[2]));
	uint64_t n_cells = g.res[0] * g.res[1] * g.res[2];
	if (n_cells > (1ull << 31)) return 5;
	std::vector<std::vector<uint64_t>> naive(n_cells);
	for (size_t index = 0; index < tris.size(); index++) {
		AABB b = triangle_bounds(tris[index]);
		V3 lo = div_ew(b.min - g.bounding_box.min, g.cell_size);
		V3 hi = div_ew(b.max - g.bounding_box.min, g.cell_size);
		uint64_t cmin[3], cmax[3];
		if (!cast_usize(lo.x, cmin[0]) || !cast_usize(lo.y, cmin[1]) || !cast_usize(lo.z, cmin[2])) return 5;
		if (!cast_usize(hi.x, cmax[0]) || !cast_usize(hi.y, cmax[1]) || !cast_usize(hi.z, cmax[2])) return 5;
		for (int i = 0; i < 3; i++) {
			cmin[i] = std::min(std::max<uint64_t>(cmin[i], 0), g.res[i] - 1);
			cmax[i] = std::min(std::max<uint64_t>(cmax[i], 0), g.res[i] - 1);
		}
		for (uint64_t z = cmin[2]; z <= cmax[2]; z++)
			for (uint64_t y = cmin[1]; y <= cmax[1]; y++)
				for (uint64_t x = cmin[0]; x <= cmax[0]; x++) {
					uint64_t idx = x + g.res[0] * (y + z * g.res[2]); /* Q5: res.z, not res.y */
					if (idx >= n_cells) return 5;
					naive[idx].push_back(index);
				}
	}
	g.cells.clear();
	g.mapping_table.clear();
	for (const auto &c : naive) {
		g.cells.push_back(g.mapping_table.size());
		g.mapping_table.push_back(c.size());
		for (uint64_t i : c) g.mapping_table.push_back(i);
	}
	g.triangles = std::move(tris);
	return 0;
}

/* acc_grid.rs:89-185 (Q5-Q8) */
Hit grid_intersects(const AccGrid &g, const Ray &ray) {
	Hit outer = aabb_intersects(g.bounding_box, ray);
	if (!outer.some) return hit_none();
	ORC_COUNT(K_WALKS, 1);
	V3 outer_pos = ray.origin + ray.direction * outer.distance;
	V3 start = ray.origin - g.bounding_box.min;
	int32_t cx, cy, cz;
	{
		V3 c = div_ew(start, g.cell_size);
		/* .cast::<i32>().unwrap() panics when out of range: the oracle reports a miss instead */
		if (!cast_i32(c.x, cx) || !cast_i32(c.y, cy) || !cast_i32(c.z, cz)) return hit_none();
	}
	if (cx < 0 || cy < 0 || cz < 0) {
		start = outer_pos - g.bounding_box.min;
		V3 c = div_ew(start, g.cell_size);
		if (!cast_i32(c.x, cx) || !cast_i32(c.y, cy) || !cast_i32(c.z, cz)) return hit_none();
	}
	/* f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN (cast panics) */
	auto signum = [](double v, int32_t &out) -> bool {
		if (!(v == v)) return false;
		out = std::signbit(v) ? -1 : 1;
		return true;
	};
	int32_t sx, sy, sz;
	if (!signum(ray.direction.x, sx) || !signum(ray.direction.y, sy) || !signum(ray.direction.z, sz)) return hit_none();

	double t_delta_x = (ray.direction.x < 0.0 ? -g.cell_size.x : g.cell_size.x) / ray.direction.x;
	double t_delta_y = (ray.direction.y < 0.0 ? -g.cell_size.y : g.cell_size.y) / ray.direction.y;
	double t_delta_z = (ray.direction.z < 0.0 ? -g.cell_size.z : g.cell_size.z) / ray.direction.z;

	double t_max_x = (((double)(cx + (ray.direction.x < 0.0 ? 0 : 1)) * g.cell_size.x) - start.x) / ray.direction.x;
	double t_max_y = (((double)(cy + (ray.direction.y < 0.0 ? 0 : 1)) * g.cell_size.y) - start.y) / ray.direction.y;
	double t_max_z = (((double)(cz + (ray.direction.z < 0.0 ? 0 : 1)) * g.cell_size.z) - start.z) / ray.direction.z;

	const int32_t rx = (int32_t)g.res[0], ry = (int32_t)g.res[1], rz = (int32_t)g.res[2];
	uint64_t w_cells = 0, w_nonempty = 0, w_tests = 0, w_maxc = 0;
	struct WalkStat {
		uint64_t &c, &n, &t, &m;
		bool hit = false;
#ifndef ORC_NO_COUNTERS
		~WalkStat() { hist_add(0, c), hist_add(1, n), hist_add(2, t), hist_add(3, m); if (hit) g_walk_hits++; }
#endif
	} wstat{w_cells, w_nonempty, w_tests, w_maxc};
	(void)wstat;
	[[maybe_unused]] uint64_t prev_cell = ~0ull; /* counters only: the mapping-table run of the cell visited before this one */
	for (;;) {
		/* `as usize` of a negative i32 sign-extends; the index arithmetic wraps (release build) */
		uint64_t x = (uint64_t)(int64_t)cx, y = (uint64_t)(int64_t)cy, z = (uint64_t)(int64_t)cz;
		uint64_t idx = x + g.res[0] * (y + z * g.res[2]);
		if (idx >= g.cells.size()) return hit_none();
		ORC_COUNT(K_CELLS, 1);
		uint64_t cell = g.cells[idx];
		uint64_t count = g.mapping_table[cell];
#ifndef ORC_NO_COUNTERS
		if (prev_cell != ~0ull)
			for (uint64_t i = 1; i <= count; i++)
				for (uint64_t j = 1; j <= g.mapping_table[prev_cell]; j++)
					if (g.mapping_table[prev_cell + j] == g.mapping_table[cell + i]) {
						ORC_COUNT(K_RETESTS, 1);
						break;
					}
		prev_cell = cell;
#endif
		w_cells++, w_tests += count, w_nonempty += count > 0, w_maxc = std::max(w_maxc, count);
		if (count > 0) ORC_COUNT(K_OCCUPIED_CELLS, 1);
		double closest = 5712515.0;
		Hit closest_hit = hit_none();
		for (uint64_t i = 1; i <= count; i++) {
			uint64_t ti = g.mapping_table[cell + i];
			ORC_COUNT(K_TRI_TESTS, 1);
			Hit h = triangle_intersects(g.triangles[ti], ray);
			if (h.some) {
				if (h.distance < closest) {
					closest = h.distance;
					closest_hit = Hit{true, h.distance, (size_t)ti};
				}
			}
		}
		if (closest_hit.some) {
			wstat.hit = true;
			return closest_hit; /* Q7: first cell with any hit wins */
		}

		if (t_max_x < t_max_y) {
			if (t_max_x < t_max_z) {
				cx += sx;
				if (cx >= rx || cx < 0) return hit_none();
				t_max_x += t_delta_x;
			} else {
				cz += sz;
				if (cz >= rz || cz < 0) return hit_none();
				t_max_z += t_delta_z;
			}
		} else {
			if (t_max_y < t_max_z) {
				cy += sy;
				if (cy >= ry || cy < 0) return hit_none();
				t_max_y += t_delta_y;
			} else {
				cz += sz;
				if (cz >= rz || cz < 0) return hit_none();
				t_max_z += t_delta_z;
			}
		}
	}
}

/* ------------------------------------------------------------------ Scene (core/src/scene.rs) */
struct Material {
	uint32_t kind; /* core/src/lib.rs:21-26 */
	V3 color;
	double roughness;
};
struct Object {
	uint32_t geometry_kind; /* scene.rs:9-13 */
	Sphere sphere;
	Plane plane;
	std::shared_ptr<AccGrid> grid; /* Arc<AccGrid> */
	Material material;
};
struct Scene {
	std::vector<Object> objects;
	std::vector<std::shared_ptr<AccGrid>> grids;
};

/* scene.rs:16-22 */
Hit geometry_intersects(const Object &o, const Ray &ray) {
	switch (o.geometry_kind) {
	case RMD_GEOM_PLANE: return plane_intersects(o.plane, ray);
	case RMD_GEOM_SPHERE: return sphere_intersects(o.sphere, ray);
	default: return grid_intersects(*o.grid, ray);
	}
}
/* scene.rs:24-30; plane.rs:28-32; sphere.rs:31-35; acc_grid.rs:85-87 */
V3 geometry_normal(const Object &o, const Ray &ray, const Hit &hit) {
	switch (o.geometry_kind) {
	case RMD_GEOM_PLANE: return o.plane.normal;
	case RMD_GEOM_SPHERE: return sphere_normal(o.sphere, ray, hit.distance);
	default: ORC_COUNT(K_MESH_HITS, 1); return triangle_normal(o.grid->triangles[hit.subobject_index], ray, hit.distance);
	}
}
/* scene.rs:54-74: linear closest hit, strict '<' keeps the first object on ties */
int scene_intersect(const Scene &scene, const Ray &ray, Hit &out) {
	ORC_COUNT(K_SEGMENTS, 1);
	double closest_distance = F_MAX;
	int closest_object = -1;
	for (size_t i = 0; i < scene.objects.size(); i++) {
		Hit h = geometry_intersects(scene.objects[i], ray);
		if (h.some) {
			if (h.distance < closest_distance) {
				closest_distance = h.distance;
				closest_object = (int)i;
				out = h;
			}
		}
	}
	return closest_object;
}

/* ------------------------------------------------------------------ BRDF + samplers (src/trace.rs:362-416) */
double lerp(double mn, double mx, double a) { return mn + a * (mx - mn); } /* :392-394 */
V3 lerp_vec(V3 mn, V3 mx, double a) { return v3(lerp(mn.x, mx.x, a), lerp(mn.y, mx.y, a), lerp(mn.z, mx.z, a)); } /* :388-390 */

/* :362-370 (Q4: a2 = roughness^2).  NdotH.powf(2.0): LLVM folds pow(x, 2.0) to x*x
 * unconditionally, and so does gcc for std::pow(x, 2.0) — written as a product here. */
double ggx_distribution(V3 n, V3 h, double roughness) {
	double a2 = roughness * roughness;
	if (mut(MUT_Q4_A2_IS_ALPHA_SQUARED)) a2 = a2 * a2;
	double NdotH = dot(n, h);
	double nominator = a2;
	double denominator = (NdotH * NdotH) * (a2 - 1.0) + 1.0;
	denominator = rmax(PI * denominator * denominator, 1e-7);
	return nominator / denominator;
}
/* :372-378 */
double geometry_schlick_ggx(V3 n, V3 v, double r) {
	double numerator = rmax(dot(n, v), 0.0);
	double k = (r * r) / 8.0;
	if (mut(MUT_Q4_K_DIRECT_LIGHTING)) k = ((r + 1.0) * (r + 1.0)) / 8.0;
	double denominator = numerator * (1.0 - k) + k;
	return numerator / denominator;
}
/* :380-382 */
double geometry_smith(V3 n, V3 v, V3 l, double r) { return geometry_schlick_ggx(n, v, r) * geometry_schlick_ggx(n, l, r); }
/* :384-386 — powf(5.0) is a libm pow call in a rustc release build */
V3 fresnel_schlick(double cos_theta, V3 F0) { return F0 + (v3(1.0, 1.0, 1.0) - F0) * std::pow(1.0 - cos_theta, 5.0); }

/* :408-416 (Duff/Frisvad ONB) */
void create_coordinate_system_of_n(V3 n, V3 &t, V3 &b) {
	double sign = n.z > 0.0 ? 1.0 : -1.0;
	double a = -1.0 / (sign + n.z);
	double bb = n.x * n.y * a;
	t = v3(1.0 + sign * n.x * n.x * a, sign * bb, -sign * n.x);
	b = v3(bb, sign + n.y * n.y * a, -n.y);
}
/* cgmath Matrix3::from_cols(c0,c1,c2) * v */
V3 mat3_mul(V3 c0, V3 c1, V3 c2, V3 v) {
	return v3((c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z,
	          (c0.z * v.x + c1.z * v.y) + c2.z * v.z);
}
/* :396-406 (Q2: cosine-weighted despite the name; pdf = sqrt(r1)) */
void uniform_sample_hemisphere(double r1, double r2, V3 &cartesian, double &pdf) {
	double theta = std::acos(std::sqrt(r1));
	double phi = 2.0 * PI * r2;
	pdf = std::sqrt(r1);
	if (mut(MUT_Q2_PDF_WITH_PI)) pdf = pdf / PI;
	if (mut(MUT_Q2_UNIFORM_SAMPLER)) theta = std::acos(r1), pdf = 0.5;
	cartesian = v3(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
}
/* :286-296 (Q3: theta = a*sqrt(r2/(1-r2)) used directly as an angle) */
V3 importance_sample_ggx(V3 reflect, double roughness, double r1, double r2) {
	double a = roughness * roughness;
	double phi = 2.0 * PI * r1;
	double theta = a * std::sqrt(r2 / (1.0 - r2));
	if (mut(MUT_Q3_GGX_ATAN)) theta = std::atan(theta);
	V3 h = v3(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
	V3 tangent, bitangent;
	create_coordinate_system_of_n(reflect, tangent, bitangent);
	return normalize(mat3_mul(tangent, reflect, bitangent, h));
}

/* ------------------------------------------------------------------ ray generation (src/trace.rs:322-360) */
Ray generate_primary_ray_u(uint32_t xi, uint32_t yi, const rmd_camera &cam, double u0, double u1) {
	double width = (double)cam.backbuffer_width;
	double height = (double)cam.backbuffer_height;
	double aspect = width / height;
	double x = (double)xi + (u0 - 0.5);
	double y = (double)yi + (u1 - 0.5);
	double px = (2.0 * ((x + 0.5) / width) - 1.0) * std::tan(cam.fov_vert / 2.0 * PI / 180.0) * aspect;
	double py = (1.0 - 2.0 * ((y + 0.5) / height)) * std::tan(cam.fov_vert / 2.0 * PI / 180.0);
	V3 pos = v3(cam.position[0], cam.position[1], cam.position[2]);
	return Ray{pos, normalize(v3(px, py, 1.0))};
}
Ray generate_primary_ray(uint32_t x, uint32_t y, const rmd_camera &cam, Rng &rng) {
	double u0, u1; /* :326, :327 */
	rng.next2(u0, u1);
	return generate_primary_ray_u(x, y, cam, u0, u1);
}
/* :335-360 (Q12).  `ok` false where the reference's unwrap() would panic. */
Ray generate_primary_ray_with_dof(uint32_t x, uint32_t y, const rmd_camera &cam, Rng &rng, bool &ok) {
	Ray primary = generate_primary_ray(x, y, cam, rng);
	V3 pos = v3(cam.position[0], cam.position[1], cam.position[2]);
	V3 start = pos;
	/* unbounded in the reference; 4096 rounds (acceptance pi/4 each) is never reached */
	for (int guard = 0; guard < 4096; guard++) {
		double r1, r2; /* :340-341 */
		rng.next2(r1, r2);
		r1 = r1 * 2.0 - 1.0, r2 = r2 * 2.0 - 1.0;
		double ax = pos.x + r1 * cam.aperture_radius;
		double ay = pos.y + r2 * cam.aperture_radius;
		start = v3(ax, ay, pos.z);
		if (distance(start, pos) < cam.aperture_radius) break;
	}
	Plane focal_plane{pos + v3(0.0, 0.0, 1.0) * cam.focal_length, v3(0.0, 0.0, -1.0)};
	Hit fh = plane_intersects(focal_plane, primary);
	ok = fh.some;
	if (!fh.some) return primary;
	V3 end = pos + fh.distance * primary.direction;
	return Ray{start, normalize(end - start)};
}

/* ------------------------------------------------------------------ trace (src/trace.rs:232-320) */
struct TraceContext {
	const Scene *scene;
	const rmd_camera *cam;
	uint32_t bounce_limit;
	int32_t *path_obj = nullptr;
	uint32_t *path_sub = nullptr;
	int32_t path_len = 0;
};

V3 trace(const Ray &ray, TraceContext &ctx, Rng &rng, uint32_t depth) {
	if (depth > ctx.bounce_limit) return v3(0.0, 0.0, 0.0); /* :235-237 */

	Hit hit;
	int oi = scene_intersect(*ctx.scene, ray, hit); /* :239 */
	if (ctx.path_obj) {
		ctx.path_obj[ctx.path_len] = oi;
		ctx.path_sub[ctx.path_len] = oi >= 0 ? (uint32_t)hit.subobject_index : 0u;
		ctx.path_len++;
	}
	if (oi < 0) return v3(0.0, 0.0, 0.0); /* :242 */
	const Object &object = ctx.scene->objects[oi];
	V3 normal = geometry_normal(object, ray, hit);                    /* :244-245 */
	V3 fragment_position = ray.origin + ray.direction * hit.distance; /* :246 */
	if (object.material.kind == RMD_MAT_EMISSION) { /* :250-252 */
		if (mut(MUT_Q14_EMISSION_FRONT_ONLY) && !(dot(normal, -ray.direction) > 0.0)) return v3(0.0, 0.0, 0.0);
		return object.material.color;
	}
	V3 material_color = object.material.color;
	double material_roughness = object.material.roughness;
	double material_metalness = object.material.kind == RMD_MAT_METAL ? 1.0 : 0.0; /* :248-249 */
	ORC_COUNT(K_BOUNCES, 1);

	V3 cam_pos = v3(ctx.cam->position[0], ctx.cam->position[1], ctx.cam->position[2]);
	V3 view_dir = normalize(cam_pos - fragment_position); /* :256 (Q1) */
	if (mut(MUT_Q1_VIEW_FROM_RAY)) view_dir = -ray.direction;
	V3 f0 = lerp_vec(v3(0.04, 0.04, 0.04), material_color, material_metalness); /* :257-258 */
	if (mut(MUT_F0_ZERO)) f0 = lerp_vec(v3(0.0, 0.0, 0.0), material_color, material_metalness);
	double r, r1, r2; /* :260, then :397-398 (diffuse) or :287-288 (specular): the depth's three draws come from one block */
	rng.next3(r, r1, r2);
	V3 lc_t, lc_b;
	create_coordinate_system_of_n(normal, lc_t, lc_b); /* :261-262: Matrix3::from_cols(t, normal, b) */
	double prob_d = lerp(0.5, 0.0, material_metalness); /* :263 */
	if (mut(MUT_PROB_D_ALL_DIFFUSE)) prob_d = 1.0 - material_metalness;
	const double off_d = mut(MUT_OFFSETS_1E3) ? 0.001 : 0.00001, off_s = mut(MUT_OFFSETS_1E3) ? 0.001 : 0.0001;
	if (r < prob_d) {
		/* :265-282 diffuse */
		V3 sample;
		double pdf;
		uniform_sample_hemisphere(r1, r2, sample, pdf);
		V3 sample_world = normalize(mat3_mul(lc_t, normal, lc_b, sample));
		bool black = false;
#ifndef ORC_NO_COUNTERS
		{ /* is this bounce's weight (:279-282) exactly zero?  (for the counters only: the values below are computed again, in the reference's place) */
			V3 hw = normalize(sample_world + view_dir);
			V3 fr = fresnel_schlick(rmax(dot(hw, view_dir), 0.0), f0);
			V3 a = mul_ew((v3(1.0, 1.0, 1.0) - fr) * (1.0 - material_metalness), material_color);
			black = (a.x == 0.0 && a.y == 0.0 && a.z == 0.0) || rmax(dot(normal, sample_world), 0.0) == 0.0;
			if (black) ORC_COUNT(K_ZERO_DIFFUSE, 1);
		}
#endif
		BlackScope scope(black);
		V3 radiance = trace(Ray{fragment_position + normal * off_d, sample_world}, ctx, rng, depth + 1);
		double cos_theta = rmax(dot(normal, sample_world), 0.0);
		V3 halfway = normalize(sample_world + view_dir);
		V3 fresnel = fresnel_schlick(rmax(dot(halfway, view_dir), 0.0), f0);
		V3 specular_part = fresnel;
		V3 diffuse_part = v3(1.0, 1.0, 1.0) - specular_part;
		diffuse_part = diffuse_part * (1.0 - material_metalness);
		V3 output = mul_ew(mul_ew(diffuse_part, material_color), radiance) * cos_theta;
		return output / (prob_d * pdf);
	} else {
		/* :283-319 specular */
		V3 reflect = normalize(-view_dir - 2.0 * (-dot(view_dir, normal) * normal));
		V3 sample_world = importance_sample_ggx(reflect, material_roughness, r1, r2);
		bool black = false;
#ifndef ORC_NO_COUNTERS
		{ /* is this bounce's weight (:301-318) exactly zero?  D G F cos / pdf with G = Schlick-GGX(max(n.v, 0)) x Schlick-GGX(max(n.l, 0)) */
			V3 hw = normalize(normalize(sample_world) + view_dir);
			V3 fr = fresnel_schlick(dot(hw, view_dir), f0);
			double n = ((((material_roughness * material_roughness) * rmax(dot(normal, view_dir), 0.0)) * rmax(dot(normal, sample_world), 0.0)) * dot(normal, sample_world)) *
			           (4.0 * dot(hw, view_dir));
			black = n == 0.0 || (fr.x == 0.0 && fr.y == 0.0 && fr.z == 0.0);
			if (black) ORC_COUNT(K_ZERO_SPECULAR, 1);
		}
#endif
		BlackScope scope(black);
		V3 radiance = trace(Ray{fragment_position + normal * off_s, sample_world}, ctx, rng, depth + 1);
		double cos_theta = dot(normal, sample_world);
		if (mut(MUT_Q4_CLAMPED_SPEC_COS)) cos_theta = rmax(cos_theta, 0.0);
		V3 light_dir = normalize(sample_world);
		V3 halfway = normalize(light_dir + view_dir);
		V3 F = fresnel_schlick(dot(halfway, view_dir), f0);
		double D = ggx_distribution(normal, halfway, material_roughness);
		double G = geometry_smith(normal, view_dir, sample_world, material_roughness);
		V3 nominator = D * G * F;
		double denominator = 4.0 * dot(normal, view_dir) * cos_theta + (mut(MUT_Q4_NO_EPSILONS) ? 0.0 : 0.001);
		V3 specular = nominator / denominator;
		V3 output = mul_ew(specular, radiance) * cos_theta;
		double pdf = (D * dot(normal, halfway)) / (4.0 * dot(halfway, view_dir)) + (mut(MUT_Q4_NO_EPSILONS) ? 0.0 : 0.0001);
		return output / (1.0 - prob_d) / pdf;
	}
}

/* src/trace.rs:199-200 for one (pixel, sample) */
V3 sample_pixel(const Scene &scene, const rmd_camera &cam, const rmd_settings &st, uint32_t x, uint32_t y, uint32_t s,
                int32_t *path_obj = nullptr, uint32_t *path_sub = nullptr, int32_t *path_len = nullptr) {
	ORC_COUNT(K_SAMPLES, 1);
	Rng rng(st.seed, y * cam.backbuffer_width + x, s);
	Ray primary;
	if ((st.flags & RMD_RENDER_DOF) && cam.aperture_radius > 0.0) { /* render_tiled itself only ever calls the pinhole generator (:199) */
		bool ok = true;
		primary = generate_primary_ray_with_dof(x, y, cam, rng, ok);
		if (!ok) return v3(0.0, 0.0, 0.0);
	} else {
		primary = generate_primary_ray(x, y, cam, rng);
	}
	TraceContext ctx{&scene, &cam, st.bounce_limit, path_obj, path_sub, 0};
	V3 out = trace(primary, ctx, rng, mut(MUT_Q14_ONE_MORE_SEGMENT) ? 0 : 1);
	if (path_len) *path_len = ctx.path_len;
	return out;
}

Ray ray_from(const double *r) { return Ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])}; }
V3 v3_from(const double *p) { return v3(p[0], p[1], p[2]); }
void v3_store(double *p, V3 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }
Triangle tri_from(const double *pos9, const double *nrm9) {
	Triangle t;
	t.p0 = v3_from(pos9), t.p1 = v3_from(pos9 + 3), t.p2 = v3_from(pos9 + 6);
	if (nrm9) t.n0 = v3_from(nrm9), t.n1 = v3_from(nrm9 + 3), t.n2 = v3_from(nrm9 + 6);
	else t.n0 = t.n1 = t.n2 = v3(0, 0, 0);
	return t;
}

/* ------------------------------------------------------------------ Mesh (core/src/geometry/mesh.rs) */
/* str::split_whitespace */
std::vector<std::string> split_ws(const std::string &line) {
	std::vector<std::string> out;
	size_t i = 0;
	while (i < line.size()) {
		while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r' || line[i] == '\n' || line[i] == '\f' || line[i] == '\v')) i++;
		size_t j = i;
		while (j < line.size() && !(line[j] == ' ' || line[j] == '\t' || line[j] == '\r' || line[j] == '\n' || line[j] == '\f' || line[j] == '\v')) j++;
		if (j > i) out.push_back(line.substr(i, j - i));
		i = j;
	}
	return out;
}
/* str::parse::<f64>(): decimal notation with optional exponent, "inf" / "infinity" / "nan" in any case, the whole token,
 * correctly rounded — which glibc's strtod is too; hexadecimal floats are not Rust syntax. */
bool parse_f64(const std::string &t, double &out) {
	if (t.empty() || t.find_first_of("xX") != std::string::npos) return false;
	for (char c : t)
		if (c == '(' || c == ')') return false; /* strtod's "nan(...)" form */
	char *end = nullptr;
	out = std::strtod(t.c_str(), &end);
	return end == t.c_str() + t.size();
}
/* str::parse::<u32>() / ::<usize>(): optional '+', decimal digits only, no overflow */
bool parse_uint(const std::string &t, uint64_t limit, uint64_t &out) {
	size_t i = (!t.empty() && t[0] == '+') ? 1 : 0;
	if (i >= t.size()) return false;
	uint64_t v = 0;
	for (; i < t.size(); i++) {
		if (t[i] < '0' || t[i] > '9') return false;
		if (v > (limit - (uint64_t)(t[i] - '0')) / 10) return false;
		v = v * 10 + (uint64_t)(t[i] - '0');
	}
	out = v;
	return true;
}
struct PlyVertex {
	V3 position, normal; /* Vertex.uv / .tangent (vertex.rs:8-9, mesh.rs:87-88,:101-108) are never read on the radiance path */
};
/* Mesh::load_ply, mesh.rs:58-121.  false where the reference panics (an unwrap() on a missing token / failed parse, an
 * index past `values` or `vertices`). */
bool load_ply_text(const std::string &buffer, std::vector<Triangle> &faces) {
	/* str::lines(): split at '\n', a trailing "\r" is dropped, no empty last line after a final newline */
	std::vector<std::string> lines;
	for (size_t i = 0; i < buffer.size();) {
		size_t j = buffer.find('\n', i);
		std::string l = buffer.substr(i, (j == std::string::npos ? buffer.size() : j) - i);
		if (!l.empty() && l.back() == '\r') l.pop_back();
		lines.push_back(l);
		if (j == std::string::npos) break;
		i = j + 1;
	}
	size_t li = 0;
	uint64_t capacity = 0; /* vertices.capacity() after the reserve_exact calls of :72 (len is 0 there) */
	/* :67-78 header: only `element vertex N` is read */
	while (li < lines.size()) {
		std::vector<std::string> tok = split_ws(lines[li++]);
		if (tok.empty()) return false; /* tokens.next().unwrap() */
		if (tok[0] == "element") {
			if (tok.size() < 2) return false;
			if (tok[1] == "vertex") {
				uint64_t n;
				if (tok.size() < 3 || !parse_uint(tok[2], UINT64_MAX, n)) return false;
				capacity = std::max(capacity, n);
			}
		} else if (tok[0] == "end_header") {
			break;
		}
	}
	/* :81-91 vertices: x y z nx ny nz [s t] */
	std::vector<PlyVertex> vertices;
	for (uint64_t k = 0; k < capacity; k++) {
		if (li >= lines.size()) return false; /* lines.next().unwrap() */
		std::vector<std::string> tok = split_ws(lines[li++]);
		std::vector<double> values;
		for (const std::string &t : tok) {
			double v;
			if (!parse_f64(t, v)) return false;
			values.push_back(v);
		}
		if (values.size() < 6) return false; /* values[5] */
		vertices.push_back(PlyVertex{v3(values[0], values[1], values[2]), v3(values[3], values[4], values[5])});
	}
	/* :94-118 faces: `3 i j k`; lines with another first value are skipped */
	faces.clear();
	while (li < lines.size()) {
		std::vector<std::string> tok = split_ws(lines[li++]);
		std::vector<uint64_t> values;
		for (const std::string &t : tok) {
			uint64_t v;
			if (!parse_uint(t, 0xFFFFFFFFull, v)) return false;
			values.push_back(v);
		}
		if (values.empty()) return false; /* values[0] */
		if (values[0] == 3) {
			if (values.size() < 4) return false;
			for (int k = 1; k <= 3; k++)
				if (values[k] >= vertices.size()) return false;
			const PlyVertex &a = vertices[values[1]], &b = vertices[values[2]], &c = vertices[values[3]];
			faces.push_back(Triangle{a.position, b.position, c.position, a.normal, b.normal, c.normal});
		}
	}
	return true;
}
/* Mesh::bake_transform, mesh.rs:48-56 */
void bake_transform(std::vector<Triangle> &tris, V3 translate) {
	for (Triangle &t : tris) t.p0 = t.p0 + translate, t.p1 = t.p1 + translate, t.p2 = t.p2 + translate;
}

} // namespace

struct orc_grid {
	AccGrid g;
};
struct orc_mesh {
	std::vector<Triangle> triangles;
	std::vector<double> pos, nrm;
	void refresh() {
		pos.resize(triangles.size() * 9), nrm.resize(triangles.size() * 9);
		for (size_t i = 0; i < triangles.size(); i++) {
			const Triangle &t = triangles[i];
			v3_store(&pos[9 * i], t.p0), v3_store(&pos[9 * i + 3], t.p1), v3_store(&pos[9 * i + 6], t.p2);
			v3_store(&nrm[9 * i], t.n0), v3_store(&nrm[9 * i + 3], t.n1), v3_store(&nrm[9 * i + 6], t.n2);
		}
	}
};
struct orc_scene {
	Scene scene;
};

extern "C" {

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
void orc_block_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t block, double out[3]) {
	uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
	Rng::at(key, pixel, sample, block, out[0], out[1], out[2]);
}

void orc_sphere_intersect(size_t n, const double *sphere4, const double *ray6, int32_t *hit, double *t) {
	for (size_t i = 0; i < n; i++) {
		Sphere s{v3_from(sphere4 + 4 * i), sphere4[4 * i + 3]};
		Hit h = sphere_intersects(s, ray_from(ray6 + 6 * i));
		hit[i] = h.some, t[i] = h.distance;
	}
}
void orc_sphere_normal(size_t n, const double *sphere4, const double *ray6, const double *t, double *n3) {
	for (size_t i = 0; i < n; i++) {
		Sphere s{v3_from(sphere4 + 4 * i), sphere4[4 * i + 3]};
		v3_store(n3 + 3 * i, sphere_normal(s, ray_from(ray6 + 6 * i), t[i]));
	}
}
void orc_plane_intersect(size_t n, const double *plane6, const double *ray6, int32_t *hit, double *t) {
	for (size_t i = 0; i < n; i++) {
		Plane p{v3_from(plane6 + 6 * i), v3_from(plane6 + 6 * i + 3)};
		Hit h = plane_intersects(p, ray_from(ray6 + 6 * i));
		hit[i] = h.some, t[i] = h.distance;
	}
}
void orc_aabb_intersect(size_t n, const double *aabb6, const double *ray6, int32_t *hit, double *t) {
	for (size_t i = 0; i < n; i++) {
		AABB b{v3_from(aabb6 + 6 * i), v3_from(aabb6 + 6 * i + 3)};
		Hit h = aabb_intersects(b, ray_from(ray6 + 6 * i));
		hit[i] = h.some, t[i] = h.distance;
	}
}
void orc_triangle_intersect(size_t n, const double *pos9, const double *ray6, int32_t *hit, double *t) {
	for (size_t i = 0; i < n; i++) {
		Hit h = triangle_intersects(tri_from(pos9 + 9 * i, nullptr), ray_from(ray6 + 6 * i));
		hit[i] = h.some, t[i] = h.distance;
	}
}
void orc_triangle_normal(size_t n, const double *pos9, const double *nrm9, const double *ray6, const double *t,
                         double *n3) {
	for (size_t i = 0; i < n; i++)
		v3_store(n3 + 3 * i, triangle_normal(tri_from(pos9 + 9 * i, nrm9 + 9 * i), ray_from(ray6 + 6 * i), t[i]));
}
void orc_onb(size_t n, const double *n3, double *t3, double *b3) {
	for (size_t i = 0; i < n; i++) {
		V3 t, b;
		create_coordinate_system_of_n(v3_from(n3 + 3 * i), t, b);
		v3_store(t3 + 3 * i, t), v3_store(b3 + 3 * i, b);
	}
}
void orc_cosine_hemisphere(size_t n, const double *r1, const double *r2, double *dir3, double *pdf) {
	for (size_t i = 0; i < n; i++) {
		V3 d;
		uniform_sample_hemisphere(r1[i], r2[i], d, pdf[i]);
		v3_store(dir3 + 3 * i, d);
	}
}
void orc_importance_sample_ggx(size_t n, const double *reflect3, const double *rough, const double *r1,
                               const double *r2, double *dir3) {
	for (size_t i = 0; i < n; i++)
		v3_store(dir3 + 3 * i, importance_sample_ggx(v3_from(reflect3 + 3 * i), rough[i], r1[i], r2[i]));
}
void orc_ggx_distribution(size_t n, const double *n3, const double *h3, const double *rough, double *out) {
	for (size_t i = 0; i < n; i++) out[i] = ggx_distribution(v3_from(n3 + 3 * i), v3_from(h3 + 3 * i), rough[i]);
}
void orc_geometry_smith(size_t n, const double *n3, const double *v3_, const double *l3, const double *rough,
                        double *out) {
	for (size_t i = 0; i < n; i++)
		out[i] = geometry_smith(v3_from(n3 + 3 * i), v3_from(v3_ + 3 * i), v3_from(l3 + 3 * i), rough[i]);
}
void orc_fresnel_schlick(size_t n, const double *cos_theta, const double *f0_3, double *out3) {
	for (size_t i = 0; i < n; i++) v3_store(out3 + 3 * i, fresnel_schlick(cos_theta[i], v3_from(f0_3 + 3 * i)));
}
void orc_primary_ray(size_t n, const rmd_camera *cam, const uint32_t *xy2, const double *u2, double *ray6) {
	for (size_t i = 0; i < n; i++) {
		Ray r = generate_primary_ray_u(xy2[2 * i], xy2[2 * i + 1], *cam, u2[2 * i], u2[2 * i + 1]);
		v3_store(ray6 + 6 * i, r.origin), v3_store(ray6 + 6 * i + 3, r.direction);
	}
}

int32_t orc_grid_build(const double *tri_pos, const double *tri_nrm, uint64_t n_tris, orc_grid **out) {
	std::vector<Triangle> tris(n_tris);
	for (uint64_t i = 0; i < n_tris; i++) tris[i] = tri_from(tri_pos + 9 * i, tri_nrm ? tri_nrm + 9 * i : nullptr);
	auto g = std::make_unique<orc_grid>();
	int rc = build_from_mesh(std::move(tris), g->g);
	if (rc != 0) return rc;
	AccGrid &a = g->g;
	a.cells32.assign(a.cells.begin(), a.cells.end());
	a.map32.assign(a.mapping_table.begin(), a.mapping_table.end());
	a.pos.resize(a.triangles.size() * 9), a.nrm.resize(a.triangles.size() * 9);
	for (size_t i = 0; i < a.triangles.size(); i++) {
		const Triangle &t = a.triangles[i];
		v3_store(&a.pos[9 * i], t.p0), v3_store(&a.pos[9 * i + 3], t.p1), v3_store(&a.pos[9 * i + 6], t.p2);
		v3_store(&a.nrm[9 * i], t.n0), v3_store(&a.nrm[9 * i + 3], t.n1), v3_store(&a.nrm[9 * i + 6], t.n2);
	}
	*out = g.release();
	return 0;
}
void orc_grid_describe(const orc_grid *g, rmd_grid_desc *d) {
	const AccGrid &a = g->g;
	std::memset(d, 0, sizeof(*d));
	v3_store(d->bbox_min, a.bounding_box.min), v3_store(d->bbox_max, a.bounding_box.max);
	for (int i = 0; i < 3; i++) d->resolution[i] = (uint32_t)a.res[i];
	v3_store(d->cell_size, a.cell_size);
	d->cells = a.cells32.data(), d->n_cells = a.cells32.size();
	d->mapping_table = a.map32.data(), d->n_mapping = a.map32.size();
	d->tri_pos = a.pos.data(), d->tri_nrm = a.nrm.data(), d->n_tris = a.triangles.size();
}
void orc_grid_destroy(orc_grid *g) { delete g; }

orc_scene *orc_scene_create(const rmd_object *objects, uint32_t n_objects, const rmd_grid_desc *grids,
                            uint32_t n_grids) {
	auto s = std::make_unique<orc_scene>();
	for (uint32_t gi = 0; gi < n_grids; gi++) {
		const rmd_grid_desc &d = grids[gi];
		auto g = std::make_shared<AccGrid>();
		g->bounding_box = AABB{v3_from(d.bbox_min), v3_from(d.bbox_max)};
		for (int i = 0; i < 3; i++) g->res[i] = d.resolution[i];
		g->cell_size = v3_from(d.cell_size);
		g->cells.assign(d.cells, d.cells + d.n_cells);
		g->mapping_table.assign(d.mapping_table, d.mapping_table + d.n_mapping);
		g->triangles.resize(d.n_tris);
		for (uint64_t i = 0; i < d.n_tris; i++) g->triangles[i] = tri_from(d.tri_pos + 9 * i, d.tri_nrm + 9 * i);
		s->scene.grids.push_back(g);
	}
	for (uint32_t i = 0; i < n_objects; i++) {
		const rmd_object &o = objects[i];
		Object obj;
		obj.geometry_kind = o.geometry_kind;
		obj.sphere = Sphere{v3_from(o.origin), o.radius};
		obj.plane = Plane{v3_from(o.origin), v3_from(o.normal)};
		if (o.geometry_kind == RMD_GEOM_GRID) {
			if (o.grid_index >= n_grids) return nullptr;
			obj.grid = s->scene.grids[o.grid_index];
		}
		obj.material = Material{o.material.kind, v3_from(o.material.color), o.material.roughness};
		s->scene.objects.push_back(obj);
	}
	return s.release();
}
void orc_scene_destroy(orc_scene *s) { delete s; }

void orc_scene_intersect(const orc_scene *s, size_t n, const double *ray6, int32_t *obj, double *t, uint32_t *sub) {
	for (size_t i = 0; i < n; i++) {
		Hit h;
		obj[i] = scene_intersect(s->scene, ray_from(ray6 + 6 * i), h);
		t[i] = obj[i] >= 0 ? h.distance : 0.0;
		sub[i] = obj[i] >= 0 ? (uint32_t)h.subobject_index : 0u;
	}
}
void orc_grid_intersect(const orc_scene *s, uint32_t g, size_t n, const double *ray6, int32_t *hit, double *t,
                        uint32_t *tri) {
	for (size_t i = 0; i < n; i++) {
		Hit h = grid_intersects(*s->scene.grids[g], ray_from(ray6 + 6 * i));
		hit[i] = h.some, t[i] = h.distance, tri[i] = (uint32_t)h.subobject_index;
	}
}

int32_t orc_trace_sample(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, uint32_t x, uint32_t y,
                         uint32_t sample, double rgb[3], int32_t *path_obj, uint32_t *path_sub) {
	int32_t len = 0;
	V3 c = sample_pixel(s->scene, *cam, *st, x, y, sample, path_obj, path_sub, &len);
	v3_store(rgb, c);
	return len;
}
void orc_trace_samples(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, size_t n,
                       const uint32_t *xy2, const uint32_t *sample, double *rgb_out) {
	for (size_t i = 0; i < n; i++)
		v3_store(rgb_out + 3 * i, sample_pixel(s->scene, *cam, *st, xy2[2 * i], xy2[2 * i + 1], sample[i]));
}

/* render_tiled's worker pool, src/trace.rs:137-230.  Tiles arrive already generated (the
 * caller reproduces :142-173).  Each worker clones nothing here (the scene is read-only),
 * pops a tile, adds ONE sample to every pixel (:197-205), and re-queues it until
 * sample_count passes are done (:207-220). */
void orc_render_tiles(const orc_scene *s, const rmd_camera *cam, const rmd_settings *st, const rmd_tile_rect *tiles,
                      uint32_t n_tiles, double *accum, uint32_t n_threads) {
	struct Work {
		rmd_tile_rect rect;
		uint32_t sample_count;
	};
	std::deque<Work> queue;
	std::mutex qm;
	for (uint32_t i = 0; i < n_tiles; i++) queue.push_back(Work{tiles[i], 0});
	if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency()); /* num_cpus::get(), :44 */
	const uint32_t W = cam->backbuffer_width;
	auto worker = [&]() {
		for (;;) {
			Work w;
			{
				std::lock_guard<std::mutex> lock(qm);
				if (queue.empty()) break; /* try_pop() == None -> thread exits, :189-195 */
				w = queue.front();
				queue.pop_front();
			}
			if (st->sample_count == 0) continue;
			uint32_t sidx = st->sample_begin + w.sample_count;
			for (uint32_t y = w.rect.top; y < w.rect.top + w.rect.height; y++)
				for (uint32_t x = w.rect.left; x < w.rect.left + w.rect.width; x++) {
					V3 c = sample_pixel(s->scene, *cam, *st, x, y, sidx);
					double *px = accum + ((size_t)x + (size_t)y * W) * 3;
					px[0] += c.x, px[1] += c.y, px[2] += c.z; /* :203 */
				}
			w.sample_count++;
			if (w.sample_count != st->sample_count) {
				std::lock_guard<std::mutex> lock(qm);
				queue.push_back(w);
			}
		}
		flush_counters();
	};
	std::vector<std::thread> pool;
	for (uint32_t i = 0; i < n_threads; i++) pool.emplace_back(worker);
	for (auto &t : pool) t.join();
}

/* Mesh::load_ply (mesh.rs:58-121) from the text of a file */
int32_t orc_mesh_load_ply_text(const char *text, size_t n_bytes, orc_mesh **out) {
	auto m = std::make_unique<orc_mesh>();
	if (!load_ply_text(std::string(text, n_bytes), m->triangles)) return 1;
	m->refresh();
	*out = m.release();
	return 0;
}
int32_t orc_mesh_load_ply(const char *path, orc_mesh **out) {
	FILE *f = std::fopen(path, "rb");
	if (!f) return 1; /* fs::read_to_string(path).unwrap() */
	std::string buf;
	char chunk[65536];
	size_t n;
	while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, n);
	std::fclose(f);
	return orc_mesh_load_ply_text(buf.data(), buf.size(), out);
}
void orc_mesh_bake_transform(orc_mesh *m, const double translate[3]) {
	bake_transform(m->triangles, v3_from(translate));
	m->refresh();
}
uint64_t orc_mesh_size(const orc_mesh *m) { return m->triangles.size(); }
void orc_mesh_arrays(const orc_mesh *m, const double **tri_pos, const double **tri_nrm) { *tri_pos = m->pos.data(), *tri_nrm = m->nrm.data(); }
/* Mesh::find_mesh_bounds, mesh.rs:123-140 (Q9 seed constants) */
void orc_mesh_bounds(const orc_mesh *m, double bbox_min[3], double bbox_max[3]) {
	AABB b = mesh_bounds(m->triangles);
	v3_store(bbox_min, b.min), v3_store(bbox_max, b.max);
}
void orc_mesh_destroy(orc_mesh *m) { delete m; }

/* TaskHandle::await's division (src/trace.rs:95) followed by cli_old/src/main.rs:161-181:
 *   tone_mapped = 1 - exp(p * -1.0 * exposure);  tone_mapped = tone_mapped.powf(1.0 / gamma);
 *   (tone_mapped * 255.0).cast::<u8>()  -> None (pixel stays (0,0,0), :176-181) if any channel is NaN or outside (-1, 256) */
void orc_resolve_tonemap(const double *accum, size_t n_pixels, double sample_count, double exposure, double gamma, uint8_t *rgb8) {
	for (size_t i = 0; i < n_pixels; i++) {
		double v[3];
		bool ok = true;
		for (int c = 0; c < 3; c++) {
			double p = accum[i * 3 + c] / sample_count;
			double tm = 1.0 - std::exp(p * -1.0 * exposure);
			tm = std::pow(tm, 1.0 / gamma);
			v[c] = tm * 255.0;
			ok = ok && (v[c] > -1.0 && v[c] < 256.0);
		}
		for (int c = 0; c < 3; c++) rgb8[i * 3 + c] = ok ? (uint8_t)v[c] : (uint8_t)0;
	}
}

/* 1: the work counters include the segments behind a zero bounce weight (everything the reference executes); 0 (default): they stop there,
 * like the product's default (see ORC_COUNT) */
void orc_count_black_paths(int32_t on) { g_count_black_paths = on ? 1 : 0; }

/* tools/mutation_pins.py only.  Returns the number of mutations, or -1 in the build that has none. */
int32_t orc_set_mutation(int32_t k) {
#ifdef ORC_NO_COUNTERS
	(void)k;
	return -1;
#else
	g_mutation = (k > 0 && k < MUT_COUNT) ? k : MUT_NONE;
	return MUT_COUNT;
#endif
}

void orc_walk_hist(uint64_t out[4 * 65 + 1]) {
	for (int k = 0; k < 4; k++)
		for (int b = 0; b < 65; b++) out[k * 65 + b] = g_walk_hist[k][b].load();
	out[4 * 65] = g_walk_hits.load();
}
void orc_counters_reset(void) {
	for (int k = 0; k < 4; k++)
		for (int b = 0; b < 65; b++) g_walk_hist[k][b] = 0;
	g_walk_hits = 0;
	flush_counters();
	std::lock_guard<std::mutex> lock(g_counter_mutex);
	g_counters = Counters();
}
void orc_counters_get(uint64_t out[12]) {
	flush_counters();
	std::lock_guard<std::mutex> lock(g_counter_mutex);
	for (int i = 0; i < K_COUNT; i++) out[i] = g_counters.c[i];
}

} // extern "C"
