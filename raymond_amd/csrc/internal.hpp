// Library-internal declarations shared by api.cpp, probe.cpp, comm.cpp and grid_build.cpp.
#pragma once
#include <stdint.h>

#include <cmath>
#include <limits>
#include <algorithm>
#include <cstring>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/raymond_hip.h"

#ifdef RMD_WITH_HIP
#include <hip/hip_runtime_api.h>

#include "device_types.hpp"

struct rmd_context {
	int device = 0;
	hipStream_t stream = nullptr;
	bool owns_stream = false;
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	bool timed = false;
	std::string last_error;
	// cached wave-tile table (device) for the rect list of the previous call
	std::vector<rmd_tile_rect> cached_rects;
	uint32_t cached_W = 0, cached_H = 0;
	rmd::WaveTile *d_wave_tiles = nullptr;
	uint32_t n_wave_tiles = 0;
	size_t wave_tiles_capacity = 0;
	// per-sample radiance scratch of split launches (api.cpp: choose_split)
	double *d_sample_buf = nullptr;
	size_t sample_buf_bytes = 0;
	uint32_t wave_slots = 0; // CUs x waves per CU the render kernels can keep resident
	uint32_t n_cus = 0;
	size_t hbm_bytes = 0; // totalGlobalMem of the device
	uint32_t *d_work_counter = nullptr; // next work item of a persistent launch (render_kernel.hpp)
	unsigned char *d_queue_buf = nullptr; // the resident waves' path queues (render_kernel.hpp: render_wave_queued), allocated at the first launch that uses them
	size_t queue_buf_bytes = 0;
	uint32_t *d_tile_done = nullptr;    // split launches: finished waves per wave tile (render_kernel.hpp)
	size_t tile_done_words = 0;
	unsigned long long *d_debug_counters = nullptr; // walk diagnostics (DIAG builds, RMD_DEBUG=8|16)
	// fault words (device_types.hpp: kFault*): pinned host memory mapped into the device's address space.  A wave whose loop runs past its bound
	// writes here; the host looks after every wait for the stream (api.cpp: check_fault) — a plain host load, no copy
	uint32_t *h_fault = nullptr, *d_fault = nullptr;
	// tile transfers (rmd_framebuffer_{download,upload}_tiles): two staging slots — packed pixels, rect table, prefix table on the device — so that a
	// second download can be packed while the first is still being copied; the copy stream; per slot the event its copy ends with
	struct TransferSlot {
		double *d_packed = nullptr;
		size_t packed_bytes = 0;
		void *d_table = nullptr; // n_rects x (rmd_tile_rect, uint64 first)
		size_t table_bytes = 0;
		hipEvent_t packed_ready = nullptr, copied = nullptr;
		bool in_flight = false;
		std::vector<unsigned char> h_table; // the host copy of the table stays alive until its upload has completed
	} transfer[2];
	uint32_t next_transfer = 0;
	hipStream_t copy_stream = nullptr;
	// Tunables (include/raymond_hip.h: rmd_context_set_tunable).  Defaults come from the environment, read ONCE when the
	// context is created; none of them changes a result.
	int64_t tunable[RMD_TUNE_COUNT] = {};
	uint32_t debug_flags = 0; // RMD_DEBUG, honoured by DIAG builds only
	rmd_launch_info last_launch = {}; // rmd_last_launch_info
};

struct rmd_scene {
	rmd_context *ctx = nullptr;
	uint32_t n_objects = 0, n_grids = 0;
	uint32_t n_grid_objects = 0; // objects whose geometry is a grid
	uint32_t mask_words_total = 0; // LDS words of the grids' occupancy masks
	uint32_t axis_pairs = 0; // RenderParams::axis_pairs
	unsigned long long visit_mask = ~0ull, grid_mask = ~0ull; // RenderParams::visit_mask / grid_mask
	uint32_t walk_steps_bound = 0; // RenderParams::walk_steps_bound
	bool regular = true; // every parameter the kernel reads is finite and inside the class for which ending zero-throughput paths is exact (api.cpp: rmd_scene_create)
	rmd::DevObject *d_objects = nullptr;
	rmd::DevGrid *d_grids = nullptr;
	std::vector<void *> owned; // every device allocation of this scene
};
#endif

// Result of a grid build (host or GPU): owns the arrays a rmd_grid_desc points at.
struct rmd_grid_build {
	double bbox_min[3], bbox_max[3], cell_size[3];
	uint32_t res[3];
	std::vector<uint32_t> cells, mapping;
	std::vector<double> pos, nrm;
	// what rmd_scene_create derives from these arrays (api.cpp: GridDerived), kept for the next upload of the same grid
	std::shared_ptr<void> derived;
	std::mutex derived_mutex;
};

namespace rmd {
// Per-triangle constants of the Heron normal (triangle.rs:47-68): the two sides and the area that do not depend on the
// hit point, with exactly the operations of device_core.hpp (dist = sqrt(((dx*dx + dy*dy) + dz*dz)), heron_area_of_sides):
// every operation is a correctly rounded IEEE one and this file is compiled with -ffp-contract=off, so the values are
// bit-identical to what the kernel would compute.  out = { |p0p1|, |p0p2|, area(p0,p1,p2), exact_reciprocal(area) }.
// r = 1.0 / b for use by the device's div_by(a, b, r) (device_core.hpp), or NaN when that shortcut is not exact for this
// divisor: Markstein's correction needs the correctly rounded reciprocal (this division) and fails for a divisor whose
// significand is all ones; zero, non-finite and extreme divisors are left to the IEEE division as well.
inline double exact_reciprocal(double b) {
	uint64_t bits;
	std::memcpy(&bits, &b, 8);
	const double m = std::fabs(b);
	if (!(m >= 0x1p-500 && m <= 0x1p500) || (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) return std::nan("");
	return 1.0 / b;
}
// A sphere around a triangle for the walk's pre-test (grid_walk.hpp): a (ray, triangle) pair whose LINE passes the centre at more than
//     sqrt(r2a + kb * |centre - origin|^2)
// cannot pass triangle.rs:11-44 and is not tested.  What has to hold is that the reference's test, AS COMPUTED in binary64, fails for every pair
// the pre-test drops.  In exact arithmetic the test accepts lines through the triangle, all of which pass within r of the centre of any sphere
// that contains the three vertices.  As computed, the barycentric coordinates carry an error of at most ~4 eps |s| |h| / |a| with |a| >= 1e-8
// (EPSILON), |h| <= L (longest edge) and |s| <= |centre - origin| + L: a point up to 4.4e-8 L^2 (|d| + L) outside the triangle can still be
// accepted.  Both terms are taken a hundredfold: the radius grows by 4.4e-6 L^3 (and 1e-3 of itself, and an ulp-sized absolute term), the
// allowance k |d| with k = 4.4e-6 L^2 enters through (r + k|d|)^2 <= 1.01 r^2 + 101 k^2 |d|^2, and 32 eps |d|^2 covers the cancellation in
// |d|^2 - (d.rd)^2 of the pre-test itself.  The sphere is the smallest one around the vertices (the longest edge's, or the circumscribed one).
inline void triangle_sphere(const double *p9, double out[4], double &kb) {
	const double *P[3] = {p9, p9 + 3, p9 + 6};
	auto sub = [](const double *a, const double *b, double *o) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; };
	auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
	auto cross = [](const double *a, const double *b, double *o) { o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0]; };
	double e[3][3], l2[3];
	sub(P[2], P[1], e[0]), sub(P[0], P[2], e[1]), sub(P[1], P[0], e[2]); // edge i is opposite vertex i
	for (int i = 0; i < 3; i++) l2[i] = dot(e[i], e[i]);
	const int longest = l2[0] >= l2[1] && l2[0] >= l2[2] ? 0 : (l2[1] >= l2[2] ? 1 : 2);
	const double L = std::sqrt(l2[longest]);
	double c[3];
	for (int i = 0; i < 3; i++) c[i] = 0.5 * (P[(longest + 1) % 3][i] + P[(longest + 2) % 3][i]);
	double dv[3];
	sub(P[longest], c, dv);
	if (dot(dv, dv) > 0.25 * l2[longest]) { // acute: the circumscribed sphere's centre, A + (|b|^2 (a x b) x a + |a|^2 b x (a x b)) / (2 |a x b|^2)
		double a[3], b[3], n[3], t1[3], t2[3];
		sub(P[1], P[0], a), sub(P[2], P[0], b), cross(a, b, n);
		const double n2 = dot(n, n);
		cross(n, a, t1), cross(b, n, t2);
		if (n2 > 0.0 && std::isfinite(n2))
			for (int i = 0; i < 3; i++) c[i] = P[0][i] + (dot(b, b) * t1[i] + dot(a, a) * t2[i]) / (2.0 * n2);
	}
	double r2 = 0.0;
	for (int v = 0; v < 3; v++) { // whatever the centre came out as: the radius that contains the vertices
		sub(P[v], c, dv);
		r2 = std::max(r2, dot(dv, dv));
	}
	const double cmax = std::max(std::fabs(c[0]), std::max(std::fabs(c[1]), std::fabs(c[2])));
	const double r = std::sqrt(r2) * 1.001 + 4.4e-6 * L * L * L + 1e-12 * (1.0 + cmax);
	out[0] = c[0], out[1] = c[1], out[2] = c[2], out[3] = 1.01 * r * r;
	const double k = 4.4e-6 * L * L;
	kb = 101.0 * k * k + 32.0 * 2.220446049250313e-16;
	if (!(std::isfinite(out[0]) && std::isfinite(out[1]) && std::isfinite(out[2]) && std::isfinite(out[3]) && std::isfinite(kb)))
		out[0] = out[1] = out[2] = 0.0, out[3] = std::numeric_limits<double>::infinity(), kb = 0.0; // a triangle with non-finite vertices: never dropped
}
inline void triangle_aux(const double *p9, double out[4]) {
	auto dist = [](const double *a, const double *b) {
		const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
		return std::sqrt((dx * dx + dy * dy) + dz * dz);
	};
	const double ab = dist(p9, p9 + 3), ac = dist(p9, p9 + 6), bc = dist(p9 + 3, p9 + 6);
	const double s = (ab + ac + bc) / 2.0;
	out[0] = ab, out[1] = ac, out[2] = std::sqrt(s * (s - ab) * (s - ac) * (s - bc)), out[3] = exact_reciprocal(out[2]);
}
// Records `text` as the last error of `ctx` (or of the calling thread when ctx is null) and returns `status`.
rmd_status fail(rmd_context *ctx, rmd_status status, const std::string &text);
// The same for a caller that must not throw (the catch blocks of `guarded`): when even the message cannot be stored the status still comes back.
rmd_status fail_noexcept(rmd_context *ctx, rmd_status status, const char *what, const char *text) noexcept;
// "Nothing throws or aborts across the boundary" (include/raymond_hip.h): every extern "C" entry point that allocates with throwing containers
// (std::vector, std::string) runs its body through this — std::bad_alloc becomes RMD_ERR_OUT_OF_MEMORY, anything else RMD_ERR_HIP with the
// exception's text; the caller's process (a Rust or ctypes host) is never terminated by an unwinding C++ exception.
template <class F>
rmd_status guarded(rmd_context *ctx, const char *what, F &&body) noexcept {
	try {
		return body();
	} catch (const std::bad_alloc &) {
		return fail_noexcept(ctx, RMD_ERR_OUT_OF_MEMORY, what, "the host ran out of memory");
	} catch (const std::exception &e) {
		return fail_noexcept(ctx, RMD_ERR_HIP, what, e.what());
	} catch (...) {
		return fail_noexcept(ctx, RMD_ERR_HIP, what, "unknown exception");
	}
}
#ifdef RMD_WITH_HIP
// After a wait for the context's stream: RMD_ERR_DEVICE_FAULT (and the fault words cleared) when a wave of a launch reported one, else RMD_OK.
rmd_status check_fault(rmd_context *ctx);
#endif
} // namespace rmd
