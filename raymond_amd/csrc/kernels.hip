// HIP kernels of the radiance integrator for gfx950 (CDNA4).
//
// render_kernel<MODE=tiles|tiles-buffered, GRID, PERSIST> — the hot path (render_kernel.hpp).  Persistent 16-wave workgroups, one per CU, whose
//   waves draw (8x8 wave tile, sample range) work items from a counter.  trace()'s recursion (src/trace.rs:232-320) becomes a running
//   throughput: each bounce's weight is multiplied in when it is produced and the terminal radiance is scaled by the product (DESIGN.md
//   section 3).  The object table is staged into LDS once per workgroup; the uniform closest-hit loop reads it through scalar loads, the
//   divergent post-hit lookup reads the LDS copy.  Two wave bodies:
//     render_wave_sorted (split launches of scenes without grids — the spheres kernel): the wave's paths live in a pool of slots in LDS and a trip is
//       64 new samples (primary ray, intersection, classification) or 64 parked hits (shading, intersection, classification): every lane
//       of a trip needs the same thing;
//     render_wave (everything else): a lane keeps its path and regenerates the next sample's primary ray in place; GRID = true adds the
//       wave-cooperative grid walk (grid_walk.hpp) and shares the grids' occupancy masks through LDS.
//   MFMA is not used: there is no dense contraction anywhere on this path.
// (The kernel template itself is in render_kernel.hpp; its list instantiation — an explicit (x, y, sample) per lane, for the
// parity probes — and the probe kernels are compiled into libraymond_hip_probe.so from probe_kernels.hip, not into this library.)
// sum_kernel, tonemap_kernel — the ordered per-sample sum of split launches of mesh scenes (the spheres kernel does it itself) and
// the output stage.
#include <hip/hip_runtime.h>

#include "render_kernel.hpp"

namespace rmd {

// Second half of a split launch: pixel += sample(s) for s = sample_begin .. +sample_count-1, strictly in that order
// (src/trace.rs:203), reading the per-sample radiance the render kernel stored.  One wavefront per wave tile, lane = pixel;
// every load is 24 contiguous bytes per lane, 1.5 KiB contiguous per wave.
__global__ __launch_bounds__(64) void sum_kernel(RenderParams P, const WaveTile *__restrict__ tiles, const double *__restrict__ buf,
                                                 double *__restrict__ out) {
	const uint32_t wt = blockIdx.x, lane = threadIdx.x;
	const WaveTile t = tiles[wt];
	const uint32_t lx = lane & 7u, ly = lane >> 3;
	if (lx >= t.w || ly >= t.h) return;
	const size_t pix = ((size_t)(t.x0 + lx) + (size_t)(t.y0 + ly) * P.W) * 3;
	V3 acc = ld3(out + pix);
	const double *src = buf + ((size_t)wt * P.sample_count * 64u + lane) * kSampleStride;
	for (uint32_t s = 0; s < P.sample_count; s++) {
		acc = acc + ld3(src);
		src += 64u * kSampleStride;
	}
	out[pix + 0] = acc.x, out[pix + 1] = acc.y, out[pix + 2] = acc.z;
}

// ---------------------------------------------------------------- resolve + tone-map (cli_old/src/main.rs:161-181, src/trace.rs:95)
// The reference's 8-bit value is trunc(255 * (1 - exp(-p * exposure))^(1/gamma)) with the HOST libm's exp and powf.  The device's exp / pow
// differ from a host libm by an ulp or two, which can only change the byte when 255 * tm lies within a few 1e-13 of an integer.  So the
// kernel computes every pixel and FLAGS the few whose value is within kTonemapGuard of a truncation boundary (1 .. 255); the host
// recomputes exactly those with its libm (api.cpp: rmd_resolve_tonemap) — byte-exact output, and a pixel in ~10^6 takes the slow road.
// Two exact cases need no flag: an argument <= -40 gives exp < 2^-57, 1 - exp = 1, pow(1, y) = 1 and 255 exactly with ANY libm
// (saturated pixels); and values that truncate to 0 either way (|255 tm| < 1 - guard: black pixels).
__global__ __launch_bounds__(256) void tonemap_kernel(const double *__restrict__ accum, uint8_t *__restrict__ rgb8, size_t n_pixels,
                                                       double sample_count, double exposure, double inv_gamma, uint32_t *__restrict__ flagged,
                                                       uint32_t *__restrict__ n_flagged) {
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_pixels) return;
	double v[3];
	bool ok = true, flag = false;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		double p = accum[i * 3 + c] / sample_count; // TaskHandle::await, src/trace.rs:95
		const double arg = p * -1.0 * exposure;
		double tm = 1.0 - exp(arg);
		tm = pow(tm, inv_gamma);
		v[c] = tm * 255.0;
		ok = ok && (v[c] > -1.0 && v[c] < 256.0);
		const double r = __builtin_rint(v[c]);
		flag = flag || (r >= 1.0 && __builtin_fabs(v[c] - r) < kTonemapGuard && !(arg <= -40.0));
	}
	// Vector3<f64>.cast::<u8>() is None if any channel is NaN / out of range: the pixel then stays (0,0,0)
#pragma unroll
	for (int c = 0; c < 3; c++) rgb8[i * 3 + c] = ok ? (uint8_t)v[c] : (uint8_t)0;
	if (flag) flagged[atomicAdd(n_flagged, 1u)] = (uint32_t)i; // the list has room for every pixel
}

// ---------------------------------------------------------------- tile rectangles <-> a packed buffer (rmd_framebuffer_{download,upload}_tiles)
// One workgroup per rectangle; rect i's pixels row-major at packed + first[i] * 3 — the layout of core::tile::Tile.data (core/src/tile.rs:13).
// Every access is 24 contiguous bytes per lane, a row of the rectangle (<= 32 pixels = 768 bytes) contiguous on both sides.
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void tile_copy_kernel(double *__restrict__ frame, double *__restrict__ packed, const rmd_tile_rect *__restrict__ rects,
                                                         const uint64_t *__restrict__ first, uint32_t W) {
	const rmd_tile_rect r = rects[blockIdx.x];
	const uint64_t base = first[blockIdx.x];
	const uint32_t n = r.width * r.height;
	for (uint32_t i = threadIdx.x; i < n; i += 256u) {
		const uint32_t x = i % r.width, y = i / r.width;
		double *f = frame + ((size_t)(r.left + x) + (size_t)(r.top + y) * W) * 3, *p = packed + (base + i) * 3;
		if (TO_PACKED) p[0] = f[0], p[1] = f[1], p[2] = f[2];
		else f[0] = p[0], f[1] = p[1], f[2] = p[2];
	}
}
hipError_t launch_tile_copy(hipStream_t stream, bool to_packed, double *frame, double *packed, const rmd_tile_rect *rects, const uint64_t *first,
                            uint32_t n_rects, uint32_t W) {
	if (n_rects == 0) return hipSuccess;
	if (to_packed) hipLaunchKernelGGL(tile_copy_kernel<true>, dim3(n_rects), dim3(256), 0, stream, frame, packed, rects, first, W);
	else hipLaunchKernelGGL(tile_copy_kernel<false>, dim3(n_rects), dim3(256), 0, stream, frame, packed, rects, first, W);
	return hipGetLastError();
}

// ---------------------------------------------------------------- launchers
size_t render_lds_bytes(uint32_t n_objects, uint32_t mask_words_total, uint32_t waves_per_wg) {
	// a scene has grids exactly when it has mask words reserved (an all-empty grid still reserves some)
	return (size_t)n_objects * sizeof(DevObject) + (size_t)((mask_words_total + 3u) & ~3u) * 4u +
	       (size_t)waves_per_wg * wave_lds_bytes(mask_words_total ? 1u : 0u);
}

// Waves per workgroup: one for grid-less scenes (finest load balance); with grids the waves of a workgroup share the
// occupancy masks staged in LDS.
uint32_t render_waves_per_wg(uint32_t n_objects, uint32_t mask_words_total) {
	if (mask_words_total == 0) return 1;
	uint32_t w = kGridWavesPerWg;
	while (w > 1 && render_lds_bytes(n_objects, mask_words_total, w) > kLdsBudgetBytes) w--;
	return w;
}

hipError_t launch_render_tiles(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids,
                               const WaveTile *wave_tiles, double *accum, uint32_t n_cus, LaunchShape *shape) {
	if (P.n_work == 0) return hipSuccess;
	const bool buffered = P.buffered != 0u;
	const uint32_t n_waves = P.n_work * (buffered ? P.split_k : 1u);
	hipError_t e;
	if (P.n_grids) e = buffered ? launch_render<kModeTilesBuffered, true>(stream, P, objs, grids, wave_tiles, n_waves, accum, nullptr, nullptr, n_cus, shape)
	                            : launch_render<kModeTiles, true>(stream, P, objs, grids, wave_tiles, n_waves, accum, nullptr, nullptr, n_cus, shape);
	else e = buffered ? launch_render<kModeTilesBuffered, false>(stream, P, objs, grids, wave_tiles, n_waves, accum, nullptr, nullptr, n_cus, shape)
	                  : launch_render<kModeTiles, false>(stream, P, objs, grids, wave_tiles, n_waves, accum, nullptr, nullptr, n_cus, shape);
	if (e != hipSuccess || !buffered || P.tile_done != nullptr) return e; // with tile_done the render kernel's waves add the samples themselves
	hipLaunchKernelGGL(sum_kernel, dim3(P.n_work), dim3(64), 0, stream, P, wave_tiles, (const double *)P.sample_buf, accum);
	return hipGetLastError();
}

hipError_t launch_sum(hipStream_t stream, const RenderParams &P, const WaveTile *wave_tiles, double *accum) {
	if (P.n_work == 0) return hipSuccess;
	hipLaunchKernelGGL(sum_kernel, dim3(P.n_work), dim3(64), 0, stream, P, wave_tiles, (const double *)P.sample_buf, accum);
	return hipGetLastError();
}

hipError_t launch_tonemap(hipStream_t stream, const double *accum, uint8_t *rgb8, size_t n, double sample_count, double exposure,
                          double inv_gamma, uint32_t *flagged, uint32_t *n_flagged) {
	// n = number of pixels; flagged: n entries, n_flagged: one zeroed word
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, accum, rgb8, n, sample_count, exposure,
	                   inv_gamma, flagged, n_flagged);
	return hipGetLastError();
}

} // namespace rmd
