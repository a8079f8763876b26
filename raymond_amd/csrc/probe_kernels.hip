// Kernels of the diagnostic probes (include/raymond_hip_probe.h) — libraymond_hip_probe.so, test infrastructure, not part of
// the product library: one device function per element (probe_kernel), Scene::intersect / AccGrid::intersects per ray
// (probe_scene_kernel), and the render kernel's list instantiation (one explicit (x, y, sample) per lane, hit sequence recorded).
#include "render_kernel.hpp"

namespace rmd {

// ---------------------------------------------------------------- probes
__global__ __launch_bounds__(64) void probe_kernel(int op, uint32_t n, const double *__restrict__ in, int in_stride,
                                                   double *__restrict__ out, int out_stride, RenderParams P) {
	uint32_t i = blockIdx.x * 64u + threadIdx.x;
	if (i >= n) return;
	const double *a = in + (size_t)i * in_stride;
	double *o = out + (size_t)i * out_stride;
	switch (op) {
	case PROBE_PHILOX: {
		uint32_t w0, w1, w2, w3;
		philox4x32_10((uint32_t)a[0], (uint32_t)a[1], (uint32_t)a[2], (uint32_t)a[3], (uint32_t)a[4], (uint32_t)a[5], w0, w1, w2, w3);
		o[0] = w0, o[1] = w1, o[2] = w2, o[3] = w3;
	} break;
	case PROBE_UNIFORM: {
		// a = key0, key1, pixel, sample, block: the block's two 53-bit uniforms, its 22-bit uniform, and the same through next3()
		Rng r;
		r.init((uint32_t)a[2], (uint32_t)a[3]);
		r.block = (uint32_t)a[4];
		r.next2((uint32_t)a[0], (uint32_t)a[1], o[0], o[1]);
		r.block = (uint32_t)a[4];
		r.next3((uint32_t)a[0], (uint32_t)a[1], o[2], o[3], o[4]);
	} break;
	case PROBE_SPHERE_INTERSECT: {
		double t = 0.0;
		bool h = sphere_intersect(ld3(a), a[3], ld3(a + 4), ld3(a + 7), t);
		o[0] = h, o[1] = h ? t : 0.0;
	} break;
	case PROBE_SPHERE_NORMAL: {
		V3 frag = ld3(a + 4) + ld3(a + 7) * a[10];
		V3 nn = normalize(frag - ld3(a));
		o[0] = nn.x, o[1] = nn.y, o[2] = nn.z;
	} break;
	case PROBE_PLANE_INTERSECT: {
		double t = 0.0;
		bool h = plane_intersect(ld3(a), ld3(a + 3), ld3(a + 6), ld3(a + 9), t);
		o[0] = h, o[1] = h ? t : 0.0;
	} break;
	case PROBE_AABB_INTERSECT: {
		double t = 0.0;
		bool h = aabb_intersect(ld3(a), ld3(a + 3), ld3(a + 6), ld3(a + 9), t);
		o[0] = h, o[1] = h ? t : 0.0;
	} break;
	case PROBE_TRIANGLE_INTERSECT: {
		V3 p0 = ld3(a), p1 = ld3(a + 3), p2 = ld3(a + 6);
		double t = 0.0;
		bool h = triangle_intersect(p0, p1 - p0, p2 - p0, ld3(a + 9), ld3(a + 12), t);
		o[0] = h, o[1] = h ? t : 0.0;
	} break;
	case PROBE_PRETEST_PAIR: {
		// a = sphere (centre, r2a), kb, triangle positions (9), ray (6): the walk's pre-test in the device's own arithmetic (grid_walk.hpp:
		// sphere_pretest, the function the chunk loop calls) and the reference's test on the same pair (the record the upload makes: v0, edge1, edge2)
		V3 p0 = ld3(a + 5), p1 = ld3(a + 8), p2 = ld3(a + 11);
		const V3 pro = ld3(a + 14), prd = ld3(a + 17);
		double t = 0.0;
		const bool h = triangle_intersect(p0, p1 - p0, p2 - p0, pro, prd, t);
		o[0] = sphere_pretest(ld3(a), a[3], a[4], pro, prd) ? 1.0 : 0.0, o[1] = h, o[2] = h ? t : 0.0;
	} break;
	case PROBE_TRIANGLE_NORMAL: {
		V3 frag = ld3(a + 18) + ld3(a + 21) * a[24];
		V3 nn = triangle_normal(a, a + 9, a + 25, frag); // sides/area precomputed on the host, as for an uploaded scene
		o[0] = nn.x, o[1] = nn.y, o[2] = nn.z;
	} break;
	case PROBE_ONB: {
		V3 t, b;
		onb(ld3(a), t, b);
		o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = b.x, o[4] = b.y, o[5] = b.z;
	} break;
	case PROBE_COSINE_HEMISPHERE: {
		V3 d;
		double pdf;
		cosine_hemisphere(a[0], a[1], d, pdf);
		o[0] = d.x, o[1] = d.y, o[2] = d.z, o[3] = pdf;
	} break;
	case PROBE_SAMPLE_GGX: {
		V3 d = importance_sample_ggx(ld3(a), a[3], a[4], a[5]);
		o[0] = d.x, o[1] = d.y, o[2] = d.z;
	} break;
	case PROBE_GGX_DISTRIBUTION: o[0] = ggx_distribution(ld3(a), ld3(a + 3), a[6]); break;
	case PROBE_GEOMETRY_SMITH: o[0] = geometry_smith(ld3(a), ld3(a + 3), ld3(a + 6), a[9]); break;
	case PROBE_FRESNEL_SCHLICK: {
		V3 f = fresnel_schlick(a[0], ld3(a + 1));
		o[0] = f.x, o[1] = f.y, o[2] = f.z;
	} break;
	case PROBE_PRIMARY_RAY: {
		V3 ro, rd;
		primary_ray(P, (uint32_t)a[0], (uint32_t)a[1], a[2], a[3], ro, rd);
		o[0] = ro.x, o[1] = ro.y, o[2] = ro.z, o[3] = rd.x, o[4] = rd.y, o[5] = rd.z;
	} break;
	case PROBE_ELEMENTARY: { // the device's own sqrt / sin / cos (device_core.hpp)
		o[0] = sqrt64(a[0]);
		sincos_cw(a[0], o[1], o[2]);
		sqrt_and_inverse(a[0], o[3], o[4]);
	} break;
	default: break;
	}
}

// mode 0: Scene::intersect -> (obj, t, sub);  mode 1: AccGrid::intersects on grid `g` -> (hit, t, tri).
// 64-thread workgroups; the grids' occupancy masks are read from LDS exactly as in the render kernel when `use_masks`.
__global__ __launch_bounds__(64) void probe_scene_kernel(int mode, uint32_t g, uint32_t n, const DevObject *__restrict__ objs,
                                                         uint32_t n_objects, const DevGrid *__restrict__ grids, uint32_t n_grids,
                                                         uint32_t mask_words_total, uint32_t axis_pairs, const double *__restrict__ rays,
                                                         double *__restrict__ out) {
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t *lmasks = reinterpret_cast<uint32_t *>(smem);
	for (uint32_t gi = 0; gi < n_grids && mask_words_total; gi++) {
		const DevGrid &gg = grids[gi];
		if (gg.mask_lds_word == 0xFFFFFFFFu) continue;
		for (uint32_t i = threadIdx.x; i < gg.mask_n_words; i += 64u) lmasks[gg.mask_lds_word + i] = as_global(gg.mask_words)[i];
	}
	__syncthreads();
	const uint32_t *lds_masks = mask_words_total ? lmasks : nullptr;
	WalkScratch &scr = *reinterpret_cast<WalkScratch *>(smem + (size_t)((mask_words_total + 3u) & ~3u) * 4u);
	uint32_t i = blockIdx.x * 64u + threadIdx.x;
	const bool want = i < n;
	const size_t ri = want ? i : 0;
	V3 ro = ld3(rays + ri * 6), rd = ld3(rays + ri * 6 + 3);
	double o0, o1, o2;
	if (mode == 0) {
		double t;
		uint32_t sub;
		int oi = scene_intersect_wave<true>(objs, n_objects, grids, lds_masks, scr, want, ro, rd, t, sub, axis_pairs, 0u, nullptr, true); // (a test's rays: any bit pattern)
		o0 = oi, o1 = oi >= 0 ? t : 0.0, o2 = oi >= 0 ? sub : 0u;
	} else {
		double t = 0.0;
		uint32_t tri = 0;
		bool h = false;
		const DevGrid &gg = grids[g];
		const uint32_t *mask = (lds_masks && gg.mask_lds_word != 0xFFFFFFFFu) ? lds_masks + gg.mask_lds_word : nullptr;
		grid_intersect_wave(gg, mask, scr, want, ro, rd, h, t, tri);
		o0 = h, o1 = h ? t : 0.0, o2 = h ? tri : 0u;
	}
	if (want) {
		double *o = out + (size_t)i * 3;
		o[0] = o0, o[1] = o1, o[2] = o2;
	}
}

hipError_t launch_render_list(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids,
                              const ListWork *list, double *rgb_out, int32_t *path_obj, uint32_t *path_sub) {
	if (P.n_work == 0) return hipSuccess;
	const uint32_t n_waves = (P.n_work + 63u) / 64u;
	if (P.n_grids) return launch_render<kModeList, true>(stream, P, objs, grids, list, n_waves, rgb_out, path_obj, path_sub);
	return launch_render<kModeList, false>(stream, P, objs, grids, list, n_waves, rgb_out, path_obj, path_sub);
}

hipError_t launch_probe(hipStream_t stream, int op, uint32_t n, const double *in, int in_stride, double *out, int out_stride,
                        const RenderParams &P) {
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(probe_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, op, n, in, in_stride, out, out_stride, P);
	return hipGetLastError();
}

hipError_t launch_probe_scene(hipStream_t stream, int mode, uint32_t g, uint32_t n, const DevObject *objs, uint32_t n_objects,
                              const DevGrid *grids, uint32_t n_grids, uint32_t mask_words_total, uint32_t axis_pairs, const double *rays, double *out) {
	if (n == 0) return hipSuccess;
	const size_t probe_lds = (size_t)((mask_words_total + 3u) & ~3u) * 4u + sizeof(WalkScratch);
	if (probe_lds > 64u * 1024u) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_scene_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(probe_scene_kernel, dim3((n + 63u) / 64u), dim3(64), probe_lds, stream, mode, g, n, objs,
	                   n_objects, grids, n_grids, mask_words_total, axis_pairs, rays, out);
	return hipGetLastError();
}

} // namespace rmd
