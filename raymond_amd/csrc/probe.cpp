// Diagnostic entry points (include/raymond_hip_probe.h): pack host inputs, run one device
// function per element on the GPU, unpack.  Used by the parity tests only.
#define RMD_WITH_HIP 1
#include <cstring>
#include <vector>

#include "../../include/raymond_hip_probe.h"
#include "internal.hpp"
#include "launch.hpp"

namespace rmd {
RenderParams make_params(const rmd_context *ctx, const rmd_scene *scene, const rmd_camera *cam, const rmd_settings *st);
}

namespace {

#define RMD_HIP(ctx, call)                                                                            \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) return rmd::fail(ctx, RMD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
	} while (0)

struct DevBuf {
	void *p = nullptr;
	~DevBuf() {
		if (p) (void)hipFree(p);
	}
};

struct Column {
	const double *src;
	int width;
};

// rows of `in_stride` doubles built from the given columns; out rows of `out_stride` doubles
rmd_status run_probe(rmd_context *ctx, int op, size_t n, const std::vector<Column> &cols, int out_stride, std::vector<double> &out,
                     const rmd::RenderParams *params = nullptr) {
	if (!ctx) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "probe: null context");
	RMD_HIP(ctx, hipSetDevice(ctx->device));
	int in_stride = 0;
	for (const Column &c : cols) {
		if (!c.src && n) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: null input array");
		in_stride += c.width;
	}
	std::vector<double> in(n * in_stride);
	for (size_t i = 0; i < n; i++) {
		double *row = in.data() + i * in_stride;
		for (const Column &c : cols) {
			std::memcpy(row, c.src + i * c.width, sizeof(double) * c.width);
			row += c.width;
		}
	}
	out.assign(n * out_stride, 0.0);
	if (n == 0) return RMD_OK;
	DevBuf din, dout;
	RMD_HIP(ctx, hipMalloc(&din.p, in.size() * sizeof(double)));
	RMD_HIP(ctx, hipMalloc(&dout.p, out.size() * sizeof(double)));
	RMD_HIP(ctx, hipMemcpyAsync(din.p, in.data(), in.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, hipMemsetAsync(dout.p, 0, out.size() * sizeof(double), ctx->stream));
	rmd::RenderParams P;
	std::memset(&P, 0, sizeof(P));
	if (params) P = *params;
	RMD_HIP(ctx, rmd::launch_probe(ctx->stream, op, (uint32_t)n, (const double *)din.p, in_stride, (double *)dout.p, out_stride, P));
	RMD_HIP(ctx, hipMemcpyAsync(out.data(), dout.p, out.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return RMD_OK;
}

std::vector<double> widen(const uint32_t *src, size_t n) {
	std::vector<double> v(n);
	for (size_t i = 0; i < n; i++) v[i] = (double)src[i];
	return v;
}

rmd_status hit_t_probe(rmd_context *ctx, int op, size_t n, const double *shape, int shape_w, const double *ray6, int32_t *hit, double *t) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, op, n, {{shape, shape_w}, {ray6, 6}}, 2, out)) return s;
	for (size_t i = 0; i < n; i++) hit[i] = out[2 * i] != 0.0, t[i] = out[2 * i + 1];
	return RMD_OK;
}

void unpack(const std::vector<double> &out, int stride, int offset, int width, double *dst, size_t n) {
	for (size_t i = 0; i < n; i++) std::memcpy(dst + i * width, out.data() + i * stride + offset, sizeof(double) * width);
}

} // namespace

extern "C" {

rmd_status rmd_probe_philox4x32_10(rmd_context *ctx, size_t n, const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4) {
	if (!ctr4 || !key2 || !out4) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: null array");
	std::vector<double> c = widen(ctr4, n * 4), k = widen(key2, n * 2), out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_PHILOX, n, {{c.data(), 4}, {k.data(), 2}}, 4, out)) return s;
	for (size_t i = 0; i < n * 4; i++) out4[i] = (uint32_t)out[i];
	return RMD_OK;
}

rmd_status rmd_probe_block_uniforms(rmd_context *ctx, uint64_t seed, size_t n, const uint32_t *pixel, const uint32_t *sample,
                                    const uint32_t *block, double *out5) {
	if (!pixel || !sample || !block || !out5) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: null array");
	std::vector<double> k0(n, (double)(uint32_t)seed), k1(n, (double)(uint32_t)(seed >> 32));
	std::vector<double> p = widen(pixel, n), s = widen(sample, n), d = widen(block, n), out;
	if (rmd_status st = run_probe(ctx, rmd::PROBE_UNIFORM, n, {{k0.data(), 1}, {k1.data(), 1}, {p.data(), 1}, {s.data(), 1}, {d.data(), 1}}, 5, out))
		return st;
	std::memcpy(out5, out.data(), n * 5 * sizeof(double));
	return RMD_OK;
}

rmd_status rmd_probe_sphere_intersect(rmd_context *ctx, size_t n, const double *sphere4, const double *ray6, int32_t *hit, double *t) {
	return hit_t_probe(ctx, rmd::PROBE_SPHERE_INTERSECT, n, sphere4, 4, ray6, hit, t);
}
rmd_status rmd_probe_sphere_normal(rmd_context *ctx, size_t n, const double *sphere4, const double *ray6, const double *t, double *n3) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_SPHERE_NORMAL, n, {{sphere4, 4}, {ray6, 6}, {t, 1}}, 3, out)) return s;
	unpack(out, 3, 0, 3, n3, n);
	return RMD_OK;
}
rmd_status rmd_probe_plane_intersect(rmd_context *ctx, size_t n, const double *plane6, const double *ray6, int32_t *hit, double *t) {
	return hit_t_probe(ctx, rmd::PROBE_PLANE_INTERSECT, n, plane6, 6, ray6, hit, t);
}
rmd_status rmd_probe_aabb_intersect(rmd_context *ctx, size_t n, const double *aabb6, const double *ray6, int32_t *hit, double *t) {
	return hit_t_probe(ctx, rmd::PROBE_AABB_INTERSECT, n, aabb6, 6, ray6, hit, t);
}
rmd_status rmd_probe_triangle_intersect(rmd_context *ctx, size_t n, const double *pos9, const double *ray6, int32_t *hit, double *t) {
	return hit_t_probe(ctx, rmd::PROBE_TRIANGLE_INTERSECT, n, pos9, 9, ray6, hit, t);
}
rmd_status rmd_probe_triangle_normal(rmd_context *ctx, size_t n, const double *pos9, const double *nrm9, const double *ray6,
                                     const double *t, double *n3) {
	std::vector<double> out;
	std::vector<double> aux(n * 4);
	for (size_t i = 0; i < n; i++) rmd::triangle_aux(pos9 + i * 9, aux.data() + i * 4); // as rmd_scene_create does
	if (rmd_status s = run_probe(ctx, rmd::PROBE_TRIANGLE_NORMAL, n, {{pos9, 9}, {nrm9, 9}, {ray6, 6}, {t, 1}, {aux.data(), 4}}, 3, out)) return s;
	unpack(out, 3, 0, 3, n3, n);
	return RMD_OK;
}
rmd_status rmd_probe_onb(rmd_context *ctx, size_t n, const double *n3, double *t3, double *b3) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_ONB, n, {{n3, 3}}, 6, out)) return s;
	unpack(out, 6, 0, 3, t3, n);
	unpack(out, 6, 3, 3, b3, n);
	return RMD_OK;
}
rmd_status rmd_probe_cosine_hemisphere(rmd_context *ctx, size_t n, const double *r1, const double *r2, double *dir3, double *pdf) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_COSINE_HEMISPHERE, n, {{r1, 1}, {r2, 1}}, 4, out)) return s;
	unpack(out, 4, 0, 3, dir3, n);
	unpack(out, 4, 3, 1, pdf, n);
	return RMD_OK;
}
rmd_status rmd_probe_importance_sample_ggx(rmd_context *ctx, size_t n, const double *reflect3, const double *rough, const double *r1,
                                           const double *r2, double *dir3) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_SAMPLE_GGX, n, {{reflect3, 3}, {rough, 1}, {r1, 1}, {r2, 1}}, 3, out)) return s;
	unpack(out, 3, 0, 3, dir3, n);
	return RMD_OK;
}
rmd_status rmd_probe_ggx_distribution(rmd_context *ctx, size_t n, const double *n3, const double *h3, const double *rough, double *o) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_GGX_DISTRIBUTION, n, {{n3, 3}, {h3, 3}, {rough, 1}}, 1, out)) return s;
	unpack(out, 1, 0, 1, o, n);
	return RMD_OK;
}
rmd_status rmd_probe_geometry_smith(rmd_context *ctx, size_t n, const double *n3, const double *v3, const double *l3,
                                    const double *rough, double *o) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_GEOMETRY_SMITH, n, {{n3, 3}, {v3, 3}, {l3, 3}, {rough, 1}}, 1, out)) return s;
	unpack(out, 1, 0, 1, o, n);
	return RMD_OK;
}
rmd_status rmd_probe_fresnel_schlick(rmd_context *ctx, size_t n, const double *cos_theta, const double *f0_3, double *out3) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_FRESNEL_SCHLICK, n, {{cos_theta, 1}, {f0_3, 3}}, 3, out)) return s;
	unpack(out, 3, 0, 3, out3, n);
	return RMD_OK;
}
rmd_status rmd_probe_elementary(rmd_context *ctx, size_t n, const double *x, double *sqrt_out, double *sin_out, double *cos_out,
                                double *root_out, double *inv_root_out) {
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_ELEMENTARY, n, {{x, 1}}, 5, out)) return s;
	unpack(out, 5, 0, 1, sqrt_out, n);
	unpack(out, 5, 1, 1, sin_out, n);
	unpack(out, 5, 2, 1, cos_out, n);
	unpack(out, 5, 3, 1, root_out, n);
	unpack(out, 5, 4, 1, inv_root_out, n);
	return RMD_OK;
}
rmd_status rmd_probe_primary_ray(rmd_context *ctx, size_t n, const rmd_camera *cam, const uint32_t *xy2, const double *u2, double *ray6) {
	if (!cam || !xy2) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: null argument");
	rmd_settings st;
	std::memset(&st, 0, sizeof(st));
	rmd::RenderParams P = rmd::make_params(ctx, nullptr, cam, &st);
	std::vector<double> xy = widen(xy2, n * 2), out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_PRIMARY_RAY, n, {{xy.data(), 2}, {u2, 2}}, 6, out, &P)) return s;
	unpack(out, 6, 0, 6, ray6, n);
	return RMD_OK;
}

static rmd_status scene_probe(rmd_context *ctx, const rmd_scene *scene, int mode, uint32_t g, size_t n, const double *ray6, int32_t *a,
                              double *t, uint32_t *b) {
	if (!ctx || !scene || scene->ctx != ctx || !ray6 || !a || !t || !b) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: bad argument");
	if (mode == 1 && g >= scene->n_grids) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: grid index out of range");
	RMD_HIP(ctx, hipSetDevice(ctx->device));
	if (n == 0) return RMD_OK;
	DevBuf din, dout;
	RMD_HIP(ctx, hipMalloc(&din.p, n * 6 * sizeof(double)));
	RMD_HIP(ctx, hipMalloc(&dout.p, n * 3 * sizeof(double)));
	RMD_HIP(ctx, hipMemcpyAsync(din.p, ray6, n * 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, rmd::launch_probe_scene(ctx->stream, mode, g, (uint32_t)n, scene->d_objects, scene->n_objects, scene->d_grids,
	                                     scene->n_grids, scene->mask_words_total, scene->axis_pairs, (const double *)din.p, (double *)dout.p));
	std::vector<double> out(n * 3);
	RMD_HIP(ctx, hipMemcpyAsync(out.data(), dout.p, out.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (size_t i = 0; i < n; i++) a[i] = (int32_t)out[3 * i], t[i] = out[3 * i + 1], b[i] = (uint32_t)out[3 * i + 2];
	return RMD_OK;
}

rmd_status rmd_probe_scene_intersect(rmd_context *ctx, const rmd_scene *scene, size_t n, const double *ray6, int32_t *obj, double *t,
                                     uint32_t *sub) {
	return scene_probe(ctx, scene, 0, 0, n, ray6, obj, t, sub);
}
rmd_status rmd_probe_grid_intersect(rmd_context *ctx, const rmd_scene *scene, uint32_t g, size_t n, const double *ray6, int32_t *hit,
                                    double *t, uint32_t *tri) {
	return scene_probe(ctx, scene, 1, g, n, ray6, hit, t, tri);
}

rmd_status rmd_probe_triangle_sphere(size_t n, const double *pos9, double *out5) {
	if (n != 0 && (!pos9 || !out5)) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "probe: bad argument");
	for (size_t i = 0; i < n; i++) {
		double kb = 0.0;
		rmd::triangle_sphere(pos9 + i * 9, out5 + i * 5, kb);
		out5[i * 5 + 4] = kb;
	}
	return RMD_OK;
}

rmd_status rmd_probe_pretest_pairs(rmd_context *ctx, size_t n, const double *sphere5, const double *pos9, const double *ray6, int32_t *pass, int32_t *hit,
                                   double *t) {
	if (n != 0 && (!sphere5 || !pos9 || !ray6 || !pass || !hit || !t)) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: bad argument");
	std::vector<double> out;
	if (rmd_status s = run_probe(ctx, rmd::PROBE_PRETEST_PAIR, n, {{sphere5, 5}, {pos9, 9}, {ray6, 6}}, 3, out)) return s;
	for (size_t i = 0; i < n; i++) pass[i] = out[3 * i] != 0.0, hit[i] = out[3 * i + 1] != 0.0, t[i] = out[3 * i + 2];
	return RMD_OK;
}

rmd_status rmd_probe_trace_samples(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *cam, const rmd_settings *settings,
                                   size_t n, const uint32_t *xy2, const uint32_t *sample, double *rgb_out, int32_t *path_obj,
                                   uint32_t *path_sub) {
	if (!ctx || !scene || scene->ctx != ctx || !cam || !settings || !xy2 || !sample || !rgb_out || (!path_obj != !path_sub))
		return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "probe: bad argument");
	if (settings->bounce_limit > RMD_MAX_BOUNCE_LIMIT) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "probe: bounce_limit above RMD_MAX_BOUNCE_LIMIT");
	RMD_HIP(ctx, hipSetDevice(ctx->device));
	if (n == 0) return RMD_OK;
	std::vector<rmd::ListWork> list(n);
	for (size_t i = 0; i < n; i++) list[i] = rmd::ListWork{xy2[2 * i], xy2[2 * i + 1], sample[i], 0u};
	size_t n_pad = (n + 63) / 64 * 64;
	DevBuf dl, drgb, dpo, dps;
	RMD_HIP(ctx, hipMalloc(&dl.p, n * sizeof(rmd::ListWork)));
	RMD_HIP(ctx, hipMalloc(&drgb.p, n_pad * 3 * sizeof(double)));
	RMD_HIP(ctx, hipMemcpyAsync(dl.p, list.data(), n * sizeof(rmd::ListWork), hipMemcpyHostToDevice, ctx->stream));
	if (path_obj) {
		RMD_HIP(ctx, hipMalloc(&dpo.p, n_pad * RMD_PATH_STRIDE * sizeof(int32_t)));
		RMD_HIP(ctx, hipMalloc(&dps.p, n_pad * RMD_PATH_STRIDE * sizeof(uint32_t)));
		std::vector<int32_t> init(n_pad * RMD_PATH_STRIDE, -2);
		RMD_HIP(ctx, hipMemcpyAsync(dpo.p, init.data(), init.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
		RMD_HIP(ctx, hipMemsetAsync(dps.p, 0, n_pad * RMD_PATH_STRIDE * sizeof(uint32_t), ctx->stream));
		RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	rmd::RenderParams P = rmd::make_params(ctx, scene, cam, settings);
	P.n_work = (uint32_t)n;
	RMD_HIP(ctx, rmd::launch_render_list(ctx->stream, P, scene->d_objects, scene->d_grids, (const rmd::ListWork *)dl.p, (double *)drgb.p,
	                                     (int32_t *)dpo.p, (uint32_t *)dps.p));
	RMD_HIP(ctx, hipMemcpyAsync(rgb_out, drgb.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (path_obj) {
		RMD_HIP(ctx, hipMemcpyAsync(path_obj, dpo.p, n * RMD_PATH_STRIDE * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
		RMD_HIP(ctx, hipMemcpyAsync(path_sub, dps.p, n * RMD_PATH_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	}
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return RMD_OK;
}

} // extern "C"
