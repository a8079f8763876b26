// Streaming ("wavefront") evaluation of the radiance loop for scenes with grids.
//
// The megakernel (kernels.hip) ties a lane to a pixel: when a few lanes of a wave need a grid walk the whole wave
// steps until the longest of those walks ends (~5 of 64 lanes active).  Here a path's state lives in HBM
// (~110 bytes per pixel; MI355X has 288 GB) and each path segment is evaluated in two dense stages:
//
//   wf_step   one thread per path.  Finishes the previous segment with the hit recorded for it — miss / emission /
//             depth cut-off: pixel += T (.) L, next sample's primary ray; otherwise shade, T <- T (.) w, bounce ray —
//             then starts the next segment: closest hit over planes and spheres, bounding-box test of every grid
//             object; a ray that enters a grid's box is appended to the walk queue.
//   wf_walk   one lane per queued ray, densely packed: the wave-cooperative grid walk of grid_walk.hpp, merged into the
//             path's hit record.
//
// Scene::intersect (core/src/scene.rs:54-74) keeps the first object on distance ties, i.e. it returns the lexicographic
// minimum of (distance, object index); the two stages merge their candidates with exactly that rule, so evaluating the
// grid objects after the others changes nothing.  Every path runs its samples in order and adds them to its pixel in
// order, with the same device functions as the megakernel: the frame is bit-identical
// (tests/test_gpu_parity.py::test_wavefront_mode_is_bit_identical).
#define RMD_WITH_HIP 1
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "grid_walk.hpp"
#include "internal.hpp"
#include "launch.hpp"

namespace rmd {

// Path state, structure of arrays over S = n_wave_tiles * 64 slots (slot = wave tile * 64 + lane, lane = pixel as in
// the megakernel).
struct WfState {
	double *ray;      // [6][S] ro.xyz rd.xyz
	double *thr;      // [3][S] throughput T
	double *hit_t;    // [S]
	int32_t *hit_obj; // [S] closest object so far, -1 none
	uint32_t *hit_sub;
	uint32_t *sample; // [S] current sample index
	uint32_t *block;  // [S] rng: index of the sample's next Philox block
	uint32_t *flags;  // [S] bit 0 alive, bit 1 fresh (needs a primary ray), bits 8.. depth
	uint32_t *queue;  // [S] slots that need a grid walk this step
	uint32_t *counters; // [0],[1]: walk queue lengths (alternating per step), [2]: active paths
	uint32_t S;
};

constexpr uint32_t kAlive = 1u, kFresh = 2u;

RMD_DEV bool lex_less(double t, int obj, double t_best, int obj_best) {
	return t < t_best || (t == t_best && obj < obj_best); // Scene::intersect's strict '<' in object order
}

__global__ __launch_bounds__(256) void wf_init(RenderParams P, WfState st, const WaveTile *__restrict__ tiles) {
	const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
	if (slot >= st.S) return;
	const WaveTile t = tiles[slot >> 6];
	const uint32_t lane = slot & 63u, lx = lane & 7u, ly = lane >> 3;
	const bool alive = lx < t.w && ly < t.h && P.sample_count > 0u;
	st.flags[slot] = alive ? (kAlive | kFresh) : 0u;
	st.sample[slot] = P.sample_begin;
	if (alive) atomicAdd(&st.counters[2], 1u);
}

// step_parity selects which walk-queue counter this step fills; the other one is cleared for the next step.
__global__ __launch_bounds__(256) void wf_step(RenderParams P, WfState st, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                                                const WaveTile *__restrict__ tiles, double *__restrict__ out, uint32_t step_parity) {
	extern __shared__ __align__(16) unsigned char smem[];
	DevObject *lobjs = reinterpret_cast<DevObject *>(smem);
	{
		const double *src = reinterpret_cast<const double *>(objs);
		double *dst = reinterpret_cast<double *>(lobjs);
		for (uint32_t i = threadIdx.x; i < P.n_objects * 16u; i += 256u) dst[i] = src[i];
	}
	__syncthreads();
	const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
	if (slot == 0) st.counters[step_parity ^ 1u] = 0u; // the previous step's queue has been consumed by its wf_walk
	if (slot >= st.S) return;
	uint32_t flags = st.flags[slot];
	if (!(flags & kAlive)) return;
	const uint32_t S = st.S;
	const WaveTile tile = tiles[slot >> 6];
	const uint32_t lane = slot & 63u;
	const uint32_t x = tile.x0 + (lane & 7u), y = tile.y0 + (lane >> 3);
	const uint32_t pixel = y * P.W + x;
	const size_t pix = ((size_t)x + (size_t)y * P.W) * 3;
	const V3 cam_pos = ld3(P.cam_pos);

	Rng rng;
	rng.pixel = pixel, rng.sample = st.sample[slot], rng.block = st.block[slot];
	uint32_t s = rng.sample;
	const uint32_t s_end = P.sample_begin + P.sample_count;
	uint32_t depth = flags >> 8;
	V3 ro = mk(st.ray[0 * (size_t)S + slot], st.ray[1 * (size_t)S + slot], st.ray[2 * (size_t)S + slot]);
	V3 rd = mk(st.ray[3 * (size_t)S + slot], st.ray[4 * (size_t)S + slot], st.ray[5 * (size_t)S + slot]);
	V3 T = mk(st.thr[0 * (size_t)S + slot], st.thr[1 * (size_t)S + slot], st.thr[2 * (size_t)S + slot]);
	bool fresh = (flags & kFresh) != 0u;

	if (!fresh) {
		// ---- finish the segment whose closest hit was recorded (src/trace.rs:239-319)
		const int oi = st.hit_obj[slot];
		const double t = st.hit_t[slot];
		const uint32_t sub = st.hit_sub[slot];
		bool terminal = false;
		V3 L = mk(0.0, 0.0, 0.0);
		if (oi < 0) {
			terminal = true; // :242
		} else {
			const DevObject &o = lobjs[oi];
			const V3 frag = ro + rd * t; // :246
			if (o.material_kind == 2u) {
				L = ld3(o.color); // :250-252
				terminal = true;
			} else {
				V3 normal;
				if (o.geometry_kind == 0u) normal = ld3(o.normal);
				else if (o.geometry_kind == 1u) normal = normalize(frag - ld3(o.origin));
				else {
					const DevGrid &g = grids[o.grid_index];
					normal = triangle_normal(as_global(g.tri_pos) + (size_t)sub * 9, as_global(g.tri_nrm) + (size_t)sub * 9, as_global(g.tri_aux) + (size_t)sub * 4, frag);
				}
				shade(P, normal, frag, ld3(o.color), o.roughness, o.metalness, cam_pos, rng, ro, rd, T); // T <- T (.) weight, bounce ray
				depth++;
				if (depth > P.bounce_limit) terminal = true; // :235-237
			}
		}
		if (terminal) {
			L = hadamard(T, L);
			out[pix + 0] = out[pix + 0] + L.x, out[pix + 1] = out[pix + 1] + L.y, out[pix + 2] = out[pix + 2] + L.z; // :203
			s++;
			fresh = true;
		}
	}
	// ---- primary ray(s): a sample that ends before its first intersection (bounce_limit 0, failed DoF) is consumed here
	while (fresh && s != s_end) {
		rng.init(pixel, s);
		bool ok = true;
		if (P.use_dof) {
			ok = primary_ray_dof(P, x, y, rng, ro, rd);
		} else {
			double u0, u1;
			rng.next2(P.key0, P.key1, u0, u1);
			primary_ray(P, x, y, u0, u1, ro, rd);
		}
		depth = 1;
		T = mk(1.0, 1.0, 1.0);
		if (!ok || P.bounce_limit == 0u) {
			s++; // contributes exactly zero: pixel + 0 = pixel
			continue;
		}
		fresh = false;
	}
	if (fresh) { // all samples done
		st.flags[slot] = 0u;
		atomicSub(&st.counters[2], 1u);
		return;
	}
	// ---- start the segment: planes and spheres now, grids through the walk queue
	double best_t = kFMax;
	int best = -1;
	bool walk = false;
	for (uint32_t i = 0; i < P.n_objects; i++) {
		const DevObject &o = objs[i];
		double t = 0.0;
		bool hit = false;
		if (o.geometry_kind == 0u) hit = plane_intersect(ld3(o.origin), ld3(o.normal), ro, rd, t);
		else if (o.geometry_kind == 1u) hit = sphere_intersect(ld3(o.origin), o.radius, ro, rd, t);
		else {
			const DevGrid &g = grids[o.grid_index];
			double t_outer;
			walk = walk || aabb_intersect(ld3(g.bbox_min), ld3(g.bbox_max), ro, rd, t_outer); // acc_grid.rs:90
		}
		if (hit && lex_less(t, (int)i, best_t, best)) best_t = t, best = (int)i;
	}
	st.hit_t[slot] = best_t, st.hit_obj[slot] = best, st.hit_sub[slot] = 0u;
	st.ray[0 * (size_t)S + slot] = ro.x, st.ray[1 * (size_t)S + slot] = ro.y, st.ray[2 * (size_t)S + slot] = ro.z;
	st.ray[3 * (size_t)S + slot] = rd.x, st.ray[4 * (size_t)S + slot] = rd.y, st.ray[5 * (size_t)S + slot] = rd.z;
	st.thr[0 * (size_t)S + slot] = T.x, st.thr[1 * (size_t)S + slot] = T.y, st.thr[2 * (size_t)S + slot] = T.z;
	st.sample[slot] = s, st.block[slot] = rng.block;
	st.flags[slot] = kAlive | (depth << 8);
	if (walk) st.queue[atomicAdd(&st.counters[step_parity], 1u)] = slot;
}

// Persistent workgroups: each stages the grids' occupancy masks into LDS once, then its waves take 64-entry chunks of
// the walk queue until it is exhausted.
__global__ __launch_bounds__(256) void wf_walk(RenderParams P, WfState st, const DevObject *__restrict__ objs, const DevGrid *__restrict__ grids,
                                                uint32_t step_parity) {
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t *lmasks = reinterpret_cast<uint32_t *>(smem);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, waves_per_wg = blockDim.x >> 6;
	WalkScratch &scr = *reinterpret_cast<WalkScratch *>(smem + (size_t)((P.mask_words_total + 3u) & ~3u) * 4u + (size_t)wave * sizeof(WalkScratch));
	const uint32_t count = st.counters[step_parity];
	if (blockIdx.x * waves_per_wg * 64u >= count) return; // nothing for this workgroup (uniform)
	for (uint32_t gi = 0; gi < P.n_grids; gi++) {
		const DevGrid &g = grids[gi];
		if (g.mask_lds_word == 0xFFFFFFFFu) continue;
		for (uint32_t i = tid; i < g.mask_n_words; i += blockDim.x) lmasks[g.mask_lds_word + i] = as_global(g.mask_words)[i];
	}
	__syncthreads();
	const uint32_t S = st.S;
	for (uint32_t chunk = blockIdx.x * waves_per_wg + wave; chunk * 64u < count; chunk += gridDim.x * waves_per_wg) {
		const uint32_t q = chunk * 64u + lane;
		const bool valid = q < count;
		const uint32_t slot = st.queue[valid ? q : 0u];
		const V3 ro = mk(st.ray[0 * (size_t)S + slot], st.ray[1 * (size_t)S + slot], st.ray[2 * (size_t)S + slot]);
		const V3 rd = mk(st.ray[3 * (size_t)S + slot], st.ray[4 * (size_t)S + slot], st.ray[5 * (size_t)S + slot]);
		double best_t = st.hit_t[slot];
		int best = st.hit_obj[slot];
		uint32_t best_sub = 0;
		for (uint32_t i = 0; i < P.n_objects; i++) { // uniform: the grid objects of the scene, in order
			const DevObject &o = objs[i];
			if (o.geometry_kind != 2u) continue;
			const DevGrid &g = grids[o.grid_index];
			const uint32_t *mask = g.mask_lds_word != 0xFFFFFFFFu ? lmasks + g.mask_lds_word : nullptr;
			bool hit = false;
			double t = 0.0;
			uint32_t tri = 0;
			grid_intersect_wave(g, mask, scr, valid, ro, rd, hit, t, tri);
			if (valid && hit && lex_less(t, (int)i, best_t, best)) best_t = t, best = (int)i, best_sub = tri;
		}
		if (valid && best != st.hit_obj[slot]) st.hit_t[slot] = best_t, st.hit_obj[slot] = best, st.hit_sub[slot] = best_sub;
	}
}

// ---------------------------------------------------------------- host driver
size_t wavefront_workspace_bytes(uint32_t n_wave_tiles) {
	const size_t S = (size_t)n_wave_tiles * 64u;
	return S * (6 + 3 + 1) * sizeof(double) + S * (1 + 1 + 1 + 1 + 1 + 1) * sizeof(uint32_t) + 64;
}

hipError_t launch_wavefront(hipStream_t stream, const RenderParams &P, const DevObject *objs, const DevGrid *grids, const WaveTile *wave_tiles,
                            double *accum, void *workspace, uint32_t n_cus) {
	if (P.n_work == 0 || P.sample_count == 0) return hipSuccess;
	const size_t S = (size_t)P.n_work * 64u;
	WfState st;
	unsigned char *p = static_cast<unsigned char *>(workspace);
	auto take = [&](size_t bytes) {
		void *r = p;
		p += bytes;
		return r;
	};
	st.ray = (double *)take(S * 6 * sizeof(double));
	st.thr = (double *)take(S * 3 * sizeof(double));
	st.hit_t = (double *)take(S * sizeof(double));
	st.hit_obj = (int32_t *)take(S * 4), st.hit_sub = (uint32_t *)take(S * 4), st.sample = (uint32_t *)take(S * 4), st.block = (uint32_t *)take(S * 4);
	st.flags = (uint32_t *)take(S * 4), st.queue = (uint32_t *)take(S * 4);
	st.counters = (uint32_t *)take(64);
	st.S = (uint32_t)S;
	hipError_t e = hipMemsetAsync(st.counters, 0, 64, stream);
	if (e != hipSuccess) return e;
	const unsigned blocks = (unsigned)((S + 255) / 256);
	hipLaunchKernelGGL(wf_init, dim3(blocks), dim3(256), 0, stream, P, st, wave_tiles);
	const size_t step_lds = (size_t)P.n_objects * sizeof(DevObject);
	const size_t walk_lds = (size_t)((P.mask_words_total + 3u) & ~3u) * 4u + 4u * sizeof(WalkScratch);
	if (walk_lds > 64u * 1024u) {
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wf_walk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes);
		if (e != hipSuccess) return e;
	}
	const unsigned walk_blocks = n_cus * 4u; // 16 waves per CU, as in the megakernel's grid instantiation
	uint32_t *h_active = nullptr;
	e = hipHostMalloc((void **)&h_active, sizeof(uint32_t), hipHostMallocDefault);
	if (e != hipSuccess) return e;
	*h_active = 1;
	// Upper bound on the number of steps: every path needs sample_count * (bounce_limit + 1) + 1 at most.
	const uint64_t max_steps = (uint64_t)P.sample_count * (P.bounce_limit + 1u) + 2u;
	for (uint64_t step = 0; step < max_steps; step++) {
		const uint32_t parity = (uint32_t)(step & 1u);
		hipLaunchKernelGGL(wf_step, dim3(blocks), dim3(256), step_lds, stream, P, st, objs, grids, wave_tiles, accum, parity);
		hipLaunchKernelGGL(wf_walk, dim3(walk_blocks), dim3(256), walk_lds, stream, P, st, objs, grids, parity);
		if ((step & 31u) == 31u) { // every 32 steps: has every path finished?
			e = hipMemcpyAsync(h_active, st.counters + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
			if (e == hipSuccess) e = hipStreamSynchronize(stream);
			if (e != hipSuccess || *h_active == 0u) break;
		}
	}
	if (e == hipSuccess) e = hipGetLastError();
	(void)hipHostFree(h_active);
	return e;
}

} // namespace rmd
