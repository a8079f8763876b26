// C-ABI implementation (include/raymond_hip.h): contexts, scene upload, the render entry points,
// framebuffer helpers and the resolve/tone-map epilogue.  Host code only; kernels are in kernels.hip.
#define RMD_WITH_HIP 1
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "internal.hpp"
#include "launch.hpp"

namespace {
thread_local std::string tl_last_error;

#define RMD_HIP(ctx, call)                                                                                      \
	do {                                                                                                        \
		hipError_t e_ = (call);                                                                                 \
		if (e_ != hipSuccess) return rmd::fail(ctx, e_ == hipErrorOutOfMemory ? RMD_ERR_OUT_OF_MEMORY : RMD_ERR_HIP, \
		                                       std::string(#call) + ": " + hipGetErrorString(e_));              \
	} while (0)

rmd_status bind(rmd_context *ctx) {
	if (!ctx) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "null context");
	RMD_HIP(ctx, hipSetDevice(ctx->device));
	return RMD_OK;
}

rmd_status context_create(int32_t device, hipStream_t stream, bool own_stream, rmd_context **out) {
	if (!out) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_context_create: null out pointer");
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return rmd::fail(nullptr, RMD_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
	if (device < 0 || device >= count) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_context_create: device ordinal out of range");
	RMD_HIP(nullptr, hipSetDevice(device));
	hipDeviceProp_t prop;
	RMD_HIP(nullptr, hipGetDeviceProperties(&prop, device));
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return rmd::fail(nullptr, RMD_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
	rmd_context *ctx = new (std::nothrow) rmd_context();
	if (!ctx) return rmd::fail(nullptr, RMD_ERR_OUT_OF_MEMORY, "rmd_context_create: allocation failed");
	ctx->device = device;
	ctx->n_cus = (uint32_t)prop.multiProcessorCount;
	ctx->wave_slots = (uint32_t)prop.multiProcessorCount * 16u; // both render kernels fit 4 waves per SIMD (<= 128 VGPRs)
	ctx->hbm_bytes = (size_t)prop.totalGlobalMem;
	if (own_stream) {
		hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
		if (se != hipSuccess) {
			delete ctx;
			return rmd::fail(nullptr, RMD_ERR_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(se));
		}
		ctx->owns_stream = true;
	} else {
		ctx->stream = stream;
	}
	// environment hooks are read here, once; rmd_context_set_tunable changes them afterwards
	{
		auto env_int = [](const char *name) -> int64_t {
			const char *v = std::getenv(name);
			return v ? (int64_t)std::atoll(v) : 0;
		};
		ctx->tunable[RMD_TUNE_SAMPLE_SPLIT] = env_int("RMD_SAMPLE_SPLIT");
		ctx->tunable[RMD_TUNE_WALK_BATCH] = env_int("RMD_WALK_BATCH");
		ctx->tunable[RMD_TUNE_MASK_BUDGET] = env_int("RMD_MASK_BUDGET");
		ctx->tunable[RMD_TUNE_SCRATCH_CAP_MB] = env_int("RMD_SCRATCH_CAP_MB");
		ctx->tunable[RMD_TUNE_WALK_CUT] = env_int("RMD_WALK_CUT");
		ctx->tunable[RMD_TUNE_SPLIT_MIN_SAMPLES] = env_int("RMD_SPLIT_MIN_SAMPLES");
		ctx->tunable[RMD_TUNE_CHAIN_ITEMS] = env_int("RMD_CHAIN_ITEMS");
		ctx->tunable[RMD_TUNE_AXIS_PAIRS] = env_int("RMD_AXIS_PAIRS");
		ctx->tunable[RMD_TUNE_PATH_QUEUES] = env_int("RMD_PATH_QUEUES");
		const char *form = std::getenv("RMD_LAUNCH_FORM");
		ctx->tunable[RMD_TUNE_LAUNCH_FORM] = !form ? 0 : (std::strcmp(form, "per-item") == 0 || std::strcmp(form, "1") == 0) ? 1 : (std::strcmp(form, "persistent") == 0 || std::strcmp(form, "2") == 0) ? 2 : 0;
#if RMD_DIAG
		ctx->debug_flags = (uint32_t)env_int("RMD_DEBUG"); // DIAG builds only: 1 | 2 are timing ablations that change results, 8 | 16 count events
#endif
	}
	if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) {
		rmd_context_destroy(ctx);
		return rmd::fail(nullptr, RMD_ERR_HIP, "hipEventCreate failed");
	}
	// the fault words: pinned host memory the device writes through (coherent: visible to the host once the launch has completed)
	{
		void *h = nullptr, *d = nullptr;
		hipError_t fe = hipHostMalloc(&h, rmd::kFaultWords * sizeof(uint32_t), hipHostMallocMapped);
		if (fe == hipSuccess) {
			ctx->h_fault = (uint32_t *)h;
			std::memset(h, 0, rmd::kFaultWords * sizeof(uint32_t));
			fe = hipHostGetDevicePointer(&d, h, 0);
		}
		if (fe != hipSuccess) {
			rmd_context_destroy(ctx);
			return rmd::fail(nullptr, RMD_ERR_HIP, std::string("fault words (hipHostMalloc): ") + hipGetErrorString(fe));
		}
		ctx->d_fault = (uint32_t *)d;
	}
	*out = ctx;
	return RMD_OK;
}

// Splits host tiles (core::tile::Tile rectangles) into 8x8 wave tiles; cached per context while the
// rect list stays the same (a progressive render re-submits the same tiles every pass).
rmd_status prepare_wave_tiles(rmd_context *ctx, const rmd_camera *cam, const rmd_tile_rect *tiles, uint32_t n_tiles) {
	const uint32_t W = cam->backbuffer_width, H = cam->backbuffer_height;
	if (ctx->d_wave_tiles && ctx->cached_W == W && ctx->cached_H == H && ctx->cached_rects.size() == n_tiles &&
	    (n_tiles == 0 || std::memcmp(ctx->cached_rects.data(), tiles, sizeof(rmd_tile_rect) * n_tiles) == 0))
		return RMD_OK;
	std::vector<rmd::WaveTile> wt;
	for (uint32_t i = 0; i < n_tiles; i++) {
		const rmd_tile_rect &r = tiles[i];
		if (r.width == 0 || r.height == 0) continue;
		if ((uint64_t)r.left + r.width > W || (uint64_t)r.top + r.height > H)
			return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_render_tiles: tile rectangle outside the backbuffer");
		for (uint32_t y = r.top; y < r.top + r.height; y += 8)
			for (uint32_t x = r.left; x < r.left + r.width; x += 8) {
				rmd::WaveTile t;
				t.x0 = (uint16_t)x, t.y0 = (uint16_t)y;
				t.w = (uint8_t)((r.left + r.width - x) < 8u ? (r.left + r.width - x) : 8u);
				t.h = (uint8_t)((r.top + r.height - y) < 8u ? (r.top + r.height - y) : 8u);
				t._pad = 0;
				wt.push_back(t);
			}
	}
	if (wt.size() > ctx->wave_tiles_capacity) {
		if (ctx->d_wave_tiles) RMD_HIP(ctx, hipFree(ctx->d_wave_tiles));
		ctx->d_wave_tiles = nullptr, ctx->wave_tiles_capacity = 0;
		RMD_HIP(ctx, hipMalloc((void **)&ctx->d_wave_tiles, wt.size() * sizeof(rmd::WaveTile)));
		ctx->wave_tiles_capacity = wt.size();
	}
	if (!wt.empty()) {
		RMD_HIP(ctx, hipMemcpyAsync(ctx->d_wave_tiles, wt.data(), wt.size() * sizeof(rmd::WaveTile), hipMemcpyHostToDevice, ctx->stream));
		RMD_HIP(ctx, hipStreamSynchronize(ctx->stream)); // wt is a stack-owned staging vector
	}
	ctx->n_wave_tiles = (uint32_t)wt.size();
	ctx->cached_rects.assign(tiles, tiles + n_tiles);
	ctx->cached_W = W, ctx->cached_H = H;
	return RMD_OK;
}

rmd_status check_render_args(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *cam, const rmd_settings *st) {
	if (!scene || !cam || !st) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: null scene/camera/settings");
	if (scene->ctx != ctx) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: scene belongs to another context");
	if (cam->backbuffer_width == 0 || cam->backbuffer_height == 0 || cam->backbuffer_width > 65535u || cam->backbuffer_height > 65535u)
		return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: backbuffer size must be 1..65535 per axis");
	if (st->bounce_limit > RMD_MAX_BOUNCE_LIMIT) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "render: bounce_limit above RMD_MAX_BOUNCE_LIMIT");
	if ((uint64_t)st->sample_begin + st->sample_count > 0xFFFFFFFFull) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: sample range overflows u32");
	if (st->flags & ~(RMD_RENDER_DOF | RMD_RENDER_TRACE_BLACK_PATHS | RMD_RENDER_END_BLACK_PATHS)) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: unknown bits in rmd_settings.flags");
	if ((st->flags & RMD_RENDER_TRACE_BLACK_PATHS) && (st->flags & RMD_RENDER_END_BLACK_PATHS))
		return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "render: RMD_RENDER_TRACE_BLACK_PATHS and RMD_RENDER_END_BLACK_PATHS exclude each other");
	if (rmd::render_lds_bytes(scene->n_objects, scene->mask_words_total, 1) + (scene->n_grids == 0u ? rmd::kSortPoolBytes : 0u) > rmd::kLdsBudgetBytes) // (one wave's area, head included)
		return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "render: object table + grid masks exceed the 160 KiB LDS of a CU");
	return RMD_OK;
}

} // namespace

namespace rmd {

rmd_status fail(rmd_context *ctx, rmd_status status, const std::string &text) {
	if (ctx) ctx->last_error = text;
	tl_last_error = text;
	return status;
}

rmd_status fail_noexcept(rmd_context *ctx, rmd_status status, const char *what, const char *text) noexcept {
	try {
		return fail(ctx, status, std::string(what) + ": " + text);
	} catch (...) { // not even the message: the status alone
		try {
			if (ctx) ctx->last_error.clear();
			tl_last_error.clear();
		} catch (...) {
		}
		return status;
	}
}

static_assert(RMD_MAX_BOUNCE_LIMIT_DEV == RMD_MAX_BOUNCE_LIMIT, "launch.hpp mirrors include/raymond_hip.h");

// Called after every wait for the context's stream.  The words are host memory: two loads when nothing happened.
rmd_status check_fault(rmd_context *ctx) {
	if (!ctx || !ctx->h_fault) return RMD_OK;
	volatile uint32_t *f = ctx->h_fault;
	const uint32_t code = f[0];
	if (code == 0u) return RMD_OK;
	const uint32_t item = f[1], reports = f[2];
	f[0] = 0u, f[1] = 0u, f[2] = 0u; // the next launch starts clean
	std::string what;
	if (code & kFaultTripLoop) what += " trip loop of render_wave;";
	if (code & kFaultSortedTripLoop) what += " trip loop of render_wave_sorted;";
	if (code & kFaultWorkLoop) what += " work loop of a persistent workgroup;";
	if (code & kFaultWalkRounds) what += " round loop of a grid walk;";
	if (code & kFaultQueuedTripLoop) what += " trip loop of render_wave_queued;";
	if (code & ~(kFaultTripLoop | kFaultSortedTripLoop | kFaultWorkLoop | kFaultWalkRounds | kFaultQueuedTripLoop)) what += " unknown code;";
	char num[96];
	std::snprintf(num, sizeof(num), " %u wave(s) reported, the last one at work item %u (code 0x%x)", reports, item, code);
	return fail(ctx, RMD_ERR_DEVICE_FAULT,
	            "device fault: a loop of the render kernel ran past its bound —" + what + num +
	                "; the launch was cut short and the framebuffer it wrote to is not valid");
}

// generate_primary_ray's loop-invariant terms (src/trace.rs:323-330), evaluated with the host libm
RenderParams make_params(const rmd_context *ctx, const rmd_scene *scene, const rmd_camera *cam, const rmd_settings *st) {
	const double PI = 3.14159265358979323846;
	RenderParams P;
	std::memset(&P, 0, sizeof(P));
	for (int a = 0; a < 3; a++) P.cam_pos[a] = cam->position[a];
	P.width = (double)cam->backbuffer_width;
	P.height = (double)cam->backbuffer_height;
	P.inv_width = rmd::exact_reciprocal(P.width), P.inv_height = rmd::exact_reciprocal(P.height);
	P.aspect = P.width / P.height;
	P.tan_half_fov = std::tan(cam->fov_vert / 2.0 * PI / 180.0);
	P.focal_length = cam->focal_length;
	P.aperture_radius = cam->aperture_radius;
	P.W = cam->backbuffer_width, P.H = cam->backbuffer_height;
	P.bounce_limit = st->bounce_limit;
	P.sample_begin = st->sample_begin, P.sample_count = st->sample_count;
	P.n_objects = scene ? scene->n_objects : 0;
	P.n_grids = scene ? scene->n_grids : 0;
	P.mask_words_total = scene ? scene->mask_words_total : 0;
	P.key0 = (uint32_t)st->seed, P.key1 = (uint32_t)(st->seed >> 32);
	P.use_dof = ((st->flags & RMD_RENDER_DOF) && cam->aperture_radius > 0.0) ? 1u : 0u; // off by default, as in the reference's loop (:199)
	// flags 0 = reference-identical: zero-throughput paths are ended only where that provably changes no sample — a scene without grids
	// (its one NaN source, the interpolated normal of a mesh hit, does not exist there); END opts in for scenes with grids, TRACE never ends
	// — AND whose parameters are all inside the class for which that is proved (rmd_scene::regular: an Emission of (inf, 0, 0), a NaN colour, a
	// material of roughness 0 or a sphere of radius 0 make non-finite radiance reachable without a mesh; such a scene is traced like one with a grid)
	P.end_black_paths = (st->flags & RMD_RENDER_TRACE_BLACK_PATHS) ? 0u : ((st->flags & RMD_RENDER_END_BLACK_PATHS) || !scene || (scene->n_grids == 0u && scene->regular)) ? 1u : 0u;
	P.shade_last_depth = (scene && !scene->regular) ? 1u : 0u;
	P.axis_pairs = scene ? scene->axis_pairs : 0u;
	P.visit_mask = scene ? scene->visit_mask : ~0ull, P.grid_mask = scene ? scene->grid_mask : ~0ull;
	P.fault = ctx ? ctx->d_fault : nullptr;
	P.walk_batch = rmd::kWalkBatchDefault;
	if (ctx && ctx->tunable[RMD_TUNE_WALK_BATCH] > 0) P.walk_batch = (uint32_t)ctx->tunable[RMD_TUNE_WALK_BATCH]; // any value gives the same image
	// walks put aside for the wave's next walk call (grid_walk.hpp): the carried state is ONE walk's, so only in scenes with one grid object.
	// Low byte: a round's stepping ends under its last K lanes; next byte: a call ends when at most 2K lanes are still walking.
	{
		uint32_t k = rmd::kWalkCutDefault;
		if (ctx && ctx->tunable[RMD_TUNE_WALK_CUT] > 0) k = (uint32_t)std::min<int64_t>(ctx->tunable[RMD_TUNE_WALK_CUT] - 1, 31); // any value gives the same image
		P.walk_cut = (scene && scene->n_grid_objects == 1u) ? (k | (2u * k) << 8) : 0u;
		P.walk_steps_bound = scene ? scene->walk_steps_bound : 0u;
	}
#if RMD_DIAG
	if (ctx) P.debug_flags = ctx->debug_flags;
#endif
	return P;
}

} // namespace rmd

extern "C" {

uint32_t rmd_abi_version(void) { return RMD_ABI_VERSION; }

rmd_status rmd_context_create(int32_t device_ordinal, rmd_context **out) { return context_create(device_ordinal, nullptr, true, out); }

rmd_status rmd_context_create_on_stream(int32_t device_ordinal, void *hip_stream, rmd_context **out) {
	return context_create(device_ordinal, (hipStream_t)hip_stream, false, out);
}

void rmd_context_destroy(rmd_context *ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream || !ctx->owns_stream) (void)hipStreamSynchronize(ctx->stream);
	if (ctx->d_wave_tiles) (void)hipFree(ctx->d_wave_tiles);
	if (ctx->d_sample_buf) (void)hipFree(ctx->d_sample_buf);
	if (ctx->d_debug_counters) (void)hipFree(ctx->d_debug_counters);
	if (ctx->d_work_counter) (void)hipFree(ctx->d_work_counter);
	if (ctx->d_queue_buf) (void)hipFree(ctx->d_queue_buf);
	if (ctx->d_tile_done) (void)hipFree(ctx->d_tile_done);
	if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
	if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
	for (auto &sl : ctx->transfer) {
		if (sl.d_packed) (void)hipFree(sl.d_packed);
		if (sl.d_table) (void)hipFree(sl.d_table);
		if (sl.packed_ready) (void)hipEventDestroy(sl.packed_ready);
		if (sl.copied) (void)hipEventDestroy(sl.copied);
	}
	if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
	if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
	if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
	if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

const char *rmd_last_error(const rmd_context *ctx) { return ctx ? ctx->last_error.c_str() : tl_last_error.c_str(); }

// What rmd_scene_create derives from a grid description before it uploads it (device_types.hpp: DevGrid): validated tables, the triangle
// records, the per-cell index lists with their entries, the Heron constants.  ~50 ms of host time for the 99k-triangle benchmark mesh — a host
// scheduler creates a scene per render_tiled call and per GPU (the counterpart of the reference's per-worker `scene.clone()`, src/trace.rs:182-185),
// so a description that comes from a rmd_grid_build (rmd_grid_desc::built) keeps them in that object: derived once, shared by every upload.
struct GridDerived {
	std::vector<unsigned char> recs;
	std::vector<rmd::CellEntry> entries;
	std::vector<uint32_t> ids;
	std::vector<double> aux;
	std::vector<double> spheres; // per triangle: centre and squared radius of a sphere around it, inflated (device_types.hpp: DevGrid::tri_sph)
	double sphere_kb = 0.0;      // ... and the grid's distance-proportional allowance (DevGrid::sph_kb)
};
static rmd_status derive_grid_tables(rmd_context *ctx, const rmd_grid_desc &g, GridDerived &out) {
	{
		if (!g.cells || !g.mapping_table || !g.tri_pos || !g.tri_nrm || g.n_tris == 0 ||
		    g.n_cells != (uint64_t)g.resolution[0] * g.resolution[1] * g.resolution[2]) {
			return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: inconsistent grid description");
		}
		// validate the CSR-like table so that the kernel's gathers stay in bounds
		for (uint64_t c = 0; c < g.n_cells; c++) {
			uint64_t off = g.cells[c];
			if (off >= g.n_mapping || off + (uint64_t)g.mapping_table[off] >= g.n_mapping) {
				return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: cells/mapping_table run out of range");
			}
			uint32_t cnt = g.mapping_table[off];
			for (uint32_t k = 1; k <= cnt; k++)
				if (g.mapping_table[off + k] >= g.n_tris) {
					return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: triangle index out of range in mapping_table");
				}
		}
		// the kernel keeps the reference's linear cell index x + res.x*(y + z*res.z) in 32 bits
		if ((uint64_t)g.resolution[0] * ((uint64_t)g.resolution[1] + (uint64_t)g.resolution[2] * g.resolution[2]) >= (1ull << 31) ||
		    g.n_cells > (1ull << 31)) {
			return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: grid resolution too large for 31-bit cell indices");
		}
		// triangle records, per-cell lists of triangle indices and the cell entries (device_types.hpp): record = v0, edge1 = v1 - v0,
		// edge2 = v2 - v0 (triangle.rs:16-17, same subtraction, done once); entry slot 0 = the cell's list in mapping_table order, slots 1..6 =
		// that list without the triangles the cell at index c - delta_s lists too (the cell a walk came from: tested already, missed)
		const uint64_t n_refs = g.n_mapping - g.n_cells;
		if (n_refs > 0xFFFFFFFFull) {
			return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: more than 2^32-1 cell->triangle references");
		}
		std::vector<unsigned char> &recs = out.recs;
		recs.assign((size_t)g.n_tris * rmd::kTriRecStride, 0);
		for (uint64_t ti = 0; ti < g.n_tris; ti++) {
			const double *p = g.tri_pos + (size_t)ti * 9;
			double q[9];
			for (int a = 0; a < 3; a++) q[a] = p[a], q[3 + a] = p[3 + a] - p[a], q[6 + a] = p[6 + a] - p[a];
			std::memcpy(recs.data() + (size_t)ti * rmd::kTriRecStride, q, sizeof(q));
		}
		std::vector<rmd::CellEntry> &entries = out.entries;
		entries.assign((size_t)g.n_cells * rmd::kEntrySlots, rmd::CellEntry{0u, 0u});
		std::vector<uint32_t> &ids = out.ids;
		ids.clear();
		ids.reserve((size_t)n_refs * 3);
		{
			uint64_t refs_seen = 0;
			for (uint64_t c = 0; c < g.n_cells; c++) { // slot 0: the full lists, validated against overlapping runs
				const uint32_t off = g.cells[c], cnt = g.mapping_table[off];
				refs_seen += cnt;
				if (refs_seen > n_refs) { // cells sharing a run: the tables are not the builder's; refuse rather than overflow
							return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: cells/mapping_table runs overlap");
				}
				entries[c * rmd::kEntrySlots] = rmd::CellEntry{(uint32_t)ids.size(), cnt};
				ids.insert(ids.end(), g.mapping_table + off + 1, g.mapping_table + off + 1 + cnt);
			}
			const int64_t sx = (int64_t)g.resolution[0], sxz = (int64_t)g.resolution[0] * (int64_t)g.resolution[2]; // Q5: res.z where res.y is meant
			const int64_t delta[7] = {0, 1, -1, sx, -sx, sxz, -sxz};
			std::vector<uint32_t> fresh;
			std::vector<uint64_t> listed_by((size_t)g.n_tris, ~0ull); // triangle -> the (cell, slot) pass that last saw it in a predecessor's list
			for (uint64_t c = 0; c < g.n_cells; c++) {
				const rmd::CellEntry full = entries[c * rmd::kEntrySlots];
				const uint32_t *mine = g.mapping_table + g.cells[c] + 1;
				for (uint32_t s = 1; s < rmd::kEntrySlots; s++) {
					rmd::CellEntry &e = entries[c * rmd::kEntrySlots + s];
					e = full;
					if (s == 7u || full.count == 0u) continue;
					const int64_t prev = (int64_t)c - delta[s];
					if (prev < 0 || prev >= (int64_t)g.n_cells || prev == (int64_t)c) continue; // no such predecessor: never asked for
					const uint32_t poff = g.cells[prev], pcnt = g.mapping_table[poff];
					if (pcnt == 0u) continue;
					const uint32_t *theirs = g.mapping_table + poff + 1;
					fresh.clear();
					const uint64_t pass = c * rmd::kEntrySlots + s;
					for (uint32_t j = 0; j < pcnt; j++) listed_by[theirs[j]] = pass;
					for (uint32_t k = 0; k < full.count; k++)
						if (listed_by[mine[k]] != pass) fresh.push_back(mine[k]);
					if (fresh.size() == full.count) continue; // nothing to leave out: shares the full list
					e = rmd::CellEntry{(uint32_t)ids.size(), (uint32_t)fresh.size()};
					ids.insert(ids.end(), fresh.begin(), fresh.end());
					if (ids.size() > 0xFFFFFFFFull) {
									return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: triangle index lists exceed 2^32 entries");
					}
				}
			}
		}
		out.aux.assign((size_t)g.n_tris * 4, 0.0);
		for (uint64_t ti = 0; ti < g.n_tris; ti++) rmd::triangle_aux(g.tri_pos + (size_t)ti * 9, out.aux.data() + (size_t)ti * 4);
		out.spheres.assign((size_t)g.n_tris * 4, 0.0);
		out.sphere_kb = 0.0;
		for (uint64_t ti = 0; ti < g.n_tris; ti++) {
			double kb = 0.0;
			rmd::triangle_sphere(g.tri_pos + (size_t)ti * 9, out.spheres.data() + (size_t)ti * 4, kb);
			out.sphere_kb = std::max(out.sphere_kb, kb);
		}
	}
	return RMD_OK;
}

// (`sc`: the scene under construction, owned by the caller's frame so that an exception — std::bad_alloc from one of the host-side tables: a
// 256^3 grid needs a gigabyte for its cell entries alone — can still release what has been uploaded)
static rmd_status scene_create_impl(rmd_context *ctx, const rmd_object *objects, uint32_t n_objects, const rmd_grid_desc *grids, uint32_t n_grids,
                                    rmd_scene **out, rmd_scene *&sc) {
	if (rmd_status s = bind(ctx)) return s;
	if (!out || (n_objects && !objects) || (n_grids && !grids)) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: null argument");
	*out = nullptr;
	std::vector<rmd::DevObject> hobj(n_objects);
	bool regular = true;
	for (uint32_t i = 0; i < n_objects; i++) {
		const rmd_object &o = objects[i];
		rmd::DevObject &d = hobj[i];
		std::memset(&d, 0, sizeof(d));
		if (o.geometry_kind > RMD_GEOM_GRID) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: unknown geometry kind");
		if (o.material.kind > RMD_MAT_EMISSION) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: unknown material kind");
		if (o.geometry_kind == RMD_GEOM_GRID && o.grid_index >= n_grids) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_scene_create: grid_index out of range");
		d.geometry_kind = o.geometry_kind, d.grid_index = o.grid_index, d.material_kind = o.material.kind;
		for (int a = 0; a < 3; a++) d.origin[a] = o.origin[a], d.normal[a] = o.normal[a], d.color[a] = o.material.color[a];
		d.radius = o.radius;
		// the GGX angle roughness^2 * sqrt(u / (1 - u)) must stay inside sincos_cw's reduction range (device_core.hpp)
		if (o.material.kind != RMD_MAT_EMISSION && std::fabs(o.material.roughness) > rmd::kMaxRoughness)
			return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: |roughness| > 512 is not supported");
		// (the bounce weight is formed as ONE quotient whose denominator holds (roughness^2 / 8)^2 for a surface seen from behind: it must not underflow)
		if (o.material.kind != RMD_MAT_EMISSION && o.material.roughness != 0.0 && std::fabs(o.material.roughness) < rmd::kMinRoughness)
			return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: 0 < |roughness| < 1e-12 is not supported");
		d.roughness = o.material.roughness;
		d.metalness = o.material.kind == RMD_MAT_METAL ? 1.0 : 0.0; // src/trace.rs:248-249
		// Is this object inside the class of parameters for which "a path whose throughput is exactly (0, 0, 0) contributes exactly (0, 0, 0)" is
		// proved?  The proof needs every later vertex of such a path to have a FINITE radiance (0 x NaN = NaN, src/trace.rs:
		// 281-282, :315-318).  Outside the class — an Emission of (inf, 0, 0), a NaN or infinite colour / position / normal, coordinates so large that a
		// hit point overflows, a sphere of radius 0 (normalize(P - centre) of a hit at the centre), a material of roughness 0 (geometry_schlick_ggx is
		// 0 / 0 for a surface seen from behind, :372-378) — the reference's sample can be NaN without any mesh, so flags 0 traces such a scene's
		// black paths on, as it does a scene with a grid (make_params).
		{
			auto tame = [](double v) { return std::fabs(v) <= 1e150; }; // finite, and no sum or product of two of them overflows
			bool ok = true;
			if (o.geometry_kind != RMD_GEOM_GRID)
				for (int a = 0; a < 3; a++) ok = ok && tame(o.origin[a]);
			if (o.geometry_kind == RMD_GEOM_PLANE)
				for (int a = 0; a < 3; a++) ok = ok && tame(o.normal[a]);
			if (o.geometry_kind == RMD_GEOM_SPHERE) ok = ok && tame(o.radius) && o.radius != 0.0 && o.radius * o.radius > 0.0;
			for (int a = 0; a < 3; a++) ok = ok && tame(o.material.color[a]);
			if (o.material.kind != RMD_MAT_EMISSION) ok = ok && tame(o.material.roughness) && o.material.roughness != 0.0;
			regular = regular && ok;
		}
		if (o.material.kind == RMD_MAT_DIFFUSE && o.material.color[0] == 0.0 && o.material.color[1] == 0.0 && o.material.color[2] == 0.0) d.flags |= rmd::kObjBlackDiffuse;
	}
	// pair_opposite_planes: plane j is tested together with the first earlier, still unpaired plane i whose normal is its exact negation
	// (device_core.hpp: plane_pair_intersect — the facing conditions of such planes exclude each other, one division serves both)
#ifndef RMD_PAIR_PLANES
#define RMD_PAIR_PLANES 1
#endif
	for (uint32_t j = 0; j < n_objects && RMD_PAIR_PLANES; j++) {
		if (hobj[j].geometry_kind != RMD_GEOM_PLANE) continue;
		for (uint32_t i = 0; i < j; i++) {
			if (hobj[i].geometry_kind != RMD_GEOM_PLANE || hobj[i].pair_info != 0u) continue;
			bool opposite = true; // NaN components compare unequal: never paired
			for (int a = 0; a < 3; a++) opposite = opposite && hobj[j].normal[a] == -hobj[i].normal[a];
			if (!opposite) continue;
			hobj[i].pair_info = rmd::kPairTestedAtPartner | j, hobj[j].pair_info = i + 1u;
			break;
		}
	}
	// ... and a pair whose normals are exactly +e_k and -e_k — the walls of an axis-aligned room — is tested with one component of the ray instead of
	// three dot products (scene_split.hpp: axis_pairs_visit has the argument for "same bits"): one pair per axis, in scenes of regular parameters
	uint32_t axis_pairs = 0;
#ifndef RMD_AXIS_PAIRS
#define RMD_AXIS_PAIRS 1
#endif
	for (uint32_t j = 0; j < n_objects && j < 1023u && regular && RMD_AXIS_PAIRS && ctx->tunable[RMD_TUNE_AXIS_PAIRS] != 1; j++) {
		if (hobj[j].geometry_kind != RMD_GEOM_PLANE || hobj[j].pair_info == 0u || (hobj[j].pair_info & rmd::kPairTestedAtPartner)) continue;
		const uint32_t i = hobj[j].pair_info - 1u;
		int k = -1, nonzero = 0;
		for (int a = 0; a < 3; a++)
			if (hobj[i].normal[a] != 0.0) nonzero++, k = a;
		if (nonzero != 1 || (hobj[i].normal[k] != 1.0 && hobj[i].normal[k] != -1.0) || ((axis_pairs >> (10 * k)) & 1023u) != 0u) continue;
		axis_pairs |= (j + 1u) << (10 * k);
		hobj[j].flags |= rmd::kObjAxisPair | ((uint32_t)k << rmd::kObjAxisShift) | (hobj[i].normal[k] == 1.0 ? rmd::kObjAxisEarlierIsPlus : 0u);
		hobj[i].flags |= rmd::kObjAxisPair;
		hobj[j].partner_origin_k = hobj[i].origin[k];
		hobj[j].pair_info = rmd::kPairTestedAtPartner | rmd::kPairAxis | i; // the object loops pass both planes by ("tested at its partner's turn"):
		hobj[i].pair_info |= rmd::kPairAxis;                                 // their turn is axis_pairs_visit's, ahead of the loop
	}
	sc = new (std::nothrow) rmd_scene();
	if (!sc) return rmd::fail(ctx, RMD_ERR_OUT_OF_MEMORY, "rmd_scene_create: allocation failed");
	sc->ctx = ctx, sc->n_objects = n_objects, sc->n_grids = n_grids, sc->regular = regular;
	sc->axis_pairs = axis_pairs;
	// the object loops' turns (device_types.hpp: RenderParams::visit_mask): a plane tested at its partner's turn or by the axis rule has none
	// the axis pairs as the axis rule reads them, behind the table's last record (device_types.hpp: AxisWalls)
	rmd::DevObject walls_block;
	std::memset(&walls_block, 0, sizeof(walls_block));
	{
		rmd::AxisWalls walls[3];
		std::memset(walls, 0, sizeof(walls));
		for (int k = 0; k < 3; k++) {
			const uint32_t f = (axis_pairs >> (10 * k)) & 1023u;
			if (f == 0u) continue;
			const uint32_t j = f - 1u, i = hobj[j].pair_info & 0x3FFFFFFFu; // the later and the earlier plane of the pair
			const bool e_plus = (hobj[j].flags & rmd::kObjAxisEarlierIsPlus) != 0u;
			walls[k].o_plus = e_plus ? hobj[i].origin[k] : hobj[j].origin[k], walls[k].o_minus = e_plus ? hobj[j].origin[k] : hobj[i].origin[k];
			walls[k].idx_plus = e_plus ? i : j, walls[k].idx_minus = e_plus ? j : i;
		}
		std::memcpy(&walls_block, walls, sizeof(walls));
	}
	sc->visit_mask = 0ull, sc->grid_mask = 0ull;
	for (uint32_t i = 0; i < n_objects && i < 64u; i++) {
		if (!(hobj[i].geometry_kind == RMD_GEOM_PLANE && (hobj[i].pair_info & rmd::kPairTestedAtPartner))) sc->visit_mask |= 1ull << i;
		if (hobj[i].geometry_kind == RMD_GEOM_GRID) sc->grid_mask |= 1ull << i;
	}
	for (uint32_t i = 0; i < n_objects; i++) sc->n_grid_objects += objects[i].geometry_kind == RMD_GEOM_GRID ? 1u : 0u;
	auto upload = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
		*dst = nullptr;
		hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
		if (e != hipSuccess) return e;
		sc->owned.push_back(*dst);
		if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
		return e;
	};
#define RMD_SCENE_HIP(call)                                                                     \
	do {                                                                                        \
		hipError_t e_ = (call);                                                                 \
		if (e_ != hipSuccess) {                                                                 \
			rmd_scene_destroy(sc), sc = nullptr;                                                              \
			return rmd::fail(ctx, e_ == hipErrorOutOfMemory ? RMD_ERR_OUT_OF_MEMORY : RMD_ERR_HIP, \
			                 std::string("rmd_scene_create: ") + hipGetErrorString(e_));         \
		}                                                                                       \
	} while (0)

	std::vector<rmd::DevGrid> hgrid(n_grids);
	for (uint32_t gi = 0; gi < n_grids; gi++) {
		const rmd_grid_desc &g = grids[gi];
		rmd::DevGrid &d = hgrid[gi];
		std::memset(&d, 0, sizeof(d));
		// the derived tables: from the rmd_grid_build this description came from (derived once, kept there), or made here
		std::shared_ptr<const GridDerived> derived;
		{
			rmd_grid_build *gb = const_cast<rmd_grid_build *>(g.built);
			const bool from_build = gb && gb->cells.data() == g.cells && gb->cells.size() == g.n_cells && gb->mapping.data() == g.mapping_table &&
			                        gb->mapping.size() == g.n_mapping && gb->pos.data() == g.tri_pos && gb->nrm.data() == g.tri_nrm && gb->pos.size() == g.n_tris * 9;
			std::unique_lock<std::mutex> lock;
			if (from_build) {
				lock = std::unique_lock<std::mutex>(gb->derived_mutex);
				derived = std::static_pointer_cast<const GridDerived>(gb->derived);
			}
			if (!derived) {
				auto fresh = std::make_shared<GridDerived>();
				if (rmd_status st = derive_grid_tables(ctx, g, *fresh)) {
					rmd_scene_destroy(sc), sc = nullptr;
					return st;
				}
				derived = fresh;
				if (from_build) gb->derived = fresh;
			}
		}
		const std::vector<unsigned char> &recs = derived->recs;
		const std::vector<rmd::CellEntry> &entries = derived->entries;
		const std::vector<uint32_t> &ids = derived->ids;
		const std::vector<double> &aux = derived->aux;
		for (int a = 0; a < 3; a++) {
			d.bbox_min[a] = g.bbox_min[a], d.bbox_max[a] = g.bbox_max[a], d.cell_size[a] = g.cell_size[a];
			d.inv_cell_size[a] = rmd::exact_reciprocal(g.cell_size[a]);
			d.res[a] = g.resolution[a];
			if (a == 2) sc->walk_steps_bound = std::max<uint32_t>(sc->walk_steps_bound, (uint32_t)std::min<uint64_t>((uint64_t)g.resolution[0] + g.resolution[1] + g.resolution[2] + 3u, 0xFFFFFFFFull));
		}
		d.n_cells = g.n_cells, d.n_tris = g.n_tris;
		uint64_t last_nonempty = 0;
		bool any_nonempty = false;
		for (uint64_t c = 0; c < g.n_cells; c++)
			if (entries[c * rmd::kEntrySlots].count) last_nonempty = c, any_nonempty = true;
		// occupancy bitmask for LDS: bit i covers cells [i << shift, (i+1) << shift); only up to the last non-empty cell
		// the mask covers every cell of the array when that fits the budget (the stepping loop's lean form then needs no clamp of the index),
		// else the cells up to the last non-empty one (indices past it read the all-zero word that ends every mask)
		uint64_t covered = any_nonempty ? last_nonempty + 1 : 0;
		size_t budget_bytes = rmd::kMaskBudgetBytes;
		if (ctx->tunable[RMD_TUNE_MASK_BUDGET] > 0) budget_bytes = (size_t)ctx->tunable[RMD_TUNE_MASK_BUDGET]; // forces coarser masks
		if (budget_bytes < 64) budget_bytes = 64;
		if (budget_bytes > rmd::kMaskBudgetBytes) budget_bytes = rmd::kMaskBudgetBytes;
		const size_t budget_words = budget_bytes / 4 / n_grids;
		if (budget_words < 2) { // one data word + the all-zero pad word is the smallest mask
			rmd_scene_destroy(sc), sc = nullptr;
			return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_scene_create: too many grids for the LDS occupancy-mask budget");
		}
		if ((g.n_cells + 31) / 32 + 1 <= budget_words) covered = g.n_cells;
		uint32_t shift = 0;
		while (shift < 63 && ((covered >> shift) + 31) / 32 + 1 > budget_words) shift++;
		const uint64_t bits = covered ? ((covered - 1) >> shift) + 1 : 0;
		std::vector<uint32_t> mask((size_t)((bits + 31) / 32) + 1, 0u); // + one all-zero word: indices past the mask read it
		for (uint64_t c = 0; c < covered; c++)
			if (entries[c * rmd::kEntrySlots].count) mask[(c >> shift) >> 5] |= 1u << ((c >> shift) & 31u);
		d.mask_bits = (uint32_t)bits, d.mask_shift = shift, d.mask_n_words = (uint32_t)mask.size();
		d.mask_lds_word = sc->mask_words_total;
		sc->mask_words_total += (uint32_t)((mask.size() + 3) & ~(size_t)3);
		void *p = nullptr;
		RMD_SCENE_HIP(upload(entries.data(), entries.size() * sizeof(rmd::CellEntry), &p));
		d.cell_entries = (const rmd::CellEntry *)p;
		RMD_SCENE_HIP(upload(ids.data(), ids.size() * sizeof(uint32_t), &p));
		d.tri_ids = (const uint32_t *)p;
		RMD_SCENE_HIP(upload(recs.data(), recs.size(), &p));
		d.tri_recs = p;
		RMD_SCENE_HIP(upload(mask.data(), mask.size() * sizeof(uint32_t), &p));
		d.mask_words = (const uint32_t *)p;
		RMD_SCENE_HIP(upload(g.tri_pos, g.n_tris * 9 * sizeof(double), &p));
		d.tri_pos = (const double *)p;
		RMD_SCENE_HIP(upload(g.tri_nrm, g.n_tris * 9 * sizeof(double), &p));
		d.tri_nrm = (const double *)p;
		RMD_SCENE_HIP(upload(aux.data(), aux.size() * sizeof(double), &p));
		d.tri_aux = (const double *)p;
		RMD_SCENE_HIP(upload(derived->spheres.data(), derived->spheres.size() * sizeof(double), &p));
		d.tri_sph = (const double *)p, d.sph_kb = derived->sphere_kb;
	}
	void *p = nullptr;
	hobj.push_back(walls_block); // (uploaded behind the table; n_objects does not count it)
	RMD_SCENE_HIP(upload(hobj.data(), hobj.size() * sizeof(rmd::DevObject), &p));
	sc->d_objects = (rmd::DevObject *)p;
	RMD_SCENE_HIP(upload(hgrid.data(), hgrid.size() * sizeof(rmd::DevGrid), &p));
	sc->d_grids = (rmd::DevGrid *)p;
#undef RMD_SCENE_HIP
	*out = sc;
	sc = nullptr; // handed over
	return RMD_OK;
}

rmd_status rmd_scene_create(rmd_context *ctx, const rmd_object *objects, uint32_t n_objects, const rmd_grid_desc *grids, uint32_t n_grids,
                            rmd_scene **out) {
	rmd_scene *sc = nullptr;
	try {
		return scene_create_impl(ctx, objects, n_objects, grids, n_grids, out, sc);
	} catch (const std::bad_alloc &) {
		if (sc) rmd_scene_destroy(sc);
		if (out) *out = nullptr;
		return rmd::fail(ctx, RMD_ERR_OUT_OF_MEMORY, "rmd_scene_create: the host ran out of memory building the scene's tables");
	} catch (const std::exception &e) { // nothing throws across the boundary
		if (sc) rmd_scene_destroy(sc);
		if (out) *out = nullptr;
		return rmd::fail(ctx, RMD_ERR_HIP, std::string("rmd_scene_create: ") + e.what());
	}
}

void rmd_scene_destroy(rmd_scene *scene) {
	if (!scene) return;
	if (scene->ctx) {
		(void)hipSetDevice(scene->ctx->device);
		(void)hipStreamSynchronize(scene->ctx->stream);
	}
	for (void *p : scene->owned) (void)hipFree(p);
	delete scene;
}

rmd_status rmd_framebuffer_alloc(rmd_context *ctx, uint32_t width, uint32_t height, double **out_dev) {
	if (rmd_status s = bind(ctx)) return s;
	if (!out_dev || width == 0 || height == 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_alloc: bad argument");
	size_t bytes = (size_t)width * height * 3 * sizeof(double);
	RMD_HIP(ctx, hipMalloc((void **)out_dev, bytes));
	RMD_HIP(ctx, hipMemsetAsync(*out_dev, 0, bytes, ctx->stream));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return RMD_OK;
}
rmd_status rmd_framebuffer_free(rmd_context *ctx, double *dev) {
	if (rmd_status s = bind(ctx)) return s;
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	RMD_HIP(ctx, hipFree(dev));
	return RMD_OK;
}
rmd_status rmd_framebuffer_zero(rmd_context *ctx, double *dev, size_t n) {
	if (rmd_status s = bind(ctx)) return s;
	if (!dev) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_zero: null pointer");
	RMD_HIP(ctx, hipMemsetAsync(dev, 0, n * sizeof(double), ctx->stream));
	return RMD_OK;
}
rmd_status rmd_framebuffer_download(rmd_context *ctx, const double *dev, double *host, size_t n) {
	if (rmd_status s = bind(ctx)) return s;
	if (!dev || !host) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_download: null pointer");
	RMD_HIP(ctx, hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return rmd::check_fault(ctx); // a frame cut short by a device fault is not handed out as a result
}
rmd_status rmd_framebuffer_upload(rmd_context *ctx, const double *host, double *dev, size_t n) {
	if (rmd_status s = bind(ctx)) return s;
	if (!dev || !host) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_upload: null pointer");
	RMD_HIP(ctx, hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return RMD_OK;
}

// A work item is (wave tile, sample range): a tile's samples are split over K items whose waves store per-sample radiance to an HBM
// scratch buffer, added to the pixels in sample order afterwards (by the wave that finishes a tile last, or by sum_kernel) — bit-identical
// to the unsplit launch (tests/test_gpu_parity.py::test_sample_split_is_bit_exact).  It buys enough items to level the launch over 256 CUs
// when wave tiles are few (an N-way shard) or very uneven (a mesh), and lanes that draw (pixel, sample) pairs from the item's pool instead
// of idling until the tile's longest pixel is done.  RMD_TUNE_SAMPLE_SPLIT forces K.
// `buffered` (out): whether the launch runs the tiles-buffered instantiation.  A split launch (K > 1) always does; a launch too short to split
// does too — as ONE item per wave tile — in scenes with grids (the direct instantiation of the mesh kernel is the slower one at any size: C3 at
// 4 spp 5.6 ms direct, 4.5 ms as two items of 2 samples; progressive passes of a host scheduler are such launches) and, from kSortedMinSamples
// samples on, in scenes without.  RMD_TUNE_SAMPLE_SPLIT = 1 still forces the direct mode (the scratch-free route).
static uint32_t choose_split(const rmd_context *ctx, bool has_grid, uint32_t n_wave_tiles, uint32_t sample_count, bool *buffered = nullptr) {
	uint32_t k = 1;
	bool may_buffer_unsplit = n_wave_tiles != 0 && sample_count >= (has_grid ? 1u : rmd::kSortedMinSamples);
	if (ctx->tunable[RMD_TUNE_SAMPLE_SPLIT] > 0) {
		k = (uint32_t)ctx->tunable[RMD_TUNE_SAMPLE_SPLIT];
		if (k > sample_count / 4u) k = sample_count / 4u; // a forced split keeps >= 4 samples (256 pool items) per wave
		may_buffer_unsplit = false;
	} else if (n_wave_tiles != 0) {
		// Mesh kernel: about 64 work items per wave slot level the tail of a launch of persistent workgroups (tools/split_sweep.py — full C3 frame:
		// 512.8 ms at 32, 508.1 at 64 .. 128, 517.8 at 500), of at least 4 samples each.  Spheres kernel (trips sorted by role,
		// render_kernel.hpp): an item ends with a few ever emptier trips while its last paths finish, a release and, for a tile's last item, the
		// ordered sum, so its items are larger — about 24 per wave slot, at least 64 samples each (round 4, full C2 frame: 52.9 ms at 3 .. 6
		// items per wave tile, 55.2 at 9, 59.1 at 12; an N-way tile share, tools/shard_split.py: N = 8 7.4 ms at 7 .. 10, 8.5 at 16, 9.2 at 3;
		// N = 2 27.3 at 6 .. 7, 29.0 at 2, 29.8 at 12)
		const uint32_t waves_per_slot = has_grid ? 256u : 24u; // (grid scenes: 64 until the work items were chained — an item's end costs a chaining wave nothing, and smaller items even the launch's tail out: C3 382.8 ms at 9 items per wave tile, 381.3 at 32, 381.0 at 64)
		k = (waves_per_slot * ctx->wave_slots + n_wave_tiles - 1u) / n_wave_tiles;
		if (has_grid && k < 2u) k = 2u; // the mesh kernel's direct instantiation is the slower one at any size
		// (launches short enough for the instantiation whose waves chain their work items — an item's end costs them nothing — take items of 2)
		uint32_t min_samples = has_grid ? (sample_count <= rmd::kChainMaxSamples && ctx->tunable[RMD_TUNE_CHAIN_ITEMS] != 1 ? 2u : rmd::kSplitMinSamplesGrid) : 64u;
		if (ctx->tunable[RMD_TUNE_SPLIT_MIN_SAMPLES] > 0) min_samples = (uint32_t)std::min<int64_t>(ctx->tunable[RMD_TUNE_SPLIT_MIN_SAMPLES], 1 << 20);
		if (k > sample_count / min_samples) k = sample_count / min_samples;
		// ... but two items per wave tile while each still holds two samples (C3 at 4 spp: 4.5 ms as two items of 2 samples, 5.1 as one item of 4:
		// with one item per wave tile the mesh tiles' items are the launch's tail)
		if (has_grid && k < 2u && sample_count >= 4u && ctx->tunable[RMD_TUNE_SPLIT_MIN_SAMPLES] <= 0) k = 2u;
	}
	if (k > 64u) k = 64u;
	while (k > 1u && (uint64_t)n_wave_tiles * k > 0x7FFFFFFFull) k--; // work items are indexed in 32 bits
	if (k < 2u) k = 1u;
	if (buffered) *buffered = k > 1u || may_buffer_unsplit;
	return k;
}

static rmd_status render_tiles_async_impl(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera, const rmd_settings *settings,
                                          const rmd_tile_rect *tiles, uint32_t n_tiles, double *accum_dev) {
	if (rmd_status s = bind(ctx)) return s;
	if (rmd_status s = check_render_args(ctx, scene, camera, settings)) return s;
	if (!accum_dev || (n_tiles && !tiles)) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_render_tiles: null tiles/accum pointer");
	if (rmd_status s = prepare_wave_tiles(ctx, camera, tiles, n_tiles)) return s;
	if (settings->bounce_limit == 0u) { // trace(.., 1) with depth 1 > bounce_limit returns (0, 0, 0) unintersected (src/trace.rs:235-237): pixel += 0
		RMD_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
		RMD_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
		ctx->timed = true;
		return RMD_OK;
	}
	rmd::RenderParams P = rmd::make_params(ctx, scene, camera, settings);
	P.n_work = ctx->n_wave_tiles;
	if (P.debug_flags & 24u) {
		if (!ctx->d_debug_counters) RMD_HIP(ctx, hipMalloc((void **)&ctx->d_debug_counters, 40 * sizeof(unsigned long long)));
		RMD_HIP(ctx, hipMemsetAsync(ctx->d_debug_counters, 0, 40 * sizeof(unsigned long long), ctx->stream));
		P.debug_counters = ctx->d_debug_counters;
	}
	bool buffered = false;
	uint32_t split = choose_split(ctx, scene->n_grids != 0, P.n_work, P.sample_count, &buffered);
	// default form: persistent workgroups, one per CU, whose waves draw their work items from a counter (form 1: one wave per work
	// item — round 1's; 4.7 % slower on the benchmark mesh, 2.4 % on the spheres frame, where a workgroup launch per item cost ~100 us
	// of a wave slot each: 152 vs 113.6 ms at 32 items per wave tile)
	const bool persistent = ctx->tunable[RMD_TUNE_LAUNCH_FORM] != 1;
	if (persistent && !ctx->d_work_counter) RMD_HIP(ctx, hipMalloc((void **)&ctx->d_work_counter, 256));
	// Samples per pass of a split launch: the scratch buffer holds n_wave_tiles x 64 x samples x 32 bytes (the whole C3 frame at 500 spp
	// is 33 GB, one launch).  By default it may take an eighth of the device memory that is free right now (RMD_TUNE_SCRATCH_CAP_MB
	// overrides), never less than 8 samples per pass; what does not fit runs as several passes.  The device may still refuse the
	// allocation — other contexts, ranks or tenants hold memory, or the cap was set beyond it: the pass is then halved until a buffer
	// can be had (the old one stays until a larger one exists), and when not even 8 samples fit the launch runs unsplit (one wave per
	// wave tile, no scratch).  Every route gives the same frame bit for bit.
	uint32_t per_pass = P.sample_count;
	if (buffered) {
		const size_t bytes_per_sample = (size_t)P.n_work * 64u * rmd::kSampleStride * sizeof(double);
		size_t cap;
		if (ctx->tunable[RMD_TUNE_SCRATCH_CAP_MB] > 0) cap = (size_t)ctx->tunable[RMD_TUNE_SCRATCH_CAP_MB] << 20;
		else {
			size_t free_b = 0, total_b = 0;
			RMD_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
			cap = (free_b + ctx->sample_buf_bytes) / 8; // the buffer this context already holds counts as available to it
		}
		if (bytes_per_sample * per_pass > cap) per_pass = (uint32_t)(cap / bytes_per_sample);
		// (the mesh kernel numbers a pass's 32-byte sectors in 32 bits: render_kernel.hpp, PathId)
		const uint64_t sectors_per_sample = (uint64_t)P.n_work * 64u;
		if (sectors_per_sample * per_pass > 0xFFFFFFFFull) per_pass = (uint32_t)(0xFFFFFFFFull / sectors_per_sample);
		if (per_pass < 8u) per_pass = 8u;
		if (per_pass > P.sample_count) per_pass = P.sample_count;
		// (a frame of more than 2^32 / 8 / 64 = 8.4 M wave tiles — 537 Mpixel — cannot number even the smallest pass's sectors in 32 bits: unsplit, no scratch)
		if (sectors_per_sample * per_pass > 0xFFFFFFFFull) split = 1u, buffered = false, per_pass = P.sample_count;
		while (buffered && bytes_per_sample * per_pass > ctx->sample_buf_bytes) {
			double *fresh = nullptr;
			const hipError_t e = hipMalloc((void **)&fresh, bytes_per_sample * per_pass);
			if (e == hipSuccess) {
				if (ctx->d_sample_buf) (void)hipFree(ctx->d_sample_buf);
				ctx->d_sample_buf = fresh, ctx->sample_buf_bytes = bytes_per_sample * per_pass;
				break;
			}
			(void)hipGetLastError(); // the failure is handled here: it must not surface at the next launch check
			if (e != hipErrorOutOfMemory) return rmd::fail(ctx, RMD_ERR_HIP, std::string("hipMalloc(sample scratch): ") + hipGetErrorString(e));
			if (per_pass <= 8u) { // not even the smallest pass: render unsplit
				split = 1u, buffered = false, per_pass = P.sample_count;
				break;
			}
			per_pass = per_pass / 2u < 8u ? 8u : per_pass / 2u;
		}
	}
	ctx->last_launch = rmd_launch_info{};
	ctx->last_launch.end_black_paths = P.end_black_paths, ctx->last_launch.has_grid = scene->n_grids != 0u;
	RMD_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
	for (uint64_t done = 0; done < settings->sample_count || done == 0; done += per_pass) { // 64-bit: sample_count may be close to 2^32
		rmd::RenderParams Q = P;
		Q.sample_begin = settings->sample_begin + (uint32_t)done;
		Q.sample_count = settings->sample_count - done < per_pass ? (uint32_t)(settings->sample_count - done) : per_pass;
		Q.split_k = split > 1u ? choose_split(ctx, scene->n_grids != 0, P.n_work, Q.sample_count) : 1u;
		Q.sample_magic = Q.sample_count > 1u ? ~0ull / Q.sample_count + 1ull : 0ull; // floor(2^64 / d) + 1 for d >= 2 (2^64 - 1 and 2^64 have the same quotient unless d divides 2^64: then + 1 overshoots by one and is still exact for dividends below 2^32)
		Q.buffered = buffered ? 1u : 0u;
		{ // split launches of grid scenes chain their work items (launch.hpp: kChainMaxSamples; RMD_TUNE_CHAIN_ITEMS: 1 = never, 2 = always)
			const int64_t force = ctx->tunable[RMD_TUNE_CHAIN_ITEMS];
			Q.chain_items = (buffered && scene->n_grids != 0u && (force == 2 || (force == 0 && Q.sample_count <= rmd::kChainMaxSamples))) ? 1u : 0u;
		}
		Q.sample_buf = ctx->d_sample_buf;
		// (a launch with fewer work items than the device has wave slots spreads better as one wave per item)
		const bool persistent_pass = persistent && ((uint64_t)P.n_work * Q.split_k >= ctx->wave_slots || ctx->tunable[RMD_TUNE_LAUNCH_FORM] == 2);
		if (persistent_pass) {
			RMD_HIP(ctx, hipMemsetAsync(ctx->d_work_counter, 0, sizeof(uint32_t), ctx->stream));
			Q.work_counter = ctx->d_work_counter;
		}
		// Persistent split launches of scenes with grids keep their paths in queues in device memory (render_kernel.hpp: render_wave_queued;
		// RMD_TUNE_PATH_QUEUES: 1 = never): kQueuePaths entries on each of a resident wave's two stacks, 192 bytes a path — 49 KB a wave, 200 MB
		// for the 4,096 resident waves of an MI355X — allocated at the first launch that wants them.  A device that cannot provide them runs
		// the lane-per-path form (render_wave): the same frame bit for bit.
		if (persistent_pass && buffered && scene->n_grids != 0u && ctx->tunable[RMD_TUNE_PATH_QUEUES] != 1) {
			const size_t wave_bytes = rmd::path_queue_bytes_host(rmd::kQueuePaths), need = wave_bytes * ctx->wave_slots;
			if (ctx->queue_buf_bytes < need) {
				if (ctx->d_queue_buf) RMD_HIP(ctx, hipFree(ctx->d_queue_buf));
				ctx->d_queue_buf = nullptr, ctx->queue_buf_bytes = 0;
				const hipError_t qe = hipMalloc((void **)&ctx->d_queue_buf, need);
				if (qe == hipSuccess) {
					ctx->queue_buf_bytes = need;
					// (zeroed once: a trip's idle lanes read the trip's first entry, never one nobody wrote — but a fresh allocation should not hold another process's data)
					RMD_HIP(ctx, hipMemsetAsync(ctx->d_queue_buf, 0, need, ctx->stream));
				} else {
					(void)hipGetLastError();
					ctx->d_queue_buf = nullptr;
					if (qe != hipErrorOutOfMemory) return rmd::fail(ctx, RMD_ERR_HIP, std::string("hipMalloc(path queues): ") + hipGetErrorString(qe));
				}
			}
			if (ctx->d_queue_buf) Q.queue_buf = ctx->d_queue_buf, Q.queue_wave_bytes = (uint32_t)wave_bytes, Q.queue_paths = rmd::kQueuePaths;
		}
		// spheres kernel: the wave that finishes a wave tile last adds the tile's samples to the pixels itself (no second kernel: 120.3 ->
		// 117.0 ms per C2 frame).  Mesh scenes keep sum_kernel: their kernel waits on memory a third of the time, and the sum's 33 GB of
		// streaming reads in between cost it more (491.4 vs 487.3 ms on C3) than the separate kernel's 5.5 ms
		if (buffered && scene->n_grids == 0) {
			if (ctx->tile_done_words < P.n_work) {
				if (ctx->d_tile_done) RMD_HIP(ctx, hipFree(ctx->d_tile_done));
				ctx->d_tile_done = nullptr, ctx->tile_done_words = 0;
				RMD_HIP(ctx, hipMalloc((void **)&ctx->d_tile_done, (size_t)P.n_work * sizeof(uint32_t)));
				ctx->tile_done_words = P.n_work;
			}
			RMD_HIP(ctx, hipMemsetAsync(ctx->d_tile_done, 0, (size_t)P.n_work * sizeof(uint32_t), ctx->stream));
			Q.tile_done = ctx->d_tile_done;
		}
		rmd::LaunchShape shape;
		RMD_HIP(ctx, rmd::launch_render_tiles(ctx->stream, Q, scene->d_objects, scene->d_grids, ctx->d_wave_tiles, accum_dev, persistent_pass ? ctx->n_cus : 0u, &shape));
		ctx->last_launch.passes++, ctx->last_launch.split_k = Q.split_k, ctx->last_launch.buffered = Q.buffered;
		ctx->last_launch.persistent = shape.persistent, ctx->last_launch.waves_per_workgroup = shape.waves_per_wg; // the form it was launched in, not the one asked for
		ctx->last_launch.queued = shape.queued;
		ctx->last_launch.chained = (shape.persistent && !shape.queued) ? Q.chain_items : 0u;
		if (settings->sample_count == 0) break;
	}
	RMD_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
	ctx->timed = true;
	if (P.debug_flags & 24u) {
		unsigned long long h[40];
		RMD_HIP(ctx, hipMemcpyAsync(h, ctx->d_debug_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
		RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if ((P.debug_flags & 16u) && ctx->last_launch.queued) {
			static const char *kinds[3] = {"GEN  ", "SHADE", "WALK "};
			for (int k = 0; k < 3; k++)
				std::fprintf(stderr, "[rmd queued stamps, cycles] %s trips=%llu fetch=%llu make_ray=%llu simple=%llu walk=%llu ray_push=%llu classify=%llu stores=%llu\n", kinds[k], h[k * 8 + 7],
				             h[k * 8 + 0], h[k * 8 + 1], h[k * 8 + 2], h[k * 8 + 3], h[k * 8 + 4], h[k * 8 + 5], h[k * 8 + 6]);
			std::fprintf(stderr, "[rmd queued stamps, cycles] inside the walks: init=%llu stepping=%llu entry_wait=%llu scan=%llu (chunk search+load+test)=%llu hits=%llu tail=%llu\n",
			             h[24], h[25], h[26], h[27], h[29], h[30], h[31]);
		} else if (P.debug_flags & 16u)
			std::fprintf(stderr, "[rmd stamps, cycles] wave_total=%llu next_ray=%llu simple=%llu walk=%llu classify=%llu | walk: init=%llu stepping=%llu entry_wait=%llu scan=%llu (chunk search+load+test)=%llu hits=%llu tail=%llu\n",
			             h[0], h[1], h[2], h[3], h[4], h[8], h[9], h[10], h[11], h[13], h[14], h[15]);
		else
			std::fprintf(stderr, "[rmd debug] walk_calls=%llu walkers=%llu calls_with_walkers=%llu rounds=%llu wave_steps=%llu lane_steps=%llu test_rounds=%llu tests=%llu chunks=%llu | main_iterations=%llu live_lanes=%llu lanes_with_ray=%llu | stepping iterations with <= 4 / 8 / 16 lanes: %llu / %llu / %llu, with <= 8 lanes while tests wait: %llu\n",
			             h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[8], h[9], h[10], h[11], h[12], h[7], h[13], h[14], h[15]);
		if ((P.debug_flags & 16u) == 0u)
			std::fprintf(stderr, "[rmd debug] sphere pre-test: pairs passed=%llu full chunks=%llu | pairs dropped that pass the reference's test (flag 64; must be 0)=%llu\n", h[17], h[18], h[16]);
		if ((P.debug_flags & 16u) == 0u && ctx->last_launch.queued)
			std::fprintf(stderr, "[rmd debug] path queues: rays pushed=%llu (of them walks put aside=%llu) rays held in their lanes=%llu hits pushed=%llu hits held in their lanes=%llu\n", h[19], h[22], h[23], h[20], h[21]);
	}
	return RMD_OK;
}

rmd_status rmd_render_tiles_async(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera, const rmd_settings *settings,
                                  const rmd_tile_rect *tiles, uint32_t n_tiles, double *accum_dev) {
	// (the wave-tile table and the cached rectangle list are std::vectors: nothing throws across the boundary)
	return rmd::guarded(ctx, "rmd_render_tiles", [&] { return render_tiles_async_impl(ctx, scene, camera, settings, tiles, n_tiles, accum_dev); });
}

rmd_status rmd_context_set_tunable(rmd_context *ctx, uint32_t key, int64_t value) {
	if (!ctx) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_context_set_tunable: null context");
	if (key >= RMD_TUNE_COUNT || value < 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_context_set_tunable: unknown key or negative value");
	if (key == RMD_TUNE_LAUNCH_FORM && value > 2) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_context_set_tunable: launch form is 0 .. 2");
	ctx->tunable[key] = value;
	return RMD_OK;
}
rmd_status rmd_context_get_tunable(const rmd_context *ctx, uint32_t key, int64_t *out_value) {
	if (!ctx || !out_value || key >= RMD_TUNE_COUNT) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_context_get_tunable: bad argument");
	*out_value = ctx->tunable[key];
	return RMD_OK;
}

rmd_status rmd_context_memory_info(rmd_context *ctx, uint64_t *out_free_bytes, uint64_t *out_total_bytes) {
	if (rmd_status s = bind(ctx)) return s;
	if (!out_free_bytes || !out_total_bytes) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_context_memory_info: null pointer");
	size_t free_b = 0, total_b = 0;
	RMD_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
	*out_free_bytes = free_b, *out_total_bytes = total_b;
	return RMD_OK;
}

rmd_status rmd_context_synchronize(rmd_context *ctx) {
	if (rmd_status s = bind(ctx)) return s;
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return rmd::check_fault(ctx);
}

rmd_status rmd_render_tiles(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera, const rmd_settings *settings,
                            const rmd_tile_rect *tiles, uint32_t n_tiles, double *accum_dev) {
	if (rmd_status s = rmd_render_tiles_async(ctx, scene, camera, settings, tiles, n_tiles, accum_dev)) return s;
	return rmd_context_synchronize(ctx);
}

rmd_status rmd_render_tiles_host(rmd_context *ctx, const rmd_scene *scene, const rmd_camera *camera, const rmd_settings *settings,
                                 const rmd_tile_rect *tiles, uint32_t n_tiles, double *accum_host) {
	if (rmd_status s = bind(ctx)) return s;
	if (!camera || !accum_host) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_render_tiles_host: null argument");
	size_t n = (size_t)camera->backbuffer_width * camera->backbuffer_height * 3;
	double *dev = nullptr;
	RMD_HIP(ctx, hipMalloc((void **)&dev, n * sizeof(double)));
	rmd_status s = rmd_framebuffer_upload(ctx, accum_host, dev, n);
	if (!s) s = rmd_render_tiles(ctx, scene, camera, settings, tiles, n_tiles, dev);
	if (!s) s = rmd_framebuffer_download(ctx, dev, accum_host, n);
	(void)hipFree(dev);
	return s;
}

// ---------------------------------------------------------------- tile rectangles <-> packed host buffers
namespace {
// Checks the rects, builds the table {rect, first pixel} and makes slot `sl`'s device buffers large enough.  Returns the packed pixel count.
rmd_status prepare_transfer(rmd_context *ctx, rmd_context::TransferSlot &sl, uint32_t W, uint32_t H, const rmd_tile_rect *rects, uint32_t n_rects, uint64_t &n_pixels) {
	n_pixels = 0;
	sl.h_table.resize((size_t)n_rects * (sizeof(rmd_tile_rect) + sizeof(uint64_t)));
	rmd_tile_rect *hr = reinterpret_cast<rmd_tile_rect *>(sl.h_table.data());
	uint64_t *hf = reinterpret_cast<uint64_t *>(sl.h_table.data() + (size_t)n_rects * sizeof(rmd_tile_rect));
	for (uint32_t i = 0; i < n_rects; i++) {
		const rmd_tile_rect &r = rects[i];
		if ((uint64_t)r.left + r.width > W || (uint64_t)r.top + r.height > H)
			return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "tile transfer: tile rectangle outside the framebuffer");
		hr[i] = r, hf[i] = n_pixels;
		n_pixels += (uint64_t)r.width * r.height;
	}
	const size_t need = (size_t)n_pixels * 3 * sizeof(double);
	if (need > sl.packed_bytes) {
		if (sl.d_packed) RMD_HIP(ctx, hipFree(sl.d_packed));
		sl.d_packed = nullptr, sl.packed_bytes = 0;
		RMD_HIP(ctx, hipMalloc((void **)&sl.d_packed, need));
		sl.packed_bytes = need;
	}
	if (sl.h_table.size() > sl.table_bytes) {
		if (sl.d_table) RMD_HIP(ctx, hipFree(sl.d_table));
		sl.d_table = nullptr, sl.table_bytes = 0;
		RMD_HIP(ctx, hipMalloc(&sl.d_table, sl.h_table.size()));
		sl.table_bytes = sl.h_table.size();
	}
	if (!sl.packed_ready) RMD_HIP(ctx, hipEventCreateWithFlags(&sl.packed_ready, hipEventDisableTiming));
	if (!sl.copied) RMD_HIP(ctx, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
	if (!ctx->copy_stream) RMD_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
	return RMD_OK;
}
rmd_status wait_slot(rmd_context *ctx, rmd_context::TransferSlot &sl) {
	if (sl.in_flight) {
		RMD_HIP(ctx, hipEventSynchronize(sl.copied));
		sl.in_flight = false;
	}
	return RMD_OK;
}
} // namespace

static rmd_status download_tiles_async_impl(rmd_context *ctx, const double *dev, uint32_t width, uint32_t height, const rmd_tile_rect *rects,
                                            uint32_t n_rects, double *host_packed) {
	if (rmd_status s = bind(ctx)) return s;
	if (!dev || !host_packed || (n_rects && !rects) || width == 0 || height == 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_download_tiles: bad argument");
	if (n_rects == 0) return RMD_OK;
	rmd_context::TransferSlot &sl = ctx->transfer[ctx->next_transfer];
	ctx->next_transfer ^= 1u;
	if (rmd_status s = wait_slot(ctx, sl)) return s; // a third download waits for the first
	uint64_t n_pixels = 0;
	if (rmd_status s = prepare_transfer(ctx, sl, width, height, rects, n_rects, n_pixels)) return s;
	const rmd_tile_rect *d_rects = reinterpret_cast<const rmd_tile_rect *>(sl.d_table);
	const uint64_t *d_first = reinterpret_cast<const uint64_t *>(reinterpret_cast<const unsigned char *>(sl.d_table) + (size_t)n_rects * sizeof(rmd_tile_rect));
	// main stream: table, pack (behind the renders enqueued before); copy stream: the download (renders enqueued after this overlap it)
	RMD_HIP(ctx, hipMemcpyAsync(sl.d_table, sl.h_table.data(), sl.h_table.size(), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, rmd::launch_tile_copy(ctx->stream, true, const_cast<double *>(dev), sl.d_packed, d_rects, d_first, n_rects, width));
	RMD_HIP(ctx, hipEventRecord(sl.packed_ready, ctx->stream));
	RMD_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, sl.packed_ready, 0));
	RMD_HIP(ctx, hipMemcpyAsync(host_packed, sl.d_packed, (size_t)n_pixels * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
	RMD_HIP(ctx, hipEventRecord(sl.copied, ctx->copy_stream));
	sl.in_flight = true;
	return RMD_OK;
}

rmd_status rmd_framebuffer_download_tiles_async(rmd_context *ctx, const double *dev, uint32_t width, uint32_t height, const rmd_tile_rect *rects,
                                                uint32_t n_rects, double *host_packed) {
	return rmd::guarded(ctx, "rmd_framebuffer_download_tiles", [&] { return download_tiles_async_impl(ctx, dev, width, height, rects, n_rects, host_packed); });
}

rmd_status rmd_context_wait_transfers(rmd_context *ctx) {
	if (rmd_status s = bind(ctx)) return s;
	for (auto &sl : ctx->transfer)
		if (rmd_status s = wait_slot(ctx, sl)) return s;
	return rmd::check_fault(ctx); // the tiles came from launches that have completed by now
}

rmd_status rmd_framebuffer_download_tiles(rmd_context *ctx, const double *dev, uint32_t width, uint32_t height, const rmd_tile_rect *rects,
                                          uint32_t n_rects, double *host_packed) {
	if (rmd_status s = rmd_framebuffer_download_tiles_async(ctx, dev, width, height, rects, n_rects, host_packed)) return s;
	return rmd_context_wait_transfers(ctx);
}

static rmd_status upload_tiles_impl(rmd_context *ctx, const double *host_packed, double *dev, uint32_t width, uint32_t height,
                                    const rmd_tile_rect *rects, uint32_t n_rects) {
	if (rmd_status s = bind(ctx)) return s;
	if (!dev || !host_packed || (n_rects && !rects) || width == 0 || height == 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_framebuffer_upload_tiles: bad argument");
	if (n_rects == 0) return RMD_OK;
	rmd_context::TransferSlot &sl = ctx->transfer[ctx->next_transfer];
	ctx->next_transfer ^= 1u;
	if (rmd_status s = wait_slot(ctx, sl)) return s;
	uint64_t n_pixels = 0;
	if (rmd_status s = prepare_transfer(ctx, sl, width, height, rects, n_rects, n_pixels)) return s;
	const rmd_tile_rect *d_rects = reinterpret_cast<const rmd_tile_rect *>(sl.d_table);
	const uint64_t *d_first = reinterpret_cast<const uint64_t *>(reinterpret_cast<const unsigned char *>(sl.d_table) + (size_t)n_rects * sizeof(rmd_tile_rect));
	RMD_HIP(ctx, hipMemcpyAsync(sl.d_table, sl.h_table.data(), sl.h_table.size(), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, hipMemcpyAsync(sl.d_packed, host_packed, (size_t)n_pixels * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	RMD_HIP(ctx, rmd::launch_tile_copy(ctx->stream, false, dev, sl.d_packed, d_rects, d_first, n_rects, width));
	RMD_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the caller's buffer and the table are free again
	return rmd::check_fault(ctx); // (this wait has also waited for every render enqueued before it)
}
rmd_status rmd_framebuffer_upload_tiles(rmd_context *ctx, const double *host_packed, double *dev, uint32_t width, uint32_t height,
                                        const rmd_tile_rect *rects, uint32_t n_rects) {
	return rmd::guarded(ctx, "rmd_framebuffer_upload_tiles", [&] { return upload_tiles_impl(ctx, host_packed, dev, width, height, rects, n_rects); });
}

rmd_status rmd_host_alloc(rmd_context *ctx, size_t bytes, void **out_host) {
	if (rmd_status s = bind(ctx)) return s;
	if (!out_host || bytes == 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_host_alloc: bad argument");
	*out_host = nullptr;
	RMD_HIP(ctx, hipHostMalloc(out_host, bytes, hipHostMallocDefault));
	return RMD_OK;
}
rmd_status rmd_host_free(rmd_context *ctx, void *host) { // (ctx may be NULL: a block may outlive the context it was allocated through)
	if (ctx)
		if (rmd_status s = bind(ctx)) return s;
	if (host) RMD_HIP(ctx, hipHostFree(host));
	return RMD_OK;
}

rmd_status rmd_last_kernel_ms(rmd_context *ctx, float *out_ms) {
	if (rmd_status s = bind(ctx)) return s;
	if (!out_ms) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_last_kernel_ms: null pointer");
	if (!ctx->timed) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_last_kernel_ms: no render has been enqueued on this context");
	RMD_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
	RMD_HIP(ctx, hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
	return rmd::check_fault(ctx);
}

rmd_status rmd_last_launch_info(const rmd_context *ctx, rmd_launch_info *out) {
	if (!ctx || !out) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_last_launch_info: null argument");
	*out = ctx->last_launch;
	return RMD_OK;
}

static rmd_status resolve_tonemap_impl(rmd_context *ctx, const double *accum_dev, uint32_t width, uint32_t height, uint32_t sample_count,
                                       double exposure, double gamma, uint8_t *out_rgb8_host, uint8_t *&d) {
	if (rmd_status s = bind(ctx)) return s;
	if (!accum_dev || !out_rgb8_host || width == 0 || height == 0 || sample_count == 0)
		return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_resolve_tonemap: bad argument");
	const size_t n_pixels = (size_t)width * height;
	if (n_pixels > 0xFFFFFFFFull) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_resolve_tonemap: more than 2^32-1 pixels");
	// device buffer: [rgb8: 3 n bytes, padded to 4][count: 1 word][flagged pixel indices: n words]
	const size_t rgb_bytes = (n_pixels * 3 + 3) & ~(size_t)3;
	RMD_HIP(ctx, hipMalloc((void **)&d, rgb_bytes + 4 + n_pixels * 4));
	uint32_t *d_count = reinterpret_cast<uint32_t *>(d + rgb_bytes), *d_list = d_count + 1;
	const double sc = (double)sample_count, inv_gamma = 1.0 / gamma;
	uint32_t n_flagged = 0;
	std::vector<uint32_t> list;
	std::vector<double> px;
	hipError_t e = hipMemsetAsync(d_count, 0, 4, ctx->stream);
	if (e == hipSuccess) e = rmd::launch_tonemap(ctx->stream, accum_dev, d, n_pixels, sc, exposure, inv_gamma, d_list, d_count);
	if (e == hipSuccess) e = hipMemcpyAsync(out_rgb8_host, d, n_pixels * 3, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(&n_flagged, d_count, 4, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess && n_flagged != 0) {
		// the pixels whose byte an ulp of exp / pow could decide: recomputed with the host libm, as the reference does every pixel
		list.resize(n_flagged);
		e = hipMemcpy(list.data(), d_list, (size_t)n_flagged * 4, hipMemcpyDeviceToHost);
		const bool whole = n_flagged > 1024u; // many (a synthetic frame): one download of the frame instead of a copy per pixel
		if (e == hipSuccess && whole) {
			px.resize(n_pixels * 3);
			e = hipMemcpy(px.data(), accum_dev, n_pixels * 3 * sizeof(double), hipMemcpyDeviceToHost);
		}
		for (uint32_t k = 0; k < n_flagged && e == hipSuccess; k++) {
			const size_t i = list[k];
			double a[3];
			if (whole) a[0] = px[i * 3], a[1] = px[i * 3 + 1], a[2] = px[i * 3 + 2];
			else e = hipMemcpy(a, accum_dev + i * 3, sizeof(a), hipMemcpyDeviceToHost);
			if (e != hipSuccess) break;
			double v[3];
			bool ok = true;
			for (int c = 0; c < 3; c++) {
				const double p = a[c] / sc; // src/trace.rs:95
				double tm = 1.0 - std::exp(p * -1.0 * exposure); // cli_old/src/main.rs:165
				tm = std::pow(tm, inv_gamma);                    // :166
				v[c] = tm * 255.0;
				ok = ok && (v[c] > -1.0 && v[c] < 256.0); // :176 cast::<u8>()
			}
			for (int c = 0; c < 3; c++) out_rgb8_host[i * 3 + c] = ok ? (uint8_t)v[c] : (uint8_t)0;
		}
	}
	RMD_HIP(ctx, e);
	// the frame came from launches this call has waited for: one that was cut short by a device fault is not handed out as an image
	return rmd::check_fault(ctx);
}
rmd_status rmd_resolve_tonemap(rmd_context *ctx, const double *accum_dev, uint32_t width, uint32_t height, uint32_t sample_count,
                               double exposure, double gamma, uint8_t *out_rgb8_host) {
	uint8_t *d = nullptr; // the device scratch: freed here whichever way the body leaves
	const rmd_status s = rmd::guarded(ctx, "rmd_resolve_tonemap", [&] { return resolve_tonemap_impl(ctx, accum_dev, width, height, sample_count, exposure, gamma, out_rgb8_host, d); });
	if (d) (void)hipFree(d);
	return s;
}

} // extern "C"
