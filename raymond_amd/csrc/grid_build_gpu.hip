// GPU-side AccGrid::build_from_mesh (reference core/src/geometry/acc_grid.rs:36-83) — SURVEY.md section 8f row N4.
// Produces the same cells / mapping_table bytes as the host builder (grid_build.cpp) and the oracle:
//   1. bounds_kernel   Mesh::find_mesh_bounds (mesh.rs:123-140): min/max over all vertices, seeded with the reference's
//                      odd constants (Q9); per-block partials, finished on the host (min/max are order-independent).
//   2. host            estimate_grid_resolution + cell_size (acc_grid.rs:6-17, :38) with the host libm pow, exactly as
//                      grid_build.cpp does, so the truncations agree.
//   3. count_kernel    per triangle: its cell range (:43-56) and one atomicAdd per overlapped cell.
//   4. scan            cells[c] = sum over c' < c of (1 + count[c'])  (:67-74), three-phase device scan.
//   5. fill_kernel     per triangle: claim a slot in each overlapped cell's run with an atomic cursor.
//   6. sort_kernel     per cell: sort its run ascending — the reference pushes indices while iterating the triangles in
//                      order (:42,:61), so every run is ascending; the atomics above fill it in arbitrary order.
#define RMD_WITH_HIP 1
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "internal.hpp"


namespace {

__constant__ double kSeedMinDev[3] = {125125.0, 1251251.0, 12512512.0};
__constant__ double kSeedMaxDev[3] = {-123125.0, -125123.0, -512123.0};

struct GridDims {
	double bmin[3], cell[3];
	unsigned long long res[3], n_cells;
};

__global__ __launch_bounds__(256) void bounds_kernel(const double *__restrict__ pos, unsigned long long n_tris, double *__restrict__ partial) {
	double mn[3] = {kSeedMinDev[0], kSeedMinDev[1], kSeedMinDev[2]}, mx[3] = {kSeedMaxDev[0], kSeedMaxDev[1], kSeedMaxDev[2]};
	for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_tris; i += (unsigned long long)gridDim.x * 256)
		for (int k = 0; k < 3; k++)
			for (int a = 0; a < 3; a++) {
				double v = pos[i * 9 + k * 3 + a];
				mn[a] = fmin(mn[a], v), mx[a] = fmax(mx[a], v);
			}
	__shared__ double red[6][256];
	for (int a = 0; a < 3; a++) red[a][threadIdx.x] = mn[a], red[3 + a][threadIdx.x] = mx[a];
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s)
			for (int a = 0; a < 3; a++) {
				red[a][threadIdx.x] = fmin(red[a][threadIdx.x], red[a][threadIdx.x + s]);
				red[3 + a][threadIdx.x] = fmax(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
			}
		__syncthreads();
	}
	if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

// triangle.rs:70-84 + acc_grid.rs:43-56: cell range of one triangle; false where the reference's usize cast fails
__device__ bool tri_range(const double *p9, const GridDims &g, unsigned lo[3], unsigned hi[3]) {
	for (int a = 0; a < 3; a++) {
		double mn = fmin(fmin(fmin(kSeedMinDev[a], p9[a]), p9[3 + a]), p9[6 + a]);
		double mx = fmax(fmax(fmax(kSeedMaxDev[a], p9[a]), p9[3 + a]), p9[6 + a]);
		double l = (mn - g.bmin[a]) / g.cell[a], h = (mx - g.bmin[a]) / g.cell[a];
		if (!(l > -1.0 && l < 18446744073709551616.0) || !(h > -1.0 && h < 18446744073709551616.0)) return false;
		unsigned long long ul = (unsigned long long)l, uh = (unsigned long long)h;
		lo[a] = (unsigned)(ul < g.res[a] - 1 ? ul : g.res[a] - 1);
		hi[a] = (unsigned)(uh < g.res[a] - 1 ? uh : g.res[a] - 1);
	}
	return true;
}

// mode 0: count[idx]++ ; mode 1: mapping[cells[idx] + 1 + cursor[idx]++] = triangle
__global__ __launch_bounds__(256) void scatter_kernel(int mode, const double *__restrict__ pos, unsigned long long n_tris, GridDims g,
                                                      unsigned *__restrict__ count, const unsigned *__restrict__ cells,
                                                      unsigned *__restrict__ mapping, int *__restrict__ error) {
	unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_tris) return;
	unsigned lo[3], hi[3];
	if (!tri_range(pos + i * 9, g, lo, hi)) {
		atomicExch(error, 1); // "Failed to cast cell bounds to usize" (acc_grid.rs:44-51)
		return;
	}
	for (unsigned long long z = lo[2]; z <= hi[2]; z++)
		for (unsigned long long y = lo[1]; y <= hi[1]; y++)
			for (unsigned long long x = lo[0]; x <= hi[0]; x++) {
				unsigned long long idx = x + g.res[0] * (y + z * g.res[2]); // :61 — res.z where res.y is meant (Q5)
				if (idx >= g.n_cells) {
					atomicExch(error, 2); // the reference panics: index out of bounds
					return;
				}
				unsigned slot = atomicAdd(&count[idx], 1u);
				if (mode == 1) mapping[cells[idx] + 1u + slot] = (unsigned)i;
			}
}

// three-phase exclusive scan of (count[c] + 1) in blocks of 1024 elements (256 threads x 4)
__global__ __launch_bounds__(256) void scan_block_sums(const unsigned *__restrict__ count, unsigned long long n, unsigned long long *__restrict__ block_sums) {
	__shared__ unsigned long long red[256];
	unsigned long long base = (unsigned long long)blockIdx.x * 1024, sum = 0;
	for (int k = 0; k < 4; k++) {
		unsigned long long i = base + threadIdx.x * 4 + k;
		if (i < n) sum += (unsigned long long)count[i] + 1ull;
	}
	red[threadIdx.x] = sum;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0];
}
__global__ void scan_of_sums(unsigned long long *block_sums, unsigned n_blocks, unsigned long long *total) { // one thread: n_blocks is small
	unsigned long long run = 0;
	for (unsigned b = 0; b < n_blocks; b++) {
		unsigned long long v = block_sums[b];
		block_sums[b] = run;
		run += v;
	}
	*total = run;
}
__global__ __launch_bounds__(256) void scan_write(const unsigned *__restrict__ count, unsigned long long n, const unsigned long long *__restrict__ block_sums,
                                                  unsigned *__restrict__ cells, unsigned *__restrict__ mapping) {
	__shared__ unsigned long long pre[256];
	unsigned long long base = (unsigned long long)blockIdx.x * 1024, v[4], sum = 0;
	for (int k = 0; k < 4; k++) {
		unsigned long long i = base + threadIdx.x * 4 + k;
		v[k] = i < n ? (unsigned long long)count[i] + 1ull : 0ull;
		sum += v[k];
	}
	pre[threadIdx.x] = sum;
	__syncthreads();
	// Hillis-Steele inclusive scan over the 256 thread sums
	for (int d = 1; d < 256; d <<= 1) {
		unsigned long long add = (int)threadIdx.x >= d ? pre[threadIdx.x - d] : 0ull;
		__syncthreads();
		pre[threadIdx.x] += add;
		__syncthreads();
	}
	unsigned long long run = block_sums[blockIdx.x] + pre[threadIdx.x] - sum;
	for (int k = 0; k < 4; k++) {
		unsigned long long i = base + threadIdx.x * 4 + k;
		if (i < n) {
			cells[i] = (unsigned)run;       // acc_grid.rs:69
			mapping[run] = count[i];        // :70
			run += v[k];
		}
	}
}

__global__ __launch_bounds__(256) void sort_kernel(const unsigned *__restrict__ cells, unsigned long long n_cells, unsigned *__restrict__ mapping) {
	unsigned long long c = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
	if (c >= n_cells) return;
	unsigned off = cells[c], n = mapping[off];
	unsigned *run = mapping + off + 1;
	for (unsigned i = 1; i < n; i++) { // insertion sort: runs hold a handful to a few dozen indices
		unsigned key = run[i], j = i;
		while (j > 0 && run[j - 1] > key) {
			run[j] = run[j - 1];
			j--;
		}
		run[j] = key;
	}
}

struct Dev {
	std::vector<void *> ptrs;
	~Dev() {
		for (void *p : ptrs) (void)hipFree(p);
	}
	hipError_t alloc(void **p, size_t bytes) {
		hipError_t e = hipMalloc(p, bytes ? bytes : 16);
		if (e == hipSuccess) ptrs.push_back(*p);
		return e;
	}
};

#define RMD_HIP(ctx, call)                                                                            \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) return rmd::fail(ctx, RMD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
	} while (0)

} // namespace

static rmd_status grid_build_from_mesh_gpu_impl(rmd_context *ctx, const double *tri_pos, const double *tri_nrm, uint64_t n_tris, rmd_grid_build **out) {
	if (!ctx) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_grid_build_from_mesh_gpu: null context");
	if (!tri_pos || !tri_nrm || !out || n_tris == 0) return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_grid_build_from_mesh_gpu: null/empty input");
	if (n_tris >= (1ull << 32)) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "rmd_grid_build_from_mesh_gpu: more than 2^32-1 triangles");
	RMD_HIP(ctx, hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	Dev dev;
	double *d_pos = nullptr, *d_partial = nullptr;
	RMD_HIP(ctx, dev.alloc((void **)&d_pos, n_tris * 9 * sizeof(double)));
	RMD_HIP(ctx, hipMemcpyAsync(d_pos, tri_pos, n_tris * 9 * sizeof(double), hipMemcpyHostToDevice, st));
	const unsigned bounds_blocks = (unsigned)std::min<uint64_t>((n_tris + 255) / 256, 1024);
	RMD_HIP(ctx, dev.alloc((void **)&d_partial, bounds_blocks * 6 * sizeof(double)));
	hipLaunchKernelGGL(bounds_kernel, dim3(bounds_blocks), dim3(256), 0, st, d_pos, (unsigned long long)n_tris, d_partial);
	RMD_HIP(ctx, hipGetLastError());
	std::vector<double> partial(bounds_blocks * 6);
	RMD_HIP(ctx, hipMemcpyAsync(partial.data(), d_partial, partial.size() * sizeof(double), hipMemcpyDeviceToHost, st));
	RMD_HIP(ctx, hipStreamSynchronize(st));

	std::unique_ptr<rmd_grid_build> g(new (std::nothrow) rmd_grid_build());
	if (!g) return rmd::fail(ctx, RMD_ERR_OUT_OF_MEMORY, "rmd_grid_build_from_mesh_gpu: allocation failed");
	for (int a = 0; a < 3; a++) {
		g->bbox_min[a] = partial[a], g->bbox_max[a] = partial[3 + a];
		for (unsigned b = 1; b < bounds_blocks; b++) {
			g->bbox_min[a] = std::fmin(g->bbox_min[a], partial[b * 6 + a]);
			g->bbox_max[a] = std::fmax(g->bbox_max[a], partial[b * 6 + 3 + a]);
		}
	}
	// estimate_grid_resolution (acc_grid.rs:6-17) and cell_size (:38): same expressions, same libm as grid_build.cpp
	double size[3];
	for (int a = 0; a < 3; a++) size[a] = g->bbox_max[a] - g->bbox_min[a];
	const double volume = std::fabs(size[0] * size[1] * size[2]);
	const double density = std::pow((3.0 * (double)n_tris) / volume, 1.0 / 3.0);
	GridDims dims;
	for (int a = 0; a < 3; a++) {
		double v = std::fabs(size[a]) * density;
		uint64_t r = (!(v == v) || v <= 0.0) ? 0 : (v >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)v);
		if (r == 0 || r > 0xFFFFFFFFull)
			return rmd::fail(ctx, RMD_ERR_GRID_INDEX, "grid resolution has a zero axis (reference underflows `grid_res[i] - 1`, acc_grid.rs:54)");
		dims.res[a] = r, g->res[a] = (uint32_t)r;
		g->cell_size[a] = size[a] / (double)r;
		dims.bmin[a] = g->bbox_min[a], dims.cell[a] = g->cell_size[a];
	}
	dims.n_cells = dims.res[0] * dims.res[1] * dims.res[2];
	if (dims.n_cells > (1ull << 31)) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "grid has more than 2^31 cells");

	const unsigned long long n_cells = dims.n_cells;
	const unsigned tri_blocks = (unsigned)((n_tris + 255) / 256), scan_blocks = (unsigned)((n_cells + 1023) / 1024);
	unsigned *d_count = nullptr, *d_cells = nullptr, *d_map = nullptr;
	unsigned long long *d_sums = nullptr, *d_total = nullptr;
	int *d_err = nullptr;
	RMD_HIP(ctx, dev.alloc((void **)&d_count, n_cells * sizeof(unsigned)));
	RMD_HIP(ctx, dev.alloc((void **)&d_cells, n_cells * sizeof(unsigned)));
	RMD_HIP(ctx, dev.alloc((void **)&d_sums, scan_blocks * sizeof(unsigned long long)));
	RMD_HIP(ctx, dev.alloc((void **)&d_total, sizeof(unsigned long long)));
	RMD_HIP(ctx, dev.alloc((void **)&d_err, sizeof(int)));
	RMD_HIP(ctx, hipMemsetAsync(d_count, 0, n_cells * sizeof(unsigned), st));
	RMD_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(int), st));
	hipLaunchKernelGGL(scatter_kernel, dim3(tri_blocks), dim3(256), 0, st, 0, d_pos, (unsigned long long)n_tris, dims, d_count, (const unsigned *)nullptr,
	                   (unsigned *)nullptr, d_err);
	hipLaunchKernelGGL(scan_block_sums, dim3(scan_blocks), dim3(256), 0, st, d_count, n_cells, d_sums);
	hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(1), 0, st, d_sums, scan_blocks, d_total);
	RMD_HIP(ctx, hipGetLastError());
	unsigned long long total = 0;
	int err = 0;
	RMD_HIP(ctx, hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, st));
	RMD_HIP(ctx, hipMemcpyAsync(&err, d_err, sizeof(err), hipMemcpyDeviceToHost, st));
	RMD_HIP(ctx, hipStreamSynchronize(st));
	if (err == 1) return rmd::fail(ctx, RMD_ERR_GRID_INDEX, "cell bound does not fit usize (reference: \"Failed to cast cell bounds to usize\", acc_grid.rs:44-51)");
	if (err == 2) return rmd::fail(ctx, RMD_ERR_GRID_INDEX, "cell index past the cell array (reference panics at acc_grid.rs:61)");
	if (total > 0xFFFFFFFFull) return rmd::fail(ctx, RMD_ERR_UNSUPPORTED, "mapping_table exceeds 2^32 entries");
	RMD_HIP(ctx, dev.alloc((void **)&d_map, total * sizeof(unsigned)));
	hipLaunchKernelGGL(scan_write, dim3(scan_blocks), dim3(256), 0, st, d_count, n_cells, d_sums, d_cells, d_map);
	RMD_HIP(ctx, hipMemsetAsync(d_count, 0, n_cells * sizeof(unsigned), st)); // reused as the per-cell fill cursor
	hipLaunchKernelGGL(scatter_kernel, dim3(tri_blocks), dim3(256), 0, st, 1, d_pos, (unsigned long long)n_tris, dims, d_count, d_cells, d_map, d_err);
	hipLaunchKernelGGL(sort_kernel, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, st, d_cells, n_cells, d_map);
	RMD_HIP(ctx, hipGetLastError());
	g->cells.resize(n_cells);
	g->mapping.resize(total);
	RMD_HIP(ctx, hipMemcpyAsync(g->cells.data(), d_cells, n_cells * sizeof(unsigned), hipMemcpyDeviceToHost, st));
	RMD_HIP(ctx, hipMemcpyAsync(g->mapping.data(), d_map, total * sizeof(unsigned), hipMemcpyDeviceToHost, st));
	RMD_HIP(ctx, hipStreamSynchronize(st));
	g->pos.assign(tri_pos, tri_pos + n_tris * 9);
	g->nrm.assign(tri_nrm, tri_nrm + n_tris * 9);
	*out = g.release();
	return RMD_OK;
}
// (host vectors for the partial bounds and the downloaded tables, device buffers behind RAII: nothing throws across the boundary)
extern "C" rmd_status rmd_grid_build_from_mesh_gpu(rmd_context *ctx, const double *tri_pos, const double *tri_nrm, uint64_t n_tris,
                                                   rmd_grid_build **out) {
	if (out) *out = nullptr;
	return rmd::guarded(ctx, "rmd_grid_build_from_mesh_gpu", [&] { return grid_build_from_mesh_gpu_impl(ctx, tri_pos, tri_nrm, n_tris, out); });
}
