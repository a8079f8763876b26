// Host-side AccGrid::build_from_mesh (reference core/src/geometry/acc_grid.rs:6-83) producing the
// compact u32 layout of rmd_grid_desc.  Two counting passes build the cells/mapping_table arrays
// directly (the reference goes through a Vec<Vec<usize>> intermediate, :39,:58-74); the resulting
// bytes are the same: cells in index order, each run = [count, triangle indices ascending].
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/raymond_hip.h"
#include "internal.hpp"


namespace {

// Triangle::find_bounds / Mesh::find_mesh_bounds share these seeds (triangle.rs:71-72, mesh.rs:124-125)
const double kSeedMin[3] = {125125.0, 1251251.0, 12512512.0};
const double kSeedMax[3] = {-123125.0, -125123.0, -512123.0};

inline void tri_bounds(const double *p9, double mn[3], double mx[3]) {
	for (int a = 0; a < 3; a++) {
		mn[a] = std::fmin(std::fmin(std::fmin(kSeedMin[a], p9[a]), p9[3 + a]), p9[6 + a]);
		mx[a] = std::fmax(std::fmax(std::fmax(kSeedMax[a], p9[a]), p9[3 + a]), p9[6 + a]);
	}
}

// num-traits NumCast f64 -> usize: truncation, None on NaN / negative beyond -1 / too large
inline bool to_usize(double v, uint64_t &out) {
	if (!(v > -1.0 && v < 18446744073709551616.0)) return false;
	out = (uint64_t)v;
	return true;
}
// Rust `as usize`: saturating, NaN -> 0
inline uint64_t as_usize(double v) {
	if (!(v == v) || v <= 0.0) return 0;
	if (v >= 18446744073709551616.0) return UINT64_MAX;
	return (uint64_t)v;
}

} // namespace

static rmd_status grid_build_from_mesh_impl(const double *tri_pos, const double *tri_nrm, uint64_t n_tris, rmd_grid_build **out, rmd_grid_build *&g) {
	if (!tri_pos || !tri_nrm || !out || n_tris == 0) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_grid_build_from_mesh: null/empty input");
	if (n_tris >= (1ull << 32)) return rmd::fail(nullptr, RMD_ERR_UNSUPPORTED, "rmd_grid_build_from_mesh: more than 2^32-1 triangles");
	g = new (std::nothrow) rmd_grid_build();
	if (!g) return rmd::fail(nullptr, RMD_ERR_OUT_OF_MEMORY, "rmd_grid_build_from_mesh: allocation failed");

	// Mesh::find_mesh_bounds (mesh.rs:123-140)
	for (int a = 0; a < 3; a++) g->bbox_min[a] = kSeedMin[a], g->bbox_max[a] = kSeedMax[a];
	for (uint64_t i = 0; i < n_tris; i++)
		for (int k = 0; k < 3; k++)
			for (int a = 0; a < 3; a++) {
				double v = tri_pos[i * 9 + k * 3 + a];
				g->bbox_min[a] = std::fmin(g->bbox_min[a], v);
				g->bbox_max[a] = std::fmax(g->bbox_max[a], v);
			}

	// estimate_grid_resolution (acc_grid.rs:6-17)
	double size[3];
	for (int a = 0; a < 3; a++) size[a] = g->bbox_max[a] - g->bbox_min[a];
	double volume = std::fabs(size[0] * size[1] * size[2]);
	double density = std::pow((3.0 * (double)n_tris) / volume, 1.0 / 3.0);
	uint64_t res[3];
	for (int a = 0; a < 3; a++) res[a] = as_usize(std::fabs(size[a]) * density);
	if (res[0] == 0 || res[1] == 0 || res[2] == 0 || res[0] > 0xFFFFFFFFull || res[1] > 0xFFFFFFFFull || res[2] > 0xFFFFFFFFull) {
		return rmd::fail(nullptr, RMD_ERR_GRID_INDEX, "grid resolution has a zero axis (reference underflows `grid_res[i] - 1`, acc_grid.rs:54)");
	}
	const uint64_t n_cells = res[0] * res[1] * res[2];
	if (n_cells > (1ull << 31)) {
		return rmd::fail(nullptr, RMD_ERR_UNSUPPORTED, "grid has more than 2^31 cells");
	}
	for (int a = 0; a < 3; a++) {
		g->res[a] = (uint32_t)res[a];
		g->cell_size[a] = size[a] / (double)res[a]; // :38
	}

	// pass 1: per-triangle cell ranges (:43-56) and per-cell counts
	struct Range {
		uint32_t lo[3], hi[3];
	};
	std::vector<Range> ranges(n_tris);
	std::vector<uint32_t> count(n_cells, 0u);
	for (uint64_t i = 0; i < n_tris; i++) {
		double mn[3], mx[3];
		tri_bounds(tri_pos + i * 9, mn, mx);
		for (int a = 0; a < 3; a++) {
			uint64_t lo, hi;
			if (!to_usize((mn[a] - g->bbox_min[a]) / g->cell_size[a], lo) || !to_usize((mx[a] - g->bbox_min[a]) / g->cell_size[a], hi)) {
				return rmd::fail(nullptr, RMD_ERR_GRID_INDEX, "cell bound does not fit usize (reference: \"Failed to cast cell bounds to usize\", acc_grid.rs:44-51)");
			}
			ranges[i].lo[a] = (uint32_t)(lo < res[a] - 1 ? lo : res[a] - 1);
			ranges[i].hi[a] = (uint32_t)(hi < res[a] - 1 ? hi : res[a] - 1);
		}
		const Range &r = ranges[i];
		for (uint64_t z = r.lo[2]; z <= r.hi[2]; z++)
			for (uint64_t y = r.lo[1]; y <= r.hi[1]; y++)
				for (uint64_t x = r.lo[0]; x <= r.hi[0]; x++) {
					uint64_t idx = x + res[0] * (y + z * res[2]); // :61 — res.z where res.y is meant
					if (idx >= n_cells) {
						return rmd::fail(nullptr, RMD_ERR_GRID_INDEX, "cell index past the cell array (reference panics at acc_grid.rs:61)");
					}
					count[idx]++;
				}
	}
	// offsets (:67-74): cells[c] = start of the run, run = [count, indices...]
	uint64_t total = 0;
	g->cells.resize(n_cells);
	for (uint64_t c = 0; c < n_cells; c++) {
		if (total > 0xFFFFFFFFull) break;
		g->cells[c] = (uint32_t)total;
		total += 1ull + count[c];
	}
	if (total > 0xFFFFFFFFull) {
		return rmd::fail(nullptr, RMD_ERR_UNSUPPORTED, "mapping_table exceeds 2^32 entries");
	}
	g->mapping.assign(total, 0u);
	std::vector<uint32_t> fill(n_cells, 0u);
	for (uint64_t c = 0; c < n_cells; c++) g->mapping[g->cells[c]] = count[c];
	// pass 2: triangle indices in ascending order per cell (the reference pushes while iterating triangles in order)
	for (uint64_t i = 0; i < n_tris; i++) {
		const Range &r = ranges[i];
		for (uint64_t z = r.lo[2]; z <= r.hi[2]; z++)
			for (uint64_t y = r.lo[1]; y <= r.hi[1]; y++)
				for (uint64_t x = r.lo[0]; x <= r.hi[0]; x++) {
					uint64_t idx = x + res[0] * (y + z * res[2]);
					g->mapping[g->cells[idx] + 1u + fill[idx]++] = (uint32_t)i;
				}
	}
	g->pos.assign(tri_pos, tri_pos + n_tris * 9);
	g->nrm.assign(tri_nrm, tri_nrm + n_tris * 9);
	*out = g;
	g = nullptr; // handed over
	return RMD_OK;
}
// (the per-cell lists are a std::vector of std::vectors over every cell: nothing throws across the boundary — a mesh whose tables the host cannot
// hold comes back as RMD_ERR_OUT_OF_MEMORY where the reference's `Vec` would abort the process)
extern "C" rmd_status rmd_grid_build_from_mesh(const double *tri_pos, const double *tri_nrm, uint64_t n_tris, rmd_grid_build **out) {
	if (out) *out = nullptr;
	rmd_grid_build *g = nullptr;
	const rmd_status s = rmd::guarded(nullptr, "rmd_grid_build_from_mesh", [&] { return grid_build_from_mesh_impl(tri_pos, tri_nrm, n_tris, out, g); });
	delete g; // (whatever a failed build has left)
	return s;
}

extern "C" rmd_status rmd_grid_build_describe(const rmd_grid_build *g, rmd_grid_desc *d) {
	if (!g || !d) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_grid_build_describe: null argument");
	std::memset(d, 0, sizeof(*d));
	for (int a = 0; a < 3; a++) {
		d->bbox_min[a] = g->bbox_min[a], d->bbox_max[a] = g->bbox_max[a];
		d->resolution[a] = g->res[a], d->cell_size[a] = g->cell_size[a];
	}
	d->cells = g->cells.data(), d->n_cells = g->cells.size();
	d->mapping_table = g->mapping.data(), d->n_mapping = g->mapping.size();
	d->tri_pos = g->pos.data(), d->tri_nrm = g->nrm.data(), d->n_tris = g->pos.size() / 9;
	d->built = g; // rmd_scene_create keeps what it derives from these arrays in the build (valid until rmd_grid_build_destroy, like the pointers above)
	return RMD_OK;
}

extern "C" void rmd_grid_build_destroy(rmd_grid_build *g) { delete g; }
