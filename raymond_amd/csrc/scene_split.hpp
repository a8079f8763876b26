// Scene::intersect (core/src/scene.rs:54-74) in two parts, for the render loop of scenes with grids (render_kernel.hpp).
#pragma once
#include "device_core.hpp"
#include "grid_walk.hpp"

namespace rmd {

// The same scan in two parts, for the render loop of a scene with grids.  Scene::intersect keeps the first object on
// distance ties, i.e. it returns the lexicographic minimum of (distance, object index) — so the objects may be visited
// in any order as long as candidates are merged with that rule.  intersect_simple() visits planes and spheres and
// reports whether the ray enters any grid's bounding box (acc_grid.rs:90); intersect_grids() later runs the cooperative
// walks of the grid objects for the lanes that do, and merges.  Splitting the scan lets a lane WAIT for its walk until
// enough other lanes of the wave need one too (RenderParams::walk_batch): a walk phase costs about the same for 15 rays as for 45,
// because the wave steps until its longest walk ends either way.
RMD_DEV bool lex_less(double t, int obj, double t_best, int obj_best) { return t < t_best || (t == t_best && obj < obj_best); }
#ifndef RMD_FLAT_OBJECT_TESTS
#define RMD_FLAT_OBJECT_TESTS 1
#endif
RMD_DEV bool intersect_simple(const DevObject *__restrict__ objs, uint32_t n_objects, const DevGrid *__restrict__ grids, bool want, V3 ro, V3 rd,
                              double &closest, int &best) {
	closest = scalar_const(kFMax), best = -1;
	bool enters = false;
#if RMD_FLAT_OBJECT_TESTS
	// tests without control flow, the running minimum updated by selects (device_core.hpp: *_test_flat)
	for (uint32_t i = 0; i < n_objects; i++) {
		const DevObject &o = objs[i];
		double t;
		bool ok;
		int idx = (int)i;
		if (o.geometry_kind == 0u) {
			if (o.pair_info != 0u) {
				if (o.pair_info & kPairTestedAtPartner) continue;
				const uint32_t e = o.pair_info - 1u;
				bool first;
				ok = plane_pair_test_flat(ld3(objs[e].origin), ld3(objs[e].normal), ld3(o.origin), ld3(o.normal), ro, rd, t, first);
				idx = first ? (int)e : (int)i;
				ok = ok && want && lex_less(t, idx, closest, best);
			} else {
				ok = plane_test_flat(ld3(o.origin), ld3(o.normal), ro, rd, t) && want && t < closest; // index order + strict '<' = the lexicographic minimum so far
			}
		} else if (o.geometry_kind == 1u) {
			ok = sphere_test_flat(ld3(o.origin), o.radius, ro, rd, t) && want && t < closest;
		} else {
			const DevGrid &g = grids[o.grid_index];
			double t_outer;
			if (want && aabb_intersect(ld3(g.bbox_min), ld3(g.bbox_max), ro, rd, t_outer)) enters = true;
			continue;
		}
		closest = ok ? t : closest, best = ok ? idx : best;
	}
	return enters;
#endif
	for (uint32_t i = 0; i < n_objects; i++) {
		const DevObject &o = objs[i];
		if (o.geometry_kind == 0u) {
			if (o.pair_info != 0u) { // a plane with an exactly opposite partner (device_core.hpp: plane_pair_visit), tested at the later one's turn
				if (o.pair_info & kPairTestedAtPartner) continue;
				const uint32_t e = o.pair_info - 1u;
				if (want)
					plane_pair_visit(ld3(objs[e].origin), ld3(objs[e].normal), ld3(o.origin), ld3(o.normal), ro, rd, [&](double t, bool first) {
						const int idx = first ? (int)e : (int)i;
						if (lex_less(t, idx, closest, best)) closest = t, best = idx;
					});
				continue;
			}
			// index order + strict '<' = the lexicographic minimum so far
			if (want) plane_visit(ld3(o.origin), ld3(o.normal), ro, rd, [&](double t) { if (t < closest) closest = t, best = (int)i; });
		} else if (o.geometry_kind == 1u) {
			if (want) sphere_visit(ld3(o.origin), o.radius, ro, rd, [&](double t) { if (t < closest) closest = t, best = (int)i; });
		} else {
			const DevGrid &g = grids[o.grid_index];
			double t_outer;
			if (want && aabb_intersect(ld3(g.bbox_min), ld3(g.bbox_max), ro, rd, t_outer)) enters = true;
		}
	}
	return enters;
}
template <bool DEEP = false>
RMD_DEV void intersect_grids(const DevObject *__restrict__ objs, uint32_t n_objects, const DevGrid *__restrict__ grids, const uint32_t *lds_masks,
                             WalkScratch &scr, bool walkers, V3 ro, V3 rd, double &closest, int &best, uint32_t &sub, uint32_t debug_flags,
                             unsigned long long *dbg, uint32_t cut_lanes = 0u, WalkCarry *carry = nullptr, bool *carried = nullptr, uint32_t cut_round = 0u) {
	for (uint32_t i = 0; i < n_objects; i++) {
		const DevObject &o = objs[i];
		if (o.geometry_kind != 2u) continue; // uniform
		const DevGrid &g = grids[o.grid_index];
		bool hit = false;
		double t = 0.0;
		uint32_t tri = 0;
		// (walks are only put aside in scenes with ONE grid object — api.cpp: walk_cut — so `carry` holds the state of this object's walk)
		grid_intersect_wave<DEEP>(g, lds_masks + g.mask_lds_word, scr, walkers, ro, rd, hit, t, tri, debug_flags, dbg, cut_lanes, carry, carried, cut_round);
		if (walkers && hit && lex_less(t, (int)i, closest, best)) closest = t, best = (int)i, sub = tri;
	}
}

} // namespace rmd
