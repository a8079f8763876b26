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
// An object loop's next turn: objects 0 .. 63 by the bits of a mask made at scene creation (RenderParams::visit_mask / grid_mask: an object the
// loop has nothing to do for would cost it a scalar round trip to find that out), objects from 64 on one by one.  Start with i = ~0u.  All-ones =
// every object (the probes).  The order of the turns is the index order either way.
#ifndef RMD_VISIT_MASKS
#define RMD_VISIT_MASKS 1
#endif
RMD_DEV uint32_t next_turn(uint32_t i, unsigned long long &m) {
#if RMD_VISIT_MASKS
	if (m != 0ull) {
		i = (uint32_t)__builtin_ctzll(m);
		m &= m - 1ull;
		return i;
	}
	return i + 1u < 64u ? 64u : i + 1u;
#else
	return i + 1u;
#endif
}
RMD_DEV bool lex_less(double t, int obj, double t_best, int obj_best) { return (t < t_best) | ((t == t_best) & (obj < obj_best)); } // (no short circuit: three compares and two scalar mask operations, no branch)
// The walls of an axis-aligned room: up to three pairs of opposite planes with normals exactly +e_k / -e_k (RenderParams::axis_pairs, made by
// rmd_scene_create in scenes of regular parameters), tested ahead of the object loops — which pass these planes by — with ONE component of the ray
// where plane.rs:11-24 takes three dot products (19 f64 operations fewer per pair).  Same bits, because:
//   * the scene is regular and a ray's origin and direction are therefore finite in every component, or NaN in every component (a degenerate
//     mesh normal); with n = +-e_k the dot products (n.x*v.x + n.y*v.y) + n.z*v.z have two terms that are +-0 and one that is +-v_k, and
//     x + (+-0) = x for every x that is not itself a zero — a NaN stays a NaN either way;
//   * a ZERO denominator may come out with the other sign, which no comparison sees (`denom > 1e-6`), and a zero NUMERATOR (a ray that starts on the
//     plane's coordinate) too, which the quotient's sign would show: one ballot sends the wave down the general test for that pair (never, in practice);
//   * the plane with normal +e_k faces rays with -rd_k > 1e-6 and its numerator dot(o - ro, -n) is -(o_k - ro_k); the other one faces rd_k > 1e-6
//     with numerator o_k - ro_k; the facing conditions exclude each other and the division is numerator / |rd_k| — plane_pair_test_flat's operands.
// Hits are merged with the lexicographic rule (distance, object index), so the order in which the planes are visited does not matter
// (core/src/scene.rs:54-74 keeps the first object of the closest distance).
template <int K>
RMD_DEV void axis_pair_test(const DevObject *__restrict__ objs, [[maybe_unused]] const AxisWalls *__restrict__ walls, uint32_t j, bool want, V3 ro, V3 rd, double &closest, int &best, bool arbitrary_rays) {
#ifndef RMD_AXIS_WALLS_TABLE
#define RMD_AXIS_WALLS_TABLE 1
#endif
#if RMD_AXIS_WALLS_TABLE
	const AxisWalls &w = walls[K]; // (device_types.hpp: behind the table's last record)
	const double o_plus = w.o_plus, o_minus = w.o_minus; // the planes with normal +e_k / -e_k (uniform)
	const int idx_plus = (int)w.idx_plus, idx_minus = (int)w.idx_minus;
#else
	const DevObject &o = objs[j];
	const uint32_t e = o.pair_info & 0x3FFFFFFFu; // the earlier plane of the pair
	const bool e_plus = (o.flags & kObjAxisEarlierIsPlus) != 0u;
	const double o_e = o.partner_origin_k, o_j = o.origin[K]; // (the partner's coordinate sits in this object's record: one round of scalar loads, not two)
	const double o_plus = e_plus ? o_e : o_j, o_minus = e_plus ? o_j : o_e; // the planes with normal +e_k / -e_k (uniform)
	const int idx_plus = e_plus ? (int)e : (int)j, idx_minus = e_plus ? (int)j : (int)e;
#endif
	const double rk = K == 0 ? rd.x : K == 1 ? rd.y : rd.z, pk = K == 0 ? ro.x : K == 1 ? ro.y : ro.z;
	const bool faces_plus = -rk > 1e-6, faces_minus = rk > 1e-6;
	double num_minus = o_minus - pk, num_plus = -(o_plus - pk);
	asm volatile("" : "+v"(num_minus), "+v"(num_plus)); // (both computed, one select: with the walls' coordinates loaded the compiler makes two divergent branches of it)
	const double num = faces_minus ? num_minus : num_plus;
	// (arbitrary_rays: the probes' Scene::intersect on rays given by a test, which may be non-finite in SOME components: always the general test)
	// The numerator's class as ONE unsigned range test of its high word: 2^-700 <= |num| < 2^700 (0x143 .. 0x6BB biased).  Anything else — a ZERO, whose
	// sign is the full dot product's, a denormal, an infinity, a NaN (a lane that carries no ray may hold anything) — sends the wave down the general
	// test for this pair; inside the range the quotient below is the IEEE quotient without the division's scaling and special-case instructions
	// (device_core.hpp: div_lean — the divisor of a lane whose quotient counts is |rd_k| in (1e-6, 1]).
	const uint32_t num_hi = (uint32_t)(__builtin_bit_cast(unsigned long long, num) >> 32) & 0x7FFFFFFFu;
	// (only lanes that carry a ray count: the others may hold anything, and two ballots and a scalar AND cost no vector instruction)
	if (RMD_UNLIKELY(arbitrary_rays || (__builtin_amdgcn_ballot_w64(num_hi - 0x14300000u >= 0x6BB00000u - 0x14300000u) & __builtin_amdgcn_ballot_w64(want)) != 0ull)) {
		double t;
		bool first;
#if RMD_AXIS_WALLS_TABLE
		const DevObject &o = objs[j];
		const uint32_t e = o.pair_info & 0x3FFFFFFFu; // the earlier plane of the pair
#endif
		const bool hit = plane_pair_test_flat(ld3(objs[e].origin), ld3(objs[e].normal), ld3(o.origin), ld3(o.normal), ro, rd, t, first);
		const int idx = first ? (int)e : (int)j;
		const bool ok = hit && want && lex_less(t, idx, closest, best);
		closest = ok ? t : closest, best = ok ? idx : best;
		return;
	}
	const double t = div_lean(num, __builtin_fabs(rk)); // (a lane that faces neither wall divides by whatever its |rd_k| is: nobody reads it)
	const int idx = faces_plus ? idx_plus : idx_minus;
	const bool ok = (faces_plus || faces_minus) && t >= 0.0 && want && lex_less(t, idx, closest, best);
	closest = ok ? t : closest, best = ok ? idx : best;
}
RMD_DEV void axis_pairs_visit(const DevObject *__restrict__ objs, uint32_t n_objects, uint32_t axis_pairs, bool want, V3 ro, V3 rd, double &closest, int &best,
                              bool arbitrary_rays = false) {
	const AxisWalls *walls = reinterpret_cast<const AxisWalls *>(objs + n_objects); // (one address for the three pairs)
	if (axis_pairs & 1023u) axis_pair_test<0>(objs, walls, (axis_pairs & 1023u) - 1u, want, ro, rd, closest, best, arbitrary_rays);
	if ((axis_pairs >> 10) & 1023u) axis_pair_test<1>(objs, walls, ((axis_pairs >> 10) & 1023u) - 1u, want, ro, rd, closest, best, arbitrary_rays);
	if ((axis_pairs >> 20) & 1023u) axis_pair_test<2>(objs, walls, ((axis_pairs >> 20) & 1023u) - 1u, want, ro, rd, closest, best, arbitrary_rays);
}
#ifndef RMD_FLAT_OBJECT_TESTS
#define RMD_FLAT_OBJECT_TESTS 1
#endif
RMD_DEV bool intersect_simple(const DevObject *__restrict__ objs, uint32_t n_objects, const DevGrid *__restrict__ grids, bool want, V3 ro, V3 rd,
                              double &closest, int &best, uint32_t axis_pairs, unsigned long long turns = ~0ull) {
	closest = scalar_const(kFMax), best = -1;
	bool enters = false;
	axis_pairs_visit(objs, n_objects, axis_pairs, want, ro, rd, closest, best);
#if RMD_FLAT_OBJECT_TESTS
	// tests without control flow, the running minimum updated by selects (device_core.hpp: *_test_flat)
	for (uint32_t i = next_turn(~0u, turns); i < n_objects; i = next_turn(i, turns)) {
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
				ok = plane_test_flat(ld3(o.origin), ld3(o.normal), ro, rd, t) && want && lex_less(t, idx, closest, best); // (the axis pairs have had their turn ahead of the loop: the rule that does not depend on the order)
			}
		} else if (o.geometry_kind == 1u) {
			ok = sphere_test_flat(ld3(o.origin), o.radius, ro, rd, t) && want && lex_less(t, idx, closest, best);
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
	for (uint32_t i = next_turn(~0u, turns); i < n_objects; i = next_turn(i, turns)) {
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
			if (want) plane_visit(ld3(o.origin), ld3(o.normal), ro, rd, [&](double t) { if (lex_less(t, (int)i, closest, best)) closest = t, best = (int)i; });
		} else if (o.geometry_kind == 1u) {
			if (want) sphere_visit(ld3(o.origin), o.radius, ro, rd, [&](double t) { if (lex_less(t, (int)i, closest, best)) closest = t, best = (int)i; });
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
                             unsigned long long *dbg, uint32_t cut_lanes = 0u, WalkCarry *carry = nullptr, bool *carried = nullptr, uint32_t cut_round = 0u,
                             unsigned long long turns = ~0ull) {
	for (uint32_t i = next_turn(~0u, turns); i < n_objects; i = next_turn(i, turns)) {
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
