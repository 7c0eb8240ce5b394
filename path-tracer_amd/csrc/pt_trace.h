// Ray / box / triangle / sphere tests and the short-stack KD traversal, one ray per lane.
//
// Semantics follow the reference function by function (cited at each one); what is new is
// the data layout (pt_scene.h), the LDS-resident traversal stack (one column per lane,
// stack[level][lane] so a wave's push/pop hits 64 different banks... 32 banks twice),
// and the loop shape: a lane runs "descend until a leaf" and "test the leaf's triangles"
// as two inner loops so that lanes of a wave spend their iterations in the same kind of
// work. The order of operations seen by ONE ray is exactly the reference's, so hit ids,
// visited leaves and tie-breaks are bit-identical whatever the scheduling.
#pragma once
#include "pt_math.h"
#include "pt_scene.h"

#define PT_TRI_EPSILON 0.00000005f   // reference: src/intersect.cuh:10

// PT_F_ALLTRIS (segment kernels): single-leaf meshes by the reference's loop over EVERY triangle of the leaf instead of through
// their pair boxes (pt_wavefront.h) - what trav_variant 15 / 7 / 9, the strict schedule, asks for
enum { PT_F_STATS = 1, PT_F_DEBUGMASK = 2, PT_F_ALLTRIS = 4 };

struct Ray {
	f3 o, d, inv;
	bool sx, sy, sz; // inv component < 0 (reference: src/pathtracing.cu:28-30)
};

// reference: src/pathtracing.cu:23-31 init_ray
PT_DEV void init_ray(Ray& r, f3 orig, f3 dir) {
	r.d = dir;
	r.o = orig;
	r.inv = pt_rcp3(dir);   // == 1.0f / dir.x, ... bit for bit (pt_math.h)
	r.sx = r.inv.x < 0.0f;
	r.sy = r.inv.y < 0.0f;
	r.sz = r.inv.z < 0.0f;
}

// Origin and direction only: for code that shades (the BSDFs, src/surfaces.cuh) and hands the ray on. The inverse and its
// signs are then NOT valid; whoever intersects next calls init_ray(r, r.o, r.d) - once, where the reference's BSDFs each
// end in init_ray and the wavefront kernels used to add two more (3 exact reciprocals each).
PT_DEV void set_ray(Ray& r, f3 orig, f3 dir) {
	r.d = dir;
	r.o = orig;
}

struct TraceCounters {
	uint32_t n_inner, n_box, n_tri, n_leaf, n_sphere, leaf_hash;
};

PT_DEV void fnv_word(uint32_t& h, uint32_t w) {
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		h ^= (w >> (8 * i)) & 0xFFu;
		h *= 16777619u;
	}
}

// reference: src/intersect.cuh:87-114 testBBoxIntersection. Written with the reference's
// comparisons (not fmin/fmax) so NaNs from 0*inf take the same branches; a box wholly behind
// the origin still "hits" with tmin = 0, as in the reference.
PT_DEV bool box_test(float minx, float miny, float minz, float maxx, float maxy, float maxz,
                     const Ray& r, float& tmin) {
	float t1 = ((r.sx ? maxx : minx) - r.o.x) * r.inv.x;
	float t2 = ((r.sx ? minx : maxx) - r.o.x) * r.inv.x;
	float t3 = ((r.sy ? maxy : miny) - r.o.y) * r.inv.y;
	float t4 = ((r.sy ? miny : maxy) - r.o.y) * r.inv.y;
	if ((t1 > t4) || (t3 > t2)) return false;
	if (t3 > t1) t1 = t3;
	if (t4 < t2) t2 = t4;
	float t5 = ((r.sz ? maxz : minz) - r.o.z) * r.inv.z;
	float t6 = ((r.sz ? minz : maxz) - r.o.z) * r.inv.z;
	if ((t1 > t6) || (t5 > t2)) return false;
	if (t5 > t1) t1 = t5;
	tmin = t1 < 0.0f ? 0.0f : t1;
	return true;
}

// reference: src/intersect.cuh:27-56 rayIntersectsTriangle (Moeller-Trumbore on a, e1, e2;
// __fdividef(1, a) is taken as the IEEE quotient 1.0f / a).
PT_DEV bool tri_test(const Ray& r, f3 va, f3 e1, f3 e2, float& res_t, float& ru, float& rv) {
	f3 h = cross3(r.d, e2);
	float a = dot3(e1, h);
	if (a > -PT_TRI_EPSILON && a < PT_TRI_EPSILON) return false;
	float f = pt_rcp_det(a); // == 1.0f / a bit for bit (pt_math.h)
	f3 s = r.o - va;
	float u = f * dot3(s, h);
	if (u < 0.0f || u > 1.0f) return false;
	f3 q = cross3(s, e1);
	float v = f * dot3(r.d, q);
	if (v < 0.0f || u + v > 1.0f) return false;
	float t = f * dot3(e2, q);
	if (t > PT_TRI_EPSILON) {
		res_t = t; ru = u; rv = v;
		return true;
	}
	return false;
}

// The same test without early exits, for callers whose lanes test DIFFERENT triangles against one
// ray (a wave pays for its slowest lane anyway, so the exits only add exec-mask bookkeeping and
// serialise the operand loads behind them). Each rejection is the reference's comparison, negated
// as a whole, so NaN/inf operands (a ~ 0 gives f = inf) take the same decisions as the early returns.
PT_DEV bool tri_test_flat(const Ray& r, f3 va, f3 e1, f3 e2, float& res_t, float& ru, float& rv) {
	const f3 h = cross3(r.d, e2);
	const float a = dot3(e1, h);
	const bool ok_a = !(a > -PT_TRI_EPSILON && a < PT_TRI_EPSILON);
	const float f = pt_rcp_det(a); // == 1.0f / a bit for bit wherever ok_a holds (pt_math.h)
	const f3 s = r.o - va;
	const float u = f * dot3(s, h);
	const bool ok_u = !(u < 0.0f || u > 1.0f);
	const f3 q = cross3(s, e1);
	const float v = f * dot3(r.d, q);
	const bool ok_v = !(v < 0.0f || u + v > 1.0f);
	const float t = f * dot3(e2, q);
	res_t = t; ru = u; rv = v;
	return (ok_a & ok_u) & (ok_v & (t > PT_TRI_EPSILON));
}

// Two triangles against one ray per call, component pairs in 64-bit registers so that the
// arithmetic issues as packed FP32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE results per instruction,
// same rounding as the scalar forms; no contraction). Per component this is tri_test_flat, operation
// for operation; bit k of the result = triangle k hit.
typedef float v2f __attribute__((ext_vector_type(2)));
struct f3x2 { v2f x, y, z; };
PT_DEV v2f splat2(float s) { v2f r = {s, s}; return r; }
// Both child boxes of an inner node against one ray: the (left, right) pairs of every component sit in
// 64-bit register pairs (node layout, pt_scene.h), so the subtractions and multiplications issue as
// packed FP32. Per box this is box_test operation for operation WITHOUT the early returns: the same
// comparisons on the same values in the same order, so NaNs decide as in the reference; bit k of the
// result = box k is hit, tmin[k] its entry distance (meaningful only then).
PT_DEV uint32_t box_test_x2(const float4 q0, const float4 q1, const float4 q2, float rox, float roy, float roz,
                            float rix, float riy, float riz, bool sx, bool sy, bool sz, float& tmin_l, float& tmin_r) {
	const v2f minx = {q0.x, q0.y}, miny = {q0.z, q0.w}, minz = {q1.x, q1.y};
	const v2f maxx = {q1.z, q1.w}, maxy = {q2.x, q2.y}, maxz = {q2.z, q2.w};
	const v2f ox = splat2(rox), oy = splat2(roy), oz = splat2(roz);
	const v2f ix = splat2(rix), iy = splat2(riy), iz = splat2(riz);
	const v2f t1 = ((sx ? maxx : minx) - ox) * ix, t2 = ((sx ? minx : maxx) - ox) * ix;
	const v2f t3 = ((sy ? maxy : miny) - oy) * iy, t4 = ((sy ? miny : maxy) - oy) * iy;
	const v2f t5 = ((sz ? maxz : minz) - oz) * iz, t6 = ((sz ? minz : maxz) - oz) * iz;
	uint32_t bits = 0;
	{
		float a1 = t1.x, a2 = t2.x;
		bool fail = (a1 > t4.x) | (t3.x > a2);
		if (t3.x > a1) a1 = t3.x;
		if (t4.x < a2) a2 = t4.x;
		fail = fail | (a1 > t6.x) | (t5.x > a2);
		if (t5.x > a1) a1 = t5.x;
		tmin_l = a1 < 0.0f ? 0.0f : a1;
		if (!fail) bits |= 1u;
	}
	{
		float a1 = t1.y, a2 = t2.y;
		bool fail = (a1 > t4.y) | (t3.y > a2);
		if (t3.y > a1) a1 = t3.y;
		if (t4.y < a2) a2 = t4.y;
		fail = fail | (a1 > t6.y) | (t5.y > a2);
		if (t5.y > a1) a1 = t5.y;
		tmin_r = a1 < 0.0f ? 0.0f : a1;
		if (!fail) bits |= 2u;
	}
	return bits;
}
PT_DEV f3x2 cross3x2(const f3x2& a, const f3x2& b) {
	f3x2 r;
	r.x = a.y * b.z - a.z * b.y;
	r.y = a.z * b.x - a.x * b.z;
	r.z = a.x * b.y - a.y * b.x;
	return r;
}
PT_DEV v2f dot3x2(const f3x2& a, const f3x2& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV v2f rcp_det_x2(v2f a) {
	const float hi = __builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y));
	if (__ballot(hi > PT_RCP_HI) != 0ull) { v2f q = {1.0f / a.x, 1.0f / a.y}; return q; }
	v2f r = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
	const v2f one = {1.0f, 1.0f};
	v2f e = __builtin_elementwise_fma(-a, r, one);
	r = __builtin_elementwise_fma(e, r, r);
	e = __builtin_elementwise_fma(-a, r, one);
	r = __builtin_elementwise_fma(e, r, r);
	return r;
}
// o2 / d: the ray's origin and direction with each component in both halves of a pair
PT_DEV uint32_t tri_test_flat_x2p(const f3x2& o2, const f3x2& d, const f3x2& va, const f3x2& e1, const f3x2& e2, v2f& res_t, v2f& ru, v2f& rv) {
	f3x2 s;
	const f3x2 h = cross3x2(d, e2);
	const v2f a = dot3x2(e1, h);
	const v2f f = rcp_det_x2(a);
	s.x = o2.x - va.x; s.y = o2.y - va.y; s.z = o2.z - va.z;
	const v2f u = f * dot3x2(s, h);
	const f3x2 q = cross3x2(s, e1);
	const v2f v = f * dot3x2(d, q);
	const v2f t = f * dot3x2(e2, q);
	const v2f uv = u + v;
	res_t = t; ru = u; rv = v;
	uint32_t bits = 0;
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		const bool ok_a = !(a[k] > -PT_TRI_EPSILON && a[k] < PT_TRI_EPSILON);
		const bool ok_u = !(u[k] < 0.0f || u[k] > 1.0f);
		const bool ok_v = !(v[k] < 0.0f || uv[k] > 1.0f);
		if ((ok_a & ok_u) & (ok_v & (t[k] > PT_TRI_EPSILON))) bits |= 1u << k;
	}
	return bits;
}
PT_DEV uint32_t tri_test_flat_x2(f3 o, f3 dir, const f3x2& va, const f3x2& e1, const f3x2& e2, v2f& res_t, v2f& ru, v2f& rv) {
	f3x2 d, o2;
	d.x = splat2(dir.x); d.y = splat2(dir.y); d.z = splat2(dir.z);
	o2.x = splat2(o.x); o2.y = splat2(o.y); o2.z = splat2(o.z);
	return tri_test_flat_x2p(o2, d, va, e1, e2, res_t, ru, rv);
}

// reference: src/intersect.cuh:59-83 testSphereIntersection
PT_DEV bool sphere_test(const Ray& r, float rad, f3 center, f3& hit_point, f3& hit_norm, float& t_out) {
	f3 op = center - r.o;
	const float epsilon = 0.0001f;
	float b = dot3(op, r.d);
	float disc = b * b - dot3(op, op) + rad * rad;
	if (disc < 0.0f) return false;
	disc = pt_sqrt(disc);
	float t = b - disc;
	if (t < epsilon) t = b + disc;
	if (t < epsilon) return false;
	hit_point = r.o + r.d * t;
	hit_norm = normalize3(hit_point - center);
	t_out = t;
	return true;
}

// reference: src/intersect.cuh:119-212 rayIntersectsKDNodeLOOP, after the root box test.
// stack: this lane's column in LDS, element i at stack[i * stride].
// Returns "some triangle test succeeded"; best_tri (index into tri_geom/tri_shade), bu, bv and
// tmin are updated only when a strictly closer hit is found (strict <, first wins ties).
//
// Loop shape ("while-while"): a lane first DESCENDS through inner nodes until it holds a leaf, then
// tests that leaf's triangles, then pops. The lanes of a wave therefore run the two kinds of work in
// two separate inner loops instead of serialising a leaf body and an inner-node body every
// iteration. A node record is fetched whole (4 x 16 B, one 64-B line) before the leaf/inner decision
// so the decision does not add a dependent round trip; the next triangle is fetched while the
// current one is tested.
struct TravState {
	int curr, sp;       // current node (scene-global, -1 = none) and number of stack entries
	int tri;            // best triangle so far (device index), -1 = none
	float tmin, bu, bv;
	bool intersects;
};

PT_DEV void trav_begin(TravState& ts, const DevObject& ob) {
	ts.curr = ob.root_node;
	ts.sp = 0;
	ts.tri = -1;
	ts.tmin = __builtin_huge_valf();
	ts.bu = ts.bv = 0.0f;
	ts.intersects = false;
}

// One unit of traversal: descend to the next leaf, test its triangles, pop. Returns true when the
// traversal is finished (stack empty). Running it to completion is rayIntersectsKDNodeLOOP.
template <int FLAGS, int PREFETCH = 1>
PT_DEV bool traverse_step(const DevScene& sc, const DevObject& ob, int obj_index, const Ray& ray, int* stack,
                          int stride, TravState& ts, TraceCounters& cnt, float& dbg, uint32_t& steps) {
	const float4* __restrict__ nodes = sc.nodes;
	const float4* __restrict__ geom = sc.tri_geom;
	int curr = ts.curr, sp = ts.sp;
	int first = 0, count = -1;
	// ---- descend until a leaf is reached or the traversal is over -----------------------
	while (curr >= 0) {
		const float4 q3 = nodes[4 * curr + 3];
		const float4 q0 = nodes[4 * curr + 0];
		const float4 q1 = nodes[4 * curr + 1];
		const float4 q2 = nodes[4 * curr + 2];
		if (FLAGS & PT_F_STATS) steps++;
		const int left = __float_as_int(q3.x), right = __float_as_int(q3.y);
		if (left == -1 && right == -1) {
			first = __float_as_int(q3.z);
			count = __float_as_int(q3.w);
			if (FLAGS & PT_F_STATS) {
				cnt.n_leaf++;
				cnt.n_tri += (uint32_t)count;
				fnv_word(cnt.leaf_hash, (uint32_t)obj_index);
				fnv_word(cnt.leaf_hash, (uint32_t)(curr - ob.first_node));
			}
			break;
		}
		// inner node: test both child boxes (stored in this record)
		if (FLAGS & PT_F_STATS) cnt.n_inner++;
		float left_dist = __builtin_huge_valf(), right_dist = __builtin_huge_valf();
		bool left_res = false, right_res = false;
		if (left != -1) {
			if (FLAGS & PT_F_STATS) cnt.n_box++;
			left_res = box_test(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, ray, left_dist) && (left_dist < ts.tmin);
		}
		if (right != -1) {
			if (FLAGS & PT_F_STATS) cnt.n_box++;
			right_res = box_test(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, ray, right_dist) && (right_dist < ts.tmin);
		}
		if (left_res && right_res) {
			// nearer child first; tie -> left first (reference :183-187)
			int further = right;
			int next = right;
			if (left_dist > right_dist) further = left; else next = left;
			stack[sp * stride] = further;
			++sp;
			curr = next;
			if (FLAGS & PT_F_DEBUGMASK) dbg *= 0.9025f;
		} else if (left_res) {
			curr = left;
			if (FLAGS & PT_F_DEBUGMASK) dbg *= 0.95f;
		} else if (right_res) {
			curr = right;
			if (FLAGS & PT_F_DEBUGMASK) dbg *= 0.95f;
		} else {
			if (sp == 0) { curr = -1; break; }
			--sp;
			curr = stack[sp * stride];
		}
	}
	if (count < 0) { ts.curr = -1; ts.sp = 0; return true; } // no leaf: traversal finished

	// ---- leaf: test all triangles of the node, in list order -----------------------------
	// The next triangle is fetched while the current one is tested (PREFETCH = 1). Deeper, grouped
	// prefetch (2-8 triangles ahead, two register buffers) was measured and is SLOWER on MI355X
	// (704 -> 630/594/524/406 Mrays/s for groups of 2/3/4/6): the registers it takes cost more
	// occupancy than the extra loads in flight buy. No prefetch at 8 waves/SIMD is slower too (485).
	if (PREFETCH) {
		if (count > 0) {
			float4 g0 = geom[3 * first + 0], g1 = geom[3 * first + 1], g2 = geom[3 * first + 2];
			for (int i = 0; i < count; ++i) {
				float4 n0 = g0, n1 = g1, n2 = g2;
				if (i + 1 < count) {
					n0 = geom[3 * (first + i + 1) + 0];
					n1 = geom[3 * (first + i + 1) + 1];
					n2 = geom[3 * (first + i + 1) + 2];
				}
				if (FLAGS & PT_F_STATS) steps++;
				float t, u, v;
				if (tri_test(ray, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), mk3(g1.z, g1.w, g2.x), t, u, v)) {
					ts.intersects = true;
					if (t < ts.tmin) { ts.tmin = t; ts.tri = first + i; ts.bu = u; ts.bv = v; }
				}
				g0 = n0; g1 = n1; g2 = n2;
			}
		}
	} else {
		for (int i = 0; i < count; ++i) {
			const float4 g0 = geom[3 * (first + i) + 0], g1 = geom[3 * (first + i) + 1], g2 = geom[3 * (first + i) + 2];
			if (FLAGS & PT_F_STATS) steps++;
			float t, u, v;
			if (tri_test(ray, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), mk3(g1.z, g1.w, g2.x), t, u, v)) {
				ts.intersects = true;
				if (t < ts.tmin) { ts.tmin = t; ts.tri = first + i; ts.bu = u; ts.bv = v; }
			}
		}
	}
	// pop
	if (sp == 0) { ts.curr = -1; ts.sp = 0; return true; }
	--sp;
	ts.curr = stack[sp * stride];
	ts.sp = sp;
	return false;
}

template <int FLAGS>
PT_DEV bool traverse_nodes(const DevScene& sc, const DevObject& ob, int obj_index, const Ray& ray,
                           int* stack, int stride, float& tmin, int& best_tri, float& bu, float& bv,
                           TraceCounters& cnt, float& dbg, uint32_t& steps) {
	TravState ts;
	trav_begin(ts, ob);
	ts.tmin = tmin;
	while (!traverse_step<FLAGS>(sc, ob, obj_index, ray, stack, stride, ts, cnt, dbg, steps)) {}
	if (ts.tri >= 0) { tmin = ts.tmin; best_tri = ts.tri; bu = ts.bu; bv = ts.bv; }
	return ts.intersects;
}

// root box test (reference :130-135) + traversal
template <int FLAGS>
PT_DEV bool traverse_mesh(const DevScene& sc, const DevObject& ob, int obj_index, const Ray& ray,
                          int* stack, int stride, float& tmin, int& best_tri, float& bu, float& bv,
                          TraceCounters& cnt, float& dbg) {
	if (ob.root_node < 0) return false;
	float dist;
	if (FLAGS & PT_F_STATS) cnt.n_box++;
	if (!box_test(ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], ray, dist))
		return false;
	if (FLAGS & PT_F_DEBUGMASK) dbg *= 0.95f;
	uint32_t steps = 0;
	return traverse_nodes<FLAGS>(sc, ob, obj_index, ray, stack, stride, tmin, best_tri, bu, bv, cnt, dbg, steps);
}

struct HitInfo {
	f3 point, normal, bary;
	int object;   // -1 = miss
	int tri;      // device triangle index (meshes), -1 for spheres
	int material; // scene-global material index
	float dist;   // length(point - origin)
};

// reference: src/intersect.cuh:322-350 rayIntersectsScene + :260-319 rayIntersectsObject.
// only_object >= 0 restricts the loop to that object (parity entry point).
template <int FLAGS>
PT_DEV bool intersect_scene(const DevScene& sc, const Ray& ray, int* stack, int stride, int only_object,
                            HitInfo& hi, TraceCounters& cnt, float& dbg) {
	bool intersects = false;
	float closest = __builtin_huge_valf();
	hi.object = -1;
	hi.tri = -1;
	hi.material = 0;
	hi.dist = closest;
	const int n = sc.num_objects;
	for (int i = 0; i < n; ++i) {
		if (only_object >= 0 && i != only_object) continue;
		const DevObject& ob = sc.objects[i];
		f3 p, nrm, bary = mk3(0.0f);
		int tri = -1;
		int mat = ob.mat_first;
		bool hit;
		if (ob.type == 0) {
			if (FLAGS & PT_F_STATS) cnt.n_sphere++;
			float ts;
			hit = sphere_test(ray, ob.radius, mk3(ob.px, ob.py, ob.pz), p, nrm, ts);
		} else {
			float tmin = __builtin_huge_valf();
			float bu = 0.0f, bv = 0.0f;
			hit = traverse_mesh<FLAGS>(sc, ob, i, ray, stack, stride, tmin, tri, bu, bv, cnt, dbg);
			// a "hit" at t = +inf never beats tmin; the reference would dereference a NULL triangle there
			hit = hit && tri >= 0;
			if (hit) {
				// reference :300-305
				p = ray.o + tmin * ray.d;
				bary.y = bu;
				bary.z = bv;
				bary.x = 1.0f - bary.y - bary.z;
				const float4 s0 = sc.tri_shade[4 * tri + 0];
				const float4 s1 = sc.tri_shade[4 * tri + 1];
				const float4 s2 = sc.tri_shade[4 * tri + 2];
				const float4 s3 = sc.tri_shade[4 * tri + 3];
				nrm = normalize3(bary.x * mk3(s0.x, s0.y, s0.z) + bary.y * mk3(s1.x, s1.y, s1.z) +
				                 bary.z * mk3(s2.x, s2.y, s2.z));
				mat = __float_as_int(s3.w);
			}
		}
		if (hit) {
			intersects = true;
			float d = length3(p - ray.o);
			if (d < closest) {
				closest = d;
				hi.object = i;
				hi.point = p;
				hi.normal = nrm;
				hi.bary = bary;
				hi.material = mat;
				hi.tri = tri;
				hi.dist = d;
			}
		}
	}
	return intersects;
}
