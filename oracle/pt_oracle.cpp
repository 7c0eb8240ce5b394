// CPU oracle for the path-tracing hot path — TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
//
// Restates filiprak/path-tracer's device algorithm in plain scalar C++. Each function cites
// the reference file:line it follows. Build: g++ -O2 -ffp-contract=off (oracle/Makefile).
// PARITY STATUS: unpinned except where oracle/Makefile's _ref target pins it (fmedian,
// Camera::init, JSON resolve) — the reference ships no tests or golden vectors, and its CUDA
// sources cannot be built here without stand-in headers.
#include "pt_oracle.h"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---- float3 helpers with the evaluation order of the reference's cutil_math.h ------------------
// reference: external/include/cutil_math.h:322-416
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 v3(float s) { return v3(s, s, s); }
inline V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(V3 v) { return sqrtf(dot(v, v)); }
inline V3 normalize(V3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return scale(v, inv); }
inline V3 reflect(V3 i, V3 n) { return sub(i, scale(scale(n, 2.0f), dot(n, i))); }
inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---- sin/cos/acos: fixed polynomials (Cephes single-precision coefficients) ---------------------
// stand in for __sinf/__sincosf/acosf of reference src/surfaces.cuh:24-29,64
void sincos_poly(float theta, float* s, float* c) {
	const float TWO_OVER_PI = 0.636619772367581343f;
	const float PIO2_HI = 1.5707962512969970703125f;
	const float PIO2_LO = 7.5497894158615963e-08f;
	float kf = rintf(theta * TWO_OVER_PI);
	int k = (int)kf;
	float r = (theta - kf * PIO2_HI) - kf * PIO2_LO;
	float z = r * r;
	float sp = -1.9515295891e-4f;
	sp = sp * z + 8.3321608736e-3f;
	sp = sp * z + -1.6666654611e-1f;
	float sr = r + (r * z) * sp;
	float cp = 2.443315711809948e-5f;
	cp = cp * z + -1.388731625493765e-3f;
	cp = cp * z + 4.166664568298827e-2f;
	float cr = (1.0f - 0.5f * z) + (z * z) * cp;
	switch (k & 3) {
		case 0: *s = sr; *c = cr; break;
		case 1: *s = cr; *c = -sr; break;
		case 2: *s = -sr; *c = -cr; break;
		default: *s = -cr; *c = sr; break;
	}
}

float acos01_poly(float x) {
	const float PIO2 = 1.5707963267948966f;
	if (x > 1.0f) x = 1.0f;
	float z, base;
	bool big = x > 0.5f;
	if (big) { z = 0.5f * (1.0f - x); base = sqrtf(z); }
	else { z = x * x; base = x; }
	float p = 4.2163199048e-2f;
	p = p * z + 2.4181311049e-2f;
	p = p * z + 4.5470025998e-2f;
	p = p * z + 7.4953002686e-2f;
	p = p * z + 1.6666752422e-1f;
	float as = base + (base * z) * p;
	return big ? 2.0f * as : PIO2 - as;
}

// ---- Philox4x32-10 ------------------------------------------------------------------------------
void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
	for (int i = 0; i < 10; ++i) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c0;
		uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
		uint32_t n1 = (uint32_t)p1;
		uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		uint32_t n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline float u01(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }

// ---- rays ------------------------------------------------------------------------------------------
struct Ray { V3 o, d, inv; int sign[3]; };

// reference: src/pathtracing.cu:23-31 init_ray
void init_ray(Ray* r, V3 orig, V3 dir) {
	r->d = dir;
	r->o = orig;
	r->inv = v3(1 / dir.x, 1 / dir.y, 1 / dir.z);
	r->sign[0] = (r->inv.x < 0);
	r->sign[1] = (r->inv.y < 0);
	r->sign[2] = (r->inv.z < 0);
}

struct Counters { uint64_t n_inner, n_box, n_tri, n_leaf, n_sphere; uint64_t t_rays, t_inner, t_box, t_tri, t_leaf; uint32_t leaf_hash; };

inline void fnv_word(uint32_t& h, uint32_t w) {
	for (int i = 0; i < 4; ++i) { h ^= (w >> (8 * i)) & 0xFFu; h *= 16777619u; }
}

// reference: src/intersect.cuh:87-114 testBBoxIntersection
bool testBBox(const float* bmin, const float* bmax, const Ray* ray, float* tmin) {
	const float* bounds[2] = {bmin, bmax};
	float t[8];
	t[1] = (bounds[ray->sign[0]][0] - ray->o.x) * ray->inv.x;
	t[2] = (bounds[1 - ray->sign[0]][0] - ray->o.x) * ray->inv.x;
	t[3] = (bounds[ray->sign[1]][1] - ray->o.y) * ray->inv.y;
	t[4] = (bounds[1 - ray->sign[1]][1] - ray->o.y) * ray->inv.y;
	if ((t[1] > t[4]) || (t[3] > t[2])) return false;
	if (t[3] > t[1]) t[1] = t[3];
	if (t[4] < t[2]) t[2] = t[4];
	t[5] = (bounds[ray->sign[2]][2] - ray->o.z) * ray->inv.z;
	t[6] = (bounds[1 - ray->sign[2]][2] - ray->o.z) * ray->inv.z;
	if ((t[1] > t[6]) || (t[5] > t[2])) return false;
	if (t[5] > t[1]) t[1] = t[5];
	if (t[6] < t[2]) t[2] = t[6];
	*tmin = t[1] < 0.0f ? 0.0f : t[1];
	return true;
}

// reference: src/intersect.cuh:27-56 rayIntersectsTriangle, EPSILON :10
const float EPSILON = 0.00000005f;
bool rayTriangle(const Ray* ray, V3 va, V3 e1, V3 e2, float* res_dist, float* ru, float* rv) {
	V3 h = cross(ray->d, e2);
	float a = dot(e1, h);
	if (a > -EPSILON && a < EPSILON) return false;
	float f = 1.0f / a; // __fdividef(1.0, a) taken as the IEEE quotient
	V3 s = sub(ray->o, va);
	float u = f * (dot(s, h));
	if (u < 0.0 || u > 1.0) return false;
	V3 q = cross(s, e1);
	float v = f * (dot(ray->d, q));
	if (v < 0.0 || u + v > 1.0) return false;
	float t = f * dot(e2, q);
	if (t > EPSILON) { *res_dist = t; *ru = u; *rv = v; return true; }
	return false;
}

// reference: src/intersect.cuh:59-83 testSphereIntersection
bool testSphere(const Ray* ray, const PtxObject* obj, V3* hit_point, V3* hit_norm) {
	float rad = obj->radius;
	V3 pos = ld3(obj->position);
	V3 op = sub(pos, ray->o);
	float t, epsilon = 0.0001f;
	float b = dot(op, ray->d);
	float disc = b * b - dot(op, op) + rad * rad;
	if (disc < 0) return false;
	else disc = sqrtf(disc);
	t = b - disc;
	if (t < epsilon) t = b + disc;
	if (t < epsilon) return false;
	*hit_point = add(ray->o, scale(ray->d, t));
	*hit_norm = normalize(sub(*hit_point, pos));
	return true;
}

// reference: src/intersect.cuh:119-212 rayIntersectsKDNodeLOOP — walks the reference's own node
// layout: each node has its OWN box; leaf triangle lists are index lists.
bool kdLoop(const PtxSceneDesc* sc, const PtxObject* ob, int obj_index, const Ray* ray, float* bary_y, float* bary_z,
            int* itrg, float* tmin, float* debug_mask, Counters* cnt) {
	const PtxKDNode* flat_nodes = sc->nodes + ob->first_node;
	const PtxTriangle* trs = sc->triangles + ob->first_triangle;
	const int32_t* leaf_tris = sc->leaf_tris + ob->first_leaf_tri;
	int curr_idx = 0;
	bool intersects = false;
	if (ob->num_nodes <= 0 || ob->num_triangles <= 0) return false;

	float dist = HUGE_VALF;
	cnt->n_box++;
	if (!testBBox(flat_nodes[curr_idx].bmin, flat_nodes[curr_idx].bmax, ray, &dist)) return intersects;
	*debug_mask *= 0.95f;
	// counters restricted to multi-node meshes (root is not a leaf): what the GPU's traversal kernel does
	const bool multi = !(flat_nodes[0].left == -1 && flat_nodes[0].right == -1);
	if (multi) cnt->t_rays++;

	int stack_pos = -1;
	int stack[64];
	memset(stack, -1, sizeof(stack));

	while (true) {
		const PtxKDNode& node = flat_nodes[curr_idx];
		if (node.left == -1 && node.right == -1) {
			float t, u, v;
			cnt->n_leaf++;
			if (multi) { cnt->t_leaf++; cnt->t_tri += (uint64_t)node.num_tris; }
			fnv_word(cnt->leaf_hash, (uint32_t)obj_index);
			fnv_word(cnt->leaf_hash, (uint32_t)curr_idx);
			for (int i = 0; i < node.num_tris; i++) {
				int ti = leaf_tris[node.first_tri + i];
				const PtxTriangle& trg = trs[ti];
				cnt->n_tri++;
				if (rayTriangle(ray, ld3(trg.a), ld3(trg.e1), ld3(trg.e2), &t, &u, &v)) {
					intersects = true;
					if (t < *tmin) {
						*tmin = t;
						*itrg = ti;
						*bary_y = u;
						*bary_z = v;
					}
				}
			}
			if (stack_pos < 0) return intersects;
			curr_idx = stack[stack_pos];
			stack[stack_pos] = -1;
			stack_pos--;
			continue;
		}

		cnt->n_inner++;
		if (multi) { cnt->t_inner++; cnt->t_box += (node.left != -1) + (node.right != -1); }
		float left_dist = HUGE_VALF, right_dist = HUGE_VALF;
		bool left_res = false, right_res = false;
		if (node.left != -1) {
			cnt->n_box++;
			left_res = testBBox(flat_nodes[node.left].bmin, flat_nodes[node.left].bmax, ray, &left_dist) && (left_dist < *tmin);
		}
		if (node.right != -1) {
			cnt->n_box++;
			right_res = testBBox(flat_nodes[node.right].bmin, flat_nodes[node.right].bmax, ray, &right_dist) && (right_dist < *tmin);
		}

		if (left_res && right_res) {
			int further_idx = node.right;
			curr_idx = node.right;
			left_dist > right_dist ? (further_idx = node.left) : (curr_idx = node.left);
			stack_pos++;
			stack[stack_pos] = further_idx;
			*debug_mask *= 0.9025f;
			continue;
		} else if (left_res) {
			curr_idx = node.left;
			*debug_mask *= 0.95f;
			continue;
		} else if (right_res) {
			curr_idx = node.right;
			*debug_mask *= 0.95f;
			continue;
		} else {
			if (stack_pos < 0) return intersects;
			curr_idx = stack[stack_pos];
			stack[stack_pos] = -1;
			stack_pos--;
		}
	}
}

struct IntersectInfo {
	V3 ipoint, normal, bary;
	int itrg;       // triangle index inside the object, -1 = none (sphere)
	int iobj_index;
	int imat;       // scene-global material index
	float dist;
	float debug_mask;
};

// reference: src/intersect.cuh:260-319 rayIntersectsObject
bool intersectObject(const PtxSceneDesc* sc, int obj_index, const Ray* ray, int* mat, int* trg, V3* hit_point, V3* hit_norm,
                     V3* bary, float* debug_mask, Counters* cnt) {
	const PtxObject* obj = &sc->objects[obj_index];
	bool intersects = false;
	if (obj->type == PTX_OBJ_SPHERE) {
		cnt->n_sphere++;
		intersects = testSphere(ray, obj, hit_point, hit_norm);
		*mat = obj->first_material;
	} else {
		float tmin = HUGE_VALF;
		intersects = kdLoop(sc, obj, obj_index, ray, &bary->y, &bary->z, trg, &tmin, debug_mask, cnt);
		if (intersects && *trg < 0) intersects = false; // a hit at t = +inf never beats tmin; the reference would dereference NULL
		if (intersects) {
			const PtxTriangle& t = sc->triangles[obj->first_triangle + *trg];
			*hit_point = add(ray->o, scale(ray->d, tmin));
			bary->x = 1.0f - bary->y - bary->z;
			*hit_norm = normalize(add(add(scale(ld3(t.na), bary->x), scale(ld3(t.nb), bary->y)), scale(ld3(t.nc), bary->z)));
			*mat = obj->first_material + t.material;
		}
	}
	return intersects;
}

// reference: src/intersect.cuh:322-350 rayIntersectsScene
bool intersectScene(const PtxSceneDesc* sc, const Ray* ray, int only_object, IntersectInfo* ii, Counters* cnt) {
	bool intersects = false;
	float closest_dist = HUGE_VALF;
	V3 inters_point = v3(0), inters_norm = v3(0), bary = v3(0);
	float debug_mask = 1.0f;
	ii->iobj_index = -1;
	ii->itrg = -1;
	ii->imat = 0;
	ii->dist = closest_dist;
	for (int i = 0; i < sc->num_objects; ++i) {
		if (only_object >= 0 && i != only_object) continue;
		int i_mat = sc->objects[i].first_material;
		int itrg = -1;
		if (intersectObject(sc, i, ray, &i_mat, &itrg, &inters_point, &inters_norm, &bary, &debug_mask, cnt)) {
			intersects = true;
			float inters_dist = length(sub(inters_point, ray->o));
			if (inters_dist < closest_dist) {
				closest_dist = inters_dist;
				ii->iobj_index = i;
				ii->ipoint = inters_point;
				ii->normal = inters_norm;
				ii->imat = i_mat;
				ii->bary = bary;
				ii->itrg = itrg;
				ii->dist = inters_dist;
			}
		}
		ii->debug_mask = debug_mask;
	}
	ii->debug_mask = debug_mask;
	return intersects;
}

// ---- BSDFs ------------------------------------------------------------------------------------------
const float PI_X2_f = 6.2831853072f; // reference: src/constants.h:5
const float PI_D2_f = 1.5707963267f; // reference: src/constants.h:6

struct Uniforms { float u[4]; int next; float draw() { return u[next++]; } };

// reference: src/surfaces.cuh:14-32 rand_cone_Dir
V3 rand_cone_Dir(V3 cone_axis, float ang_max_dev, Uniforms* rs) {
	V3 u = normalize(cone_axis.x != 0.0f || cone_axis.y != 0.0f ? v3(cone_axis.y, -cone_axis.x, 0.0f) : v3(0.0f, cone_axis.z, -cone_axis.y));
	V3 v = normalize(cross(u, cone_axis));
	float sn, cs;
	sincos_poly(ang_max_dev, &sn, &cs);
	const float r2 = sn * rs->draw();
	const float r = sqrtf(r2);
	const float theta = PI_X2_f * rs->draw();
	float sinth, costh;
	sincos_poly(theta, &sinth, &costh);
	// EPS is the double literal 0.0001 (reference :18): the sum is formed in double
	const float w2 = (float)((double)(1.0f - r2) + 0.0001);
	return normalize(add(add(scale(u, sinth * r), scale(v, costh * r)), scale(cone_axis, sqrtf(w2))));
}

// reference: src/surfaces.cuh:37-44
void Diffuse_BRDF(Ray* in_ray, V3 surf_normal, V3 inters_point, Uniforms* rs) {
	V3 rand_dir = rand_cone_Dir(surf_normal, PI_D2_f, rs);
	init_ray(in_ray, add(inters_point, scale(surf_normal, 0.00005f)), rand_dir);
}

// reference: src/surfaces.cuh:48-70
void Specular_Diffuse_BRDF(Ray* in_ray, float refl_factor, float sharpness, V3 surf_normal, V3 inters_point, Uniforms* rs) {
	V3 new_orig = add(inters_point, scale(surf_normal, 0.0001f));
	float r2 = rs->draw();
	if (r2 > refl_factor) {
		Diffuse_BRDF(in_ray, surf_normal, inters_point, rs);
	} else {
		const float max_cone_ang = PI_D2_f * (1.0f - sharpness);
		const float scatt_ang = PI_D2_f - acos01_poly(fabsf(dot(in_ray->d, surf_normal)));
		const float ang = scatt_ang < max_cone_ang ? scatt_ang : max_cone_ang; // fmin(max_cone_ang, scatt_ang)
		V3 rand_dir = rand_cone_Dir(reflect(in_ray->d, surf_normal), ang, rs);
		init_ray(in_ray, new_orig, rand_dir);
	}
}

// reference: src/surfaces.cuh:73-137
void Refractive_BRDF(Ray* in_ray, float refract_index, float refl_factor, V3* mask, V3 surf_normal, V3 inters_point, Uniforms* rs) {
	const float REFL_BIAS = 0.05f, REFL_BIAS_LOW = 0.0005f, SCENE_REFRACTION_INDEX = 1.0f;
	V3 ray_oriented_norm = dot(surf_normal, in_ray->d) < 0 ? surf_normal : scale(surf_normal, -1.0f);
	bool ray_outside = dot(surf_normal, ray_oriented_norm) > 0;
	float n_obj = refract_index;
	float nn_ratio = ray_outside ? (SCENE_REFRACTION_INDEX / n_obj) : (n_obj / SCENE_REFRACTION_INDEX);
	float cos_ray = dot(in_ray->d, ray_oriented_norm);
	float cos2refr = 1.0f - nn_ratio * nn_ratio * (1.0f - cos_ray * cos_ray);
	if (cos2refr < 0.0f) {
		init_ray(in_ray, add(inters_point, scale(ray_oriented_norm, REFL_BIAS)), reflect(in_ray->d, ray_oriented_norm));
	} else {
		float cosrefr = sqrtf(cos2refr);
		V3 trans_ray_dir = normalize(sub(scale(in_ray->d, nn_ratio),
		                                 scale(surf_normal, (ray_outside ? 1.0f : -1.0f) * (cos_ray * nn_ratio + cosrefr))));
		float ndiff = n_obj - SCENE_REFRACTION_INDEX;
		float nsum = n_obj + SCENE_REFRACTION_INDEX;
		float Refl0 = (ndiff * ndiff) / (nsum * nsum);
		float pw = 1.0f - (ray_outside ? -cos_ray : dot(trans_ray_dir, surf_normal));
		float Reflectance = Refl0 + (1.0f - Refl0) * pw * pw * pw * pw * pw;
		float EnergyScale = refl_factor + (1.0f - refl_factor - refl_factor) * Reflectance;
		if (rs->draw() < refl_factor) {
			*mask = scale(*mask, Reflectance / EnergyScale);
			init_ray(in_ray, add(inters_point, scale(ray_oriented_norm, REFL_BIAS)), reflect(in_ray->d, surf_normal));
		} else {
			*mask = scale(*mask, (1.0f - Reflectance) / (1.0f - EnergyScale));
			init_ray(in_ray, sub(inters_point, scale(ray_oriented_norm, REFL_BIAS_LOW)), trans_ray_dir);
		}
	}
}

// reference: src/world_load.cpp:177-183 — point filter, wrap addressing, normalized coordinates
void texFetch(const PtxTexture* tx, float u, float v, float out[4]) {
	float fu = u - floorf(u);
	float fv = v - floorf(v);
	int i = (int)floorf(fu * (float)tx->width);
	int j = (int)floorf(fv * (float)tx->height);
	i = i < 0 ? 0 : (i >= tx->width ? tx->width - 1 : i);
	j = j < 0 ? 0 : (j >= tx->height ? tx->height - 1 : j);
	const float* p = tx->rgba + ((size_t)j * tx->width + i) * 4;
	out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
}

void hitTexel(const PtxSceneDesc* sc, const IntersectInfo* ii, int texture, float out[4]) {
	const PtxObject* ob = &sc->objects[ii->iobj_index];
	const PtxTriangle& t = sc->triangles[ob->first_triangle + ii->itrg];
	// reference: src/radiance.cuh:74
	float tu = ii->bary.x * t.ta[0] + ii->bary.y * t.tb[0] + ii->bary.z * t.tc[0];
	float tv = ii->bary.x * t.ta[1] + ii->bary.y * t.tb[1] + ii->bary.z * t.tc[1];
	texFetch(&sc->textures[texture], tu, tv, out);
}

struct PathStats { uint64_t rays, hit_mesh, hit_sphere, hit_tex; Counters cnt; };

// reference: src/radiance.cuh:15-87 gatherRadiance
V3 gatherRadiance(const Ray* prim_ray, const PtxSceneDesc* sc, const PtxCamera* cam, uint32_t pixel, uint32_t sample,
                  uint32_t k0, uint32_t k1, PathStats* ps) {
	V3 radiance = v3(0.0f);
	V3 mask = v3(1.0f);
	Ray ray = *prim_ray;
	for (int bounce = 0; bounce < cam->max_ray_bounces; ++bounce) {
		IntersectInfo ii;
		ps->rays++;
		if (!intersectScene(sc, &ray, -1, &ii, &ps->cnt)) {
			if (cam->aabb_mode) return sub(v3(1.0f), v3(ii.debug_mask));
			return v3(0);
		}
		const PtxMaterial* mat = &sc->materials[ii.imat];
		if (cam->aabb_mode)
			return add(scale(sub(v3(1.0f), v3(ii.debug_mask)), 0.8f), scale(scale(ld3(mat->color), 0.2f), fabsf(dot(ray.d, ii.normal))));

		const bool textured = cam->texture_enabled && mat->texture > -1 && ii.itrg >= 0;
		if (ii.itrg >= 0) ps->hit_mesh++; else ps->hit_sphere++;
		if (textured) ps->hit_tex++;

		if (cam->preview_mode) {
			V3 tex_color = v3(0.0f);
			float tex_blend = 0.0f, alpha = 0.0f;
			if (textured) {
				float texel[4];
				hitTexel(sc, &ii, mat->texture, texel);
				alpha = texel[3];
				tex_color = v3(texel[0], texel[1], texel[2]);
				tex_blend = 0.5f;
			}
			float rdot = dot(ii.normal, ray.d);
			if (rdot < 0)
				return scale(add(scale(tex_color, alpha * tex_blend), scale(v3(1.0f), 1.0f - tex_blend)), -rdot);
			else
				return scale(add(scale(tex_color, tex_blend), scale(v3(1.0f, 0, 0), 1.0f - tex_blend)), rdot);
		}

		uint32_t r[4];
		philox(pixel, sample, (uint32_t)bounce, 0u, k0, k1, r);
		Uniforms rs;
		for (int k = 0; k < 4; ++k) rs.u[k] = u01(r[k]);
		rs.next = 0;
		switch (mat->type) {
			case PTX_MAT_DIFFUSE: Diffuse_BRDF(&ray, ii.normal, ii.ipoint, &rs); break;
			case PTX_MAT_SPECULAR: Specular_Diffuse_BRDF(&ray, mat->reflect_factor, mat->sharpness, ii.normal, ii.ipoint, &rs); break;
			case PTX_MAT_TRANSPARENT: Refractive_BRDF(&ray, mat->refract_index, mat->reflect_factor, &mask, ii.normal, ii.ipoint, &rs); break;
			default: break;
		}
		if (textured) {
			float texel[4];
			hitTexel(sc, &ii, mat->texture, texel);
			mask = mul(mask, mul(v3(texel[0], texel[1], texel[2]), ld3(mat->color)));
		} else {
			mask = mul(mask, ld3(mat->color));
		}
		if (mat->type == PTX_MAT_LUMINESCENT) return mul(mask, ld3(mat->emittance));
	}
	return radiance;
}

// reference: src/pathtracing.cu:73-94 generatePrimaryRays (midpoint + centre direction)
void pixelMidpoint(const PtxCamera* cam, int x, int y, V3* mid, V3* centre_dir) {
	float pix_side = cam->pixel_size;
	V3 screenDistanceVec = scale(normalize(ld3(cam->direction)), cam->screen_dist);
	V3 pixVector = add(scale(normalize(ld3(cam->right)), cam->screen_half[0] - x * pix_side - pix_side / 2.0f),
	                   scale(normalize(ld3(cam->up)), cam->screen_half[1] - y * pix_side - pix_side / 2.0f));
	*centre_dir = normalize(add(screenDistanceVec, pixVector));
	*mid = add(add(ld3(cam->position), screenDistanceVec), pixVector);
}

// reference: src/pathtracing.cu:107-129 jitterPrimaryRays
V3 jitteredDir(const PtxCamera* cam, V3 mid, uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1) {
	uint32_t r[4];
	philox(pixel, sample, 0xFFFFFFFFu, 0u, k0, k1, r);
	float sc = cam->pixel_size * cam->aa_jitter;
	float jx = sc * (u01(r[0]) - 0.5f);
	float jy = sc * (u01(r[1]) - 0.5f);
	return normalize(sub(add(add(mid, scale(ld3(cam->right), jx)), scale(ld3(cam->up), jy)), ld3(cam->position)));
}

V3 pixelSample(const PtxSceneDesc* sc, const PtxCamera* cam, int x, int y, int sample, uint32_t k0, uint32_t k1, PathStats* ps) {
	V3 mid, centre;
	pixelMidpoint(cam, x, y, &mid, &centre);
	uint32_t pixel = (uint32_t)(x + y * cam->width);
	Ray ray;
	init_ray(&ray, ld3(cam->position), jitteredDir(cam, mid, pixel, (uint32_t)sample, k0, k1));
	return gatherRadiance(&ray, sc, cam, pixel, (uint32_t)sample, k0, k1, ps);
}

int checkScene(const PtxSceneDesc* sc) {
	if (!sc) return PTX_E_INVALID;
	for (int i = 0; i < sc->num_objects; ++i) {
		const PtxObject& o = sc->objects[i];
		if (o.first_material < 0 || o.first_material + o.num_materials > sc->num_materials) return PTX_E_INVALID;
		if (o.type == PTX_OBJ_MESH) {
			if (o.first_triangle < 0 || o.first_triangle + o.num_triangles > sc->num_triangles) return PTX_E_INVALID;
			if (o.first_node < 0 || o.first_node + o.num_nodes > sc->num_nodes) return PTX_E_INVALID;
			if (o.first_leaf_tri < 0 || o.first_leaf_tri + o.num_leaf_tris > sc->num_leaf_tris) return PTX_E_INVALID;
		}
	}
	return PTX_OK;
}

template <typename F>
void parallelRows(int height, int nthreads, F fn) {
	if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
	if (nthreads < 1) nthreads = 1;
	if (nthreads > height) nthreads = height > 0 ? height : 1;
	std::atomic<int> next(0);
	std::vector<std::thread> th;
	for (int t = 0; t < nthreads; ++t)
		th.emplace_back([&, t]() {
			for (;;) {
				int y = next.fetch_add(1);
				if (y >= height) break;
				fn(y, t);
			}
		});
	for (size_t i = 0; i < th.size(); ++i) th[i].join();
}

} // namespace

// ==== exported API ===================================================================================

extern "C" int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }
extern "C" void orc_free(void* p) { free(p); }
extern "C" float orc_u01(uint32_t x) { return u01(x); }
extern "C" void orc_sincos(float theta, float* s, float* c) { sincos_poly(theta, s, c); }
extern "C" float orc_acos01(float x) { return acos01_poly(x); }
extern "C" void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
	philox(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

// reference: src/cutility.cuh:7-15
// The V3 helpers above on one pair (a, b) and scalar s, in the layout of oracle/ref_shim.cpp:ref_vec3_ops
// (which runs the reference's own external/include/cutil_math.h): dot, length(a), cross, normalize(a),
// reflect(a, b), a*s, a/s (= a * (1/s), cutil_math.h:348-352), (a+b).x, (a-b).x, (a*b).x
extern "C" void orc_vec3_ops(const float* pa, const float* pb, float s, float* out) {
	const V3 a = ld3(pa), b = ld3(pb);
	const V3 c = cross(a, b), n = normalize(a), r = reflect(a, b), m = scale(a, s), q = scale(a, 1.0f / s);
	out[0] = dot(a, b); out[1] = length(a);
	out[2] = c.x; out[3] = c.y; out[4] = c.z;
	out[5] = n.x; out[6] = n.y; out[7] = n.z;
	out[8] = r.x; out[9] = r.y; out[10] = r.z;
	out[11] = m.x; out[12] = m.y; out[13] = m.z;
	out[14] = q.x; out[15] = q.y; out[16] = q.z;
	out[17] = add(a, b).x; out[18] = sub(a, b).x; out[19] = mul(a, b).x;
}

extern "C" uint32_t orc_wang_hash(uint32_t seed) {
	seed = (seed ^ 61) ^ (seed >> 16);
	seed *= 9;
	seed = seed ^ (seed >> 4);
	seed *= 0x27d4eb2d;
	seed = seed ^ (seed >> 15);
	return seed;
}

// reference: src/utility.cpp:5-41 (quicksort + median). Sorting floats ascending gives one
// result whatever the algorithm, so std::sort-free insertion of the same recursive scheme is
// not needed: a simple in-place quicksort is restated.
static void orc_quicksort(float* arr, int left, int right) {
	int i = left, j = right;
	float pivot = arr[(left + right) / 2];
	while (i <= j) {
		while (arr[i] < pivot) i++;
		while (arr[j] > pivot) j--;
		if (i <= j) { float tmp = arr[i]; arr[i] = arr[j]; arr[j] = tmp; i++; j--; }
	}
	if (left < j) orc_quicksort(arr, left, j);
	if (i < right) orc_quicksort(arr, i, right);
}

extern "C" float orc_fmedian(float* arr, int size) {
	if (size < 1) return 0.0f;
	orc_quicksort(arr, 0, size - 1);
	int halfs = size / 2;
	if (size % 2 == 0) return ((arr[halfs - 1] + arr[halfs]) / 2.0f);
	return arr[halfs];
}

// ---- KD build (reference: src/KDTree.cpp:18-150) -------------------------------------------------
namespace {

const float BBOX_BIAS = 0.00005f;

struct KdBuilder {
	const float* abc;
	std::vector<PtxKDNode> nodes;
	std::vector<std::vector<int> > lists; // per node
	int max_depth;

	V3 vert(int t, int k) const { return ld3(abc + (size_t)t * 9 + 3 * k); }
	// reference :18-20 triangleMiddle: (a + b + c) / 3.0f with operator/ = multiply by (1.0f / 3.0f)
	V3 middle(int t) const {
		float inv = 1.0f / 3.0f;
		return scale(add(add(vert(t, 0), vert(t, 1)), vert(t, 2)), inv);
	}
	// reference :53-66 makeTrgBBox
	void trgBox(int t, float* mn, float* mx) const {
		V3 a = vert(t, 0), b = vert(t, 1), c = vert(t, 2);
		mn[0] = fminf(a.x, fminf(b.x, c.x)); mn[1] = fminf(a.y, fminf(b.y, c.y)); mn[2] = fminf(a.z, fminf(b.z, c.z));
		mx[0] = fmaxf(a.x, fmaxf(b.x, c.x)); mx[1] = fmaxf(a.y, fmaxf(b.y, c.y)); mx[2] = fmaxf(a.z, fmaxf(b.z, c.z));
		for (int k = 0; k < 3; ++k) { mn[k] += -BBOX_BIAS; mx[k] += BBOX_BIAS; }
	}

	// reference :84-150 buildKDTree
	int build(const std::vector<int>& idxs, int depth) {
		int me = (int)nodes.size();
		PtxKDNode nd;
		memset(&nd, 0, sizeof(nd));
		nd.left = nd.right = -1;
		nd.first_tri = -1;
		nd.num_tris = (int)idxs.size();
		nodes.push_back(nd);
		lists.push_back(idxs);
		const int n = (int)idxs.size();
		if (n == 0) return me;
		if (n == 1) {
			trgBox(idxs[0], nodes[me].bmin, nodes[me].bmax);
			return me;
		}
		float mn[3], mx[3];
		trgBox(idxs[0], mn, mx);
		for (int i = 0; i < n; ++i) {
			// reference :68-82 expandBBox: the triangle's biased box, biased once more
			float tmn[3], tmx[3];
			trgBox(idxs[i], tmn, tmx);
			for (int k = 0; k < 3; ++k) {
				mn[k] = fminf(mn[k], tmn[k] - BBOX_BIAS);
				mx[k] = fmaxf(mx[k], tmx[k] + BBOX_BIAS);
			}
		}
		for (int k = 0; k < 3; ++k) { nodes[me].bmin[k] = mn[k]; nodes[me].bmax[k] = mx[k]; }

		int axis = depth % 3;
		if (depth > max_depth) max_depth = depth;
		std::vector<float> coords((size_t)n);
		for (int i = 0; i < n; ++i) {
			V3 m = middle(idxs[i]);
			coords[(size_t)i] = axis == 0 ? m.x : (axis == 1 ? m.y : m.z);
		}
		std::vector<float> sorted(coords);
		float median = orc_fmedian(sorted.data(), n);

		std::vector<int> left, right;
		for (int i = 0; i < n; ++i) {
			if (coords[(size_t)i] > median) left.push_back(idxs[i]);
			else right.push_back(idxs[i]);
		}
		if ((int)left.size() > 16) { // MAX_KD_LEAF_TRGS, reference src/config.h:18
			int l = build(left, depth + 1);
			int r = build(right, depth + 1);
			nodes[me].left = l;
			nodes[me].right = r;
		}
		return me;
	}
};

} // namespace

extern "C" int orc_kd_build(const float* abc, int n, PtxKDNode** nodes_out, int32_t** leaf_tris_out, int* num_leaf_tris_out,
                            int* max_depth_out) {
	KdBuilder kb;
	kb.abc = abc;
	kb.max_depth = 0;
	std::vector<int> all((size_t)n);
	for (int i = 0; i < n; ++i) all[(size_t)i] = i;
	kb.build(all, 0);
	const int nn = (int)kb.nodes.size();
	size_t total = 0;
	for (int i = 0; i < nn; ++i)
		if (kb.nodes[i].left == -1 && kb.nodes[i].right == -1) total += kb.lists[i].size();
	PtxKDNode* nodes = (PtxKDNode*)malloc(sizeof(PtxKDNode) * (size_t)(nn > 0 ? nn : 1));
	int32_t* lt = (int32_t*)malloc(sizeof(int32_t) * (total > 0 ? total : 1));
	size_t pos = 0;
	for (int i = 0; i < nn; ++i) {
		nodes[i] = kb.nodes[i];
		if (nodes[i].left == -1 && nodes[i].right == -1) {
			nodes[i].first_tri = (int32_t)pos;
			for (size_t k = 0; k < kb.lists[i].size(); ++k) lt[pos++] = kb.lists[i][k];
		}
	}
	*nodes_out = nodes;
	*leaf_tris_out = lt;
	*num_leaf_tris_out = (int)total;
	*max_depth_out = kb.max_depth;
	return nn;
}

extern "C" int orc_trace_rays(const PtxSceneDesc* scene, int n, const float* origins, const float* dirs, int object, PtxHit* hits) {
	int rc = checkScene(scene);
	if (rc != PTX_OK) return rc;
	if (object < -1 || object >= scene->num_objects) return PTX_E_INVALID;
	for (int i = 0; i < n; ++i) {
		Ray ray;
		init_ray(&ray, ld3(origins + 3 * i), ld3(dirs + 3 * i));
		IntersectInfo ii;
		Counters cnt;
		memset(&cnt, 0, sizeof(cnt));
		cnt.leaf_hash = 2166136261u;
		bool hit = intersectScene(scene, &ray, object, &ii, &cnt);
		PtxHit h;
		memset(&h, 0, sizeof(h));
		h.hit = hit ? 1 : 0;
		h.object = hit ? ii.iobj_index : -1;
		h.prim = (hit && ii.itrg >= 0) ? ii.itrg : -1;
		memcpy(&h.t_bits, &ii.dist, 4);
		h.leaf_hash = cnt.leaf_hash;
		h.n_inner = (uint32_t)cnt.n_inner; h.n_box = (uint32_t)cnt.n_box; h.n_tri = (uint32_t)cnt.n_tri; h.n_leaf = (uint32_t)cnt.n_leaf;
		if (hit && ii.iobj_index >= 0) { // no object recorded (all successes at infinite distance): zeros, not stale memory
			h.point[0] = ii.ipoint.x; h.point[1] = ii.ipoint.y; h.point[2] = ii.ipoint.z;
			h.normal[0] = ii.normal.x; h.normal[1] = ii.normal.y; h.normal[2] = ii.normal.z;
			h.bary[0] = ii.bary.x; h.bary[1] = ii.bary.y; h.bary[2] = ii.bary.z;
		}
		hits[i] = h;
	}
	return PTX_OK;
}

extern "C" int orc_generate_rays(const PtxCamera* cam, int sample, uint64_t seed, float* origins, float* dirs) {
	const uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
	for (int y = 0; y < cam->height; ++y)
		for (int x = 0; x < cam->width; ++x) {
			int i = x + y * cam->width;
			V3 mid, centre;
			pixelMidpoint(cam, x, y, &mid, &centre);
			V3 d = sample < 0 ? centre : jitteredDir(cam, mid, (uint32_t)i, (uint32_t)sample, k0, k1);
			origins[3 * i] = cam->position[0]; origins[3 * i + 1] = cam->position[1]; origins[3 * i + 2] = cam->position[2];
			dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
		}
	return PTX_OK;
}

extern "C" int orc_bsdf_eval(int n, PtxBsdfCase* cases) {
	for (int i = 0; i < n; ++i) {
		PtxBsdfCase& c = cases[i];
		Ray ray;
		init_ray(&ray, v3(0.0f), ld3(c.dir));
		V3 mask = v3(1.0f);
		Uniforms rs;
		for (int k = 0; k < 4; ++k) rs.u[k] = c.u[k];
		rs.next = 0;
		switch (c.mat_type) {
			case PTX_MAT_DIFFUSE: Diffuse_BRDF(&ray, ld3(c.normal), ld3(c.point), &rs); break;
			case PTX_MAT_SPECULAR: Specular_Diffuse_BRDF(&ray, c.reflect_factor, c.sharpness, ld3(c.normal), ld3(c.point), &rs); break;
			case PTX_MAT_TRANSPARENT: Refractive_BRDF(&ray, c.refract_index, c.reflect_factor, &mask, ld3(c.normal), ld3(c.point), &rs); break;
			default: break;
		}
		c.out_origin[0] = ray.o.x; c.out_origin[1] = ray.o.y; c.out_origin[2] = ray.o.z;
		c.out_dir[0] = ray.d.x; c.out_dir[1] = ray.d.y; c.out_dir[2] = ray.d.z;
		c.out_mask[0] = mask.x; c.out_mask[1] = mask.y; c.out_mask[2] = mask.z;
	}
	return PTX_OK;
}

extern "C" int orc_radiance_sample(const PtxSceneDesc* scene, const PtxCamera* cam, int sample, uint64_t seed, float* rgb, int nthreads) {
	int rc = checkScene(scene);
	if (rc != PTX_OK) return rc;
	const uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
	parallelRows(cam->height, nthreads, [&](int y, int) {
		PathStats ps;
		memset(&ps, 0, sizeof(ps));
		for (int x = 0; x < cam->width; ++x) {
			V3 r = pixelSample(scene, cam, x, y, sample, k0, k1, &ps);
			size_t i = (size_t)x + (size_t)y * cam->width;
			rgb[3 * i] = r.x; rgb[3 * i + 1] = r.y; rgb[3 * i + 2] = r.z;
		}
	});
	return PTX_OK;
}

extern "C" int orc_render(const PtxSceneDesc* scene, const PtxCamera* cam, int first_sample, int num_samples, uint64_t seed,
                          float* accum, int nthreads, PtxStats* stats) {
	int rc = checkScene(scene);
	if (rc != PTX_OK) return rc;
	const uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
	int nt = nthreads <= 0 ? (int)std::thread::hardware_concurrency() : nthreads;
	if (nt < 1) nt = 1;
	std::vector<PathStats> tstats((size_t)nt);
	memset(tstats.data(), 0, sizeof(PathStats) * (size_t)nt);
	std::vector<uint64_t> tpaths((size_t)nt, 0);
	parallelRows(cam->height, nt, [&](int y, int t) {
		PathStats* ps = &tstats[(size_t)t];
		for (int x = 0; x < cam->width; ++x) {
			size_t i = (size_t)x + (size_t)y * cam->width;
			// reference: src/pathtracing.cu:151 acc_image[index] += make_float4(radiance), once per iteration
			float ax = accum[4 * i], ay = accum[4 * i + 1], az = accum[4 * i + 2], aw = accum[4 * i + 3];
			for (int s = first_sample; s < first_sample + num_samples; ++s) {
				V3 r = pixelSample(scene, cam, x, y, s, k0, k1, ps);
				ax += r.x; ay += r.y; az += r.z; aw += 0.0f;
				tpaths[(size_t)t]++;
			}
			accum[4 * i] = ax; accum[4 * i + 1] = ay; accum[4 * i + 2] = az; accum[4 * i + 3] = aw;
		}
	});
	if (stats) {
		memset(stats, 0, sizeof(*stats));
		for (int t = 0; t < nt; ++t) {
			const PathStats& p = tstats[(size_t)t];
			stats->rays += p.rays; stats->paths += tpaths[(size_t)t];
			stats->n_inner += p.cnt.n_inner; stats->n_box += p.cnt.n_box; stats->n_tri += p.cnt.n_tri;
			stats->n_sphere += p.cnt.n_sphere; stats->n_leaf += p.cnt.n_leaf;
			stats->hit_mesh += p.hit_mesh; stats->hit_sphere += p.hit_sphere; stats->hit_textured += p.hit_tex;
			stats->trav_rays += p.cnt.t_rays; stats->trav_inner += p.cnt.t_inner; stats->trav_box += p.cnt.t_box;
			stats->trav_tri += p.cnt.t_tri; stats->trav_leaf += p.cnt.t_leaf;
		}
	}
	return PTX_OK;
}

// reference: src/kernel.cu:40-54 writeImageToPBO (clamp = fmaxf(a, fminf(f, b)), cutil_math.h:66-69)
extern "C" int orc_tonemap(const float* accum, int n_pixels, int iter, float gamma, unsigned char* rgba8) {
	if (iter < 1) return PTX_E_INVALID;
	float inv_iter = 1 / (float)iter;
	for (int i = 0; i < n_pixels; ++i) {
		for (int k = 0; k < 3; ++k) {
			float v = accum[4 * i + k] * inv_iter;
			v = fmaxf(0.0f, fminf(v, 1.0f));
			rgba8[4 * i + k] = (unsigned char)(255.0f * powf(v, gamma));
		}
		rgba8[4 * i + 3] = 0;
	}
	return PTX_OK;
}
