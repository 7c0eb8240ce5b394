// BSDF sampling and per-segment shading (reference: src/surfaces.cuh:14-137,
// src/radiance.cuh:15-87), one path per lane. Uniforms come from one Philox call per
// bounce (pt_math.h) and are consumed in the order the reference draws them.
// The BSDFs leave the new ray's ORIGIN and DIRECTION in `ray` (set_ray, pt_trace.h): the caller that intersects next
// computes the inverse direction (init_ray(ray, ray.o, ray.d)), once.
#pragma once
#include "pt_trace.h"

#define PT_PI_X2 6.2831853072f   // reference: src/constants.h:5
#define PT_PI_D2 1.5707963267f   // reference: src/constants.h:6

struct UniformStream {
	float u[4];
	int next;
	PT_DEV float draw() {
		// a bounce consumes at most 3 (Specular: 1 + 2); index is compile-time after inlining
		float r = next == 0 ? u[0] : (next == 1 ? u[1] : (next == 2 ? u[2] : u[3]));
		++next;
		return r;
	}
};

// reference: src/surfaces.cuh:14-32 rand_cone_Dir
PT_DEV f3 rand_cone_dir(f3 axis, float ang_max_dev, UniformStream& rs) {
	f3 u = normalize3((axis.x != 0.0f || axis.y != 0.0f) ? mk3(axis.y, -axis.x, 0.0f) : mk3(0.0f, axis.z, -axis.y));
	f3 v = normalize3(cross3(u, axis));
	const float r2 = pt_sin(ang_max_dev) * rs.draw();
	const float r = pt_sqrt(r2);
	const float theta = PT_PI_X2 * rs.draw();
	float sinth, costh;
	pt_sincos(theta, &sinth, &costh);
	// reference :18,31 — EPS is the double literal 0.0001, so the sum is formed in double and
	// rounded to float once when passed to __fsqrt_rn
	const float w2 = (float)((double)(1.0f - r2) + 0.0001);
	return normalize3((sinth * r) * u + (costh * r) * v + pt_sqrt(w2) * axis);
}

// reference: src/surfaces.cuh:37-44 Diffuse_BRDF
PT_DEV void diffuse_brdf(Ray& ray, f3 n, f3 ip, UniformStream& rs) {
	f3 dir = rand_cone_dir(n, PT_PI_D2, rs);
	set_ray(ray, ip + n * 0.00005f, dir);
}

// reference: src/surfaces.cuh:48-70 Specular_Diffuse_BRDF
PT_DEV void specular_diffuse_brdf(Ray& ray, float refl_factor, float sharpness, f3 n, f3 ip, UniformStream& rs) {
	f3 new_orig = ip + n * 0.0001f;
	float r2 = rs.draw();
	if (r2 > refl_factor) {
		diffuse_brdf(ray, n, ip, rs);
	} else {
		const float max_cone_ang = PT_PI_D2 * (1.0f - sharpness);
		const float scatt_ang = PT_PI_D2 - pt_acos01(fabsf(dot3(ray.d, n)));
		const float ang = (scatt_ang < max_cone_ang) ? scatt_ang : max_cone_ang; // fmin(max_cone, scatt)
		f3 dir = rand_cone_dir(reflect3(ray.d, n), ang, rs);
		set_ray(ray, new_orig, dir);
	}
}

// reference: src/surfaces.cuh:76-137 Refractive_BRDF (SCENE_REFRACTION_INDEX = 1.0f, src/config.h:9)
PT_DEV void refractive_brdf(Ray& ray, float n_obj, float refl_factor, f3& mask, f3 n, f3 ip, UniformStream& rs) {
	const float REFL_BIAS = 0.05f, REFL_BIAS_LOW = 0.0005f, N_SCENE = 1.0f;
	f3 n_or = dot3(n, ray.d) < 0.0f ? n : (-1.0f) * n;
	bool outside = dot3(n, n_or) > 0.0f;
	float nn = outside ? (N_SCENE / n_obj) : (n_obj / N_SCENE);
	float cos_ray = dot3(ray.d, n_or);
	float cos2refr = 1.0f - nn * nn * (1.0f - cos_ray * cos_ray);
	if (cos2refr < 0.0f) {
		// total internal reflection
		set_ray(ray, ip + n_or * REFL_BIAS, reflect3(ray.d, n_or));
	} else {
		float cosrefr = pt_sqrt(cos2refr);
		f3 trans = normalize3(ray.d * nn - n * ((outside ? 1.0f : -1.0f) * (cos_ray * nn + cosrefr)));
		float ndiff = n_obj - N_SCENE;
		float nsum = n_obj + N_SCENE;
		float refl0 = (ndiff * ndiff) / (nsum * nsum);
		float pw = 1.0f - (outside ? -cos_ray : dot3(trans, n));
		float reflectance = refl0 + (1.0f - refl0) * pw * pw * pw * pw * pw;
		float energy = refl_factor + (1.0f - refl_factor - refl_factor) * reflectance;
		if (rs.draw() < refl_factor) {
			mask = mask * (reflectance / energy);
			set_ray(ray, ip + n_or * REFL_BIAS, reflect3(ray.d, n));
		} else {
			mask = mask * ((1.0f - reflectance) / (1.0f - energy));
			set_ray(ray, ip - n_or * REFL_BIAS_LOW, trans);
		}
	}
}

// reference: src/world_load.cpp:177-183 texture object = point filter, wrap, normalized coords.
// Texel (i, j) with i = floor(frac(u) * W), j = floor(frac(v) * H), clamped to the last texel if
// frac() rounds up to 1.
PT_DEV float4 tex_fetch(const DevTexture& tx, float u, float v) {
	float fu = u - floorf(u);
	float fv = v - floorf(v);
	int i = (int)floorf(fu * (float)tx.width);
	int j = (int)floorf(fv * (float)tx.height);
	i = i < 0 ? 0 : (i >= tx.width ? tx.width - 1 : i);
	j = j < 0 ? 0 : (j >= tx.height ? tx.height - 1 : j);
	return tx.texels[(size_t)j * (size_t)tx.width + (size_t)i];
}

struct MaterialRec {
	int type, texture;
	f3 color, emittance;
	float reflect_factor, refract_index, sharpness;
};

// `mats`: the scene's material table, or a copy of it in LDS (wf_segment_kernel: the table is small and sits in the middle
// of a chain of dependent loads - state, triangle, material, texture descriptor, texel)
PT_DEV MaterialRec load_material(const float4* mats, int idx);
PT_DEV MaterialRec load_material(const DevScene& sc, int idx);
PT_DEV MaterialRec load_material(const float4* mats, int idx) {
	const float4 m0 = mats[3 * idx + 0];
	const float4 m1 = mats[3 * idx + 1];
	const float4 m2 = mats[3 * idx + 2];
	MaterialRec m;
	m.type = __float_as_int(m0.x);
	m.color = mk3(m0.y, m0.z, m0.w);
	m.texture = __float_as_int(m1.x);
	m.emittance = mk3(m1.y, m1.z, m1.w);
	m.reflect_factor = m2.x;
	m.refract_index = m2.y;
	m.sharpness = m2.z;
	return m;
}
PT_DEV MaterialRec load_material(const DevScene& sc, int idx) { return load_material(sc.materials, idx); }

// `pre`: the triangle's four tri_shade words if the caller holds them already (else they are loaded here)
struct TriShade { float4 s0, s1, s2, s3; bool held; };   // held: the caller has the record already
PT_DEV float4 hit_texel(const DevScene& sc, const DevTexture* texs, const HitInfo& hi, int texture, const TriShade& pre) {
	const float4 s0 = pre.held ? pre.s0 : sc.tri_shade[4 * hi.tri + 0];
	const float4 s1 = pre.held ? pre.s1 : sc.tri_shade[4 * hi.tri + 1];
	const float4 s2 = pre.held ? pre.s2 : sc.tri_shade[4 * hi.tri + 2];
	const float4 s3 = pre.held ? pre.s3 : sc.tri_shade[4 * hi.tri + 3];
	// reference: src/radiance.cuh:74 texuv = bx*tx_a + by*tx_b + bz*tx_c
	float tu = hi.bary.x * s0.w + hi.bary.y * s2.w + hi.bary.z * s3.y;
	float tv = hi.bary.x * s1.w + hi.bary.y * s3.x + hi.bary.z * s3.z;
	return tex_fetch(texs[texture], tu, tv);
}

// One path segment after a successful intersection (reference: src/radiance.cuh:36-83).
// Returns true when the path ends here; `out` then holds the radiance to add.
// `mats` / `texs`: the material and texture-descriptor tables to read (the scene's, or LDS copies).
PT_DEV bool shade_segment(const DevScene& sc, const DevCamera& cam, const HitInfo& hi, Ray& ray, f3& mask,
                          const u4& rnd, f3& out, const float4* mats, const DevTexture* texs, const TriShade& pre_shade);
PT_DEV bool shade_segment(const DevScene& sc, const DevCamera& cam, const HitInfo& hi, Ray& ray, f3& mask, const u4& rnd, f3& out) {
	TriShade none;
	none.held = false;
	none.s0 = none.s1 = none.s2 = none.s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	return shade_segment(sc, cam, hi, ray, mask, rnd, out, sc.materials, sc.textures, none);
}
PT_DEV bool shade_segment(const DevScene& sc, const DevCamera& cam, const HitInfo& hi, Ray& ray, f3& mask,
                          const u4& rnd, f3& out, const float4* mats, const DevTexture* texs, const TriShade& pre_shade) {
	const MaterialRec mat = load_material(mats, hi.material);
	const bool textured = cam.texture_enabled && mat.texture > -1 && hi.tri >= 0;

	if (cam.preview_mode) {
		// reference: src/radiance.cuh:42-58
		f3 tex_color = mk3(0.0f);
		float tex_blend = 0.0f, alpha = 0.0f;
		if (textured) {
			float4 texel = hit_texel(sc, texs, hi, mat.texture, pre_shade);
			alpha = texel.w;
			tex_color = mk3(texel.x, texel.y, texel.z);
			tex_blend = 0.5f;
		}
		float rdot = dot3(hi.normal, ray.d);
		if (rdot < 0.0f)
			out = -rdot * ((alpha * tex_blend) * tex_color + (1.0f - tex_blend) * mk3(1.0f));
		else
			out = rdot * (tex_blend * tex_color + (1.0f - tex_blend) * mk3(1.0f, 0.0f, 0.0f));
		return true;
	}

	// the texel is fetched BEFORE the BSDF is sampled: the fetch is the last of a chain of dependent loads and the
	// sampling below covers its round trip (the value is used after it, as in the reference: src/radiance.cuh:72-78)
	float4 texel = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
	if (textured) texel = hit_texel(sc, texs, hi, mat.texture, pre_shade);
	UniformStream rs;
	rs.u[0] = u01(rnd.x); rs.u[1] = u01(rnd.y); rs.u[2] = u01(rnd.z); rs.u[3] = u01(rnd.w);
	rs.next = 0;
	switch (mat.type) {
		case 1: diffuse_brdf(ray, hi.normal, hi.point, rs); break;
		case 2: specular_diffuse_brdf(ray, mat.reflect_factor, mat.sharpness, hi.normal, hi.point, rs); break;
		case 3: refractive_brdf(ray, mat.refract_index, mat.reflect_factor, mask, hi.normal, hi.point, rs); break;
		default: break;
	}
	if (textured) {
		mask = mask * (mk3(texel.x, texel.y, texel.z) * mat.color);
	} else {
		mask = mask * mat.color;
	}
	if (mat.type == 0) {
		out = mask * mat.emittance;
		return true;
	}
	return false;
}
