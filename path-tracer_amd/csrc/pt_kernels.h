// __global__ kernels of the gfx950 path tracer.
//
//   render_kernel      the whole hot path: jittered primary ray -> (traverse + intersect ->
//                      shade/BSDF)* -> per-pixel accumulation, for a range of samples.
//                      Replaces generatePrimaryRays + jitterPrimaryRays + tracePaths and their
//                      three launches + three device syncs PER SAMPLE (reference:
//                      src/pathtracing.cu:73-182) by one launch per sample RANGE; rays never
//                      touch HBM, the accumulator is read and written once per launch.
//   trace_kernel       rayIntersectsScene on caller rays (parity entry point)
//   genrays_kernel     generatePrimaryRays / jitterPrimaryRays results (parity entry point)
//   bsdf_kernel        BSDF known-answer evaluation (parity entry point)
//   tonemap_kernel     writeImageToPBO (reference: src/kernel.cu:40-54)
//
// Thread -> pixel map: a 256-thread workgroup owns a 16x16 pixel tile, each of its four waves
// an 8x8 sub-tile (lane = x + 8*y), so the 64 rays of a wave start out spatially coherent.
// Path regeneration: a lane owns one pixel for the whole launch and walks its samples in
// order; its loop body is ONE path segment, and when a path ends the lane immediately starts
// the next sample of its pixel. Lanes of a wave are therefore always in "trace one segment"
// regardless of how long each individual path is, and the per-pixel sum is formed in sample
// order (bit-identical to adding one sample per launch, the reference's schedule).
#pragma once
#include "pt_shade.h"
#include "../../include/ptx_hip.h"

#define PT_BLOCK 256

PT_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

// Bijective remap of the linear workgroup id so that ids with equal (id % 8) — the workgroups
// the dispatcher places on one XCD — receive one contiguous run of tiles and so share that
// XCD's L2 for the geometry under those tiles (cdna_hip_programming.md §5.5 T1).
PT_DEV int xcd_remap(int orig, int nwg) {
	const int q = nwg / 8, r = nwg % 8;
	const int xcd = orig % 8;
	const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
	return base + orig / 8;
}

// reference: src/pathtracing.cu:84-92 (pixel midpoint and centre direction)
PT_DEV void pixel_midpoint(const DevCamera& cam, int x, int y, f3& mid, f3& centre_dir) {
	const float ps = cam.pixel_size;
	f3 sdv = cam.screen_dist * normalize3(mk3(cam.dir[0], cam.dir[1], cam.dir[2]));
	f3 pv = (cam.half_x - (float)x * ps - ps / 2.0f) * normalize3(mk3(cam.right[0], cam.right[1], cam.right[2])) +
	        (cam.half_y - (float)y * ps - ps / 2.0f) * normalize3(mk3(cam.up[0], cam.up[1], cam.up[2]));
	centre_dir = normalize3(sdv + pv);
	mid = mk3(cam.pos[0], cam.pos[1], cam.pos[2]) + sdv + pv;
}

// reference: src/pathtracing.cu:121-126 (jittered direction through the pixel)
PT_DEV f3 jittered_dir(const DevCamera& cam, f3 mid, uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1) {
	const u4 r = philox4x32_10(pixel, sample, PT_DIM_JITTER, 0u, k0, k1);
	const float sc = cam.pixel_size * cam.aa_jitter;
	const float jx = sc * (u01(r.x) - 0.5f);
	const float jy = sc * (u01(r.y) - 0.5f);
	return normalize3(mid + jx * mk3(cam.right[0], cam.right[1], cam.right[2]) +
	                  jy * mk3(cam.up[0], cam.up[1], cam.up[2]) - mk3(cam.pos[0], cam.pos[1], cam.pos[2]));
}

template <int FLAGS>
__global__ __launch_bounds__(PT_BLOCK) void render_kernel(DevScene sc, DevCamera cam, float4* __restrict__ accum,
                                                          int first_sample, int num_samples, uint32_t k0,
                                                          uint32_t k1, PtxStats* __restrict__ stats) {
	extern __shared__ int lds_stack[];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int tiles_x = (cam.width + 15) >> 4;
	const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
	const int bx = tile % tiles_x, by = tile / tiles_x;
	const int x = bx * 16 + (wave & 1) * 8 + (lane & 7);
	const int y = by * 16 + (wave >> 1) * 8 + (lane >> 3);
	const bool valid = (x < cam.width) && (y < cam.height);
	int* stack = lds_stack + tid;

	uint64_t n_rays = 0, n_paths = 0;
	TraceCounters cnt = {0, 0, 0, 0, 0, 2166136261u};
	uint64_t c_inner = 0, c_box = 0, c_tri = 0, c_sphere = 0, c_leaf = 0;
	uint64_t c_hit_mesh = 0, c_hit_sphere = 0, c_hit_tex = 0, c_active = 0, c_total = 0;

	if (valid) {
		const uint32_t pixel = (uint32_t)(x + y * cam.width);
		f3 mid, centre;
		pixel_midpoint(cam, x, y, mid, centre);
		const f3 cam_pos = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
		float4 acc = accum[pixel];

		int s = first_sample;
		const int end = first_sample + num_samples;
		int bounce = 0;
		f3 mask = mk3(1.0f);
		Ray ray;
		if (s < end) init_ray(ray, cam_pos, jittered_dir(cam, mid, pixel, (uint32_t)s, k0, k1));

		while (s < end) {
			if (FLAGS & PT_F_STATS) {
				const unsigned long long m = __ballot(1);
				c_active += 1;
				if (lane == (int)__ffsll((long long)m) - 1) c_total += 64;
			}
			f3 out = mk3(0.0f);
			bool done;
			if (bounce >= cam.max_bounces) {
				// reference: src/radiance.cuh:27,86 — loop exhausted, radiance 0
				done = true;
			} else {
				HitInfo hi;
				float dbg = 1.0f;
				n_rays++;
				const bool hit = intersect_scene<FLAGS>(sc, ray, stack, PT_BLOCK, -1, hi, cnt, dbg);
				if (!hit) {
					done = true; // reference: src/radiance.cuh:30-34
					if (FLAGS & PT_F_DEBUGMASK) out = mk3(1.0f) - mk3(dbg);
				} else if (FLAGS & PT_F_DEBUGMASK) {
					// reference: src/radiance.cuh:36-37 aabb_mode
					const MaterialRec mat = load_material(sc, hi.material);
					out = 0.8f * (mk3(1.0f) - mk3(dbg)) + (0.2f * mat.color) * fabsf(dot3(ray.d, hi.normal));
					done = true;
				} else {
					if (FLAGS & PT_F_STATS) {
						if (hi.tri >= 0) c_hit_mesh++; else c_hit_sphere++;
						const MaterialRec mt = load_material(sc, hi.material);
						if (cam.texture_enabled && mt.texture > -1 && hi.tri >= 0) c_hit_tex++;
					}
					const u4 rnd = philox4x32_10(pixel, (uint32_t)s, (uint32_t)bounce, 0u, k0, k1);
					done = shade_segment(sc, cam, hi, ray, mask, rnd, out);
					if (!done) init_ray(ray, ray.o, ray.d);   // the BSDF set origin and direction (pt_shade.h)
					bounce++;
				}
			}
			if (done) {
				// reference: src/pathtracing.cu:148-151 acc_image[index] += make_float4(radiance)
				acc.x += out.x; acc.y += out.y; acc.z += out.z; acc.w += 0.0f;
				n_paths++;
				s++;
				bounce = 0;
				mask = mk3(1.0f);
				if (s < end) init_ray(ray, cam_pos, jittered_dir(cam, mid, pixel, (uint32_t)s, k0, k1));
			}
		}
		accum[pixel] = acc;
	}

	// counters: one atomic per wave per counter
	n_rays = wave_sum_u64(n_rays);
	n_paths = wave_sum_u64(n_paths);
	if (FLAGS & PT_F_STATS) {
		c_inner = wave_sum_u64(cnt.n_inner); c_box = wave_sum_u64(cnt.n_box); c_tri = wave_sum_u64(cnt.n_tri);
		c_sphere = wave_sum_u64(cnt.n_sphere); c_leaf = wave_sum_u64(cnt.n_leaf);
		c_hit_mesh = wave_sum_u64(c_hit_mesh); c_hit_sphere = wave_sum_u64(c_hit_sphere);
		c_hit_tex = wave_sum_u64(c_hit_tex); c_active = wave_sum_u64(c_active); c_total = wave_sum_u64(c_total);
	}
	if (lane == 0) {
		atomicAdd((unsigned long long*)&stats->rays, (unsigned long long)n_rays);
		atomicAdd((unsigned long long*)&stats->paths, (unsigned long long)n_paths);
		if (FLAGS & PT_F_STATS) {
			atomicAdd((unsigned long long*)&stats->n_inner, (unsigned long long)c_inner);
			atomicAdd((unsigned long long*)&stats->n_box, (unsigned long long)c_box);
			atomicAdd((unsigned long long*)&stats->n_tri, (unsigned long long)c_tri);
			atomicAdd((unsigned long long*)&stats->n_sphere, (unsigned long long)c_sphere);
			atomicAdd((unsigned long long*)&stats->n_leaf, (unsigned long long)c_leaf);
			atomicAdd((unsigned long long*)&stats->hit_mesh, (unsigned long long)c_hit_mesh);
			atomicAdd((unsigned long long*)&stats->hit_sphere, (unsigned long long)c_hit_sphere);
			atomicAdd((unsigned long long*)&stats->hit_textured, (unsigned long long)c_hit_tex);
			atomicAdd((unsigned long long*)&stats->lane_steps_active, (unsigned long long)c_active);
			atomicAdd((unsigned long long*)&stats->lane_steps_total, (unsigned long long)c_total);
		}
	}
}

// Parity entry: rayIntersectsScene / rayIntersectsObject on caller-supplied rays.
__global__ __launch_bounds__(PT_BLOCK) void trace_kernel(DevScene sc, int n, const float* __restrict__ origins,
                                                         const float* __restrict__ dirs, int only_object,
                                                         PtxHit* __restrict__ hits) {
	extern __shared__ int lds_stack[];
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= n) return;
	int* stack = lds_stack + threadIdx.x;
	Ray ray;
	init_ray(ray, mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
	         mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
	HitInfo hi;
	TraceCounters cnt = {0, 0, 0, 0, 0, 2166136261u};
	float dbg = 1.0f;
	const bool hit = intersect_scene<PT_F_STATS>(sc, ray, stack, PT_BLOCK, only_object, hi, cnt, dbg);
	PtxHit h;
	h.hit = hit ? 1 : 0;
	h.object = hit ? hi.object : -1;
	h.prim = (hit && hi.tri >= 0) ? __float_as_int(sc.tri_geom[3 * hi.tri + 2].y) : -1;
	h.t_bits = __float_as_uint(hi.dist);
	h.leaf_hash = cnt.leaf_hash;
	h.n_inner = cnt.n_inner; h.n_box = cnt.n_box; h.n_tri = cnt.n_tri; h.n_leaf = cnt.n_leaf;
	// "hit" is the reference's return value ("some test succeeded"); when every success sat at an infinite
	// distance no object was recorded (object = -1) and the reference leaves its IntersectInfo untouched:
	// the record then carries zeros rather than stale memory
	if (hit && hi.object >= 0) {
		h.point[0] = hi.point.x; h.point[1] = hi.point.y; h.point[2] = hi.point.z;
		h.normal[0] = hi.normal.x; h.normal[1] = hi.normal.y; h.normal[2] = hi.normal.z;
		h.bary[0] = hi.bary.x; h.bary[1] = hi.bary.y; h.bary[2] = hi.bary.z;
	} else {
		h.point[0] = h.point[1] = h.point[2] = 0.0f;
		h.normal[0] = h.normal[1] = h.normal[2] = 0.0f;
		h.bary[0] = h.bary[1] = h.bary[2] = 0.0f;
	}
	hits[i] = h;
}

__global__ __launch_bounds__(PT_BLOCK) void genrays_kernel(DevCamera cam, int sample, uint32_t k0, uint32_t k1,
                                                           float* __restrict__ origins, float* __restrict__ dirs) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= cam.width * cam.height) return;
	const int x = i % cam.width, y = i / cam.width;
	f3 mid, centre;
	pixel_midpoint(cam, x, y, mid, centre);
	f3 d = sample < 0 ? centre : jittered_dir(cam, mid, (uint32_t)i, (uint32_t)sample, k0, k1);
	origins[3 * i] = cam.pos[0]; origins[3 * i + 1] = cam.pos[1]; origins[3 * i + 2] = cam.pos[2];
	dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}

__global__ __launch_bounds__(PT_BLOCK) void bsdf_kernel(int n, PtxBsdfCase* __restrict__ cases) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= n) return;
	PtxBsdfCase c = cases[i];
	Ray ray;
	init_ray(ray, mk3(0.0f), mk3(c.dir[0], c.dir[1], c.dir[2]));
	f3 mask = mk3(1.0f);
	UniformStream rs;
	rs.u[0] = c.u[0]; rs.u[1] = c.u[1]; rs.u[2] = c.u[2]; rs.u[3] = c.u[3];
	rs.next = 0;
	const f3 nrm = mk3(c.normal[0], c.normal[1], c.normal[2]);
	const f3 ip = mk3(c.point[0], c.point[1], c.point[2]);
	switch (c.mat_type) {
		case 1: diffuse_brdf(ray, nrm, ip, rs); break;
		case 2: specular_diffuse_brdf(ray, c.reflect_factor, c.sharpness, nrm, ip, rs); break;
		case 3: refractive_brdf(ray, c.refract_index, c.reflect_factor, mask, nrm, ip, rs); break;
		default: break;
	}
	c.out_origin[0] = ray.o.x; c.out_origin[1] = ray.o.y; c.out_origin[2] = ray.o.z;
	c.out_dir[0] = ray.d.x; c.out_dir[1] = ray.d.y; c.out_dir[2] = ray.d.z;
	c.out_mask[0] = mask.x; c.out_mask[1] = mask.y; c.out_mask[2] = mask.z;
	cases[i] = c;
}

// reference: src/kernel.cu:40-54 writeImageToPBO
__global__ __launch_bounds__(PT_BLOCK) void tonemap_kernel(uchar4* __restrict__ out, float gamma, int n, int iter,
                                                           const float4* __restrict__ accum) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= n) return;
	const float inv_iter = 1.0f / (float)iter;
	const float4 a = accum[i];
	uchar4 p;
	p.w = 0;
	p.x = (unsigned char)(255.0f * powf(fmaxf(0.0f, fminf(a.x * inv_iter, 1.0f)), gamma));
	p.y = (unsigned char)(255.0f * powf(fmaxf(0.0f, fminf(a.y * inv_iter, 1.0f)), gamma));
	p.z = (unsigned char)(255.0f * powf(fmaxf(0.0f, fminf(a.z * inv_iter, 1.0f)), gamma));
	out[i] = p;
}

// Exhaustive check of pt_rcp_newton against the IEEE quotient over all float bit patterns with lo <= |a| <= hi.
__global__ __launch_bounds__(PT_BLOCK) void rcp_selftest_kernel(float lo, float hi, unsigned long long* out /* checked, mismatches, first bad */) {
	unsigned long long checked = 0, bad = 0;
	unsigned long long first = ~0ull;
	const unsigned long long stride = (unsigned long long)gridDim.x * PT_BLOCK;
	for (unsigned long long b = (unsigned long long)blockIdx.x * PT_BLOCK + threadIdx.x; b < (1ull << 32); b += stride) {
		const float a = __uint_as_float((uint32_t)b);
		const float m = __builtin_fabsf(a);
		if (!(m >= lo && m <= hi)) continue;
		checked++;
		const float want = 1.0f / a;
		const float got = pt_rcp_newton(a);
		if (__float_as_uint(want) != __float_as_uint(got)) { bad++; if (b < first) first = b; }
	}
	checked = wave_sum_u64(checked);
	bad = wave_sum_u64(bad);
	if (first != ~0ull) atomicMin(&out[2], first);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&out[0], checked);
		atomicAdd(&out[1], bad);
	}
}

// The float3 helpers of pt_math.h on caller-supplied pairs, in the layout of oracle/ref_shim.cpp:ref_vec3_ops
// (the reference's own cutil_math.h compiled on the host): pins the evaluation order under every hot-path function.
__global__ __launch_bounds__(PT_BLOCK) void vec3_selftest_kernel(int n, const float* __restrict__ a3, const float* __restrict__ b3,
                                                                 const float* __restrict__ sc, float* __restrict__ out) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= n) return;
	const f3 a = mk3(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]), b = mk3(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]);
	const float s = sc[i];
	const f3 c = cross3(a, b), nn = normalize3(a), r = reflect3(a, b), m = a * s, q = a * (1.0f / s);
	float* o = out + 20 * (size_t)i;
	o[0] = dot3(a, b); o[1] = length3(a);
	o[2] = c.x; o[3] = c.y; o[4] = c.z;
	o[5] = nn.x; o[6] = nn.y; o[7] = nn.z;
	o[8] = r.x; o[9] = r.y; o[10] = r.z;
	o[11] = m.x; o[12] = m.y; o[13] = m.z;
	o[14] = q.x; o[15] = q.y; o[16] = q.z;
	o[17] = (a + b).x; o[18] = (a - b).x; o[19] = (a * b).x;
}
