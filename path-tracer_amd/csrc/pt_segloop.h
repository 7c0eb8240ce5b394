// Segment kernel with the bounce loop INSIDE the launch (round 3; replaces wf_segment_kernel on the hot path).
//
// Why. In the headline scene 63 % of all path segments never enter a multi-node mesh's root box: their closest hit
// is known as soon as the segment kernel has intersected the cheap objects (spheres, single-leaf meshes). The
// per-bounce kernel nevertheless wrote every such ray's 64-B state to HBM, ended the launch, and the next launch
// read it back to shade it: 128 B of traffic, a launch boundary and one more link in the chain of dependent loads
// per segment, for nothing (profiles/r02_pmc.json: the kernel waits 66 % of its wave life). Here a ray whose new
// segment needs no traversal stays in the workgroup: its resolved hit goes into an LDS pool, and the next iteration
// of the workgroup shades it (the reference's loop over bounces, src/radiance.cuh:27-84) next to fresh input.
// Only rays that need a traversal, and paths that end, leave the workgroup.
//
// Results are unchanged bit for bit: a path's arithmetic depends only on (seed, pixel, sample, bounce) - Philox is
// keyed by the bounce - and parked radiance is accumulated in sample order as before.
//
// State sets. A ray that needs a traversal at bounce nb is written to state set nb (one set per bounce now:
// set nb = base + nb * set_stride) and queued in queue (nb, mesh); launch nb+1 of this kernel reads set nb after
// the traversal launches of bounce nb have merged their hits into it. A launch of bounce b therefore appends to the
// sets and queues of bounces b, b+1, ... (the rays it advanced), through the same per-(bounce, list) counters the
// later launches continue to append to.
//
// One workgroup iteration (CAP = 512 entries, 256 threads):
//   0. n_pool resolved hits wait in LDS; the workgroup claims CAP - n_pool fresh entries of its input (set b-1, or
//      pixel samples for b = 0) with one atomic;
//   1. shade / generate, one entry per lane and round. Survivors are compacted into LDS as new rays (stage A),
//      already root-box-tested against the multi-node meshes: the rays that need a traversal are counted per
//      (bounce, list) so that ONE global atomic per list and iteration reserves their slots;
//   2. one new ray per lane: cheap objects inline. Rays that enter a multi-node mesh are written to their state set
//      and queues; the others go back into the pool with their hit.
// Stage A and the pool share storage (a barrier separates each batch of reads from the writes that may reuse it).
//
// Measured (headline, 128-sample passes, ms per launch averaged over the six launches of a pass): per-bounce kernel 6.41;
// this kernel 5.70 (launches of a pass: 21.6 / 3.2 / 4.3 / 2.5 / 1.8 / 0.8 ms against 8.6 / 12.5 / 10.2 / 4.0 / 2.5 / 0.9).
// 96 registers and 30.7 KB of LDS: FIVE waves per SIMD, which is where the gain comes from as much as from the saved state
// traffic - with the material / texture-descriptor tables in LDS as well (39.4 KB, four waves) it runs at 6.58 ms, so the
// tables stay in global memory here. Two restructurings were built, pass every parity test, and were slower:
//   * two LDS queues per WORKGROUP (hits to shade / rays to intersect, growing towards each other in the same 512 entries)
//     with every step a full 256 lanes, shade or intersect: 6.2-6.4 ms (five and three barriers per 256-entry step
//     instead of seven per ~800 entry-steps);
//   * the same two queues per WAVE, 128 entries, no barrier anywhere, slots and queue entries reserved per wave in runs of
//     256 (unused ones filled with a dead path id / a duplicate queue entry), persistent grid: 6.5-6.8 ms at four or five
//     waves per SIMD, with or without a separate generate step in the launch of bounce 0.
#pragma once
#include "pt_wavefront.h"

#define WSL_WORDS 15   // o (3), d (3), mask (3), path, meta, flags | tri, t, u, v
// per (bounce, list) in dynamic LDS: count of this iteration, cursor of phase 2, reserved base
#define WSL_LIST_INTS 3
// dynamic LDS of a launch, bytes
#define WSL_DYN_BYTES(maxb, nlists) ((size_t)((maxb) > 0 ? (maxb) : 1) * (size_t)(nlists) * WSL_LIST_INTS * sizeof(int))
// input cursor of the segment launch of bounce b (5th region of the counter block)
#define WF_SCUR(wp, b) ((wp).counters[4 * (PT_WF_MAX_BOUNCES + 1) * (1 + PT_MAX_BIG) + (b)])

#ifndef WSL_OCC
#define WSL_OCC 4
#endif
// The triangles of the single-leaf meshes (sky cube, floor, ...) are copied into LDS when all their runs fit: the pair
// loop of leaf_only_mesh_pairs fetches them per lane (6 x 16 B per touched pair, each a line of its own for the L1) in a
// chain of dependent loads. 16 triangles = 768 B: 5.60 -> 5.50 ms per launch; with 32 (1.5 KB) the workgroup's LDS passes
// what five workgroups per CU can have and the kernel runs at 6.24 ms. Scenes with more fall back to tri_geom.
#ifndef WSL_CAM_LDS
#define WSL_CAM_LDS 1
#endif
#ifndef WSL_LDS_TRIS
#define WSL_LDS_TRIS 16
#endif
// COMPACT: pool entries of 13 words instead of 15 - the hit POINT instead of origin and t (shading reads nothing else of the old
// ray), and bounce, object and triangle in one word (3 + 8 + 21 bits: the host picks this variant for bounce limits <= 7, < 255
// objects, < 2^21 - 1 triangles) - and no LDS copy of the cheap triangles: 26.9 KB per workgroup, SIX workgroups per CU instead of five.
#define WSL_WORDS_COMPACT 13
#define WSL_COMPACT_MAX_BOUNCES 7
#define WSL_COMPACT_MAX_OBJECTS 254
#define WSL_COMPACT_MAX_TRIS ((1 << 21) - 2)
template <int FLAGS, int LDSTAB, int COMPACT>
__global__ __launch_bounds__(PT_BLOCK, COMPACT ? 6 : WSL_OCC) void wf_segment_loop_kernel(DevScene sc, DevCamera cam, WfParams wp) {
	const int tid = threadIdx.x, lane = tid & 63;
	constexpr int CAP = WF_ROUNDS * PT_BLOCK;
	constexpr int WORDS = COMPACT ? WSL_WORDS_COMPACT : WSL_WORDS;
	constexpr int W_U = COMPACT ? 11 : 13, W_V = COMPACT ? 12 : 14;   // where a pool entry keeps u and v
	__shared__ float4 sh_mats[LDSTAB ? 3 * WF_LDS_MATS : 1];
	__shared__ DevTexture sh_texs[LDSTAB ? WF_LDS_TEXS : 1];
	__shared__ float sh_w[WORDS][CAP];
	__shared__ int sh_live, sh_pool, sh_fresh_base, sh_fresh_n, sh_fresh_want;
	extern __shared__ int sh_lists[];   // [nl_total][WSL_LIST_INTS]
	if (LDSTAB) {
		for (int i = tid; i < 3 * sc.num_materials; i += PT_BLOCK) sh_mats[i] = sc.materials[i];
		for (int i = tid; i < sc.num_textures; i += PT_BLOCK) sh_texs[i] = sc.textures[i];
	}
	const float4* mats = LDSTAB ? sh_mats : sc.materials;
	const DevTexture* texs = LDSTAB ? sh_texs : sc.textures;
	int* shi = reinterpret_cast<int*>(&sh_w[0][0]);   // integer view: word w of entry k at shi[w * CAP + k]
	const int launch_bounce = wp.bounce;
	const int maxb = wp.max_bounces;
	const int nlists = 1 + wp.num_big;
	const int nl_total = (maxb > 0 ? maxb : 1) * nlists;
	const int n_in = launch_bounce == 0 ? wp.n_total : WF_CNT(wp, launch_bounce - 1, 0);
#if !WSL_CAM_LDS
	const f3 cam_pos = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
#endif
	const size_t in_off = launch_bounce > 0 ? (size_t)(launch_bounce - 1) * wp.set_stride : 0;
	const float4* __restrict__ i0 = wp.s0 + in_off;
	const float4* __restrict__ i1 = wp.s1 + in_off;
	const float4* __restrict__ i2 = wp.s2 + in_off;
	const float4* __restrict__ i3 = wp.s3 + in_off;

	TraceCounters cnt = {0, 0, 0, 0, 0, 0};
	uint64_t n_rays = 0, n_paths = 0, c_hm = 0, c_hs = 0, c_ht = 0;
	uint64_t sh_waves = 0, sh_lanes = 0, sh_branches = 0, sh_t[4] = {0, 0, 0, 0};
#define WSL_CLOCK() (((FLAGS & PT_F_STATS) && (threadIdx.x & 63) == 0) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
	uint64_t sc0 = 0, sc1 = 0, sc2 = 0, sc3 = 0;

	// single-leaf meshes: runs (rounded up to pairs: the slot behind an odd run is tri_geom's zero record) copied in object order
	__shared__ float4 sh_tri[COMPACT ? 1 : 3 * WSL_LDS_TRIS];
	bool tris_in_lds = WSL_LDS_TRIS > 0 && !COMPACT;
	if (!COMPACT) {
		int total = 0;
		for (int oi = 0; oi < sc.num_objects; ++oi) {
			const DevObject& ob = sc.objects[oi];
			if (ob.type != 0 && ob.root_node >= 0 && ob.big_slot < 0)
				total += (__builtin_amdgcn_readfirstlane(__float_as_int(sc.nodes[4 * ob.root_node + 3].w)) + 1) & ~1;
		}
		tris_in_lds = tris_in_lds && total <= WSL_LDS_TRIS;
		if (tris_in_lds) {
			int off = 0;
			for (int oi = 0; oi < sc.num_objects; ++oi) {
				const DevObject& ob = sc.objects[oi];
				if (ob.type != 0 && ob.root_node >= 0 && ob.big_slot < 0) {
					const float4 q3 = sc.nodes[4 * ob.root_node + 3];
					const int first = __builtin_amdgcn_readfirstlane(__float_as_int(q3.z));
					const int n2 = (__builtin_amdgcn_readfirstlane(__float_as_int(q3.w)) + 1) & ~1;
					for (int i = tid; i < 3 * n2; i += PT_BLOCK) sh_tri[3 * off + i] = sc.tri_geom[3 * first + i];
					off += n2;
				}
			}
		}
	}
#if WSL_CAM_LDS
	// What primary-ray generation needs of the camera, derived once per workgroup (the same operations in the same order as
	// pixel_midpoint / jittered_dir, pt_kernels.h) and read back from LDS where a ray is generated: as kernel arguments these two
	// dozen wave-uniform values sat in scalar registers across the whole loop, and this kernel has none to spare (79 spilled).
	__shared__ float sh_cam[24];
	if (tid == 0) {
		const f3 sdv = cam.screen_dist * normalize3(mk3(cam.dir[0], cam.dir[1], cam.dir[2]));
		const f3 rn = normalize3(mk3(cam.right[0], cam.right[1], cam.right[2])), un = normalize3(mk3(cam.up[0], cam.up[1], cam.up[2]));
		sh_cam[0] = sdv.x; sh_cam[1] = sdv.y; sh_cam[2] = sdv.z; sh_cam[3] = cam.pixel_size;
		sh_cam[4] = rn.x; sh_cam[5] = rn.y; sh_cam[6] = rn.z; sh_cam[7] = cam.pixel_size / 2.0f;
		sh_cam[8] = un.x; sh_cam[9] = un.y; sh_cam[10] = un.z; sh_cam[11] = cam.half_x;
		sh_cam[12] = cam.pos[0]; sh_cam[13] = cam.pos[1]; sh_cam[14] = cam.pos[2]; sh_cam[15] = cam.half_y;
		sh_cam[16] = cam.right[0]; sh_cam[17] = cam.right[1]; sh_cam[18] = cam.right[2]; sh_cam[19] = cam.pixel_size * cam.aa_jitter;
		sh_cam[20] = cam.up[0]; sh_cam[21] = cam.up[1]; sh_cam[22] = cam.up[2]; sh_cam[23] = 0.0f;
	}
#endif
	__shared__ int sh_pool_cur, sh_fresh_cap;
	if (tid == 0) {
		sh_pool = 0; sh_live = 0;
		// SHORT INPUTS (the reference's own loop renders one sample per call, runCUDA: a launch then holds fewer entries than the
		// grid has room for): a claim is no larger than the input's share of one workgroup, in whole wave-widths, so that the
		// input spreads over the chip instead of filling a third of the workgroups. Long inputs keep CAP. Schedule only.
		const int wgs = (int)gridDim.x;
		int cap = CAP;
		if (n_in < wgs * CAP) {
			cap = (((n_in + wgs - 1) / wgs) + 63) & ~63;
			cap = cap < 64 ? 64 : cap > CAP ? CAP : cap;
		}
		sh_fresh_cap = cap;
	}
	bool exhausted = n_in <= 0;   // block-uniform: the input has no entry left to claim
	for (;;) {
		const unsigned long long k0 = WSL_CLOCK();
		__syncthreads();   // pool entries and counters of the previous iteration are complete, its reads of sh_live done
		if (tid == 0) {
			const int np = sh_pool;
			int fb = 0, fn = 0, want = 0;
			if (!exhausted && np < CAP) {
				want = CAP - np;
				want = want < sh_fresh_cap ? want : sh_fresh_cap;
				fb = atomicAdd(&WF_SCUR(wp, launch_bounce), want);
				fn = n_in - fb;
				fn = fn < 0 ? 0 : (fn > want ? want : fn);
			}
			sh_pool_cur = np;
			sh_fresh_base = fb;
			sh_fresh_n = fn;
			sh_fresh_want = want;
			sh_pool = 0;
			sh_live = 0;
		}
		for (int i = tid; i < nl_total * WSL_LIST_INTS; i += PT_BLOCK) sh_lists[i] = 0;
		__syncthreads();
		const int n_pool = sh_pool_cur, fresh_base = sh_fresh_base, n_fresh = sh_fresh_n;
		if (n_fresh < sh_fresh_want) exhausted = true;   // the cursor ran past the end of the input
		if (n_pool + n_fresh == 0) break;

		// ---- phase 1: shade pooled hits and fresh input, generate primary rays; compact the new rays (stage A) ----
		for (int round = 0; round < WF_ROUNDS; ++round) {
			const int e = round * PT_BLOCK + tid;
			const bool pooled = e < n_pool;
			const bool fresh = !pooled && (e - n_pool) < n_fresh;
			bool alive = pooled || fresh;
			int path = 0, bs = launch_bounce - 1;   // bs: bounce of the segment whose hit is shaded (-1: none, generate)
			int best_obj = -1, best_tri = -1;
			float best_t = 0.0f, b_u = 0.0f, b_v = 0.0f;
			f3 ro = mk3(0.0f), rd = mk3(1.0f), mask = mk3(1.0f);
			f3 hp = mk3(0.0f);   // COMPACT: the hit point, formed where the hit was found (same expression, same bits)
			if (pooled) {
				ro = mk3(sh_w[0][e], sh_w[1][e], sh_w[2][e]);   // COMPACT: the hit point
				rd = mk3(sh_w[3][e], sh_w[4][e], sh_w[5][e]);
				mask = mk3(sh_w[6][e], sh_w[7][e], sh_w[8][e]);
				path = shi[9 * CAP + e];
				const int meta = shi[10 * CAP + e];
				if (COMPACT) {
					hp = ro;
					bs = meta & 7;
					best_obj = ((meta >> 3) & 0xff) - 1;
					best_tri = (int)((uint32_t)meta >> 11) - 1;
				} else {
					bs = meta & 0xff;
					best_obj = (meta >> 8) - 1;
					best_tri = shi[11 * CAP + e];
					best_t = sh_w[12][e];
				}
				b_u = sh_w[W_U][e]; b_v = sh_w[W_V][e];
			} else if (fresh) {
				const int i = fresh_base + (e - n_pool);
				if (launch_bounce == 0) path = i;
				else {
					const float4 a0 = i0[i], a1 = i1[i], a2 = i2[i], a3 = i3[i];
					ro = mk3(a0.x, a0.y, a0.z); rd = mk3(a1.x, a1.y, a1.z); mask = mk3(a2.x, a2.y, a2.z);
					best_t = a1.w;
					best_obj = __float_as_int(a2.w);
					best_tri = __float_as_int(a3.x); b_u = a3.y; b_v = a3.z;
					path = __float_as_int(a3.w);
					if (COMPACT) hp = ro + rd * best_t;
				}
			}
			__syncthreads();   // this round's pool entries are in registers: stage A may overwrite them
			// path id -> (sample, pixel): sample-major, then 8x8 tiles, lane = x + 8 y inside the tile
			const int sl = wp.npt == 1 ? path : (int)(__umulhi((uint32_t)path, wp.npt_m) >> wp.npt_s), r = path - sl * wp.npt;
			const int tile = r >> 6, l = r & 63;
			const int trow = wp.tiles_x == 1 ? tile : (int)(__umulhi((uint32_t)tile, wp.tx_m) >> wp.tx_s);
			const int x = (tile - trow * wp.tiles_x) * 8 + (l & 7);
			const int y = trow * 8 + (l >> 3);
			if (x >= cam.width || y >= cam.height) alive = false;
			const uint32_t pixel = (uint32_t)(x + y * cam.width);
			const uint32_t sample = (uint32_t)(wp.first_sample + sl);
			// (sample, pixel) and (bounce, slot) indices stay below 2^32 records (wf_ensure): 32-bit products, one 64-bit shift-add for
			// the address, instead of 64-bit multiply-adds per lane
			Rad3* rad_slot = wp.rad + ((uint32_t)sl * (uint32_t)wp.npix + pixel);

			Ray ray;
			set_ray(ray, ro, rd);   // shading reads origin and direction only; the inverse is formed once, for the rays that go on
			ray.inv = mk3(0.0f); ray.sx = ray.sy = ray.sz = false;
			int shade_type = -1;
			if (alive) {
				if (bs < 0) {
#if WSL_CAM_LDS
					// pixel_midpoint + jittered_dir (reference: src/pathtracing.cu:84-92, 121-126) on the values derived above
					const float4* cl = reinterpret_cast<const float4*>(sh_cam);
					const float4 c0 = cl[0], c1 = cl[1], c2 = cl[2], c3 = cl[3], c4 = cl[4], c5 = cl[5];
					const float ps = c0.w;
					const f3 sdv = mk3(c0.x, c0.y, c0.z), cpos = mk3(c3.x, c3.y, c3.z);
					const f3 pv = (c2.w - (float)x * ps - c1.w) * mk3(c1.x, c1.y, c1.z) + (c3.w - (float)y * ps - c1.w) * mk3(c2.x, c2.y, c2.z);
					const f3 mid = cpos + sdv + pv;
					const u4 jr = philox4x32_10(pixel, sample, PT_DIM_JITTER, 0u, wp.k0, wp.k1);
					const float jx = c4.w * (u01(jr.x) - 0.5f), jy = c4.w * (u01(jr.y) - 0.5f);
					init_ray(ray, cpos, normalize3(mid + jx * mk3(c4.x, c4.y, c4.z) + jy * mk3(c5.x, c5.y, c5.z) - cpos));
#else
					f3 mid, centre;
					pixel_midpoint(cam, x, y, mid, centre);
					init_ray(ray, cam_pos, jittered_dir(cam, mid, pixel, sample, wp.k0, wp.k1));
#endif
					if (maxb == 0) {   // max_ray_bounces == 0: the bounce loop never runs
						*rad_slot = Rad3{0.0f, 0.0f, 0.0f};
						n_paths++;
						alive = false;
					}
				} else {
					// ---- resolve the closest hit of segment bs and shade it (gatherRadiance body) ----
					f3 out = mk3(0.0f);
					bool done;
					if (best_obj < 0) {
						done = true;   // miss: reference src/radiance.cuh:30-34
					} else {
						const DevObject& ob = sc.objects[best_obj];
						HitInfo hi;
						TriShade held;
						held.s0 = held.s1 = held.s2 = held.s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
						hi.object = best_obj;
						hi.dist = 0.0f;   // not read by the shading
						hi.point = COMPACT ? hp : ray.o + ray.d * best_t;
						if (ob.type == 0) {
							hi.tri = -1;
							hi.bary = mk3(0.0f);
							hi.material = ob.mat_first;
							hi.normal = normalize3(hi.point - mk3(ob.px, ob.py, ob.pz));
							if (FLAGS & PT_F_STATS) c_hs++;
						} else {
							hi.tri = best_tri;
							hi.bary.y = b_u;
							hi.bary.z = b_v;
							hi.bary.x = 1.0f - hi.bary.y - hi.bary.z;
							const float4 t0 = sc.tri_shade[4 * hi.tri + 0], t1 = sc.tri_shade[4 * hi.tri + 1];
							const float4 t2 = sc.tri_shade[4 * hi.tri + 2], t3 = sc.tri_shade[4 * hi.tri + 3];
							held.s0 = t0; held.s1 = t1; held.s2 = t2; held.s3 = t3;
							hi.normal = normalize3(hi.bary.x * mk3(t0.x, t0.y, t0.z) + hi.bary.y * mk3(t1.x, t1.y, t1.z) +
							                       hi.bary.z * mk3(t2.x, t2.y, t2.z));
							hi.material = __float_as_int(t3.w);
							if (FLAGS & PT_F_STATS) c_hm++;
						}
						if (FLAGS & PT_F_STATS) {
							const MaterialRec mt = load_material(mats, hi.material);
							if (cam.texture_enabled && mt.texture > -1 && hi.tri >= 0) c_ht++;
							shade_type = mt.type;
						}
						const u4 rnd = philox4x32_10(pixel, sample, (uint32_t)bs, 0u, wp.k0, wp.k1);
						held.held = hi.tri >= 0;
						done = shade_segment(sc, cam, hi, ray, mask, rnd, out, mats, texs, held);
						if (!done && bs + 1 >= maxb) { done = true; out = mk3(0.0f); }   // bounce loop exhausted (:86)
					}
					if (done) {
						*rad_slot = Rad3{out.x, out.y, out.z};
						n_paths++;
						alive = false;
					} else {
						init_ray(ray, ray.o, ray.d);   // the BSDF set origin and direction (pt_shade.h): the inverse for the root-box tests below
					}
				}
			}
			if (FLAGS & PT_F_STATS) {
				const unsigned long long sm = __ballot(shade_type >= 0);
				if (sm != 0ull) {
					const int nb_ = (__ballot(shade_type == 1) != 0ull) + (__ballot(shade_type == 2) != 0ull) + (__ballot(shade_type == 3) != 0ull);
					if (lane == 0) { sh_waves++; sh_branches += (uint64_t)nb_; }
					if (shade_type >= 0) { sh_lanes++; }
					if (shade_type == 0) sh_t[0]++; else if (shade_type == 1) sh_t[1]++; else if (shade_type == 2) sh_t[2]++; else if (shade_type == 3) sh_t[3]++;
				}
			}
			// the new segment (bounce nb): root boxes of the multi-node meshes decide whether it leaves the workgroup
			const int nb = bs + 1;
			uint32_t flags = 0;
			if (alive) {
				const int nobj = sc.num_objects;
				for (int oi = 0; oi < nobj; ++oi) {
					const DevObject& ob = sc.objects[oi];
					if (ob.type != 0 && ob.root_node >= 0 && ob.big_slot >= 0) {
						float rdist;
						if (FLAGS & PT_F_STATS) cnt.n_box++;
						if (box_test(ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], ray, rdist)) flags |= 2u << ob.big_slot;
					}
				}
				if (flags != 0u) {
					int* li = sh_lists + (nb * nlists) * WSL_LIST_INTS;
					atomicAdd(&li[0], 1);
					for (int L = 1; L < nlists; ++L)
						if ((flags >> L) & 1u) atomicAdd(&li[L * WSL_LIST_INTS], 1);
				}
			}
			// compact: slot = (LDS counter, one add per wave) + rank inside the wave
			const unsigned long long am = __ballot(alive);
			if (am != 0ull) {
				int base = 0;
				if (lane == 0) base = atomicAdd(&sh_live, (int)__popcll(am));
				base = __builtin_amdgcn_readfirstlane(base);
				if (alive) {
					const int k = base + (int)__popcll(am & ((1ull << lane) - 1ull));
					sh_w[0][k] = ray.o.x; sh_w[1][k] = ray.o.y; sh_w[2][k] = ray.o.z;
					sh_w[3][k] = ray.d.x; sh_w[4][k] = ray.d.y; sh_w[5][k] = ray.d.z;
					sh_w[6][k] = mask.x; sh_w[7][k] = mask.y; sh_w[8][k] = mask.z;
					shi[9 * CAP + k] = path;
					shi[10 * CAP + k] = nb;
					shi[11 * CAP + k] = (int)flags;
				}
			}
		} // rounds
		const unsigned long long k1 = WSL_CLOCK();
		__syncthreads();
		const int n_live = sh_live;
		const int n_chunks = (n_live + PT_BLOCK - 1) / PT_BLOCK;
		// ONE global atomic per (bounce, list) with entries in this iteration
		for (int i = tid; i < nl_total; i += PT_BLOCK) {
			const int c = sh_lists[i * WSL_LIST_INTS];
			if (c > 0) sh_lists[i * WSL_LIST_INTS + 2] = atomicAdd(&WF_CNT(wp, i / nlists, i % nlists), c);
		}
		__syncthreads();
		const unsigned long long k2 = WSL_CLOCK();

		// ---- phase 2: the new segments against the cheap objects; leave (state set + queues) or stay (pool) ----
		for (int chunk = 0; chunk < n_chunks; ++chunk) {
			const int k = chunk * PT_BLOCK + tid;
			const bool alive = k < n_live;
			uint32_t flags = 0;
			Ray ray;
			f3 mask = mk3(1.0f);
			int path = 0, nb = 0;
			if (alive) {
				init_ray(ray, mk3(sh_w[0][k], sh_w[1][k], sh_w[2][k]), mk3(sh_w[3][k], sh_w[4][k], sh_w[5][k]));
				mask = mk3(sh_w[6][k], sh_w[7][k], sh_w[8][k]);
				path = shi[9 * CAP + k];
				nb = shi[10 * CAP + k];
				flags = (uint32_t)shi[11 * CAP + k];
			} else {
				init_ray(ray, mk3(0.0f), mk3(1.0f));
			}
			__syncthreads();   // this chunk's stage-A entries are in registers: the pool may overwrite them
			float best_dist = __builtin_huge_valf(), best_t = 0.0f, b_u = 0.0f, b_v = 0.0f;
			int best_obj = -1, best_tri = -1;
			if (alive) n_rays++;
			const int nobj = sc.num_objects;
			int lds_off = 0;   // where the current single-leaf mesh's run starts in sh_tri (wave-uniform)
			for (int oi = 0; oi < nobj; ++oi) {
				const DevObject& ob = sc.objects[oi];
				if (ob.type == 0) {
					if (alive) {
						if (FLAGS & PT_F_STATS) cnt.n_sphere++;
						f3 p, nrm;
						float t;
						if (sphere_test(ray, ob.radius, mk3(ob.px, ob.py, ob.pz), p, nrm, t)) {
							const float d = length3(p - ray.o);
							if (d < best_dist) { best_dist = d; best_t = t; best_obj = oi; best_tri = -1; }
						}
					}
				} else if (ob.root_node >= 0 && ob.big_slot < 0) {
					bool enter = false;
					if (alive) {
						float rdist;
						if (FLAGS & PT_F_STATS) cnt.n_box++;
						enter = box_test(ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], ray, rdist);
					}
					if (enter) {
						float tmin = __builtin_huge_valf(), bu = 0.0f, bv = 0.0f;
						int tri = -1;
						const int run_off = lds_off;
						auto fetch = [&](int first, int rel, float4& g0, float4& g1, float4& g2) {
							if (tris_in_lds) { const int q = 3 * (run_off + rel); g0 = sh_tri[q]; g1 = sh_tri[q + 1]; g2 = sh_tri[q + 2]; }
							else { const int q = 3 * (first + rel); g0 = sc.tri_geom[q]; g1 = sc.tri_geom[q + 1]; g2 = sc.tri_geom[q + 2]; }
						};
						bool hit = (WF_CHEAP_PAIRS && !(FLAGS & PT_F_ALLTRIS)) ? leaf_only_mesh_pairs<FLAGS>(sc, ob, ray, tmin, tri, bu, bv, cnt, fetch) : leaf_only_mesh<FLAGS>(sc, ob, ray, tmin, tri, bu, bv, cnt);
						if (hit && tri >= 0) {
							const f3 p = ray.o + tmin * ray.d;
							const float d = length3(p - ray.o);
							if (d < best_dist) { best_dist = d; best_t = tmin; best_obj = oi; best_tri = tri; b_u = bu; b_v = bv; }
						}
					}
					if (tris_in_lds) lds_off += (__builtin_amdgcn_readfirstlane(__float_as_int(sc.nodes[4 * ob.root_node + 3].w)) + 1) & ~1;
				}
			}
			const bool leaves = alive && flags != 0u;
			const bool stays = alive && flags == 0u;
			if (leaves) {
				int* li = sh_lists + (nb * nlists) * WSL_LIST_INTS;
				const int slot = li[2] + atomicAdd(&li[1], 1);
				const uint32_t so = (uint32_t)nb * (uint32_t)wp.set_stride + (uint32_t)slot;
				wp.s0[so] = make_float4(ray.o.x, ray.o.y, ray.o.z, best_dist);
				wp.s1[so] = make_float4(ray.d.x, ray.d.y, ray.d.z, best_t);
				wp.s2[so] = make_float4(mask.x, mask.y, mask.z, __int_as_float(best_obj));
				wp.s3[so] = make_float4(__int_as_float(best_tri), b_u, b_v, __int_as_float(path));
				for (int L = 1; L < nlists; ++L)
					if ((flags >> L) & 1u) {
						const int q = li[L * WSL_LIST_INTS + 2] + atomicAdd(&li[L * WSL_LIST_INTS + 1], 1);
						wp.queue[L - 1][(uint32_t)nb * (uint32_t)wp.queue_stride + (uint32_t)q] = slot;
					}
			}
			const unsigned long long pm = __ballot(stays);
			if (pm != 0ull) {
				int base = 0;
				if (lane == 0) base = atomicAdd(&sh_pool, (int)__popcll(pm));
				base = __builtin_amdgcn_readfirstlane(base);
				if (stays) {
					const int p = base + (int)__popcll(pm & ((1ull << lane) - 1ull));
					// COMPACT: the hit point in place of the origin (hi.point of the shading: ray.o + ray.d * best_t, formed here)
					const f3 w0 = COMPACT ? ray.o + ray.d * best_t : ray.o;
					sh_w[0][p] = w0.x; sh_w[1][p] = w0.y; sh_w[2][p] = w0.z;
					sh_w[3][p] = ray.d.x; sh_w[4][p] = ray.d.y; sh_w[5][p] = ray.d.z;
					sh_w[6][p] = mask.x; sh_w[7][p] = mask.y; sh_w[8][p] = mask.z;
					shi[9 * CAP + p] = path;
					if (COMPACT) {
						shi[10 * CAP + p] = (int)((uint32_t)nb | ((uint32_t)(best_obj + 1) << 3) | ((uint32_t)(best_tri + 1) << 11));
					} else {
						shi[10 * CAP + p] = nb | ((best_obj + 1) << 8);
						shi[11 * CAP + p] = best_tri;
						sh_w[12][p] = best_t;
					}
					sh_w[W_U][p] = b_u; sh_w[W_V][p] = b_v;
				}
			}
		} // chunks
		const unsigned long long k3 = WSL_CLOCK();
		sc0 += k1 - k0; sc1 += k2 - k1; sc2 += k3 - k2;
	}
	(void)sc3;
	stats_flush(wp.stats, cnt, n_rays, n_paths, c_hm, c_hs, c_ht, 0, 0, (FLAGS & PT_F_STATS) != 0);
	if (FLAGS & PT_F_STATS) {
		uint64_t x[11] = {sh_waves, sh_lanes, sh_branches, sh_t[0], sh_t[1], sh_t[2], sh_t[3], sc0, sc1, sc2, sc3};
		unsigned long long* dst = (unsigned long long*)&wp.stats->sh_waves;
#pragma unroll
		for (int k = 0; k < 11; ++k) {
			const uint64_t v = wave_sum_u64(x[k]);
			if (lane == 0 && v) atomicAdd(dst + k, (unsigned long long)v);
		}
	}
#undef WSL_CLOCK
}
