// Wavefront path tracer: the hot path as a per-bounce pipeline over ALL paths of a sample batch.
//
// Why: in a one-lane-per-pixel megakernel the wave waits for its slowest lane. In the headline
// scene only ~1 ray in 3 enters the car's KD-tree, and those rays do hundreds of triangle tests
// while their 40-odd neighbours idle (measured: 8 % useful lanes, 68 % of wave time waiting,
// profiles/r01_baseline_*). Here rays that need the expensive traversal are compacted into a
// queue (ballot + popcount prefix inside the wave, ONE atomic per wave) and a dedicated kernel
// traverses them with full waves and few registers (high occupancy to hide L2/MALL latency).
//
// Pipeline for a batch of B samples (paths = B x padded pixel count, path id = sample-major,
// then 8x8 pixel tiles, so a wave starts on one tile of one sample):
//   for bounce = 0 .. max_bounces:
//     wf_segment_kernel   bounce 0: jittered primary ray (generatePrimaryRays + jitterPrimaryRays)
//                         bounce>0: resolve the previous segment's closest hit, shade / sample the
//                                   BSDF (gatherRadiance body), terminate on miss / light / last bounce
//                         then intersect the new ray with all CHEAP objects inline (spheres, single-leaf
//                         meshes), root-box-test the multi-node meshes and enqueue the ray for those it
//                         enters; survivors are appended to the next live list (stream compaction)
//     wf_traverse_kernel  one launch per multi-node mesh: short-stack KD traversal (stack in LDS) of the
//                         queued rays, merged into the path's closest-hit record
//   wf_accumulate_kernel  acc[pixel] += radiance[sample][pixel] in SAMPLE ORDER
//
// Determinism: a path's arithmetic does not depend on which lane runs it or in what order queues are
// filled; terminated paths park their radiance in rad[sample][pixel] and the accumulation adds them
// in sample order, so the result is bit-identical to one-sample-per-launch (the reference schedule).
// Closest hit across objects = smallest length(ipoint - origin), ties to the lower object index —
// exactly what the reference's in-order loop with strict '<' yields (src/intersect.cuh:328-346).
//
// Path state in HBM, 4 x float4 per path (64 B), SoA so lanes of a wave read consecutive 16-B words:
//   s0 = (origin.xyz, best_dist)   s1 = (dir.xyz, best_t)
//   s2 = (mask.rgb, int best_obj)  s3 = (int best_tri, bary_u, bary_v, int path id)
// The state is COMPACTED at every bounce: there are two sets of these arrays, and the paths that survive
// bounce b are written to set b & 1 at consecutive SLOTS (slot = position in that bounce's list of
// survivors; the path id - sample-major, then 8x8 pixel tiles - travels in s3.w). Bounce b + 1 reads slots
// 0 .. n-1 of that set densely, and the traversal queues hold slots. (Round 1 kept one set indexed by path id
// plus live lists of ids: once two thirds of the paths had ended, every 16-B record read or written dragged a
// 64-B line of dead neighbours along: 22 GB of HBM-side traffic per launch, profiles/r01_*, r02_pmc.json.)
#pragma once
#include "pt_kernels.h"

#define PT_MAX_BIG 31   // multi-node meshes with a traversal queue (flag bits 1..31 of a 32-bit word)
#ifndef WF_ROUNDS
#define WF_ROUNDS 2
#endif
#define PT_WF_MAX_BOUNCES 64

// parked radiance of a finished path (12 B: the fourth word of a float4 would be a quarter of this buffer's traffic)
struct Rad3 { float x, y, z; };

struct WfParams {
	float4* s0; float4* s1; float4* s2; float4* s3;         // state set this bounce WRITES (and its traversals update)
	const float4* i0; const float4* i1; const float4* i2; const float4* i3;   // state set of the previous bounce (read)
	Rad3* rad;                // [batch][npix], 12 B each: a finished path parks its radiance here
	int* queue[PT_MAX_BIG];   // per multi-node mesh: slots (in s0..s3) of the paths whose ray enters its root box
	int* retry;               // slots of the rays the wide traversal (pt_wide.h) hands to the strict kernel; shared by the queues in turn
	int use_retry;            // wf_traverse_coop_kernel: take the rays from `retry` instead of queue[slot]
	int* counters;            // [bounce][1 + PT_MAX_BIG]: live count, queue counts (zeroed per batch)
	PtxStats* stats;
	int first_sample, batch;
	int npt;                  // padded pixels per sample = tiles * 64
	int tiles_x, npix, n_total;
	// n / npt and n / tiles_x for n < 2^28 as mulhi(n, m) >> s (wf_magic_div on the host): a vector integer division is some
	// twenty instructions, and the segment kernel did two per path
	uint32_t npt_m, npt_s, tx_m, tx_s;
	uint32_t k0, k1;
	int bounce, is_last;
	int num_big;
	// one state set and one queue segment per bounce (pt_segloop.h): set b = s0..s3 + b * set_stride, queue (b, k) =
	// queue[k] + b * queue_stride; the traversal kernels and the per-bounce segment kernel get pointers the host has offset already
	size_t set_stride, queue_stride;
	int max_bounces;
};

#define WF_CNT(wp, b, k) ((wp).counters[(b) * (1 + PT_MAX_BIG) + (k)])
// dynamic work distribution of the traversal kernels: next unclaimed queue entry per (bounce, queue)
#define WF_CUR(wp, b, k) ((wp).counters[(PT_WF_MAX_BOUNCES + 1) * (1 + PT_MAX_BIG) + (b) * (1 + PT_MAX_BIG) + (k)])
// the same pair for the retry queue of (bounce, queue): its length and the next unclaimed entry
#define WF_RCNT(wp, b, k) ((wp).counters[2 * (PT_WF_MAX_BOUNCES + 1) * (1 + PT_MAX_BIG) + (b) * (1 + PT_MAX_BIG) + (k)])
#define WF_RCUR(wp, b, k) ((wp).counters[3 * (PT_WF_MAX_BOUNCES + 1) * (1 + PT_MAX_BIG) + (b) * (1 + PT_MAX_BIG) + (k)])
#define WF_COUNTER_INTS (4 * (size_t)(PT_WF_MAX_BOUNCES + 1) * (1 + PT_MAX_BIG) + (PT_WF_MAX_BOUNCES + 1))   /* + the segment launches' input cursors (pt_segloop.h) */

// Queue feed of a persistent traversal wave. A wave claims CHUNK consecutive queue entries at a time
// with one atomic (wave-uniform state: next/hi/exhausted), so every wave keeps all its lanes fed until
// the whole queue is drained — with a static split a wave owns only ~2 rays per lane and spends most
// of its life draining a tail of long rays with few lanes. CHUNK is large enough that the atomics
// (queue / CHUNK per launch, ~10^4) stay far below the rate at which one address serialises.
struct WaveFeed {
	int next, hi;
	bool exhausted;
};

PT_DEV void feed_claim(WaveFeed& f, int* cursor, int n, int chunk) {
	int base = 0;
	if ((threadIdx.x & 63) == 0) base = atomicAdd(cursor, chunk);
	base = __builtin_amdgcn_readfirstlane(base);
	f.next = base;
	f.hi = base + chunk < n ? base + chunk : n;
	if (base >= n) { f.exhausted = true; f.next = f.hi = 0; }
}

// Stream compaction inside a wave: returns the slot of this lane in the list `counter` counts, or -1.
// Must be reached by whole waves (callers keep their loops wave-uniform).
PT_DEV int wave_append(int* counter, bool pred) {
	const unsigned long long m = __ballot(pred);
	if (m == 0ull) return -1;
	const int lane = threadIdx.x & 63;
	const int leader = (int)__ffsll((long long)m) - 1;
	int base = 0;
	if (lane == leader) base = atomicAdd(counter, (int)__popcll(m));
	base = __shfl(base, leader, 64);
	const int rank = (int)__popcll(m & ((1ull << lane) - 1ull));
	return pred ? base + rank : -1;
}

PT_DEV void stats_flush(PtxStats* stats, const TraceCounters& cnt, uint64_t rays, uint64_t paths, uint64_t hm, uint64_t hs,
                        uint64_t ht, uint64_t act, uint64_t tot, bool detailed, uint64_t trav_rays = 0, bool is_traverse = false) {
	const int lane = threadIdx.x & 63;
	rays = wave_sum_u64(rays);
	paths = wave_sum_u64(paths);
	uint64_t c_inner = 0, c_box = 0, c_tri = 0, c_sphere = 0, c_leaf = 0;
	if (detailed) {
		c_inner = wave_sum_u64(cnt.n_inner); c_box = wave_sum_u64(cnt.n_box); c_tri = wave_sum_u64(cnt.n_tri);
		c_sphere = wave_sum_u64(cnt.n_sphere); c_leaf = wave_sum_u64(cnt.n_leaf);
		hm = wave_sum_u64(hm); hs = wave_sum_u64(hs); ht = wave_sum_u64(ht);
		act = wave_sum_u64(act); tot = wave_sum_u64(tot);
		trav_rays = wave_sum_u64(trav_rays);
	}
	if (lane == 0) {
		if (rays) atomicAdd((unsigned long long*)&stats->rays, (unsigned long long)rays);
		if (paths) atomicAdd((unsigned long long*)&stats->paths, (unsigned long long)paths);
		if (detailed) {
			if (trav_rays) atomicAdd((unsigned long long*)&stats->trav_rays, (unsigned long long)trav_rays);
			atomicAdd((unsigned long long*)&stats->n_inner, (unsigned long long)c_inner);
			atomicAdd((unsigned long long*)&stats->n_box, (unsigned long long)c_box);
			atomicAdd((unsigned long long*)&stats->n_tri, (unsigned long long)c_tri);
			atomicAdd((unsigned long long*)&stats->n_sphere, (unsigned long long)c_sphere);
			atomicAdd((unsigned long long*)&stats->n_leaf, (unsigned long long)c_leaf);
			atomicAdd((unsigned long long*)&stats->hit_mesh, (unsigned long long)hm);
			atomicAdd((unsigned long long*)&stats->hit_sphere, (unsigned long long)hs);
			atomicAdd((unsigned long long*)&stats->hit_textured, (unsigned long long)ht);
			atomicAdd((unsigned long long*)&stats->lane_steps_active, (unsigned long long)act);
			atomicAdd((unsigned long long*)&stats->lane_steps_total, (unsigned long long)tot);
			if (is_traverse) {
				atomicAdd((unsigned long long*)&stats->trav_inner, (unsigned long long)c_inner);
				atomicAdd((unsigned long long*)&stats->trav_box, (unsigned long long)c_box);
				atomicAdd((unsigned long long*)&stats->trav_tri, (unsigned long long)c_tri);
				atomicAdd((unsigned long long*)&stats->trav_leaf, (unsigned long long)c_leaf);
			}
		}
	}
}

PT_DEV void diag_flush(PtxStats* stats, uint64_t d0, uint64_t d1, uint64_t d2, uint64_t d3) {
	d0 = wave_sum_u64(d0); d1 = wave_sum_u64(d1); d2 = wave_sum_u64(d2); d3 = wave_sum_u64(d3);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd((unsigned long long*)&stats->diag0, (unsigned long long)d0);
		atomicAdd((unsigned long long*)&stats->diag1, (unsigned long long)d1);
		atomicAdd((unsigned long long*)&stats->diag2, (unsigned long long)d2);
		atomicAdd((unsigned long long*)&stats->diag3, (unsigned long long)d3);
	}
}

// A traversal's result goes into the first three words of the path's s3 record; the fourth carries the path id
// (written by the segment kernel) and must survive: one 12-byte store.
struct HitWords { int tri; float u, v; };
PT_DEV void store_hit(float4* rec, int tri, float u, float v) {
	HitWords h; h.tri = tri; h.u = u; h.v = v;
	*reinterpret_cast<HitWords*>(rec) = h;
}

// PtxStats.x_* (include/ptx_hip.h): what a traversal kernel executed
PT_DEV void exec_flush(PtxStats* stats, uint64_t iters, uint64_t item, uint64_t half, uint64_t wait, uint64_t usteps, uint64_t units,
                       uint64_t uphases, uint64_t turns) {
	uint64_t x[8] = {iters, item, half, wait, usteps, units, uphases, turns};
	unsigned long long* dst = (unsigned long long*)&stats->x_iters;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint64_t v = wave_sum_u64(x[k]);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst + k, (unsigned long long)v);
	}
}

// A single-leaf mesh (root is a leaf): all lanes walk the same triangle run, whose addresses are
// wave-uniform, so the compiler can fetch the triangles with scalar loads.
template <int FLAGS>
PT_DEV bool leaf_only_mesh(const DevScene& sc, const DevObject& ob, const Ray& ray, float& tmin, int& best_tri,
                           float& bu, float& bv, TraceCounters& cnt) {
	const float4 q3 = sc.nodes[4 * ob.root_node + 3];
	const int first = __builtin_amdgcn_readfirstlane(__float_as_int(q3.z));
	const int count = __builtin_amdgcn_readfirstlane(__float_as_int(q3.w));
	if (FLAGS & PT_F_STATS) { cnt.n_leaf++; cnt.n_tri += (uint32_t)count; }
	bool intersects = false;
	for (int i = 0; i < count; ++i) {
		const float4 g0 = sc.tri_geom[3 * (first + i) + 0];
		const float4 g1 = sc.tri_geom[3 * (first + i) + 1];
		const float4 g2 = sc.tri_geom[3 * (first + i) + 2];
		float t, u, v;
		// every lane tests the SAME triangle with its own ray: early exits would only fire when all 64 agree
		if (tri_test_flat(ray, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), mk3(g1.z, g1.w, g2.x), t, u, v)) {
			intersects = true;
			if (t < tmin) { tmin = t; best_tri = first + i; bu = u; bv = v; }
		}
	}
	return intersects;
}

// The same mesh through its PAIR boxes (pt_scene.h, cheap_pairs): every lane first tests the boxes of all pairs of
// consecutive triangles (wave-uniform data: scalar loads, one slab test each) and then only the pairs its ray touches
// (its own loads from tri_geom), in run order with strict '<' - what the serial loop above keeps, because a pair
// whose padded box the ray misses holds no hit - up to the SAME exception as the group boxes: at
// grazing incidence the reference's test can accept a triangle the ray passes beside, box and all, out of rounding noise.
// Never observed on a pair box (every comparison with the oracle, 105 M segments of the full frame among them, is
// identical), but the strict schedule (PT_F_ALLTRIS) takes the serial loop above to be strict by construction. A ray inside
// the sky cube tests one face instead of twelve triangles; the loop below runs once for most waves.
// `fetch(first, rel, g0, g1, g2)`: the three words of triangle first + rel of tri_geom - from global memory, or from a
// copy of the single-leaf meshes' runs in LDS (pt_segloop.h): the loop below is a chain of dependent loads.
struct GlobalTriFetch {
	const float4* tri_geom;
	PT_DEV void operator()(int first, int rel, float4& g0, float4& g1, float4& g2) const {
		g0 = tri_geom[3 * (first + rel) + 0]; g1 = tri_geom[3 * (first + rel) + 1]; g2 = tri_geom[3 * (first + rel) + 2];
	}
};
template <int FLAGS, class Fetch>
PT_DEV bool leaf_only_mesh_pairs(const DevScene& sc, const DevObject& ob, const Ray& ray, float& tmin, int& best_tri,
                                 float& bu, float& bv, TraceCounters& cnt, const Fetch& fetch) {
	const float4 q3 = sc.nodes[4 * ob.root_node + 3];
	const float4 q0 = sc.nodes[4 * ob.root_node + 0];
	const int first = __builtin_amdgcn_readfirstlane(__float_as_int(q3.z));
	const int count = __builtin_amdgcn_readfirstlane(__float_as_int(q3.w));
	const int pair_base = __builtin_amdgcn_readfirstlane(__float_as_int(q0.y));
	if (FLAGS & PT_F_STATS) { cnt.n_leaf++; cnt.n_tri += (uint32_t)count; }   // the reference's count: every triangle of the leaf
	const int npairs = (count + 1) >> 1;
	// slab test as (plane - o) * inv, the IEEE form: with a direction component of exactly zero (inv = inf; a camera ray
	// of the centre column, say) the two planes give -inf and +inf when the origin lies between them, and a NaN (0 * inf:
	// the origin ON a padded plane) drops out of min / max. Until late round 3 this was fma(plane, inv, -o * inv), one
	// instruction less per plane - and wrong for such rays: plane * inf - o * inf is inf - inf = NaN for the plane on the
	// origin's side of zero, the NaN dropped out, both bounds of the axis became the other plane's -inf and the box was
	// "missed": one black pixel in every ten samples of the headline frame (tools/strict_diff.py found it).
	// The difference form also keeps the rounding relative to the distance from the box instead of from the scene's origin.
	bool intersects = false;
	for (int base = 0; base < npairs; base += 32) {
		const int nb = npairs - base < 32 ? npairs - base : 32;
		uint32_t touched = 0;
		for (int j = 0; j < nb; ++j) {
			const float4* pr = sc.cheap_pairs + 2 * (size_t)(pair_base + base + j);
			const float4 lo = pr[0], hi = pr[1];
			const float ax = (lo.x - ray.o.x) * ray.inv.x, bx = (hi.x - ray.o.x) * ray.inv.x;
			const float ay = (lo.y - ray.o.y) * ray.inv.y, by = (hi.y - ray.o.y) * ray.inv.y;
			const float az = (lo.z - ray.o.z) * ray.inv.z, bz = (hi.z - ray.o.z) * ray.inv.z;
			const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
			const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
			if (!(tn > tf) && !(tf < 0.0f)) touched |= 1u << j;
		}
		while (__ballot(touched != 0u) != 0ull) {
			if (touched != 0u) {
				const int j = (int)__ffs((int)touched) - 1;
				touched &= touched - 1u;
				const int t0 = first + 2 * (base + j);
#pragma unroll
				for (int h = 0; h < 2; ++h) {   // the slot behind an odd run's last triangle is a zero record: never hit
					float4 g0, g1, g2;
					fetch(first, 2 * (base + j) + h, g0, g1, g2);
					float t, u, v;
					if (tri_test_flat(ray, mk3(g0.x, g0.y, g0.z), mk3(g0.w, g1.x, g1.y), mk3(g1.z, g1.w, g2.x), t, u, v)) {
						intersects = true;
						if (t < tmin) { tmin = t; best_tri = t0 + h; bu = u; bv = v; }
					}
				}
			}
		}
	}
	return intersects;
}
template <int FLAGS>
PT_DEV bool leaf_only_mesh_pairs(const DevScene& sc, const DevObject& ob, const Ray& ray, float& tmin, int& best_tri,
                                 float& bu, float& bv, TraceCounters& cnt) {
	return leaf_only_mesh_pairs<FLAGS>(sc, ob, ray, tmin, best_tri, bu, bv, cnt, GlobalTriFetch{sc.tri_geom});
}

#ifndef WF_CHEAP_PAIRS
#define WF_CHEAP_PAIRS 1   // single-leaf meshes through their pair records (0: every triangle, as round 1)
#endif
#ifndef WF_SEG_OCC
#define WF_SEG_OCC 4   // r2, with the pair filter of the single-leaf meshes: 128 registers, no spills: 6.61 ms per launch; at 5 waves/SIMD
                       // (96 registers, 33 spilled values) 7.29 ms. Without the filter: 7.13 ms at 5 (11 spilled), 7.25 at 4.
#endif
// LDSTAB: the material table and the texture descriptors are copied into LDS first (the host picks this variant when
// they fit): both are small, and both sit in the middle of the shading's chain of dependent loads
#define WF_LDS_MATS 160
#define WF_LDS_TEXS 64
template <int FLAGS, int LDSTAB>
__global__ __launch_bounds__(PT_BLOCK, WF_SEG_OCC) void wf_segment_kernel(DevScene sc, DevCamera cam, WfParams wp) {
	const int tid = threadIdx.x, lane = tid & 63;
	__shared__ float4 sh_mats[LDSTAB ? 3 * WF_LDS_MATS : 1];
	__shared__ DevTexture sh_texs[LDSTAB ? WF_LDS_TEXS : 1];
	if (LDSTAB) {
		for (int i = tid; i < 3 * sc.num_materials; i += PT_BLOCK) sh_mats[i] = sc.materials[i];
		for (int i = tid; i < sc.num_textures; i += PT_BLOCK) sh_texs[i] = sc.textures[i];
	}
	const float4* mats = LDSTAB ? sh_mats : sc.materials;
	const DevTexture* texs = LDSTAB ? sh_texs : sc.textures;
	const int bounce = wp.bounce;
	const int n_in = bounce == 0 ? wp.n_total : WF_CNT(wp, bounce - 1, 0);
	const f3 cam_pos = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);

	TraceCounters cnt = {0, 0, 0, 0, 0, 0};
	uint64_t n_rays = 0, n_paths = 0, c_hm = 0, c_hs = 0, c_ht = 0;
	uint64_t sh_waves = 0, sh_lanes = 0, sh_branches = 0, sh_t[4] = {0, 0, 0, 0};
	// counting variant only: shader clock per phase (PtxStats.seg_clk)
#define WFS_CLOCK() (((FLAGS & PT_F_STATS) && (threadIdx.x & 63) == 0) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
	uint64_t sc0 = 0, sc1 = 0, sc2 = 0, sc3 = 0;

	// A workgroup takes WF_ROUNDS x 256 paths per iteration and works in two phases:
	//   1. generate / shade: one path per lane; the rays that go on are COMPACTED into LDS
	//      (a third to two thirds of the paths end at each bounce, so without this most lanes of
	//      the intersection code below would idle next to a few live ones);
	//   2. intersect: one surviving ray per lane, full waves; writes the path state and the list flags.
	// List appends (next live list, traversal queues) are aggregated over the whole iteration: ONE atomic
	// per list per iteration. (One atomic per wave serialised the whole chip on a single counter
	// address: 91 % of this kernel's wave time was spent waiting, profiles/r01_baseline_*.)
	constexpr int CAP = WF_ROUNDS * PT_BLOCK;
	__shared__ float sh_ray[9][CAP];          // origin, direction, mask of the compacted rays (SoA: conflict-free)
	__shared__ int sh_path[CAP];
	__shared__ uint32_t sh_flags[CAP];
	__shared__ int sh_cnt[1 + PT_MAX_BIG][WF_ROUNDS * (PT_BLOCK / 64)];
	__shared__ int sh_base[1 + PT_MAX_BIG];
	__shared__ int sh_n;
	const int wave = tid >> 6;
	const int nlists = 1 + wp.num_big;
	if (tid == 0) sh_n = 0;
	__syncthreads();

	for (int blk0 = (int)blockIdx.x * CAP; blk0 < n_in; blk0 += (int)gridDim.x * CAP) {
	const unsigned long long k0 = WFS_CLOCK();
	// ---- phase 1: generate / shade, compact the survivors ---------------------------------------------
	for (int round = 0; round < WF_ROUNDS; ++round) {
		const int i = blk0 + round * PT_BLOCK + tid;
		bool alive = i < n_in;
		int path = 0;
		float4 a0, a1, a2, a3;
		if (alive) {
			if (bounce == 0) path = i;
			else {
				// slot i of the previous bounce's compacted state: dense, coalesced reads (issuing the loads of all rounds
				// up front, with or without their tri_shade records, costs registers: 28 ... 90 spilled values, 6.95 vs 6.48 ms)
				a0 = wp.i0[i]; a1 = wp.i1[i]; a2 = wp.i2[i]; a3 = wp.i3[i];
				path = __float_as_int(a3.w);
			}
		}
		// path id -> (sample, pixel): sample-major, then 8x8 tiles, lane = x + 8 y inside the tile
		const int sl = wp.npt == 1 ? path : (int)(__umulhi((uint32_t)path, wp.npt_m) >> wp.npt_s), r = path - sl * wp.npt;
		const int tile = r >> 6, l = r & 63;
		const int trow = wp.tiles_x == 1 ? tile : (int)(__umulhi((uint32_t)tile, wp.tx_m) >> wp.tx_s);
		const int x = (tile - trow * wp.tiles_x) * 8 + (l & 7);
		const int y = trow * 8 + (l >> 3);
		if (x >= cam.width || y >= cam.height) alive = false;
		const uint32_t pixel = (uint32_t)(x + y * cam.width);
		const uint32_t sample = (uint32_t)(wp.first_sample + sl);
		Rad3* rad_slot = wp.rad + (size_t)sl * (size_t)wp.npix + pixel;

		Ray ray;
		f3 mask = mk3(1.0f);
		int shade_type = -1;   // counting variant: material type of the hit this lane shades
		if (alive) {
			if (bounce == 0) {
				f3 mid, centre;
				pixel_midpoint(cam, x, y, mid, centre);
				set_ray(ray, cam_pos, jittered_dir(cam, mid, pixel, sample, wp.k0, wp.k1));   // phase 2 forms the inverse (from LDS)
				if (wp.is_last) { // max_ray_bounces == 0: the bounce loop never runs
					*rad_slot = Rad3{0.0f, 0.0f, 0.0f};
					n_paths++;
					alive = false;
				}
			} else {
				// ---- resolve the closest hit of the previous segment and shade it ------------------
				set_ray(ray, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z));
				mask = mk3(a2.x, a2.y, a2.z);
				const int best_obj = __float_as_int(a2.w);
				f3 out = mk3(0.0f);
				bool done;
				if (best_obj < 0) {
					done = true; // miss: reference src/radiance.cuh:30-34
				} else {
					const DevObject& ob = sc.objects[best_obj];
					HitInfo hi;
					TriShade held;   // the hit triangle's tri_shade record, handed on to the texture lookup
					held.s0 = held.s1 = held.s2 = held.s3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					hi.object = best_obj;
					hi.dist = a0.w;
					hi.point = ray.o + ray.d * a1.w;
					if (ob.type == 0) {
						hi.tri = -1;
						hi.bary = mk3(0.0f);
						hi.material = ob.mat_first;
						hi.normal = normalize3(hi.point - mk3(ob.px, ob.py, ob.pz));
						if (FLAGS & PT_F_STATS) c_hs++;
					} else {
						hi.tri = __float_as_int(a3.x);
						hi.bary.y = a3.y;
						hi.bary.z = a3.z;
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
					const u4 rnd = philox4x32_10(pixel, sample, (uint32_t)(bounce - 1), 0u, wp.k0, wp.k1);
					held.held = hi.tri >= 0;
					done = shade_segment(sc, cam, hi, ray, mask, rnd, out, mats, texs, held);
					if (!done && wp.is_last) { done = true; out = mk3(0.0f); } // bounce loop exhausted (:86)
				}
				if (done) {
					*rad_slot = Rad3{out.x, out.y, out.z};
					n_paths++;
					alive = false;
				}
			}
		}
		if (FLAGS & PT_F_STATS) {
			// shading divergence of this wave: which BSDF branches it had to run, for how many lanes
			const unsigned long long sm = __ballot(shade_type >= 0);
			if (sm != 0ull) {
				const int nb = (__ballot(shade_type == 1) != 0ull) + (__ballot(shade_type == 2) != 0ull) + (__ballot(shade_type == 3) != 0ull);
				if (lane == 0) { sh_waves++; sh_branches += (uint64_t)nb; }
				if (shade_type >= 0) { sh_lanes++; }
				if (shade_type == 0) sh_t[0]++; else if (shade_type == 1) sh_t[1]++; else if (shade_type == 2) sh_t[2]++; else if (shade_type == 3) sh_t[3]++;
			}
		}
		// compact: slot = (LDS counter, one add per wave) + rank inside the wave
		const unsigned long long am = __ballot(alive);
		if (am != 0ull) {
			int base = 0;
			if (lane == 0) base = atomicAdd(&sh_n, (int)__popcll(am));
			base = __builtin_amdgcn_readfirstlane(base);
			if (alive) {
				const int k = base + (int)__popcll(am & ((1ull << lane) - 1ull));
				sh_ray[0][k] = ray.o.x; sh_ray[1][k] = ray.o.y; sh_ray[2][k] = ray.o.z;
				sh_ray[3][k] = ray.d.x; sh_ray[4][k] = ray.d.y; sh_ray[5][k] = ray.d.z;
				sh_ray[6][k] = mask.x; sh_ray[7][k] = mask.y; sh_ray[8][k] = mask.z;
				sh_path[k] = path;
			}
		}
	} // rounds
		const unsigned long long k1 = WFS_CLOCK();
		__syncthreads();
		const int n_live = sh_n;
		const int n_chunks = (n_live + PT_BLOCK - 1) / PT_BLOCK;
		// every compacted ray survives phase 2, so the block's slot range in this bounce's state set (= its part of
		// the list of survivors) can be reserved now: ONE atomic per block and iteration
		if (tid == 0) sh_base[0] = n_live ? atomicAdd(&WF_CNT(wp, bounce, 0), n_live) : 0;
		__syncthreads();
		const int base0 = sh_base[0];
		const unsigned long long k2 = WFS_CLOCK();

	// ---- phase 2: intersect the new segments with the cheap objects, enqueue them for the expensive ones ----
	for (int chunk = 0; chunk < n_chunks; ++chunk) {
		const int k = chunk * PT_BLOCK + tid;
		const bool alive = k < n_live;
		uint32_t flags = 0;
		Ray ray;
		f3 mask = mk3(1.0f);
		int path = 0;
		if (alive) {
			init_ray(ray, mk3(sh_ray[0][k], sh_ray[1][k], sh_ray[2][k]), mk3(sh_ray[3][k], sh_ray[4][k], sh_ray[5][k]));
			mask = mk3(sh_ray[6][k], sh_ray[7][k], sh_ray[8][k]);
			path = sh_path[k];
		}
		float best_dist = __builtin_huge_valf(), best_t = 0.0f, b_u = 0.0f, b_v = 0.0f;
		int best_obj = -1, best_tri = -1;
		if (alive) n_rays++;
		const int nobj = sc.num_objects;
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
			} else if (ob.root_node >= 0) {
				bool enter = false;
				if (alive) {
					float rd;
					if (FLAGS & PT_F_STATS) cnt.n_box++;
					enter = box_test(ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], ray, rd);
				}
				if (ob.big_slot >= 0) {
					if (enter) flags |= 2u << ob.big_slot;
				} else if (enter) {
					float tmin = __builtin_huge_valf(), bu = 0.0f, bv = 0.0f;
					int tri = -1;
					bool hit = (WF_CHEAP_PAIRS && !(FLAGS & PT_F_ALLTRIS)) ? leaf_only_mesh_pairs<FLAGS>(sc, ob, ray, tmin, tri, bu, bv, cnt) : leaf_only_mesh<FLAGS>(sc, ob, ray, tmin, tri, bu, bv, cnt);
					if (hit && tri >= 0) {
						const f3 p = ray.o + tmin * ray.d;
						const float d = length3(p - ray.o);
						if (d < best_dist) { best_dist = d; best_t = tmin; best_obj = oi; best_tri = tri; b_u = bu; b_v = bv; }
					}
				}
			}
		}
		if (alive) {
			const int slot = base0 + k;
			wp.s0[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, best_dist);
			wp.s1[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, best_t);
			wp.s2[slot] = make_float4(mask.x, mask.y, mask.z, __int_as_float(best_obj));
			wp.s3[slot] = make_float4(__int_as_float(best_tri), b_u, b_v, __int_as_float(path));
		}
		sh_flags[chunk * PT_BLOCK + tid] = flags;
		for (int L = 1; L < nlists; ++L) {
			const unsigned long long m = __ballot((flags >> L) & 1u);
			if (lane == 0) sh_cnt[L][chunk * (PT_BLOCK / 64) + wave] = (int)__popcll(m);
		}
	} // chunks
		const unsigned long long k3 = WFS_CLOCK();
		__syncthreads();
		const unsigned long long k4 = WFS_CLOCK();
		if (tid >= 1 && tid < nlists) {   // list 0 (the survivors) was reserved before phase 2
			int total = 0;
			for (int j = 0; j < n_chunks * (PT_BLOCK / 64); ++j) {
				const int c = sh_cnt[tid][j];
				sh_cnt[tid][j] = total;
				total += c;
			}
			sh_base[tid] = total ? atomicAdd(&WF_CNT(wp, bounce, tid), total) : 0;
		}
		if (tid == 0) sh_n = 0;
		__syncthreads();
		for (int chunk = 0; chunk < n_chunks; ++chunk) {
			const int k = chunk * PT_BLOCK + tid;
			const uint32_t f = sh_flags[k];
			for (int L = 1; L < nlists; ++L) {
				const bool on = (f >> L) & 1u;
				const unsigned long long m = __ballot(on);
				if (on) {
					const int qslot = sh_base[L] + sh_cnt[L][chunk * (PT_BLOCK / 64) + wave] + (int)__popcll(m & ((1ull << lane) - 1ull));
					wp.queue[L - 1][qslot] = base0 + k;   // the ray's slot in this bounce's state set
				}
			}
		}
		__syncthreads();
		const unsigned long long k5 = WFS_CLOCK();
		sc0 += k1 - k0; sc1 += (k2 - k1) + (k4 - k3); sc2 += k3 - k2; sc3 += k5 - k4;
	}
	stats_flush(wp.stats, cnt, n_rays, n_paths, c_hm, c_hs, c_ht, 0, 0, (FLAGS & PT_F_STATS) != 0);
	if (FLAGS & PT_F_STATS) {
		uint64_t x[11] = {sh_waves, sh_lanes, sh_branches, sh_t[0], sh_t[1], sh_t[2], sh_t[3], sc0, sc1, sc2, sc3};   // seg_clk follows sh_type_lanes
		unsigned long long* dst = (unsigned long long*)&wp.stats->sh_waves;
#pragma unroll
		for (int k = 0; k < 11; ++k) {
			const uint64_t v = wave_sum_u64(x[k]);
			if (lane == 0 && v) atomicAdd(dst + k, (unsigned long long)v);
		}
	}
#undef WFS_CLOCK
}

#define WF_STAGE_F4 4
#define WF_STAGE_BYTES ((PT_BLOCK / 64) * 64 * WF_STAGE_F4 * 16)

#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 8
#endif
// LDS staging area of leaf mode 5, per wave: 64 64-bit result keys (one per lane) and a list of
// WF_UNIT_CAP 32-bit work units (owner lane | group << 6: 26 bits of group index, i.e. leaves of up to
// 2^29 triangles, more than a scene may hold)
#define WF_UNIT_CAP 320
#define WF_GRP_WAVE_BYTES 4096   /* leaf phase: 64 * 8 (+ 64 * 8 in mode 9) + WF_UNIT_CAP * 4 = 2304 B; descent of modes 6, 9: 64 node records */
#define WF_GRP_BYTES ((PT_BLOCK / 64) * WF_GRP_WAVE_BYTES)

PT_DEV uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }

// Minimum over the 4 lanes of a quad, left in every lane of the quad
PT_DEV uint32_t quad_min_u32(uint32_t x) {
	uint32_t y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0xB1, 0xf, 0xf, false); x = y < x ? y : x;    // quad_perm [1,0,3,2]
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x4E, 0xf, 0xf, false); x = y < x ? y : x;    // quad_perm [2,3,0,1]
	return x;
}

// Minimum over the 16 lanes of a DPP row, left in every lane of the row: xor 1 and xor 2 inside the
// quads, then the half-row and row mirrors (the values are already uniform per quad / per half).
PT_DEV uint32_t row_min_u32(uint32_t x) {
	uint32_t y;
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0xB1, 0xf, 0xf, false); x = y < x ? y : x;    // quad_perm [1,0,3,2]
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x4E, 0xf, 0xf, false); x = y < x ? y : x;    // quad_perm [2,3,0,1]
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x141, 0xf, 0xf, false); x = y < x ? y : x;   // row_half_mirror
	y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x140, 0xf, 0xf, false); x = y < x ? y : x;   // row_mirror
	return x;
}

// Cooperative variant of the traversal kernel: the DESCENT stays one ray per lane, the LEAVES are
// tested by the wave together.
//
// Why: a KD leaf of the reference's builder holds 17..33 triangles (children only when the left half
// is > 16, src/KDTree.cpp:96) and a ray visits ~12 leaves, so ~300 triangle tests per traversal
// against ~50 inner nodes. One ray per lane walks those triangles serially: 64 lanes fetch 64
// unrelated 48-B records per step and only a quarter of the lanes of a VALU instruction do useful
// work. Here every lane that holds a leaf hands it to the wave. How the wave tests it is the leaf
// mode FLAT (history and measurements: DESIGN.md 3.2):
//   0 / 1  two leaves per step, one per 32-lane half, lane i tests triangle i of the run (tri_soa);
//          early-exit / branch-free scalar test; 2 = mode 1 with two steps' loads in flight
//   3      four leaves per step, one per 16-lane quarter, two triangles per lane as register pairs in
//          packed FP32 (tri_pair); rays broadcast by shuffles, hits folded one by one in run order
//   4      the same step, rays handed over through LDS slots, hits reduced with DPP row minima
//   5      boxed groups: a leaf's triangles sorted and cut into groups of eight with a box each
//          (tri_grp, grp_box); the owner tests the group boxes, the wave tests only the touched
//          groups, sixteen per step                                              <- default (as 6)
//   6      mode 5 with the node records of the descent fetched by the quad through LDS
// In every mode the hits are merged into the owner's state so that the result is exactly what the
// reference's serial loop over the run with strict '<' gives (first triangle wins ties).
//
// Round structure (wave-uniform control flow throughout):
//   commit finished traversals and refill idle lanes from the wave's queue chunk (once enough lanes
//   wait) -> all lanes descend until each holds a leaf or is done (stop at ATHRESH stragglers) ->
//   the leaves are worked off -> owners pop their stacks.
template <int FLAGS, int OCC, int CHUNK, int ATHRESH, int FLAT>
__global__ __launch_bounds__(PT_BLOCK, OCC) void wf_traverse_coop_kernel(DevScene sc, WfParams wp, int obj_index, int slot_index) {
	static_assert(FLAT == 3 || FLAT == 4 || FLAT == 9, "leaf modes of the product build: 9 (default), 4 and 3 (fall-backs for deep trees); 0-2, 5-8 were A/B steps of rounds 1-2");
	extern __shared__ int lds_stack[];
	const int tid = threadIdx.x, lane = tid & 63;
	// counting variant only: constant-rate clock read by lane 0, summed per phase into PtxStats.diag0..3
	// (refill + commit, descent, leaf phase, of which unit steps of mode 5); tools/phase_clock.py
#define WF_CLOCK() (((FLAGS & PT_F_STATS) && lane == 0) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
	int* stack = lds_stack + tid;
	const DevObject& ob = sc.objects[obj_index];
	// the rays of this launch: the mesh's queue, or the retry queue the wide traversal left for it (pt_wide.h)
	const int n = wp.use_retry ? WF_RCNT(wp, wp.bounce, 1 + slot_index) : WF_CNT(wp, wp.bounce, 1 + slot_index);
	const int* __restrict__ queue = wp.use_retry ? wp.retry : wp.queue[slot_index];
	const float4* __restrict__ nodes = sc.nodes;
	const float4* __restrict__ pa = sc.tri_soa;
	const float4* __restrict__ pe1 = sc.tri_soa + sc.num_tris;
	const float4* __restrict__ pe2 = sc.tri_soa + 2 * (size_t)sc.num_tris;
	TraceCounters cnt = {0, 0, 0, 0, 0, 0};
	uint64_t act = 0, tot = 0, n_taken = 0, dg0 = 0, dg1 = 0, dg2 = 0, dg3 = 0;
	// what the kernel EXECUTES (PtxStats.x_*, bench.py's roofline): descent iterations, lanes descending in them,
	// leaf lanes fetching group boxes, -, unit steps, units, rounds with a leaf phase, commit / refill turns
	uint64_t x_iters = 0, x_item = 0, x_half = 0, x_usteps = 0, x_units = 0, x_uphases = 0, x_turns = 0;

	WaveFeed feed;
	feed.exhausted = CHUNK == 0;
	if (CHUNK == 0) {
		// static split: wave w owns queue[n*w/W, n*(w+1)/W)
		const long long nw = (long long)gridDim.x * (PT_BLOCK / 64);
		const long long gw = (long long)blockIdx.x * (PT_BLOCK / 64) + (tid >> 6);
		feed.next = (int)(((long long)n * gw) / nw);
		feed.hi = (int)(((long long)n * (gw + 1)) / nw);
	} else {
		feed.next = feed.hi = 0;
	}
	int* cursor = wp.use_retry ? &WF_RCUR(wp, wp.bounce, 1 + slot_index) : &WF_CUR(wp, wp.bounce, 1 + slot_index);

	// wave-uniform LDS byte address of this wave's staging area (leaf modes 5/6), as a scalar for the LDS-DMA base
	const uint32_t nbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(
		reinterpret_cast<unsigned char*>(lds_stack + sc.stack_entries * PT_BLOCK) + (tid >> 6) * WF_GRP_WAVE_BYTES));
	enum { IDLE = 0, DESCEND = 1, LEAF = 2, DONE = 3 };
	int state = IDLE;
	int path = 0, leaf_first = 0, leaf_count = 0, leaf_block = 0;
	float cur_d = 0.0f;
	int cur_obj_r = -1;   // leaf mode 9: the path's closest object so far, read at refill
	Ray ray;
	TravState ts;
	ts.curr = -1; ts.sp = 0; ts.tri = -1; ts.tmin = 0.0f; ts.bu = ts.bv = 0.0f; ts.intersects = false;
	init_ray(ray, mk3(0.0f), mk3(1.0f));

	// the traversal of this lane's ray is over: merge into the path's closest-hit record
	auto commit = [&]() {
		if (ts.intersects && ts.tri >= 0) {
			const f3 p = ray.o + ts.tmin * ray.d;
			const float d = length3(p - ray.o);
			// leaf mode 9 read the path's current hit object at refill (same round trip as the ray) and got u, v of
			// the winning triangle from the unit step that tested it: its commit issues no load and waits for nothing
			const int cur_obj = FLAT == 9 ? cur_obj_r : __float_as_int(wp.s2[path].w);
			if (d < cur_d || (d == cur_d && cur_obj >= 0 && obj_index < cur_obj)) {
				if (FLAT >= 5 && FLAT != 9) {
					// mode 5 keeps only (t, triangle) per leaf; u and v are the test's own values, taken
					// once for the triangle that won (same operations, same bits)
					float t_, u_, v_;
					const float4 a = pa[ts.tri], e1 = pe1[ts.tri], e2 = pe2[ts.tri];
					(void)tri_test_flat(ray, mk3(a.x, a.y, a.z), mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z), t_, u_, v_);
					ts.bu = u_; ts.bv = v_;
				}
				wp.s0[path].w = d;
				wp.s1[path].w = ts.tmin;
				wp.s2[path].w = __int_as_float(obj_index);
				store_hit(wp.s3 + path, ts.tri, ts.bu, ts.bv);
			}
		}
		state = IDLE;
	};

	for (;;) {
		const unsigned long long c0 = WF_CLOCK();
		// ---- finished traversals are committed here, once per round: inside the descent loop the
		// wave would run this block (a dependent load, up to four stores) in nearly every iteration,
		// because some lane or other finishes in most of them ----
		// ... and only when enough lanes wait for it (WF_REFILL_MIN), or nothing else is left to do: with two or
		// three traversals ending per round the whole wave would otherwise pay for this block every round.
		const int n_wait = (int)__popcll(__ballot(state == DONE || state == IDLE));
		const bool turn = n_wait >= WF_REFILL_MIN || n_wait == 64;
		if (FLAGS & PT_F_STATS) { if (turn && lane == 0) x_turns++; }
		if (turn && state == DONE) commit();
		// ---- refill idle lanes from the wave's chunk ----
		for (int attempt = 0; turn && attempt < 2; ++attempt) {
			const unsigned long long idle = __ballot(state == IDLE);
			if (idle == 0ull) break;
			if (feed.next >= feed.hi) {
				if (feed.exhausted) break;
				feed_claim(feed, cursor, n, CHUNK);
				if (feed.exhausted) break;
			}
			const int mine = feed.next + (int)__popcll(idle & ((1ull << lane) - 1ull));
			if (state == IDLE && mine < feed.hi) {
				if (FLAGS & PT_F_STATS) n_taken++;
				path = queue[mine];
				const float4 a0 = wp.s0[path], a1 = wp.s1[path];
				init_ray(ray, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z));
				cur_d = a0.w;
				if (FLAT == 9) cur_obj_r = __float_as_int(wp.s2[path].w);
				trav_begin(ts, ob);
				state = DESCEND;
			}
			feed.next += (int)__popcll(idle);
		}
		if (turn && __ballot(state != IDLE) == 0ull) break;

		const unsigned long long c1 = WF_CLOCK();
		dg0 += c1 - c0;
		// ---- descent: every lane walks inner nodes until it holds a leaf or its traversal ends ----
		// The loop stops once no more than ATHRESH lanes are still on their way: waiting for the last
		// few would idle the rest (measured with ATHRESH = 0: 19 % of the lanes busy per iteration);
		// the stragglers simply carry on in the next round, next to the lanes that popped.
		for (int iter = 0; ; ++iter) {
			const int n_desc = (int)__popcll(__ballot(state == DESCEND));
			if (n_desc == 0 || (iter > 0 && n_desc <= ATHRESH)) break;
			if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 64; x_iters++; } if (state == DESCEND) x_item++; }
			if (FLAT == 6 || FLAT == 9) {
				// Leaf mode 6 = mode 5 + node records fetched by the QUAD: in pass k the four lanes of a quad
				// read the four 16-B parts of the record lane 4q + k wants (one 64-B line: one tag lookup in
				// the L1 instead of four, the vector-memory path being this kernel's limit) straight into LDS
				// (global_load_lds: destination = pass base + lane x 16, so the record of lane 4q + k lands
				// contiguously at pass k, offset 64 q); every lane then reads its own record from LDS.
				const int mine = state == DESCEND ? ts.curr : -1;
				const uint32_t part16 = (uint32_t)(lane & 3) * 16u;   // 32-bit byte offsets from the uniform base: scalar-base addressing
#define WF_NODE_PASS(K) { \
					const int want = __builtin_amdgcn_mov_dpp(mine, (K) * 0x55, 0xf, 0xf, false);   /* quad_perm [K,K,K,K] */ \
					if (want >= 0) \
						__builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char*>(nodes) + ((uint32_t)want * 64u + part16), (__attribute__((address_space(3))) void*)(uintptr_t)(nbase + (K) * 1024u), 16, 0, 0); \
				}
				WF_NODE_PASS(0) WF_NODE_PASS(1) WF_NODE_PASS(2) WF_NODE_PASS(3)
#undef WF_NODE_PASS
			}
			if (state == DESCEND) {
				if (FLAGS & PT_F_STATS) { act++;  }
				const int curr = ts.curr;
				float4 q0, q1, q2, q3;
				if (FLAT == 6 || FLAT == 9) {
					const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<unsigned char*>(lds_stack + sc.stack_entries * PT_BLOCK) +
					                                                    (tid >> 6) * WF_GRP_WAVE_BYTES + (lane & 3) * 1024 + (lane >> 2) * 64);
					q3 = rec[3]; q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
				} else {
					q3 = nodes[4 * curr + 3];
					q0 = nodes[4 * curr + 0];
					q1 = nodes[4 * curr + 1];
					q2 = nodes[4 * curr + 2];
				}
				// Both child boxes at once, packed (pt_trace.h), BEFORE the leaf / inner decision: the test is
				// branch-free, a leaf reached through a plain link is rare, and the whole record is then
				// fetched in one round trip instead of the links first and the boxes after the branch.
				float left_dist, right_dist;
				const uint32_t bb = box_test_x2(q0, q1, q2, ray.o.x, ray.o.y, ray.o.z, ray.inv.x, ray.inv.y, ray.inv.z, ray.sx, ray.sy, ray.sz,
				                                left_dist, right_dist);
				if (__float_as_int(q3.x) == -1 && __float_as_int(q3.y) == -1) {
					// a leaf reached through a plain link (more than 63 triangles, or none)
					leaf_first = __float_as_int(q3.z);
					leaf_count = __float_as_int(q3.w);
					leaf_block = __float_as_int(q0.x);
					if (FLAGS & PT_F_STATS) {
						cnt.n_leaf++;
						cnt.n_tri += (uint32_t)leaf_count;
						fnv_word(cnt.leaf_hash, (uint32_t)obj_index);
						fnv_word(cnt.leaf_hash, (uint32_t)(curr - ob.first_node));
					}
					state = LEAF;
				} else {
					const int left = __float_as_int(q3.z), right = __float_as_int(q3.w);   // packed links: pt_scene.h
					if (FLAGS & PT_F_STATS) cnt.n_inner++;
					const bool left_res = (left != -1) & ((bb & 1u) != 0u) & (left_dist < ts.tmin);
					const bool right_res = (right != -1) & ((bb & 2u) != 0u) & (right_dist < ts.tmin);
					if (FLAGS & PT_F_STATS) cnt.n_box += (left != -1 ? 1u : 0u) + (right != -1 ? 1u : 0u);
					if (left_res && right_res) {
						int further = right, nxt = right;
						if (left_dist > right_dist) further = left; else nxt = left;
						stack[ts.sp * PT_BLOCK] = further;
						++ts.sp;
						ts.curr = nxt;
					} else if (left_res) {
						ts.curr = left;
					} else if (right_res) {
						ts.curr = right;
					} else if (ts.sp == 0) {
						state = DONE;
					} else {
						--ts.sp;
						ts.curr = stack[ts.sp * PT_BLOCK];
					}
				}
			}
			// a link that names a leaf run (0x80000000 | block << 6 | count) is taken up at once: no node fetch
			if (state == DESCEND && ts.curr < -1) {
				leaf_block = (int)(((uint32_t)ts.curr & 0x7fffffffu) >> 6);
				leaf_first = leaf_block << 5;
				leaf_count = ts.curr & 63;
				if (FLAGS & PT_F_STATS) {
					cnt.n_leaf++;
					cnt.n_tri += (uint32_t)leaf_count;
					fnv_word(cnt.leaf_hash, (uint32_t)obj_index);
					fnv_word(cnt.leaf_hash, (uint32_t)__float_as_int(sc.tri_geom[3 * leaf_first + 2].z)); // the leaf's node index
				}
				state = LEAF;
			}
		}

		// ---- leaves (by leaf mode) ----
		unsigned long long m = __ballot(state == LEAF);
		const unsigned long long c2 = WF_CLOCK();
		dg1 += c2 - c1;
		if (FLAT >= 5) {
			// Boxed groups (pt_scene.h, tri_grp / grp_box). A leaf of the reference's builder holds 17..33
			// triangles and a ray that enters its box passes most of them by: here every LEAF lane first
			// tests the boxes of its leaf's groups of eight, one group per iteration, and appends a work
			// unit (its slot, the group) to the wave's list for every box it touches; then the wave tests
			// the listed groups, SIXTEEN per step, four lanes per group and two triangles per lane in
			// packed FP32. The hits of a unit are reduced inside the quad (DPP) to the smallest t and,
			// among equal t, the smallest slot of the unsorted run, and merged into the owner's 64-bit
			// key (t bits << 32 | slot) with an LDS atomic minimum: exactly what the reference's loop over
			// the run with strict '<' keeps. Skipping a group changes nothing: its box contains its
			// triangles with padding far above the rounding of either test, so a skipped group has no hit.
			// Leaves of any size: a leaf of n triangles has ceil(n / 8) groups; when the list is full the
			// wave works it off and carries on with the remaining groups.
			unsigned char* area = reinterpret_cast<unsigned char*>(lds_stack + sc.stack_entries * PT_BLOCK) + (tid >> 6) * WF_GRP_WAVE_BYTES;
			unsigned long long* keys = reinterpret_cast<unsigned long long*>(area);          // one result key per lane
			// leaf mode 9: (u, v) of the triangle behind each lane's key, written by the unit step that holds the key
			float2* uvs = reinterpret_cast<float2*>(area + 64 * 8);
			uint32_t* units = reinterpret_cast<uint32_t*>(area + 64 * 8 + (FLAT == 9 ? 64 * 8 : 0));
			const float4* __restrict__ gb = sc.grp_box;
			const float4* __restrict__ tg = sc.tri_grp;
			const unsigned long long lt = (1ull << lane) - 1ull;
			const bool isleaf = state == LEAF;
			const int ngroups = isleaf ? (leaf_count + 7) >> 3 : 0;
			if (FLAGS & PT_F_STATS) { if (lane == 0 && m != 0ull) x_uphases++; if (ngroups > 0) x_half++; }
			// slab test of a group box on min/max (a NaN from 0 * inf drops out: that axis does not constrain)
			auto touches = [](const float4 lo, const float4 hi, f3 o, f3 inv) {
				const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
				const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
				const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
				const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
				const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
				return !(tn > tf) && !(tf < 0.0f);
			};
			// ---- fill, first pass: every LEAF lane tests the boxes of the first four groups of its leaf;
			// the four boxes are one 128-B line and are fetched together, so the pass costs one memory
			// round trip (a leaf of up to 32 triangles has no more groups) ----
			uint32_t tb = 0;
			if (FLAT == 9) {
				// Leaf mode 9: the four group boxes of the run's first block from their 48-B quantised form (pt_scene.h,
				// grp_qbox): three 16-B loads instead of eight - this kernel pays per vector-memory instruction
				// (DESIGN.md 3.2). The decoding is folded into the slab test: (U.lo + q * step - o) * inv =
				// (U.lo - o) * inv + q * (step * inv); the decoded boxes contain the fp32 ones, so a group is skipped
				// only if the ray misses a box that holds all its triangles with the usual padding.
				if (isleaf) {
					keys[lane] = ~0ull;
					if (ngroups > 0) {
						const float4* qp = sc.grp_qbox + 3 * (size_t)leaf_block;
						const float4 f0 = qp[0], f1 = qp[1], f2 = qp[2];
						const float ax = (f0.x - ray.o.x) * ray.inv.x, ay = (f0.y - ray.o.y) * ray.inv.y, az = (f0.z - ray.o.z) * ray.inv.z;
						const float sx = f0.w * ray.inv.x, sy = f1.x * ray.inv.y, sz = f1.y * ray.inv.z;
						const uint32_t w0 = __float_as_uint(f1.z), w1 = __float_as_uint(f1.w), w2 = __float_as_uint(f2.x),
						               w3 = __float_as_uint(f2.y), w4 = __float_as_uint(f2.z), w5 = __float_as_uint(f2.w);
#define WF_QB(W, K) ((float)(((W) >> (8 * (K))) & 0xffu))
						auto hit = [&](float lx, float ly, float lz, float hx, float hy, float hz) {
							const float tax = __builtin_fmaf(lx, sx, ax), tbx = __builtin_fmaf(hx, sx, ax);
							const float tay = __builtin_fmaf(ly, sy, ay), tby = __builtin_fmaf(hy, sy, ay);
							const float taz = __builtin_fmaf(lz, sz, az), tbz = __builtin_fmaf(hz, sz, az);
							const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tax, tbx), __builtin_fminf(tay, tby)), __builtin_fminf(taz, tbz));
							const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tax, tbx), __builtin_fmaxf(tay, tby)), __builtin_fmaxf(taz, tbz));
							return !(tn > tf) && !(tf < 0.0f);
						};
						// box g = bytes 6g .. 6g+5 of the 24
						tb = (hit(WF_QB(w0, 0), WF_QB(w0, 1), WF_QB(w0, 2), WF_QB(w0, 3), WF_QB(w1, 0), WF_QB(w1, 1)) ? 1u : 0u) |
						     (ngroups > 1 && hit(WF_QB(w1, 2), WF_QB(w1, 3), WF_QB(w2, 0), WF_QB(w2, 1), WF_QB(w2, 2), WF_QB(w2, 3)) ? 2u : 0u) |
						     (ngroups > 2 && hit(WF_QB(w3, 0), WF_QB(w3, 1), WF_QB(w3, 2), WF_QB(w3, 3), WF_QB(w4, 0), WF_QB(w4, 1)) ? 4u : 0u) |
						     (ngroups > 3 && hit(WF_QB(w4, 2), WF_QB(w4, 3), WF_QB(w5, 0), WF_QB(w5, 1), WF_QB(w5, 2), WF_QB(w5, 3)) ? 8u : 0u);
#undef WF_QB
					}
				}
			} else
			if (isleaf) {
				keys[lane] = ~0ull;
				if (ngroups > 0) {   // the four slots of the run's first block exist whenever the run is not empty
					const float4* bx = gb + 8 * (size_t)leaf_block;
					const float4 b0 = bx[0], b1 = bx[1], b2 = bx[2], b3 = bx[3], b4 = bx[4], b5 = bx[5], b6 = bx[6], b7 = bx[7];
					tb = (touches(b0, b1, ray.o, ray.inv) ? 1u : 0u) | (ngroups > 1 && touches(b2, b3, ray.o, ray.inv) ? 2u : 0u) |
					     (ngroups > 2 && touches(b4, b5, ray.o, ray.inv) ? 4u : 0u) | (ngroups > 3 && touches(b6, b7, ray.o, ray.inv) ? 8u : 0u);
				}
			}
			int pending = 0;   // units of the first pass, appended at the top of the loop below
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const bool on = (tb >> k) & 1u;
				const unsigned long long tm = __ballot(on);
				if (on) units[pending + (int)__popcll(tm & lt)] = (uint32_t)lane | ((uint32_t)k << 6);
				pending += (int)__popcll(tm);
			}
			int g = 4;       // per lane: next group of the own leaf beyond the first four
			for (;;) {
				int nunits = pending;
				pending = 0;
				// ---- fill, the groups beyond the fourth (leaves of more than 32 triangles: rare), one per
				// LEAF lane and iteration ----
				{
					while (nunits <= WF_UNIT_CAP - 64) {
						const bool more = g < ngroups;
						if (__ballot(more) == 0ull) break;
						bool touch = false;
						if (more) {
							const float4* bx = gb + 2 * (size_t)(4 * leaf_block + g);
							touch = touches(bx[0], bx[1], ray.o, ray.inv);
						}
						const unsigned long long tm = __ballot(touch);
						if (touch) units[nunits + (int)__popcll(tm & lt)] = (uint32_t)lane | ((uint32_t)g << 6);
						nunits += (int)__popcll(tm);
						++g;
					}
				}
				__builtin_amdgcn_wave_barrier();
				const unsigned long long c3 = WF_CLOCK();
				// ---- work the list off: 16 units per step ----
				if (FLAGS & PT_F_STATS) { if (lane == 0) x_units += (uint64_t)nunits; }
				for (int s16 = 0; s16 < nunits; s16 += 16) {
					const int ui = s16 + (lane >> 2);
					const bool on = ui < nunits;
					const uint32_t un = units[on ? ui : s16];
					const int owner = (int)(un & 63u), grp = (int)(un >> 6);
					if (FLAGS & PT_F_STATS) { if (lane == 0) x_usteps++; }
					// the owner's ray and leaf come straight out of its registers (no LDS copy: the staging
					// area stays small enough for one more wave per SIMD)
					const f3 bo = mk3(__shfl(ray.o.x, owner, 64), __shfl(ray.o.y, owner, 64), __shfl(ray.o.z, owner, 64));
					const f3 bd = mk3(__shfl(ray.d.x, owner, 64), __shfl(ray.d.y, owner, 64), __shfl(ray.d.z, owner, 64));
					const int oblock = __shfl(leaf_block, owner, 64);
					if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 128;  } }
					if (on) {   // uniform per quad
						const float4* bp = tg + ((size_t)(4 * oblock + grp) * PT_GRP_BLOCK_F4 + (lane & 3));
						const float4 c0 = bp[0], c1 = bp[4], c2 = bp[8], c3 = bp[12], c4 = bp[16];
						f3x2 va, e1, e2;
						va.x = (v2f){c0.x, c0.y}; va.y = (v2f){c0.z, c0.w}; va.z = (v2f){c1.x, c1.y};
						e1.x = (v2f){c1.z, c1.w}; e1.y = (v2f){c2.x, c2.y}; e1.z = (v2f){c2.z, c2.w};
						e2.x = (v2f){c3.x, c3.y}; e2.y = (v2f){c3.z, c3.w}; e2.z = (v2f){c4.x, c4.y};
						v2f t, u, v;
						const uint32_t hb = tri_test_flat_x2(bo, bd, va, e1, e2, t, u, v);
						if (FLAGS & PT_F_STATS) { act += 2;  }
						if (__ballot(hb != 0u) != 0ull) {
							// a hit has t > 5e-8, so its bit pattern orders like its value; 0xffffffff = no hit
							const uint32_t k0 = (hb & 1u) ? __float_as_uint(t.x) : 0xffffffffu;
							const uint32_t k1 = (hb & 2u) ? __float_as_uint(t.y) : 0xffffffffu;
							const uint32_t k = quad_min_u32(k0 < k1 ? k0 : k1);
							const uint32_t s0 = __float_as_uint(c4.z), s1 = __float_as_uint(c4.w);   // slots in the unsorted run
							const uint32_t cand = quad_min_u32(umin3(k0 == k ? s0 : 0xffffffffu, k1 == k ? s1 : 0xffffffffu, 0xffffffffu));
							if (FLAT == 9) {
								// the one lane of the quad that holds the quad's best triangle merges it and, if the key it
								// wrote is the owner's minimum after all lanes' atomics, publishes that triangle's u, v
								const bool win0 = (k0 == k) & (s0 == cand), win1 = (k1 == k) & (s1 == cand);
								if (k != 0xffffffffu && (win0 | win1)) {
									const unsigned long long key = ((unsigned long long)k << 32) | (unsigned long long)cand;
									atomicMin(&keys[owner], key);
									// re-read through an atomic load (the word was just the target of other lanes' atomics: a plain load
									// could legally be folded with the operand above); the wave executes both in lock-step
									__builtin_amdgcn_wave_barrier();
									if (__hip_atomic_load(&keys[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == key)
										uvs[owner] = win0 ? make_float2(u.x, v.x) : make_float2(u.y, v.y);
								}
							} else
							if ((lane & 3) == 0 && k != 0xffffffffu)
								atomicMin(&keys[owner], ((unsigned long long)k << 32) | (unsigned long long)cand);
						}
					}
				}
				__builtin_amdgcn_wave_barrier();
				dg3 += WF_CLOCK() - c3;
				if (__ballot(g < ngroups) == 0ull) break;
			}
			if (isleaf) {
				const unsigned long long key = keys[lane];
				if (key != ~0ull) {
					ts.intersects = true;
					const float tt = __uint_as_float((uint32_t)(key >> 32));
					if (tt < ts.tmin) {
						ts.tmin = tt; ts.tri = (int)(uint32_t)key;   // tri_grp carries device indices
						if (FLAT == 9) { const float2 w = uvs[lane]; ts.bu = w.x; ts.bv = w.y; }
					}
				}
			}
		} else
		if (FLAT == 4) {
			// Mode 3's step (four owners, one per 16-lane quarter, two triangles per lane in packed FP32)
			// with the hand-over moved to LDS. Every LEAF lane writes its ray and leaf run ONCE per round
			// into slot `rank` (its rank among the LEAF lanes) of the wave's staging area; step s serves
			// slots 4s..4s+3, quarter q reads slot 4s+q (a broadcast read) - no owner search, no per-step
			// shuffles. The hits of a quarter are reduced with DPP row operations (min t, then the lowest
			// triangle slot among equal t: the serial loop's "first strictly smaller wins"), the winning
			// lane writes (t, u, v, slot) back INTO the slot, and the owners pick their results up after
			// the last step. A slot is 4 float4, the ray components already doubled into the register
			// pairs the packed test wants: (ox ox oy oy)(oz oz dx dx)(dy dy dz dz)(block, count, key, -);
			// the result replaces the first word.
			float4* stage = reinterpret_cast<float4*>(lds_stack + sc.stack_entries * PT_BLOCK) + (tid >> 6) * (WF_STAGE_F4 * 64);
			const int quarter = lane >> 4, sub16 = lane & 15;
			const float4* __restrict__ pp = sc.tri_pair;
			const int nleaf = (int)__popcll(m);
			const int rank = (int)__popcll(m & ((1ull << lane) - 1ull));
			if (state == LEAF) {
				float4* sl = stage + WF_STAGE_F4 * rank;
				sl[0] = make_float4(ray.o.x, ray.o.x, ray.o.y, ray.o.y);
				sl[1] = make_float4(ray.o.z, ray.o.z, ray.d.x, ray.d.x);
				sl[2] = make_float4(ray.d.y, ray.d.y, ray.d.z, ray.d.z);
				sl[3] = make_float4(__int_as_float(leaf_block), __int_as_float(leaf_count), __uint_as_float(0xffffffffu), 0.0f);
			}
			__builtin_amdgcn_wave_barrier();
			for (int s4 = 0; s4 < nleaf; s4 += 4) {
				const int slot = s4 + quarter;               // <= 63
				float4* sl = stage + WF_STAGE_F4 * slot;
				const float4 r0 = sl[0], r1 = sl[1], r2 = sl[2], r3 = sl[3];
				f3x2 bo, bd;
				bo.x = (v2f){r0.x, r0.y}; bo.y = (v2f){r0.z, r0.w}; bo.z = (v2f){r1.x, r1.y};
				bd.x = (v2f){r1.z, r1.w}; bd.y = (v2f){r2.x, r2.y}; bd.z = (v2f){r2.z, r2.w};
				const int blk = __float_as_int(r3.x);
				const int count = slot < nleaf ? __float_as_int(r3.y) : 0;   // slots past nleaf hold stale data
				uint32_t best_k = 0xffffffffu;
				for (int base = 0; __ballot(base < count) != 0ull; base += 32) {
					if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 128;  } }
					if (base < count) {   // uniform per quarter (= per DPP row)
						const float4* bp = pp + ((size_t)(blk + (base >> 5)) * PT_PAIR_BLOCK_F4 + sub16);
						const float4 c0 = bp[0], c1 = bp[16], c2 = bp[32], c3 = bp[48], c4 = bp[64];
						f3x2 va, e1, e2;
						va.x = (v2f){c0.x, c0.y}; va.y = (v2f){c0.z, c0.w}; va.z = (v2f){c1.x, c1.y};
						e1.x = (v2f){c1.z, c1.w}; e1.y = (v2f){c2.x, c2.y}; e1.z = (v2f){c2.z, c2.w};
						e2.x = (v2f){c3.x, c3.y}; e2.y = (v2f){c3.z, c3.w}; e2.z = (v2f){c4.x, c4.y};
						v2f t, u, v;
						const uint32_t hb = tri_test_flat_x2p(bo, bd, va, e1, e2, t, u, v);
						if (FLAGS & PT_F_STATS) {
							const int here = (sub16 < count - base ? 1 : 0) + (sub16 + 16 < count - base ? 1 : 0);
							act += here; 
						}
						if (__ballot(hb != 0u) != 0ull) {
							// a hit has t > 5e-8, so its bit pattern orders like its value; 0xffffffff = no hit
							const uint32_t k0 = (hb & 1u) ? __float_as_uint(t.x) : 0xffffffffu;
							const uint32_t k1 = (hb & 2u) ? __float_as_uint(t.y) : 0xffffffffu;
							const uint32_t k = row_min_u32(k0 < k1 ? k0 : k1);
							const uint32_t cand = row_min_u32(k0 == k ? (uint32_t)sub16 : k1 == k ? (uint32_t)sub16 + 16u : 64u);
							if (k < best_k) {   // strict: an earlier block of the run keeps a tie
								best_k = k;
								if ((cand & 15u) == (uint32_t)sub16) {
									const bool second = cand >= 16u;
									sl[0] = make_float4(second ? t.y : t.x, second ? u.y : u.x, second ? v.y : v.x, __int_as_float(base + (int)cand));
									sl[3].z = __uint_as_float(k);
								}
							}
						}
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			if (state == LEAF) {
				const float4* sl = stage + WF_STAGE_F4 * rank;
				const float4 res = sl[0];
				const uint32_t k = __float_as_uint(sl[3].z);
				if (k != 0xffffffffu) {
					ts.intersects = true;
					if (res.x < ts.tmin) { ts.tmin = res.x; ts.tri = leaf_first + __float_as_int(res.w); ts.bu = res.y; ts.bv = res.z; }
				}
			}
		} else
		if (FLAT == 3) {
			// Four owners per step, one per 16-lane quarter; every lane tests TWO triangles of its owner's run
			// (slots p and p + 16 of a tri_pair block, pt_scene.h) with packed FP32 arithmetic. Padding
			// slots hold zero triangles, which never hit, so no per-slot validity is needed.
			const int quarter = lane >> 4, sub16 = lane & 15;
			const float4* __restrict__ pp = sc.tri_pair;
			while (m != 0ull) {
				int l[4];
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					l[k] = -1;
					if (m != 0ull) { l[k] = (int)__ffsll((long long)m) - 1; m &= m - 1ull; }
				}
				const int mine_l = quarter == 0 ? l[0] : quarter == 1 ? l[1] : quarter == 2 ? l[2] : l[3];
				const bool on = mine_l >= 0;
				const int src = on ? mine_l : l[0];
				const f3 bo = mk3(__shfl(ray.o.x, src, 64), __shfl(ray.o.y, src, 64), __shfl(ray.o.z, src, 64));
				const f3 bd = mk3(__shfl(ray.d.x, src, 64), __shfl(ray.d.y, src, 64), __shfl(ray.d.z, src, 64));
				const int blk = __shfl(leaf_block, src, 64);
				const int count_src = __shfl(leaf_count, src, 64);
				const int count = on ? count_src : 0;
				int cmax = 0;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int c = l[k] >= 0 ? __builtin_amdgcn_readlane(leaf_count, l[k]) : 0;
					cmax = c > cmax ? c : cmax;
				}
				const bool owner = state == LEAF && ((lane == l[0]) | (lane == l[1]) | (lane == l[2]) | (lane == l[3]));
				const int my_quarter = lane == l[0] ? 0 : lane == l[1] ? 1 : lane == l[2] ? 2 : 3;   // as an owner
				for (int base = 0; base < cmax; base += 32) {
					uint32_t hb = 0;
					v2f t = {0.0f, 0.0f}, u = t, v = t;
					if (base < count) {
						const float4* bp = pp + ((size_t)(blk + (base >> 5)) * PT_PAIR_BLOCK_F4 + sub16);
						const float4 c0 = bp[0], c1 = bp[16], c2 = bp[32], c3 = bp[48], c4 = bp[64];
						f3x2 va, e1, e2;
						va.x = (v2f){c0.x, c0.y}; va.y = (v2f){c0.z, c0.w}; va.z = (v2f){c1.x, c1.y};
						e1.x = (v2f){c1.z, c1.w}; e1.y = (v2f){c2.x, c2.y}; e1.z = (v2f){c2.z, c2.w};
						e2.x = (v2f){c3.x, c3.y}; e2.y = (v2f){c3.z, c3.w}; e2.z = (v2f){c4.x, c4.y};
						hb = tri_test_flat_x2(bo, bd, va, e1, e2, t, u, v);
						if (FLAGS & PT_F_STATS) {
							const int here = (sub16 < count - base ? 1 : 0) + (sub16 + 16 < count - base ? 1 : 0);
							act += here; 
						}
					}
					if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 128;  } }
					// fold in triangle order: slot 0 holds triangles base..base+15, slot 1 base+16..base+31
#pragma unroll
					for (int slot = 0; slot < 2; ++slot) {
						unsigned long long hm = __ballot((hb >> slot) & 1u);
						const float ts_ = slot ? t.y : t.x, us_ = slot ? u.y : u.x, vs_ = slot ? v.y : v.x;
						while (hm != 0ull) {
							// the lowest remaining hit of every quarter is handled in this iteration
							const uint32_t g0 = (uint32_t)hm & 0xffffu, g1 = (uint32_t)(hm >> 16) & 0xffffu;
							const uint32_t g2 = (uint32_t)(hm >> 32) & 0xffffu, g3 = (uint32_t)(hm >> 48) & 0xffffu;
							const int w0 = g0 ? (int)__ffs((int)g0) - 1 : 0, w1 = g1 ? (int)__ffs((int)g1) - 1 : 0;
							const int w2 = g2 ? (int)__ffs((int)g2) - 1 : 0, w3 = g3 ? (int)__ffs((int)g3) - 1 : 0;
							const int wq = my_quarter == 0 ? w0 : my_quarter == 1 ? w1 : my_quarter == 2 ? w2 : w3;
							const uint32_t gq = my_quarter == 0 ? g0 : my_quarter == 1 ? g1 : my_quarter == 2 ? g2 : g3;
							const int w = my_quarter * 16 + wq;
							const float tt = __shfl(ts_, w, 64), uu = __shfl(us_, w, 64), vv = __shfl(vs_, w, 64);
							if (owner && gq != 0u) {
								ts.intersects = true;
								if (tt < ts.tmin) { ts.tmin = tt; ts.tri = leaf_first + base + slot * 16 + wq; ts.bu = uu; ts.bv = vv; }
							}
							unsigned long long clr = 0ull;
							if (g0) clr |= 1ull << w0;
							if (g1) clr |= 1ull << (16 + w1);
							if (g2) clr |= 1ull << (32 + w2);
							if (g3) clr |= 1ull << (48 + w3);
							hm &= ~clr;
						}
					}
				}
			}
		}

		dg2 += WF_CLOCK() - c2;
		// ---- owners pop ----
		if (state == LEAF) {
			if (ts.sp == 0) state = DONE;
			else {
				--ts.sp;
				ts.curr = stack[ts.sp * PT_BLOCK];
				state = DESCEND;
				if (ts.curr < -1) {   // the popped link names a leaf run
					leaf_block = (int)(((uint32_t)ts.curr & 0x7fffffffu) >> 6);
					leaf_first = leaf_block << 5;
					leaf_count = ts.curr & 63;
					if (FLAGS & PT_F_STATS) {
						cnt.n_leaf++;
						cnt.n_tri += (uint32_t)leaf_count;
						fnv_word(cnt.leaf_hash, (uint32_t)obj_index);
						fnv_word(cnt.leaf_hash, (uint32_t)__float_as_int(sc.tri_geom[3 * leaf_first + 2].z));
					}
					state = LEAF;
				}
			}
		}
	}
	stats_flush(wp.stats, cnt, 0, 0, 0, 0, 0, act, tot, (FLAGS & PT_F_STATS) != 0, n_taken, true);
	if (FLAGS & PT_F_STATS) {
		diag_flush(wp.stats, dg0, dg1, dg2, dg3);
		exec_flush(wp.stats, x_iters, x_item, x_half, 0, x_usteps, x_units, x_uphases, x_turns);
	}
#undef WF_CLOCK
}

// Parity harness (ptx_trace_rays_queued): caller rays become the paths of a one-queue pass, so that the
// PRODUCTION traversal kernel can be checked ray by ray against the oracle. Fill = what the segment
// kernel does for a new ray and one multi-node mesh: empty closest-hit record, root box test, enqueue.
__global__ __launch_bounds__(PT_BLOCK) void wfq_fill_kernel(DevScene sc, WfParams wp, int n, const float* __restrict__ origins,
                                                            const float* __restrict__ dirs, int obj_index, int slot_index) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	const DevObject& ob = sc.objects[obj_index];
	bool enter = false;
	if (i < n) {
		Ray ray;
		init_ray(ray, mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
		float rd;
		enter = box_test(ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], ray, rd);
		wp.s0[i] = make_float4(ray.o.x, ray.o.y, ray.o.z, __builtin_huge_valf());
		wp.s1[i] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
		wp.s2[i] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(-1));
		wp.s3[i] = make_float4(__int_as_float(-1), 0.0f, 0.0f, 0.0f);
	}
	const int slot = wave_append(&WF_CNT(wp, 0, 1 + slot_index), enter);
	if (enter) wp.queue[slot_index][slot] = i;
}

__global__ __launch_bounds__(PT_BLOCK) void wfq_result_kernel(DevScene sc, WfParams wp, int n, int obj_index, PtxHit* __restrict__ hits) {
	const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (i >= n) return;
	const float4 a0 = wp.s0[i], a2 = wp.s2[i], a3 = wp.s3[i];
	const int obj = __float_as_int(a2.w), tri = __float_as_int(a3.x);
	PtxHit h;
	memset(&h, 0, sizeof(h));
	h.hit = obj >= 0 ? 1 : 0;
	h.object = obj;
	h.prim = (obj >= 0 && tri >= 0) ? __float_as_int(sc.tri_geom[3 * tri + 2].y) : -1;
	h.t_bits = __float_as_uint(a0.w);
	if (obj >= 0) { h.bary[0] = 1.0f - a3.y - a3.z; h.bary[1] = a3.y; h.bary[2] = a3.z; }   // src/intersect.cuh:296
	hits[i] = h;
}

// reference: src/pathtracing.cu:151 acc_image[index] += make_float4(radiance), once per iteration, in order
__global__ __launch_bounds__(PT_BLOCK) void wf_accumulate_kernel(float4* __restrict__ accum, const Rad3* __restrict__ rad,
                                                                 int npix, int batch) {
	const int p = blockIdx.x * PT_BLOCK + threadIdx.x;
	if (p >= npix) return;
	float4 acc = accum[p];
	for (int s = 0; s < batch; ++s) {
		const Rad3 r = rad[(size_t)s * (size_t)npix + p];
		acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += 0.0f;
	}
	accum[p] = acc;
}
