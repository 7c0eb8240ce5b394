// Wide traversal: the rays queued for a multi-node mesh walk a 4-WIDE tree over the groups of eight triangles
// (pt_scene.h, wide_nodes) instead of the reference's binary KD tree, and the triangle tests run ASYNCHRONOUSLY
// to the descent.
//
// Why. The cooperative kernel (pt_wavefront.h) follows the reference's tree node by node: 51 inner steps and 12
// leaf visits per traversal, every one a dependent round trip through the vector-memory path, which is what that
// kernel pays for (texture addresser 76 % busy, 33 of 64 lanes active). The RESULT of a traversal,
// however, does not depend on the tree: rayIntersectsKDNodeLOOP (reference: src/intersect.cuh:119-212) returns the
// smallest t over the triangles of all leaves it visits, and it visits every leaf that can hold a closer hit -
// a box is only skipped when the ray misses it or enters it beyond the best t so far, and every box holds its
// triangles with padding far above the rounding of either test (the argument, and the stress test, of the group
// boxes: pt_wavefront.h, tools/grazing_stress.py). So the closest hit is the minimum over ALL triangles that pass
// the reference's triangle test, and any conservative hierarchy finds it. What does depend on the tree is which
// of two triangles with the SAME t (bit for bit) wins: the reference keeps the first it tests. Inside a leaf that
// is run order, which the 64-bit keys reproduce (t bits << 32 | device index: run order); across leaves it is the
// reference's visiting order. A ray whose best t is reached by triangles of two different 32-slot blocks is
// therefore NOT decided here: it goes to the retry queue and is traversed again, from its untouched state, by the
// strict kernel (wf_traverse_coop_kernel with wp.use_retry). The same exit serves rays whose stack would overflow.
//
// The argument above needs the hit POINT of every accepted triangle to lie inside the boxes that hold the triangle.
// At grazing incidence (cos ~ 1e-4) the reference's t = f * dot(e2, q) loses most of its digits (f = 1 / det is huge):
// tools/grazing_stress.py finds triangles the reference "hits" 0.005 ... 0.25 units in FRONT of where the ray meets
// them, i.e. in front of their own boxes. Such a hit can beat a well-conditioned one although its box lies behind it,
// and one traversal may test the triangle where the other has culled its box by distance. Two measures:
//   * distance culling keeps a margin: a box is dropped only if the ray enters it beyond best t x cull_rel + cull_abs,
//     so triangles just behind the best hit are still tested;
//   * a WINNING hit whose point is not inside its triangle's own (slightly widened) vertex box sends its ray to the strict
//     kernel (checked at the commit, where the winner's record is read once more for its u, v).
// With both, the grazing stress (400 000 rays per scene) shows no difference from the oracle; without them 3 rays of the
// car and 2 of the triangle soup differed.
// WHAT REMAINS (the one documented exception to bit-identity, fenced by tests/test_gpu_wide.py): a triangle this kernel
// never tests cannot win here. At 2 000 000 adversarial rays per scene 1 (car) + 2 (soup) rays still lose a noise hit to the
// distance culling, and at 19.7 million on the car 7 lose one because the ray MISSES the triangle's group box altogether
// (float u, v inside the triangle, double u, v outside by 0.02 ... 0.37: tests/golden/grazing_hard_rays.npz). 0 pixels of
// the headline frame at 4000 spp. trav_variant 15 (every triangle of every leaf the reference visits) has no such case.
//
// Structure (wave-uniform control flow):
//   * one ray per lane walks wide nodes: one 64-B record = four child boxes in 8 bits relative to their union
//     + four links (node index | 0x80000000 + group | -1). Children that are nodes are visited nearest first
//     (sorted by entry distance; the others go on the stack as ONE entry per node with the distance of the nearest of
//     them, so that children behind the best hit are dropped at the pop without a fetch); children that are GROUPS become work units (owner lane, group)
//     in the wave's list.
//   * the lane does not wait for its units. Whenever >= DRAIN_MIN units have gathered the wave tests sixteen of
//     them per step (four lanes per unit, two triangles per lane in packed FP32, the owner's ray by bpermute) and
//     merges the hits into the owners' keys with LDS atomic minima; the owners then lower their culling bound.
//   * once TURN_MIN lanes have finished, the remaining units are tested, finished lanes commit (closest object by
//     distance, as the reference's loop over the objects: src/intersect.cuh:328-346) and idle lanes take new rays.
// Default configuration <6 waves/SIMD, chunks of 256, turn at 24 waiting lanes, drain at 16 units>: 80 registers.
#pragma once
#include "pt_wavefront.h"

#define WFW_UNIT_CAP 320
#ifndef WFW_CLAIMS_MIN
#define WFW_CLAIMS_MIN 8   /* a short queue is cut so that every wave of the grid gets this many claims (1 / 4 / 8: 16-sample pass 10.00 / 9.87 / 9.83 ms, one-sample pass 2.31 / 2.23 / 2.23; a full pass does not notice) */
#endif
#define WFW_WAVE_BYTES (64 * 8 + 64 * 4 + WFW_UNIT_CAP * 4)   /* keys, flags, units */
#define WFW_BYTES ((PT_BLOCK / 64) * WFW_WAVE_BYTES)
#define WFW_STACK_ENTRY_BYTES 6                                         /* link (4) + entry distance as bf16 (2) */
#define WFW_SCAP_MAX 10   /* 10 x 6 B x 256 lanes + the waves' areas = 25.6 KB per workgroup: six workgroups per CU */

template <int FLAGS, int OCC, int CHUNK, int TURN_MIN, int DRAIN_MIN>
__global__ __launch_bounds__(PT_BLOCK, OCC) void wf_traverse_wide_kernel(DevScene sc, WfParams wp, int obj_index, int slot_index, int scap, float cull_rel, float cull_abs) {
	static_assert(CHUNK >= 64, "a refill takes up to 64 entries of the chunk at a time");
	static_assert(DRAIN_MIN >= 16 && DRAIN_MIN <= 64 && 63 + 256 < WFW_UNIT_CAP, "the unit list must hold one more iteration's appends");
	extern __shared__ int lds_stack[];
	const int tid = threadIdx.x, lane = tid & 63;
	int* stack = lds_stack + tid;                                                              // entry i at stack[i * PT_BLOCK]
	unsigned short* sdist = reinterpret_cast<unsigned short*>(lds_stack + scap * PT_BLOCK) + tid;   // entry i at sdist[i * PT_BLOCK]
	unsigned char* area = reinterpret_cast<unsigned char*>(lds_stack) + (size_t)scap * PT_BLOCK * WFW_STACK_ENTRY_BYTES + (tid >> 6) * WFW_WAVE_BYTES;
	unsigned long long* keys = reinterpret_cast<unsigned long long*>(area);      // per lane: (t bits << 32 | triangle) of its best hit
	uint32_t* flags = reinterpret_cast<uint32_t*>(area + 64 * 8);                // != 0: the best t was reached from two blocks
	uint32_t* units = reinterpret_cast<uint32_t*>(area + 64 * 12);               // owner lane | group << 6
	const int n = WF_CNT(wp, wp.bounce, 1 + slot_index);
	const int* __restrict__ queue = wp.queue[slot_index];
	const float4* __restrict__ wn = sc.wide_nodes;
	const float4* __restrict__ tg = sc.tri_grp;
	const int root = sc.wide_root[obj_index];
	const unsigned long long lt = (1ull << lane) - 1ull;
	uint64_t n_taken = 0, tot = 0;
	// PtxStats.x_*: visit iterations, lanes visiting in them, rays sent to the retry queue, stacked nodes dropped at the
	// pop, unit steps, units, drains, commit / refill turns
	uint64_t x_iters = 0, x_item = 0, x_retry = 0, x_drop = 0, x_usteps = 0, x_units = 0, x_drains = 0, x_turns = 0;
	// counting variant only: constant-rate clock read by lane 0, summed per phase into PtxStats.diag0..2 (turns without their
	// unit steps, node visits, unit steps); diag3 = rays retried because their stack was full (the others: equal t)
#define WFW_CLOCK() (((FLAGS & PT_F_STATS) && lane == 0) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
	uint64_t dg_turn = 0, dg_visit = 0, dg_unit = 0, dg_over = 0, x_fragile = 0;

	WaveFeed feed;
	feed.exhausted = false;
	feed.next = feed.hi = 0;
	int* cursor = &WF_CUR(wp, wp.bounce, 1 + slot_index);
	// SHORT QUEUES. The reference's own loop (runCUDA: qt/cudarunthread.cpp:21-41) renders ONE sample per call: a few hundred
	// thousand rays per launch, fewer than the grid has lanes. A wave that claimed CHUNK of them would walk them four deep while
	// four waves in five found the queue drained (0.40 ms per launch, eight times the full-queue cost per ray). So a claim is no
	// larger than 1 / WFW_CLAIMS_MIN of the queue's share of one wave, in whole wave-widths (the shorter passes of the look-ahead
	// windows, ptx_render_sample_ahead, balance better for it too); long queues keep CHUNK. Only the schedule depends on it.
	int chunk = CHUNK;
	{
		const int nwaves = (int)gridDim.x * (PT_BLOCK / 64);
		if (n < nwaves * CHUNK * WFW_CLAIMS_MIN) {
			const int share = (int)__builtin_amdgcn_readfirstlane((n + nwaves * WFW_CLAIMS_MIN - 1) / (nwaves * WFW_CLAIMS_MIN));
			chunk = (share + 63) & ~63;
			chunk = chunk < 64 ? 64 : chunk > CHUNK ? CHUNK : chunk;
		}
	}

	enum { IDLE = 0, DESCEND = 1, DONE = 2 };
	int state = IDLE;
	int path = 0, curr = 0, sp = 0, cur_obj_r = -1;
	bool redo = false;
	float cur_d = 0.0f, tmin = 0.0f;   // tmin: the CULLING bound = best t so far x cull_rel + cull_abs (see the header comment)
	f3 o = mk3(0.0f), d = mk3(1.0f), inv = mk3(1.0f);
	int nunits = 0;   // wave-uniform

	// the wave tests listed units, sixteen per step, newest first; all of them, or while full steps remain
	auto unit_steps = [&](bool all) {
		__builtin_amdgcn_wave_barrier();
		const unsigned long long cu0 = WFW_CLOCK();
		if (FLAGS & PT_F_STATS) { if (lane == 0 && nunits > 0) x_drains++; }
		while (nunits >= 16 || (all && nunits > 0)) {
			const int take = nunits >= 16 ? 16 : nunits;
			const int base = nunits - take;
			const bool on = (lane >> 2) < take;
			const uint32_t un = units[on ? base + (lane >> 2) : base];
			const int owner = (int)(un & 63u);
			const uint32_t grp = un >> 6;
			const f3 bo = mk3(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
			const f3 bd = mk3(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
			if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 128; x_usteps++; x_units += (uint64_t)take; } }
			if (on) {   // uniform per quad
				// 32-bit byte offset from the (wave-uniform) base: ptx_scene_upload refuses scenes whose groups pass 4 GB
				const float4* bp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tg) + (grp * (uint32_t)(PT_GRP_BLOCK_F4 * 16) + (uint32_t)((lane & 3) << 4)));
				const float4 c0 = bp[0], c1 = bp[4], c2 = bp[8], c3 = bp[12], c4 = bp[16];
				f3x2 va, e1, e2;
				va.x = (v2f){c0.x, c0.y}; va.y = (v2f){c0.z, c0.w}; va.z = (v2f){c1.x, c1.y};
				e1.x = (v2f){c1.z, c1.w}; e1.y = (v2f){c2.x, c2.y}; e1.z = (v2f){c2.z, c2.w};
				e2.x = (v2f){c3.x, c3.y}; e2.y = (v2f){c3.z, c3.w}; e2.z = (v2f){c4.x, c4.y};
				v2f t, u, v;
				const uint32_t hb = tri_test_flat_x2(bo, bd, va, e1, e2, t, u, v);
				if (__ballot(hb != 0u) != 0ull) {
					// a hit has t > 5e-8, so its bit pattern orders like its value; 0xffffffff = no hit
					const uint32_t k0 = (hb & 1u) ? __float_as_uint(t.x) : 0xffffffffu;
					const uint32_t k1 = (hb & 2u) ? __float_as_uint(t.y) : 0xffffffffu;
					const uint32_t k = quad_min_u32(k0 < k1 ? k0 : k1);
					const uint32_t s0 = __float_as_uint(c4.z), s1 = __float_as_uint(c4.w);   // device indices: run order inside a leaf
					const uint32_t cand = quad_min_u32(umin3(k0 == k ? s0 : 0xffffffffu, k1 == k ? s1 : 0xffffffffu, 0xffffffffu));
					const bool win0 = (k0 == k) & (s0 == cand), win1 = (k1 == k) & (s1 == cand);
					if (k != 0xffffffffu && (win0 | win1)) {
						const unsigned long long key = ((unsigned long long)k << 32) | (unsigned long long)cand;
						const unsigned long long old = atomicMin(&keys[owner], key);
						// the same t from another 32-slot block (= possibly another leaf): the reference's visiting
						// order decides, which this kernel does not know
						if ((uint32_t)(old >> 32) == k && (((uint32_t)old ^ cand) >> 5) != 0u) flags[owner] = 1u;
					}
				}
			}
			nunits = base;
		}
		__builtin_amdgcn_wave_barrier();
		// owners lower their culling bound: the best t so far plus the margin (a NaN pattern = no hit yet leaves it)
		if (state == DESCEND) tmin = __builtin_fminf(tmin, __builtin_fmaf(__uint_as_float(reinterpret_cast<const uint32_t*>(keys)[2 * lane + 1]), cull_rel, cull_abs));
		dg_unit += WFW_CLOCK() - cu0;
	};

	for (;;) {
		const unsigned long long ct0 = WFW_CLOCK();
		const uint64_t du0 = dg_unit;
		const int n_desc = (int)__popcll(__ballot(state == DESCEND));
		if (n_desc == 0 || (64 - n_desc >= TURN_MIN && !feed.exhausted)) {
			// ---- turn: all listed units are tested, finished traversals commit, idle lanes take new rays ----
			if (FLAGS & PT_F_STATS) { if (lane == 0) x_turns++; }
			unit_steps(true);
			bool again = false;
			if (state == DONE) {
				const unsigned long long key = keys[lane];
				again = redo || flags[lane] != 0u;
				if (!again && key != ~0ull) {
					const float t = __uint_as_float((uint32_t)(key >> 32));
					if (t < __builtin_huge_valf()) {
						// The winner once more, from its plain record: (u, v) are the test's own values (same operations,
						// same bits as in the unit step), and the vertices decide whether the hit is ROBUST: its point o + t d
						// must lie in the box of the triangle's own vertices, widened by less than any tree's padding. Then
						// every box that holds the triangle, here and in the reference's tree, holds the point, and both
						// traversals test the triangle unless it lies beyond their best hit anyway. A winner that is not (t off
						// through cancellation at grazing incidence) may have been culled by the reference: strict kernel.
						const int tri = (int)(uint32_t)key;
						const float4 a4 = sc.tri_soa[tri], f4 = sc.tri_soa[sc.num_tris + tri], g4 = sc.tri_soa[2 * (size_t)sc.num_tris + tri];
						Ray rr;
						rr.o = o; rr.d = d; rr.inv = inv; rr.sx = rr.sy = rr.sz = false;
						float t_, u_, v_;
						(void)tri_test_flat(rr, mk3(a4.x, a4.y, a4.z), mk3(f4.x, f4.y, f4.z), mk3(g4.x, g4.y, g4.z), t_, u_, v_);
						bool bad = false;
#define WFW_AXIS(C) { \
						const float p_ = __builtin_fmaf(t, d.C, o.C); \
						const float rho_ = __builtin_fmaf(4.8e-7f, __builtin_fabsf(p_), 3e-5f); \
						const float lo_ = a4.C + __builtin_fminf(0.0f, __builtin_fminf(f4.C, g4.C)), hi_ = a4.C + __builtin_fmaxf(0.0f, __builtin_fmaxf(f4.C, g4.C)); \
						bad |= !(p_ >= lo_ - rho_ && p_ <= hi_ + rho_); }
						WFW_AXIS(x) WFW_AXIS(y) WFW_AXIS(z)
#undef WFW_AXIS
						const f3 p = o + t * d;
						const float dist = length3(p - o);
						if (bad) {
							again = true;
							if (FLAGS & PT_F_STATS) x_fragile++;
						} else if (dist < cur_d || (dist == cur_d && cur_obj_r >= 0 && obj_index < cur_obj_r)) {
							wp.s0[path].w = dist;
							wp.s1[path].w = t;
							wp.s2[path].w = __int_as_float(obj_index);
							store_hit(wp.s3 + path, tri, u_, v_);
						}
					}
				}
				state = IDLE;
			}
			{
				const int slot = wave_append(&WF_RCNT(wp, wp.bounce, 1 + slot_index), again);
				if (again) { wp.retry[slot] = path; if (FLAGS & PT_F_STATS) { x_retry++; if (redo) dg_over++; } }
			}
			// Refill: idle lanes take the next entries of the wave's chunk. (Splitting it - queue entries fetched when the chunk is
			// claimed, the ray's loads issued here and taken up after the node visit - gained nothing at 5 waves/SIMD and its 11
			// registers cost the sixth wave, which is worth 4 %.)
			for (int attempt = 0; attempt < 2; ++attempt) {
				const unsigned long long idle = __ballot(state == IDLE);
				if (idle == 0ull) break;
				if (feed.next >= feed.hi) {
					if (feed.exhausted) break;
					feed_claim(feed, cursor, n, chunk);
					if (feed.exhausted) break;
				}
				const int mine = feed.next + (int)__popcll(idle & lt);
				if (state == IDLE && mine < feed.hi) {
					if (FLAGS & PT_F_STATS) n_taken++;
					path = queue[mine];
					const float4 a0 = wp.s0[path], a1 = wp.s1[path];
					cur_obj_r = __float_as_int(wp.s2[path].w);
					o = mk3(a0.x, a0.y, a0.z);
					d = mk3(a1.x, a1.y, a1.z);
					inv = pt_rcp3(d);
					cur_d = a0.w;
					tmin = __builtin_huge_valf();
					curr = root;
					sp = 0;
					redo = false;
					keys[lane] = ~0ull;
					flags[lane] = 0u;
					state = DESCEND;
				}
				feed.next += (int)__popcll(idle);
			}
			if (__ballot(state == DESCEND) == 0ull) break;   // queue drained, nothing in flight
		}

		const unsigned long long ct1 = WFW_CLOCK();
		dg_turn += (ct1 - ct0) - (dg_unit - du0);
		const uint64_t du1 = dg_unit;
		// ---- one wide node per descending lane ----
		if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 64; x_iters++; } if (state == DESCEND) x_item++; }
		bool g0 = false, g1 = false, g2 = false, g3 = false;   // child k is a group and the ray touches its box
		int gref0 = 0, gref1 = 0, gref2 = 0, gref3 = 0;
		if (state == DESCEND) {
			const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wn) + ((uint32_t)curr << 6));   // < 2^24 records: a 32-bit byte offset
			const float4 f0 = rec[0], f1 = rec[1], f2 = rec[2], f3 = rec[3];
			const float ax = (f0.x - o.x) * inv.x, ay = (f0.y - o.y) * inv.y, az = (f0.z - o.z) * inv.z;
			const float sx = f0.w * inv.x, sy = f1.x * inv.y, sz = f1.y * inv.z;
			const uint32_t w0 = __float_as_uint(f1.z), w1 = __float_as_uint(f1.w), w2 = __float_as_uint(f2.x),
			               w3 = __float_as_uint(f2.y), w4 = __float_as_uint(f2.z), w5 = __float_as_uint(f2.w);
			const int r0 = __float_as_int(f3.x), r1 = __float_as_int(f3.y), r2 = __float_as_int(f3.z), r3 = __float_as_int(f3.w);
			gref0 = r0; gref1 = r1; gref2 = r2; gref3 = r3;
			// The six code words are axis-major (ptx_hip.hip, quantise_boxes4): the ray's NEAR face of every child on an axis is
			// the lo word if it travels upwards there and the hi word otherwise, so the near / far choice is one swap of two
			// words per axis and the per-child work is six conversions, six fma, one max3, one min3. (min / max pairs per
			// child and axis cost 24 instructions of the half-rate kind; tools/micro/valu_rate.hip.) fma is monotonic in
			// the code, so near <= far as before; a NaN (0 * inf: the ray runs in a face's plane) drops out of max3 / min3
			// as it did out of min / max: that axis does not constrain.
			const bool ngx = sx < 0.0f, ngy = sy < 0.0f, ngz = sz < 0.0f;
			const uint32_t nx = ngx ? w1 : w0, fx = ngx ? w0 : w1, ny = ngy ? w3 : w2, fy = ngy ? w2 : w3, nz = ngz ? w5 : w4, fz = ngz ? w4 : w5;
#define WFW_QB(W, K) ((float)(((W) >> (8 * (K))) & 0xffu))
			// entry distance out (clamped at the origin); hit = the slabs overlap in front of the origin, not beyond the bound
			auto hit = [&](float qnx, float qny, float qnz, float qfx, float qfy, float qfz, float& tn) {
				const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(qnx, sx, ax), __builtin_fmaf(qny, sy, ay)), __builtin_fmaf(qnz, sz, az));
				const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaf(qfx, sx, ax), __builtin_fmaf(qfy, sy, ay)), __builtin_fmaf(qfz, sz, az));
				tn = __builtin_fmaxf(tnear, 0.0f);
				return !(tn > tfar) && !(tn > tmin);
			};
			float n0, n1, n2, n3;
			const bool h0 = hit(WFW_QB(nx, 0), WFW_QB(ny, 0), WFW_QB(nz, 0), WFW_QB(fx, 0), WFW_QB(fy, 0), WFW_QB(fz, 0), n0) & (r0 != -1);
			const bool h1 = hit(WFW_QB(nx, 1), WFW_QB(ny, 1), WFW_QB(nz, 1), WFW_QB(fx, 1), WFW_QB(fy, 1), WFW_QB(fz, 1), n1) & (r1 != -1);
			const bool h2 = hit(WFW_QB(nx, 2), WFW_QB(ny, 2), WFW_QB(nz, 2), WFW_QB(fx, 2), WFW_QB(fy, 2), WFW_QB(fz, 2), n2) & (r2 != -1);
			const bool h3 = hit(WFW_QB(nx, 3), WFW_QB(ny, 3), WFW_QB(nz, 3), WFW_QB(fx, 3), WFW_QB(fy, 3), WFW_QB(fz, 3), n3) & (r3 != -1);
#undef WFW_QB
			g0 = h0 & (r0 < 0); g1 = h1 & (r1 < 0); g2 = h2 & (r2 < 0); g3 = h3 & (r3 < 0);
			// inner children, nearest first: sort (entry distance with the slot in its two lowest bits)
			auto skey = [](bool on, float tn, uint32_t slot) {
				return on ? ((__float_as_uint(tn) & ~3u) | slot) : 0xffffffffu;
			};
			uint32_t k0 = skey(h0 & (r0 >= 0), n0, 0u), k1 = skey(h1 & (r1 >= 0), n1, 1u), k2 = skey(h2 & (r2 >= 0), n2, 2u), k3 = skey(h3 & (r3 >= 0), n3, 3u);
#define WFW_CX(A, B) { const uint32_t lo_ = A < B ? A : B, hi_ = A < B ? B : A; A = lo_; B = hi_; }
			WFW_CX(k0, k1) WFW_CX(k2, k3) WFW_CX(k0, k2) WFW_CX(k1, k3) WFW_CX(k1, k2)
#undef WFW_CX
			// children that are nodes sit in slots 0.. with consecutive indices (ptx_scene_upload): link of slot s = r0 + s
			if (k0 != 0xffffffffu) {
				curr = r0 + (int)(k0 & 3u);
				const uint32_t more = (k1 != 0xffffffffu ? 1u : 0u) + (k2 != 0xffffffffu ? 1u : 0u) + (k3 != 0xffffffffu ? 1u : 0u);
				if (more != 0u) {
					if (sp >= scap) {
						redo = true;          // the stack is full: the strict kernel takes this ray
						state = DONE;
					} else {
						// ONE entry for the node: its inner children sit in slots 0.. with consecutive indices, so the
						// entry is (index of slot 0) << 8 | the slots still to visit, nearest first, 2 bits each | their
						// number; beside it the entry distance of the nearest of them, rounded DOWN to bf16
						stack[sp * PT_BLOCK] = (int)(((uint32_t)r0 << 8) | ((k3 & 3u) << 6) | ((k2 & 3u) << 4) | ((k1 & 3u) << 2) | more);
						sdist[sp * PT_BLOCK] = (unsigned short)(k1 >> 16);
						++sp;
					}
				}
			} else {
				// pop the nearest child still to visit of the newest entry. An entry whose nearest remaining child the ray
				// enters beyond the best hit so far is dropped whole (the others are farther; the distance was rounded
				// down and is kept as a lower bound for them)
				state = DONE;
				while (sp > 0) {
					const uint32_t e = (uint32_t)stack[(sp - 1) * PT_BLOCK];
					const float dn = __uint_as_float((uint32_t)sdist[(sp - 1) * PT_BLOCK] << 16);
					const uint32_t left = e & 3u;
					if (dn > tmin) { --sp; if (FLAGS & PT_F_STATS) x_drop += left; continue; }
					curr = (int)((e >> 8) + ((e >> 2) & 3u));
					if (left == 1u) --sp;
					else stack[(sp - 1) * PT_BLOCK] = (int)((e & ~0xffu) | (((e >> 4) & 0xfu) << 2) | (left - 1u));
					state = DESCEND;
					break;
				}
			}
		}
		// append the touched groups to the wave's list (ballot + popcount prefix, no atomics)
#define WFW_APPEND(ON, REF) { \
			const bool on_ = (ON); \
			const unsigned long long m_ = __ballot(on_); \
			if (on_) units[nunits + (int)__popcll(m_ & lt)] = (uint32_t)lane | (((uint32_t)(REF) & 0x03ffffffu) << 6); \
			nunits += (int)__popcll(m_); }
		WFW_APPEND(g0, gref0) WFW_APPEND(g1, gref1) WFW_APPEND(g2, gref2) WFW_APPEND(g3, gref3)
#undef WFW_APPEND
		if (nunits >= DRAIN_MIN) unit_steps(false);
		dg_visit += (WFW_CLOCK() - ct1) - (dg_unit - du1);
	}
	// lane_steps_active of this kernel = rays sent to the strict kernel because their winning hit was not robust
	stats_flush(wp.stats, TraceCounters{0, 0, 0, 0, 0, 0}, 0, 0, 0, 0, 0, x_fragile, tot, (FLAGS & PT_F_STATS) != 0, n_taken, true);
	if (FLAGS & PT_F_STATS) {
		exec_flush(wp.stats, x_iters, x_item, x_retry, x_drop, x_usteps, x_units, x_drains, x_turns);
		diag_flush(wp.stats, dg_turn, dg_visit, dg_unit, dg_over);
	}
#undef WFW_CLOCK
}
