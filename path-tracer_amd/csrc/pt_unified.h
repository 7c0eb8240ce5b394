// Unified traversal kernel: KD descent and the leaves' group-box tests in ONE loop.
//
// wf_traverse_coop_kernel (pt_wavefront.h) works in rounds: all lanes descend until each holds a leaf,
// then every leaf lane tests its leaf's group boxes, then the wave tests the touched groups, then the
// owners pop. Measured on the bench workload (profiles/r01_*): 33 of 64 lanes active per vector
// instruction, 21 rounds per traversal, and two thirds of the leaf visits touch no group at all yet
// still end their lane's descent for the round.
//
// Here a lane always holds ONE item: an inner node, or one HALF of a leaf = two group boxes stored as a
// record of the node array in the layout of an inner node's two child boxes (pt_scene.h, leaf-half
// records). Every iteration every lane that holds an item fetches its 64-B record (by the quad, through
// LDS, as leaf mode 6 does) and runs the same packed two-box test; what differs is only what the result
// means:
//   inner node   the reference's step (src/intersect.cuh:172-209): nearer child first, tie -> left,
//                `dist < tmin` culling, push / descend / pop
//   leaf half    every touched group becomes a work unit (owner lane, group) in the wave's LDS list;
//                after the leaf's last half the lane pops AT ONCE if the leaf produced no unit, else it
//                waits for the wave to test its units (it needs the new tmin for its next decision)
// The wave tests the listed groups sixteen per step (four lanes per group, two triangles per lane in
// packed FP32, as in leaf mode 5) whenever UMIN units have gathered or no lane can go on without it.
//
// What one ray sees is unchanged: the nodes it visits, the order, every culling decision (each is taken
// with the tmin of all leaves visited before, because a lane never passes a leaf with untested units)
// and the strict-'<' / first-in-run-order tie rule inside a leaf (64-bit key: t bits << 32 | slot in the
// unsorted run, LDS atomic minimum, merged into the lane's state only when the leaf is complete).
// A group's box is tested here with the reference's slab test (plus: exit distance >= 0), in leaf mode
// 5 with a min/max form; both only ever skip groups that cannot hold a hit (pt_scene.h, grp_box padding).
#pragma once
#include "pt_wavefront.h"

#define WF_UNI_UNIT_CAP 192                  /* >= UMIN - 1 + 128 (two units per lane and iteration) */
#define WF_UNI_WAVE_BYTES (64 * 8 + WF_UNI_UNIT_CAP * 4)   /* result keys, unit list */
#define WF_UNI_BYTES ((PT_BLOCK / 64) * WF_UNI_WAVE_BYTES)

// box_test_x2 (pt_trace.h) that also reports whether each box's exit distance is >= 0 (bits 2, 3):
// the same operations on the same values for the entry side, so an inner step decides exactly as before.
PT_DEV uint32_t box_test_x2f(const float4 q0, const float4 q1, const float4 q2, float rox, float roy, float roz,
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
		if (t6.x < a2) a2 = t6.x;
		tmin_l = a1 < 0.0f ? 0.0f : a1;
		if (!fail) bits |= 1u;
		if (!(a2 < 0.0f)) bits |= 4u;
	}
	{
		float a1 = t1.y, a2 = t2.y;
		bool fail = (a1 > t4.y) | (t3.y > a2);
		if (t3.y > a1) a1 = t3.y;
		if (t4.y < a2) a2 = t4.y;
		fail = fail | (a1 > t6.y) | (t5.y > a2);
		if (t5.y > a1) a1 = t5.y;
		if (t6.y < a2) a2 = t6.y;
		tmin_r = a1 < 0.0f ? 0.0f : a1;
		if (!fail) bits |= 2u;
		if (!(a2 < 0.0f)) bits |= 8u;
	}
	return bits;
}

template <int FLAGS, int OCC, int CHUNK, int UMIN>
__global__ __launch_bounds__(PT_BLOCK, OCC) void wf_traverse_unified_kernel(DevScene sc, WfParams wp, int obj_index, int slot_index) {
	static_assert(UMIN >= 1 && UMIN - 1 + 128 <= WF_UNI_UNIT_CAP, "the unit list must hold one more iteration's appends");
	extern __shared__ int lds_stack[];
	const int tid = threadIdx.x, lane = tid & 63;
	// counting variant only: constant-rate clock read by lane 0, summed per phase into PtxStats.diag0..3
	// (commit + refill, item steps, unit phase, -)
#define WFU_CLOCK() (((FLAGS & PT_F_STATS) && lane == 0) ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull)
	int* stack = lds_stack + tid;
	const DevObject& ob = sc.objects[obj_index];
	const int n = WF_CNT(wp, wp.bounce, 1 + slot_index);
	const int* __restrict__ queue = wp.queue[slot_index];
	const float4* __restrict__ nodes = sc.nodes;
	const float4* __restrict__ pa = sc.tri_soa;
	const float4* __restrict__ pe1 = sc.tri_soa + sc.num_tris;
	const float4* __restrict__ pe2 = sc.tri_soa + 2 * (size_t)sc.num_tris;
	const float4* __restrict__ tg = sc.tri_grp;
	const int half_base = sc.half_base;
	TraceCounters cnt = {0, 0, 0, 0, 0, 0};
	uint64_t act = 0, tot = 0, n_taken = 0, dg0 = 0, dg1 = 0, dg2 = 0, dg3 = 0;
	uint64_t x_iters = 0, x_item = 0, x_half = 0, x_wait = 0, x_usteps = 0, x_units = 0, x_uphases = 0, x_turns = 0;

	WaveFeed feed;
	feed.exhausted = false;
	feed.next = feed.hi = 0;
	int* cursor = &WF_CUR(wp, wp.bounce, 1 + slot_index);

	// this wave's LDS area behind the traversal stacks: one 64-bit result key per lane, then the unit list
	unsigned char* area = reinterpret_cast<unsigned char*>(lds_stack + sc.stack_entries * PT_BLOCK) + (tid >> 6) * WF_UNI_WAVE_BYTES;
	unsigned long long* keys = reinterpret_cast<unsigned long long*>(area);
	uint32_t* units = reinterpret_cast<uint32_t*>(area + 64 * 8);
	const unsigned long long lt = (1ull << lane) - 1ull;

	enum { IDLE = 0, ITEM = 1, WAIT = 2, DONE = 3 };
	int state = IDLE;
	bool is_leaf = false;        // kind of the item held: inner node / leaf half
	bool pend = false;           // units of this lane sit in the list, untested
	bool leaf_units = false;     // the current leaf has produced units (its key may hold a hit)
	int item = 0;                // record index in `nodes` to fetch next
	int path = 0, leaf_first = 0, leaf_block = 0, leaf_half = 0, leaf_groups = 0;
	float cur_d = 0.0f;
	int nunits = 0;              // wave-uniform
	Ray ray;
	TravState ts;
	ts.curr = -1; ts.sp = 0; ts.tri = -1; ts.tmin = 0.0f; ts.bu = ts.bv = 0.0f; ts.intersects = false;
	init_ray(ray, mk3(0.0f), mk3(1.0f));

	// take up a leaf: `count` triangles from device index `first` = block `block` of the group arrays
	auto enter_leaf = [&](int block, int first, int count, int node_for_hash) {
		leaf_block = block;
		leaf_first = first;
		leaf_groups = (count + 7) >> 3;
		leaf_half = 0;
		leaf_units = false;
		is_leaf = true;
		item = half_base + 2 * block;
		keys[lane] = ~0ull;
		if (FLAGS & PT_F_STATS) {
			cnt.n_leaf++;
			cnt.n_tri += (uint32_t)count;
			fnv_word(cnt.leaf_hash, (uint32_t)obj_index);
			fnv_word(cnt.leaf_hash, (uint32_t)node_for_hash);
		}
	};
	// follow a link of the tree: a node index, or the packed name of a leaf run (pt_scene.h)
	auto go_to = [&](int link) {
		state = ITEM;
		if (link >= 0) {
			is_leaf = false;
			item = link;
		} else {
			const int block = (int)(((uint32_t)link & 0x7fffffffu) >> 6);
			int hash_node = 0;
			if (FLAGS & PT_F_STATS) hash_node = __float_as_int(sc.tri_geom[3 * (block << 5) + 2].z);
			enter_leaf(block, block << 5, link & 63, hash_node);
		}
	};
	// the current leaf is complete and all its units are tested: fold its key into the traversal state
	auto merge_leaf = [&]() {
		const unsigned long long key = keys[lane];
		if (key != ~0ull) {
			ts.intersects = true;
			const float tt = __uint_as_float((uint32_t)(key >> 32));
			if (tt < ts.tmin) { ts.tmin = tt; ts.tri = (int)(uint32_t)key; }
		}
	};
	// the traversal of this lane's ray is over: merge into the path's closest-hit record
	auto commit = [&]() {
		if (ts.intersects && ts.tri >= 0) {
			const f3 p = ray.o + ts.tmin * ray.d;
			const float d = length3(p - ray.o);
			const int cur_obj = __float_as_int(wp.s2[path].w);
			if (d < cur_d || (d == cur_d && cur_obj >= 0 && obj_index < cur_obj)) {
				// only (t, triangle) are kept per leaf; u and v are the test's own values, taken once for the winner
				float t_, u_, v_;
				const float4 a = pa[ts.tri], e1 = pe1[ts.tri], e2 = pe2[ts.tri];
				(void)tri_test_flat(ray, mk3(a.x, a.y, a.z), mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z), t_, u_, v_);
				wp.s0[path].w = d;
				wp.s1[path].w = ts.tmin;
				wp.s2[path].w = __int_as_float(obj_index);
				store_hit(wp.s3 + path, ts.tri, u_, v_);
			}
		}
		state = IDLE;
	};

	for (;;) {
		const unsigned long long c0 = WFU_CLOCK();
		// ---- commit finished traversals and refill idle lanes, in batches: the whole wave pays for this block, so
		// it runs once WF_REFILL_MIN lanes wait for it, or when nothing else can be done ----
		const int n_free = (int)__popcll(__ballot(state == DONE || state == IDLE));
		const bool busy = __ballot(state == ITEM || state == WAIT) != 0ull;
		const bool turn = !busy || (n_free >= WF_REFILL_MIN && (!feed.exhausted || __ballot(state == DONE) != 0ull));
		if (turn) {
			if (FLAGS & PT_F_STATS) { if (lane == 0) x_turns++; }
			if (state == DONE) commit();
			for (int attempt = 0; attempt < 2; ++attempt) {
				const unsigned long long idle = __ballot(state == IDLE);
				if (idle == 0ull) break;
				if (feed.next >= feed.hi) {
					if (feed.exhausted) break;
					feed_claim(feed, cursor, n, CHUNK);
					if (feed.exhausted) break;
				}
				const int mine = feed.next + (int)__popcll(idle & lt);
				if (state == IDLE && mine < feed.hi) {
					if (FLAGS & PT_F_STATS) n_taken++;
					path = queue[mine];
					const float4 a0 = wp.s0[path], a1 = wp.s1[path];
					init_ray(ray, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z));
					cur_d = a0.w;
					trav_begin(ts, ob);
					pend = false;
					go_to(ts.curr);
				}
				feed.next += (int)__popcll(idle);
			}
			if (__ballot(state != IDLE) == 0ull) break;   // queue drained, nothing in flight
		}
		const unsigned long long c1 = WFU_CLOCK();
		dg0 += c1 - c0;

		// ---- item step: every lane that holds an item fetches its record and tests its two boxes ----
		// Written as selects wherever no memory is touched: the lanes of a wave hold inner nodes and leaf halves
		// side by side, and every divergent branch would be run by the whole wave.
		bool on0 = false, on1 = false;   // leaf halves: this lane's groups 2h / 2h + 1 are touched
		int g0 = 0;
		const bool resume = state == WAIT && !pend;   // the units of this lane's leaf were tested in the last unit phase
		if (__ballot(state == ITEM || resume) != 0ull) {
			if (FLAGS & PT_F_STATS) {
				if (lane == 0) { tot += 64; x_iters++; }
				if (state == ITEM) { x_item++; act++; if (is_leaf) x_half++; }
				if (state == WAIT && pend) x_wait++;
			}
			const int NO_LINK = (int)0x80000000;   // never a link: packed leaf links carry a count of 1..63 in their low bits
			int link = NO_LINK;
			bool do_pop = resume, fin_leaf = resume;
			if (state == ITEM) {
				const float4* rp = nodes + 4 * (size_t)item;
				const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];   // leaf halves carry zeros in q3
				float left_dist, right_dist;
				const uint32_t bb = box_test_x2f(q0, q1, q2, ray.o.x, ray.o.y, ray.o.z, ray.inv.x, ray.inv.y, ray.inv.z, ray.sx, ray.sy, ray.sz,
				                                 left_dist, right_dist);
				const bool leaf = is_leaf;
				const bool plain_leaf = !leaf && __float_as_int(q3.x) == -1 && __float_as_int(q3.y) == -1;
				const bool inner = !leaf & !plain_leaf;
				// inner node: the reference's step (src/intersect.cuh:172-209)
				const int left = __float_as_int(q3.z), right = __float_as_int(q3.w);   // packed links: pt_scene.h
				if (FLAGS & PT_F_STATS) { if (inner) { cnt.n_inner++; cnt.n_box += (left != -1 ? 1u : 0u) + (right != -1 ? 1u : 0u); } }
				const bool left_res = inner & (left != -1) & ((bb & 1u) != 0u) & (left_dist < ts.tmin);
				const bool right_res = inner & (right != -1) & ((bb & 2u) != 0u) & (right_dist < ts.tmin);
				const bool both = left_res & right_res;
				const bool right_first = left_dist > right_dist;   // nearer child first, tie -> left
				const int near_link = right_first ? right : left, far_link = right_first ? left : right;
				if (both) stack[ts.sp * PT_BLOCK] = far_link;
				ts.sp += both ? 1 : 0;
				if (left_res | right_res) link = both ? near_link : (left_res ? left : right);
				// leaf half: groups 2h and 2h + 1 of the current leaf; the list appends follow below, in uniform control flow
				g0 = 2 * leaf_half;
				on0 = leaf & ((bb & 5u) == 5u) & (g0 < leaf_groups);
				on1 = leaf & ((bb & 10u) == 10u) & (g0 + 1 < leaf_groups);
				pend = pend | on0 | on1;
				leaf_units = leaf_units | on0 | on1;
				leaf_half += leaf ? 1 : 0;
				item += leaf ? 1 : 0;
				const bool leaf_end = leaf & (2 * leaf_half >= leaf_groups);
				if (leaf_end & pend) state = WAIT;           // its units must be tested before its next decision
				fin_leaf = leaf_end & !pend;                 // nothing untested: the lane goes on at once
				do_pop = fin_leaf | (inner & !(left_res | right_res));
				if (plain_leaf) {
					// a leaf reached through a plain link (more than 63 triangles, or none): rare
					const int count = __float_as_int(q3.w);
					enter_leaf(__float_as_int(q0.x), __float_as_int(q3.z), count, item - ob.first_node);
					if (count <= 0) do_pop = true;
				}
			}
			// one tail for all: fold a finished leaf, pop, follow the link
			if (fin_leaf & leaf_units) merge_leaf();
			if (do_pop) {
				if (ts.sp == 0) state = DONE;
				else { --ts.sp; link = stack[ts.sp * PT_BLOCK]; }
			}
			if (link != NO_LINK) go_to(link);
			// append the touched groups to the wave's list (ballot + popcount prefix, no atomics)
			const unsigned long long m0 = __ballot(on0), m1 = __ballot(on1);
			if ((m0 | m1) != 0ull) {
				const int base1 = nunits + (int)__popcll(m0);
				if (on0) units[nunits + (int)__popcll(m0 & lt)] = (uint32_t)lane | ((uint32_t)g0 << 6);
				if (on1) units[base1 + (int)__popcll(m1 & lt)] = (uint32_t)lane | ((uint32_t)(g0 + 1) << 6);
				nunits = base1 + (int)__popcll(m1);
			}
		}
		const unsigned long long c2 = WFU_CLOCK();
		dg1 += c2 - c1;

		// ---- unit phase: the wave tests the listed groups, 16 per step, once enough have gathered or no
		// lane can go on without it (every lane that could is waiting for its units) ----
		if (nunits >= UMIN || (nunits > 0 && __ballot(state == ITEM) == 0ull)) {
			__builtin_amdgcn_wave_barrier();
			if (FLAGS & PT_F_STATS) { if (lane == 0) { x_uphases++; x_units += (uint64_t)nunits; } }
			for (int s16 = 0; s16 < nunits; s16 += 16) {
				const int ui = s16 + (lane >> 2);
				const bool on = ui < nunits;
				const uint32_t un = units[on ? ui : s16];
				const int owner = (int)(un & 63u), grp = (int)(un >> 6);
				// the owner's ray and leaf come straight out of its registers
				const f3 bo = mk3(__shfl(ray.o.x, owner, 64), __shfl(ray.o.y, owner, 64), __shfl(ray.o.z, owner, 64));
				const f3 bd = mk3(__shfl(ray.d.x, owner, 64), __shfl(ray.d.y, owner, 64), __shfl(ray.d.z, owner, 64));
				const int oblock = __shfl(leaf_block, owner, 64);
				if (FLAGS & PT_F_STATS) { if (lane == 0) { tot += 128; x_usteps++; } }
				if (on) {   // uniform per quad
					const float4* bp = tg + ((size_t)(4 * oblock + grp) * PT_GRP_BLOCK_F4 + (lane & 3));
					const float4 c0 = bp[0], c1 = bp[4], c2 = bp[8], c3 = bp[12], c4 = bp[16];
					f3x2 va, e1, e2;
					va.x = (v2f){c0.x, c0.y}; va.y = (v2f){c0.z, c0.w}; va.z = (v2f){c1.x, c1.y};
					e1.x = (v2f){c1.z, c1.w}; e1.y = (v2f){c2.x, c2.y}; e1.z = (v2f){c2.z, c2.w};
					e2.x = (v2f){c3.x, c3.y}; e2.y = (v2f){c3.z, c3.w}; e2.z = (v2f){c4.x, c4.y};
					v2f t, u, v;
					const uint32_t hb = tri_test_flat_x2(bo, bd, va, e1, e2, t, u, v);
					if (FLAGS & PT_F_STATS) act += 2;
					if (__ballot(hb != 0u) != 0ull) {
						// a hit has t > 5e-8, so its bit pattern orders like its value; 0xffffffff = no hit
						const uint32_t k0 = (hb & 1u) ? __float_as_uint(t.x) : 0xffffffffu;
						const uint32_t k1 = (hb & 2u) ? __float_as_uint(t.y) : 0xffffffffu;
						const uint32_t k = quad_min_u32(k0 < k1 ? k0 : k1);
						const uint32_t s0 = __float_as_uint(c4.z), s1 = __float_as_uint(c4.w);   // slots in the unsorted run
						const uint32_t cand = quad_min_u32(umin3(k0 == k ? s0 : 0xffffffffu, k1 == k ? s1 : 0xffffffffu, 0xffffffffu));
						if ((lane & 3) == 0 && k != 0xffffffffu)
							atomicMin(&keys[owner], ((unsigned long long)k << 32) | (unsigned long long)cand);
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			nunits = 0;
			pend = false;   // the lanes that waited for these units resume in the next item step
		}
		dg2 += WFU_CLOCK() - c2;
	}
	stats_flush(wp.stats, cnt, 0, 0, 0, 0, 0, act, tot, (FLAGS & PT_F_STATS) != 0, n_taken, true);
	if (FLAGS & PT_F_STATS) {
		diag_flush(wp.stats, dg0, dg1, dg2, dg3);
		exec_flush(wp.stats, x_iters, x_item, x_half, x_wait, x_usteps, x_units, x_uphases, x_turns);
	}
#undef WFU_CLOCK
}
