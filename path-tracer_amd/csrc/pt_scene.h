// Device-resident scene layout (what ptx_scene_upload packs; DESIGN.md "Data layout in HBM").
//
// The reference keeps 80-B KDNode / 128-B Triangle AoS records, child boxes inside the child
// nodes and one device allocation per node for its triangle index list (reference:
// src/world.h:46-78, src/KDTree.cpp:152-163). Here, per scene, five flat arrays:
//
//   nodes      4 x float4 per node (64 B, one record = half a 128-B L2 line):
//                q0 = (L.min.x, R.min.x, L.min.y, R.min.y)
//                q1 = (L.min.z, R.min.z, L.max.x, R.max.x)
//                q2 = (L.max.y, R.max.y, L.max.z, R.max.z)
//                q3 = int4(left, right, tri_first, tri_count)
//              L/R = the CHILDREN's boxes, stored in the parent so one record serves both
//              box tests of an inner step, and interleaved so that each (left, right) pair of a
//              component arrives in one 64-bit register pair: both tests run as packed FP32; left/right are scene-global node indices (-1 = none);
//              tri_first/tri_count address tri_geom/tri_shade (leaves only). In INNER nodes q3.z/q3.w hold
//              the two child links again in packed form for the cooperative traversal kernel: >= 0 node
//              index, -1 none, and for a leaf child with 1..63 triangles 0x80000000 | block << 6 | count
//              (its triangle run itself: tri_first = 32 * block, every run starts on a multiple of 32), so
//              the kernel never fetches that leaf's node record.
//   tri_geom   3 x float4 per triangle (48 B), stored in LEAF ORDER (the KD build partitions
//              triangles, every triangle is in exactly one leaf, so leaf lists become
//              contiguous runs and the reference's trg_idxs indirection disappears):
//                g0 = (a.x, a.y, a.z, e1.x)  g1 = (e1.y, e1.z, e2.x, e2.y)
//                g2 = (e2.z, int original_prim_id, int leaf_node_local_idx, 0)
//   tri_soa    the same triangles in the same order as three float4 planes of num_tris entries
//              (a.xyz|prim id), (e1.xyz|0), (e2.xyz|0): the cooperative traversal kernel tests 32
//              consecutive triangles of a leaf with 32 lanes, so each plane read is one contiguous 512 B
//   tri_pair   every leaf run once more, padded to blocks of 32 slots (zero triangles never hit: their
//              determinant is 0) and PAIR-INTERLEAVED: block = 5 planes x 16 float4 (1280 B); plane c,
//              entry p = (X[p], X[p+16], Y[p], Y[p+16]) for components (X, Y) = (a.x, a.y), (a.z, e1.x),
//              (e1.y, e1.z), (e2.x, e2.y), (e2.z, -). A lane of the cooperative kernel loads entry p of
//              the five planes and holds triangles p and p+16 as register PAIRS, which is the operand
//              form of the packed FP32 instructions. The leaf's node record carries its first block
//              in q0.x (as int; a leaf's own q0..q2 are otherwise unused).
//   tri_grp    every leaf run a third time, for leaf mode 5 of the cooperative kernel: the run's
//   grp_box    triangles SORTED along the longest axis of the leaf's box and cut into GROUPS of eight,
//              each group with a box of its own. A ray that reaches the leaf is first tested against the
//              group boxes (a few slab tests) and only the groups it touches are tested triangle by
//              triangle: ~1.2 groups of 8 instead of a block of 32 on the bench scene. Four group slots
//              per 32-triangle block of the run, so group j of a run starting at block b is entry 4b + j
//              of both arrays; a run of n triangles uses ceil(n / 8) of them.
//                grp_box: 2 float4 per group (min.xyz, -) (max.xyz, -): the triangles' vertices, padded by
//                  5e-5 + 1e-4 x the mesh diagonal (ptx_ctx_tune "grp_pad_ppm", default 100) so that a ray
//                  the triangle test accepts does not miss its group's box through the rounding of either
//                  test (tools/grazing_stress.py: no difference from the oracle even at 0 ppm; unused
//                  entries are never consulted)
//                tri_grp: PT_GRP_BLOCK_F4 float4 per group: 5 planes x 4 float4, plane c entry l =
//                  (X[l], X[l+4], Y[l], Y[l+4]) for the tri_pair component pairs; plane 4 carries in
//                  .z/.w the triangles' slots in the UNSORTED run (int), which is what a hit reports and
//                  what breaks ties between equal t (the reference tests a leaf in run order).
//              The group boxes are stored a second time as LEAF-HALF RECORDS behind the node records of
//              `nodes` (index half_base + 2 * block + h = groups 2h, 2h + 1 of that 32-triangle block), in the
//              layout of an inner node's two child boxes, so that the unified traversal kernel tests a
//              leaf's groups with the same fetch and the same packed box test as an inner step.
//   grp_qbox   the four group boxes of a 32-triangle block a third time, in 48 B (leaf mode 9): the union U of the
//              four boxes as origin and step = extent / 255 per axis (6 floats), then per box six BYTES
//              (lo.xyz rounded down, hi.xyz rounded up, in steps from U's origin): words
//                f0 = (U.lo.x, U.lo.y, U.lo.z, step.x)  f1 = (step.y, step.z, bytes 0-3, bytes 4-7)  f2 = bytes 8-23
//              box g's bytes are 6g .. 6g+5; an unused group slot is (255, 255, 255, 0, 0, 0): never touched.
//              The decoded box contains the fp32 one (outward rounding, at most 1/255 of the block's extent per
//              side), so the filter stays conservative; three 16-B loads per leaf instead of eight.
//   cheap_pairs  for the SINGLE-LEAF meshes (sky cube, floor: intersected inline by the segment kernel, every ray against
//              every such mesh): per pair of consecutive triangles (2j, 2j + 1) of the run the box of its vertices,
//              (lo.xyz, -) (hi.xyz, -), padded like a group box. The root leaf's node record carries the mesh's first
//              pair in q0.y (int). A lane tests the boxes (wave-uniform data) and then only the pairs it touches,
//              in run order: a ray inside the sky cube tests one face instead of twelve triangles.
//   wide_nodes the 4-wide tree over the groups that the wide traversal walks (pt_wide.h): 64 B per node = the four child
//              boxes in the grp_qbox form (f0..f2), but with the 24 code bytes AXIS-major (word 2k = the four lo codes of
//              axis k, word 2k + 1 the four hi codes, child g in byte g) + int4 links (>= 0 node, 0x80000000 | group, -1 none); children
//              that are nodes come first and have consecutive indices. wide_root[object] = its root (-1: none).
//   tri_shade  4 x float4 per triangle (64 B), same order, touched once per mesh hit:
//                s0 = (na.xyz, ta.x) s1 = (nb.xyz, ta.y) s2 = (nc.xyz, tb.x)
//                s3 = (tb.y, tc.x, tc.y, int material_global_idx)
//   materials  3 x float4 per material (48 B):
//                m0 = (int type, color.rgb) m1 = (int texture, emittance.rgb)
//                m2 = (reflect_factor, refract_index, sharpness, 0)
//   objects    DevObject[num_objects] (64 B each), read with wave-uniform indices (scalar loads)
//   textures   DevTexture[num_textures] + one float4 texel array each
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_PAIR_BLOCK_F4 80   /* float4 per tri_pair block: 5 planes x 16 */
#define PT_GRP_BLOCK_F4 20    /* float4 per tri_grp group: 5 planes x 4 */

struct DevObject {
	int32_t type;
	int32_t mat_first;      // scene-global index of the object's material 0
	int32_t root_node;      // scene-global node index of the root (-1 for empty meshes)
	int32_t num_triangles;
	float radius;
	float px, py, pz;
	float bmin[3];          // root box (mesh)
	int32_t first_node;     // scene-global index of node 0 (to report local leaf ids)
	float bmax[3];
	int32_t big_slot;       // >= 0: mesh with inner nodes, traversed by the wavefront's traversal kernel
	                        //       through queue `big_slot`; -1: sphere / single-leaf mesh (tested inline)
};

struct DevTexture {
	const float4* texels;
	int32_t width, height;
};

struct DevScene {
	const DevObject* objects;
	const float4* nodes;
	const float4* tri_geom;
	const float4* tri_soa;  // the same triangles as three planes: a[], e1[], e2[] (float4 each, .w of plane 0 = prim id)
	const float4* tri_pair;
	const float4* tri_grp;
	const float4* grp_box;
	const float4* grp_qbox; // 3 float4 (48 B) per 32-triangle block: the block's four group boxes quantised to 8 bits (below)
	const float4* cheap_pairs; // 2 float4 per pair of consecutive triangles of a single-leaf mesh: its padded box (below)
	const float4* wide_nodes; // 4 float4 (64 B) per node of the 4-wide tree over the groups (below)
	const int32_t* wide_root; // per object: its wide root node, -1 if it has none
	const float4* tri_shade;
	const float4* materials;
	const DevTexture* textures;
	int32_t num_objects;
	int32_t stack_entries;  // LDS traversal-stack slots per lane (>= deepest tree level, >= 1)
	int32_t num_tris;       // triangles in tri_geom / per plane of tri_soa
	int32_t half_base;      // index of the first leaf-half record in `nodes` (behind all node records)
	int32_t wide_stack;     // most entries a traversal of a wide tree can hold on its stack
	int32_t num_materials, num_textures;
};

struct DevCamera {
	float pos[3], dir[3], up[3], right[3];
	int32_t width, height;
	float pixel_size, half_x, half_y, screen_dist, aa_jitter;
	int32_t max_bounces, preview_mode, aabb_mode, texture_enabled;
};
