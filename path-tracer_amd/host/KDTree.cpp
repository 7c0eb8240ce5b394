// KD-tree build over triangle centroids (reference: src/KDTree.cpp:18-202).
//
// The tree decides which leaves a ray visits and in which order triangles are
// tested, so it is reproduced exactly (compile with -ffp-contract=off):
//   * node index = preorder creation counter                     (KDTree.cpp:89-92)
//   * 0 triangles -> bare node; 1 triangle -> leaf with that triangle's biased box (:95-103)
//   * otherwise box = triangle-0 box, then expanded by EVERY triangle's already
//     biased box minus/plus the bias once more (bias applied twice)  (:52-82,106-110)
//   * axis = depth % 3, max_depth updated before the split decision (:114-115)
//   * median of centroids (a+b+c)*(1.0f/3.0f) on that axis          (:18-50)
//   * centroid > median -> "left" list, else "right"                (:121-130)
//   * children only if left list holds more than MAX_KD_LEAF_TRGS   (:135-143);
//     otherwise the node stays a leaf with all of its triangles.
// Each triangle therefore lives in exactly one leaf (a partition, no duplication),
// which the device layout exploits by storing triangles in leaf order.
#include "config.h"
#include "utility.h"
#include "world.h"
#include "world_load.h"

#include <cstdlib>
#include <cstring>
#include <vector>

static float3 triangleMiddle(const Triangle& tr) {
	return (tr.a + tr.b + tr.c) / 3.0f;
}

static inline float axisOf(const float3& v, int axis) {
	return axis == 0 ? v.x : (axis == 1 ? v.y : v.z);
}

static float trs_median(Triangle* trs_src, int* trs_idxs, int num_trs, int axis) {
	float* coords = (float*)malloc(num_trs * sizeof(float));
	for (int i = 0; i < num_trs; i++)
		coords[i] = axisOf(triangleMiddle(trs_src[trs_idxs[i]]), axis);
	float coords_median = fmedian(coords, num_trs);
	free(coords);
	return coords_median;
}

#define BBOX_BIAS 0.00005f

static inline float min2(float a, float b) { return fminf(a, b); }
static inline float max2(float a, float b) { return fmaxf(a, b); }

static BBox makeTrgBBox(const Triangle& trg) {
	BBox bbox;
	bbox.bounds[0].x = min2(trg.a.x, min2(trg.b.x, trg.c.x));
	bbox.bounds[0].y = min2(trg.a.y, min2(trg.b.y, trg.c.y));
	bbox.bounds[0].z = min2(trg.a.z, min2(trg.b.z, trg.c.z));

	bbox.bounds[1].x = max2(trg.a.x, max2(trg.b.x, trg.c.x));
	bbox.bounds[1].y = max2(trg.a.y, max2(trg.b.y, trg.c.y));
	bbox.bounds[1].z = max2(trg.a.z, max2(trg.b.z, trg.c.z));

	bbox.bounds[0] += make_float3(-BBOX_BIAS);
	bbox.bounds[1] += make_float3(BBOX_BIAS);
	return bbox;
}

static void expandBBox(BBox& bbox, const Triangle& tr) {
	BBox trbbox = makeTrgBBox(tr);
	float3& trvmin = trbbox.bounds[0];
	float3& trvmax = trbbox.bounds[1];
	float3& vmin = bbox.bounds[0];
	float3& vmax = bbox.bounds[1];

	vmin.x = min2(vmin.x, trvmin.x - BBOX_BIAS);
	vmin.y = min2(vmin.y, trvmin.y - BBOX_BIAS);
	vmin.z = min2(vmin.z, trvmin.z - BBOX_BIAS);

	vmax.x = max2(vmax.x, trvmax.x + BBOX_BIAS);
	vmax.y = max2(vmax.y, trvmax.y + BBOX_BIAS);
	vmax.z = max2(vmax.z, trvmax.z + BBOX_BIAS);
}

KDNode* buildKDTree(Triangle* trg_src, int* trg_idxs, int num_trs, int depth, int& num_nodes, int& max_depth) {
	KDNode* node = (KDNode*)malloc(sizeof(KDNode));
	memset(&node->bbox, 0, sizeof(BBox)); // reference leaves it uninitialised for empty nodes
	node->trg_idxs = trg_idxs;
	node->num_trgs = num_trs;
	node->left = node->right = NULL;
	node->idx = num_nodes;
	node->left_idx = -1;
	node->right_idx = -1;
	num_nodes++;

	if (num_trs == 0)
		return node;

	if (num_trs == 1) {
		node->bbox = makeTrgBBox(trg_src[trg_idxs[0]]);
		return node;
	}

	node->bbox = makeTrgBBox(trg_src[trg_idxs[0]]);
	for (int i = 0; i < num_trs; i++)
		expandBBox(node->bbox, trg_src[trg_idxs[i]]);

	int axis = depth % 3;
	max_depth = max_depth > depth ? max_depth : depth;
	float median = trs_median(trg_src, trg_idxs, num_trs, axis);

	std::vector<int> trg_left_idxs, trg_right_idxs;
	for (int i = 0; i < num_trs; i++) {
		int trg_idx = trg_idxs[i];
		if (axisOf(triangleMiddle(trg_src[trg_idx]), axis) > median)
			trg_left_idxs.push_back(trg_idx);
		else
			trg_right_idxs.push_back(trg_idx);
	}

	int left_size = (int)trg_left_idxs.size();
	int right_size = (int)trg_right_idxs.size();

	if (left_size > MAX_KD_LEAF_TRGS) {
		int* left_idxs = (int*)malloc(left_size * sizeof(int));
		memcpy(left_idxs, trg_left_idxs.data(), left_size * sizeof(int));
		node->left = buildKDTree(trg_src, left_idxs, left_size, depth + 1, num_nodes, max_depth);

		int* right_idxs = (int*)malloc((right_size > 0 ? right_size : 1) * sizeof(int));
		memcpy(right_idxs, trg_right_idxs.data(), right_size * sizeof(int));
		node->right = buildKDTree(trg_src, right_idxs, right_size, depth + 1, num_nodes, max_depth);
	}

	if (node->left != NULL) node->left_idx = node->left->idx;
	if (node->right != NULL) node->right_idx = node->right->idx;
	return node;
}

// reference: KDTree.cpp:152-163 copied each node into the flat array and gave every
// node its own device allocation for trg_idxs. Here the flat array stays on the
// host (trg_idxs keep pointing at the pointer tree's lists); the device copy is
// made in ONE packed upload by ptx_scene_upload().
void flatenKDTree(KDNode* host_node, KDNode* flat_mem_ptr) {
	if (host_node == NULL) return;
	flat_mem_ptr[host_node->idx] = *host_node;
	flatenKDTree(host_node->left, flat_mem_ptr);
	flatenKDTree(host_node->right, flat_mem_ptr);
}

// reference: KDTree.cpp:165-189 built a device pointer tree that the LOOP traversal
// never dereferenced. Kept as an entry point; it returns its argument.
KDNode* copyKDTreeToCuda(KDNode* host_root) {
	return host_root;
}

void freeKDTree(KDNode* root) {
	if (root == NULL) return;
	freeKDTree(root->left);
	freeKDTree(root->right);
	if (root->trg_idxs != NULL) free(root->trg_idxs);
	free(root);
}

// Device memory of all trees belongs to the uploaded PtxScene and is released by
// freeWorldObjects(); nothing per-tree to do (reference: KDTree.cpp:204-210).
void freeCudaKDTree(KDNode*) {}
