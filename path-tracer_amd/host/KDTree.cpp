// KD-tree build over triangle centroids, behind the reference's entry points buildKDTree / flatenKDTree /
// copyKDTreeToCuda / freeKDTree / freeCudaKDTree (reference: src/KDTree.cpp:18-210).
//
// The tree decides which leaves a ray visits and in which order triangles are tested, so its OUTPUT is the
// reference's exactly (compiled with -ffp-contract=off; tests compare it node for node with the oracle's
// independently written builder and pin the median to the reference's own utility.cpp):
//   * node index = preorder creation counter                                         (KDTree.cpp:89-92)
//   * no triangle -> bare node; one triangle -> leaf with that triangle's padded box  (:95-103)
//   * else box = triangle 0's padded box grown by EVERY triangle's padded box padded once more
//     (the padding of 5e-5 is applied twice)                                          (:52-82, 106-110)
//   * split axis = depth % 3; the deepest level is recorded before the split decision (:114-115)
//   * split value = median of the centroids (a + b + c) * (1.0f / 3.0f) on that axis  (:18-50)
//   * centroid > median -> first child's list, else second child's, input order kept  (:121-130)
//   * children only if the first list holds more than MAX_KD_LEAF_TRGS triangles      (:135-143);
//     otherwise the node stays a leaf with all of its triangles.
// Every triangle therefore lives in exactly one leaf, which the device layout exploits by storing
// triangles in leaf order.
//
// How it is built here: a Grower object computes each triangle's centroid and padded box ONCE (the
// reference recomputes them at every level: the same pure float expressions, so the same values) and
// splits index lists with one stable pass into two vectors.
#include "config.h"
#include "utility.h"
#include "world.h"
#include "world_load.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

const float kBoxPad = 0.00005f;   // reference: BBOX_BIAS, src/KDTree.cpp:52

struct Bounds { float lo[3], hi[3]; };

inline float least(float a, float b) { return fminf(a, b); }
inline float most(float a, float b) { return fmaxf(a, b); }

class Grower {
public:
	Grower(const Triangle* tris, int& node_counter, int& deepest) : tris_(tris), counter_(node_counter), deepest_(deepest) {}

	KDNode* grow(int* ids, int n, int depth) {
		KDNode* node = (KDNode*)malloc(sizeof(KDNode));
		memset(&node->bbox, 0, sizeof(BBox));   // the reference leaves an empty node's box uninitialised
		node->trg_idxs = ids;
		node->num_trgs = n;
		node->left = node->right = NULL;
		node->left_idx = node->right_idx = -1;
		node->idx = counter_++;
		if (n == 0) return node;

		Bounds box = padded(ids[0]);
		if (n > 1) {
			for (int i = 0; i < n; ++i) {   // triangle 0 takes part again, as in the reference
				const Bounds t = padded(ids[i]);
				for (int k = 0; k < 3; ++k) {
					box.lo[k] = least(box.lo[k], t.lo[k] - kBoxPad);
					box.hi[k] = most(box.hi[k], t.hi[k] + kBoxPad);
				}
			}
		}
		node->bbox.bounds[0] = make_float3(box.lo[0], box.lo[1], box.lo[2]);
		node->bbox.bounds[1] = make_float3(box.hi[0], box.hi[1], box.hi[2]);
		if (n == 1) return node;

		const int axis = depth % 3;
		if (depth > deepest_) deepest_ = depth;

		// split value: the median of the centroid coordinates (fmedian sorts its argument, so it gets a copy)
		std::vector<float> coords((size_t)n);
		for (int i = 0; i < n; ++i) coords[(size_t)i] = centroid(ids[i], axis);
		std::vector<float> sorted(coords);
		const float cut = fmedian(sorted.data(), n);

		std::vector<int> above, rest;   // "above the median" becomes the first child
		for (int i = 0; i < n; ++i) (coords[(size_t)i] > cut ? above : rest).push_back(ids[i]);

		if ((int)above.size() > MAX_KD_LEAF_TRGS) {
			node->left = grow(detach(above), (int)above.size(), depth + 1);
			node->right = grow(detach(rest), (int)rest.size(), depth + 1);
			node->left_idx = node->left->idx;
			node->right_idx = node->right->idx;
		}
		return node;
	}

private:
	// (a + b + c) / 3.0f with the vector header's operator/, i.e. a multiplication by (1.0f / 3.0f)
	float centroid(int id, int axis) const {
		const Triangle& t = tris_[id];
		const float3 m = (t.a + t.b + t.c) / 3.0f;
		return axis == 0 ? m.x : (axis == 1 ? m.y : m.z);
	}

	Bounds padded(int id) const {
		const Triangle& t = tris_[id];
		const float a[3] = {t.a.x, t.a.y, t.a.z}, b[3] = {t.b.x, t.b.y, t.b.z}, c[3] = {t.c.x, t.c.y, t.c.z};
		Bounds r;
		for (int k = 0; k < 3; ++k) {
			r.lo[k] = least(a[k], least(b[k], c[k])) + -kBoxPad;
			r.hi[k] = most(a[k], most(b[k], c[k])) + kBoxPad;
		}
		return r;
	}

	// the node owns a malloc'ed copy of its list (freeKDTree releases it with free())
	static int* detach(const std::vector<int>& v) {
		int* p = (int*)malloc((v.empty() ? 1 : v.size()) * sizeof(int));
		if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(int));
		return p;
	}

	const Triangle* tris_;
	int& counter_;
	int& deepest_;
};

} // namespace

KDNode* buildKDTree(Triangle* trg_src, int* trg_idxs, int num_trs, int depth, int& num_nodes, int& max_depth) {
	return Grower(trg_src, num_nodes, max_depth).grow(trg_idxs, num_trs, depth);
}

// reference: KDTree.cpp:152-163 copied each node into the flat array and gave every node its own device
// allocation for trg_idxs. Here the flat array stays on the host (trg_idxs keep pointing at the pointer
// tree's lists); the device copy is made in ONE packed upload by ptx_scene_upload().
void flatenKDTree(KDNode* host_node, KDNode* flat_mem_ptr) {
	std::vector<KDNode*> todo;
	if (host_node) todo.push_back(host_node);
	while (!todo.empty()) {
		KDNode* n = todo.back();
		todo.pop_back();
		flat_mem_ptr[n->idx] = *n;
		if (n->right) todo.push_back(n->right);
		if (n->left) todo.push_back(n->left);
	}
}

// reference: KDTree.cpp:165-189 built a device pointer tree that the LOOP traversal never dereferenced.
// Kept as an entry point; it returns its argument.
KDNode* copyKDTreeToCuda(KDNode* host_root) { return host_root; }

void freeKDTree(KDNode* root) {
	std::vector<KDNode*> todo;
	if (root) todo.push_back(root);
	while (!todo.empty()) {
		KDNode* n = todo.back();
		todo.pop_back();
		if (n->left) todo.push_back(n->left);
		if (n->right) todo.push_back(n->right);
		if (n->trg_idxs) free(n->trg_idxs);
		free(n);
	}
}

// Device memory of all trees belongs to the uploaded PtxScene and is released by freeWorldObjects();
// nothing per-tree to do (reference: KDTree.cpp:204-210).
void freeCudaKDTree(KDNode*) {}
