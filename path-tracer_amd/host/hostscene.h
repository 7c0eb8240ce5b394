// Flat host copy of the loaded scene in the C ABI's description format (include/ptx_hip.h).
// Built by world_load.cpp after all objects are loaded; handed to ptx_scene_upload() once and
// kept so tests can read exactly what the device received.
#pragma once
#include <string>
#include <vector>
#include "../../include/ptx_hip.h"

struct HostTexture {
	int width, height;
	std::vector<float> rgba; // rows flipped as the reference uploads them
};

struct HostSceneFlat {
	std::vector<PtxObject> objects;
	std::vector<PtxMaterial> materials;
	std::vector<PtxTexture> textures;   // rgba pointers alias the loader's HostTexture storage
	std::vector<PtxTriangle> triangles;
	std::vector<float> tri_abc;         // 9 floats per triangle: a, b, c (KD build input)
	std::vector<PtxKDNode> nodes;
	std::vector<int32_t> leaf_tris;
	PtxSceneDesc desc;

	HostSceneFlat() { fillDesc(); }
	void fillDesc() {
		desc.num_objects = (int32_t)objects.size();     desc.objects = objects.empty() ? NULL : objects.data();
		desc.num_materials = (int32_t)materials.size(); desc.materials = materials.empty() ? NULL : materials.data();
		desc.num_textures = (int32_t)textures.size();   desc.textures = textures.empty() ? NULL : textures.data();
		desc.num_triangles = (int32_t)triangles.size(); desc.triangles = triangles.empty() ? NULL : triangles.data();
		desc.num_nodes = (int32_t)nodes.size();         desc.nodes = nodes.empty() ? NULL : nodes.data();
		desc.num_leaf_tris = (int32_t)leaf_tris.size(); desc.leaf_tris = leaf_tris.empty() ? NULL : leaf_tris.data();
	}
	// copies must re-point desc at their own vectors
	HostSceneFlat(const HostSceneFlat& o) { *this = o; }
	HostSceneFlat& operator=(const HostSceneFlat& o) {
		objects = o.objects; materials = o.materials; textures = o.textures; triangles = o.triangles;
		tri_abc = o.tri_abc; nodes = o.nodes; leaf_tris = o.leaf_tris;
		fillDesc();
		return *this;
	}
};

const HostSceneFlat& hostSceneFlat();
void setDeviceUploadEnabled(bool on);
int loadTexture(const std::string src_path);
// texture files named by the loaded scene that could not be read (distinct paths, load order)
const std::vector<std::string>& failedTextureFiles();
