#include "checkpoint.h"
#include "camera.h"
#include "../../include/ptx_hip.h"

#include <cstddef>
#include <cstdio>
#include <cstring>
#include <ctime>

// PTXACC2: PTXACC1 (width, height, samples, seed) + scene hash, bounce limit, texture switch
static const char CK_MAGIC[8] = {'P', 'T', 'X', 'A', 'C', 'C', '2', 0};

static void fnv(uint64_t& h, const void* p, size_t n) {
	const unsigned char* b = (const unsigned char*)p;
	for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
}

uint64_t sceneContentHash(const PtxSceneDesc& d) {
	uint64_t h = 14695981039346656037ull;
	fnv(h, &d.num_objects, 4); fnv(h, &d.num_triangles, 4); fnv(h, &d.num_nodes, 4); fnv(h, &d.num_materials, 4);
	if (d.objects) fnv(h, d.objects, sizeof(PtxObject) * (size_t)d.num_objects);
	for (int i = 0; i < d.num_materials; ++i) fnv(h, &d.materials[i], offsetof(PtxMaterial, _pad));   // not the padding word
	if (d.triangles) fnv(h, d.triangles, sizeof(PtxTriangle) * (size_t)d.num_triangles);
	if (d.nodes) fnv(h, d.nodes, sizeof(PtxKDNode) * (size_t)d.num_nodes);
	if (d.leaf_tris) fnv(h, d.leaf_tris, sizeof(int32_t) * (size_t)d.num_leaf_tris);
	for (int i = 0; i < d.num_textures; ++i) {
		fnv(h, &d.textures[i].width, 4); fnv(h, &d.textures[i].height, 4);
		// every 61st texel (the headline's textures are 252 MB of floats; a swapped or re-encoded file changes far more than one in 61)
		const size_t n = (size_t)d.textures[i].width * (size_t)d.textures[i].height;
		if (d.textures[i].rgba) for (size_t t = 0; t < n; t += 61) fnv(h, d.textures[i].rgba + 4 * t, 16);
	}
	return h ? h : 1;   // 0 means "not recorded"
}

uint64_t renderContentHash(const PtxSceneDesc& d, const Camera& cam) {
	uint64_t h = sceneContentHash(d);
	const float f[] = {cam.position.x, cam.position.y, cam.position.z, cam.direction.x, cam.direction.y, cam.direction.z, cam.up.x, cam.up.y, cam.up.z,
	                   cam.right.x, cam.right.y, cam.right.z, cam.projection.pixel_size.x, cam.projection.pixel_size.y, cam.projection.screen_halfsize.x,
	                   cam.projection.screen_halfsize.y, cam.projection.screen_dist, cam.projection.aa_jitter};
	const int32_t i[] = {cam.projection.width, cam.projection.height, cam.preview_mode ? 1 : 0, cam.aabb_mode ? 1 : 0};
	fnv(h, f, sizeof(f));
	fnv(h, i, sizeof(i));
	return h ? h : 1;
}

std::string checkpointMismatch(const AccumCheckpoint& ck, uint64_t scene_hash, int max_ray_bounces, int texture_enabled) {
	if (ck.scene_hash != 0 && scene_hash != 0 && ck.scene_hash != scene_hash) return "it was rendered from a different scene, textures or camera";
	if (ck.max_ray_bounces >= 0 && ck.max_ray_bounces != max_ray_bounces)
		return "it was rendered with max_ray_bounces " + std::to_string(ck.max_ray_bounces) + ", not " + std::to_string(max_ray_bounces);
	if (ck.texture_enabled >= 0 && ck.texture_enabled != (texture_enabled ? 1 : 0))
		return std::string("it was rendered with textures ") + (ck.texture_enabled ? "on" : "off");
	return "";
}

bool saveCheckpoint(const std::string& path, const AccumCheckpoint& ck, std::string& err) {
	if (ck.width <= 0 || ck.height <= 0 || ck.samples_done < 0 || ck.rgba.size() != (size_t)ck.width * ck.height * 4) {
		err = "saveCheckpoint: inconsistent checkpoint";
		return false;
	}
	const std::string tmp = path + ".tmp";
	FILE* f = fopen(tmp.c_str(), "wb");
	if (!f) { err = "saveCheckpoint: cannot open " + tmp; return false; }
	bool ok = fwrite(CK_MAGIC, 1, 8, f) == 8 && fwrite(&ck.width, 4, 1, f) == 1 && fwrite(&ck.height, 4, 1, f) == 1 &&
	          fwrite(&ck.samples_done, 4, 1, f) == 1 && fwrite(&ck.seed, 8, 1, f) == 1 && fwrite(&ck.scene_hash, 8, 1, f) == 1 &&
	          fwrite(&ck.max_ray_bounces, 4, 1, f) == 1 && fwrite(&ck.texture_enabled, 4, 1, f) == 1 &&
	          fwrite(ck.rgba.data(), sizeof(float), ck.rgba.size(), f) == ck.rgba.size();
	ok = (fclose(f) == 0) && ok;
	// written beside the target and renamed, so an interrupted save never clobbers the last good checkpoint
	if (ok && rename(tmp.c_str(), path.c_str()) != 0) ok = false;
	if (!ok) { remove(tmp.c_str()); err = "saveCheckpoint: write failed for " + path; }
	return ok;
}

bool loadCheckpoint(const std::string& path, AccumCheckpoint& ck, std::string& err) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) { err = "loadCheckpoint: cannot open " + path; return false; }
	char magic[8];
	bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, CK_MAGIC, 8) == 0;
	if (!ok) { fclose(f); err = "loadCheckpoint: " + path + " is not a PTXACC2 checkpoint"; return false; }
	ok = fread(&ck.width, 4, 1, f) == 1 && fread(&ck.height, 4, 1, f) == 1 && fread(&ck.samples_done, 4, 1, f) == 1 &&
	     fread(&ck.seed, 8, 1, f) == 1 && fread(&ck.scene_hash, 8, 1, f) == 1 && fread(&ck.max_ray_bounces, 4, 1, f) == 1 &&
	     fread(&ck.texture_enabled, 4, 1, f) == 1;
	if (!ok || ck.width <= 0 || ck.height <= 0 || ck.samples_done < 0 || (int64_t)ck.width * ck.height > ((int64_t)1 << 28)) {
		fclose(f);
		err = "loadCheckpoint: bad header in " + path;
		return false;
	}
	ck.rgba.resize((size_t)ck.width * ck.height * 4);
	ok = fread(ck.rgba.data(), sizeof(float), ck.rgba.size(), f) == ck.rgba.size();
	char extra;
	if (ok && fread(&extra, 1, 1, f) != 0) ok = false; // trailing bytes: not ours
	fclose(f);
	if (!ok) err = "loadCheckpoint: truncated or oversized payload in " + path;
	return ok;
}

bool autosaveDue(int iter, int every) { return iter > 0 && every > 0 && ((iter % every) == 0 || iter == 1); }

std::string autosaveFileName(const std::string& folder, const std::string& name, int iter, int day, int month, int year) {
	char buf[64];
	snprintf(buf, sizeof(buf), "_%02d-%02d-%04d_%dspp.png", day, month, year, iter);
	return folder + "/" + name + buf;
}

std::string autosaveFileName(const std::string& folder, const std::string& name, int iter) {
	time_t now = time(NULL);
	struct tm tmv;
	localtime_r(&now, &tmv);
	return autosaveFileName(folder, name, iter, tmv.tm_mday, tmv.tm_mon + 1, tmv.tm_year + 1900);
}
