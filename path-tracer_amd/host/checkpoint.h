// Float accumulation checkpoint / resume and the reference's autosave file naming.
// The reference autosaves tone-mapped PNGs only (qt/mainwindow.cpp:93-99) and cannot resume; a
// checkpoint here is the float4 accumulation buffer plus what is needed to continue the SAME sample
// sequence: resolution, samples done, master seed. Because samples are added in index order, a run
// resumed from a checkpoint is bit-identical to an uninterrupted one.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct PtxSceneDesc;

struct AccumCheckpoint {
	int32_t width, height;
	int32_t samples_done;      // the buffer holds global samples [0, samples_done)
	uint64_t seed;
	// what else the samples depend on: continuing with another scene, bounce limit or texture switch would mix
	// two different images in one buffer, so a resume checks these too (0 / -1 = "not recorded", accepted as is)
	uint64_t scene_hash = 0;          // sceneContentHash() of the scene description the samples were rendered from
	int32_t max_ray_bounces = -1;
	int32_t texture_enabled = -1;
	std::vector<float> rgba;   // width * height * 4
};

struct Camera;

// FNV-1a over the geometry, tree, materials and textures (sizes and every 61st texel) of a flat scene description (include/ptx_hip.h)
uint64_t sceneContentHash(const PtxSceneDesc& d);
// The same scene seen by another camera is another image: sceneContentHash folded with what a sample depends on of the camera
// (pose, projection, jitter width, preview / AABB switches; not gamma, which only the display uses). This is what a
// checkpoint records as scene_hash (a resume with a moved camera used to mix two images silently).
uint64_t renderContentHash(const PtxSceneDesc& d, const Camera& cam);
// Empty string if `ck` continues a render of (scene_hash, bounces, textures); else what differs.
std::string checkpointMismatch(const AccumCheckpoint& ck, uint64_t scene_hash, int max_ray_bounces, int texture_enabled);

bool saveCheckpoint(const std::string& path, const AccumCheckpoint& ck, std::string& err);
bool loadCheckpoint(const std::string& path, AccumCheckpoint& ck, std::string& err);

// reference: qt/mainwindow.cpp:93 — autosave when (iter % every == 0 || iter == 1) && iter > 0
bool autosaveDue(int iter, int every);
// reference: qt/mainwindow.cpp:96-97 — "<folder>/<name>_<dd-MM-yyyy>_<iter>spp.png" (date = today, local time)
std::string autosaveFileName(const std::string& folder, const std::string& name, int iter);
std::string autosaveFileName(const std::string& folder, const std::string& name, int iter, int day, int month, int year);
