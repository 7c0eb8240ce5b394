// See host_capi.h: thin extern "C" forwarding layer over the C++ host API.
#include "host_capi.h"

#include "hostscene.h"
#include "image_io.h"
#include "kernel.h"
#include "pathtracing.h"
#include "scenestate.h"
#include "utility.h"
#include "world_load.h"

#include <cstdlib>
#include <cstring>
#include <string>

static std::string g_err;
static DialogLogger* g_logger = NULL;
static Scene g_scene; // the render thread's clone (reference: qt/cudarunthread.cpp:16,22)
static bool g_have_scene = false;

extern "C" const char* pth_last_error(void) { return g_err.c_str(); }
extern "C" void pth_set_asset_root(const char* dir) { setAssetRoot(dir ? dir : ""); }
extern "C" void pth_set_device(int device) { hip_device_id = device; }
extern "C" void pth_set_device_upload(int enabled) { setDeviceUploadEnabled(enabled != 0); }
extern "C" void pth_set_procedural_car_triangles(int n) { setProceduralCarTriangles(n); }
extern "C" void pth_set_seed(uint64_t seed) { kernelSetSeed(seed); }
// samples the one-sample-per-call loop (pth_run_iteration = runCUDA) may trace ahead; 1 = none
extern "C" void pth_set_look_ahead(int samples) { kernelSetLookAhead(samples); }
extern "C" int pth_look_ahead(void) { return kernelLookAhead(); }

extern "C" int pth_scene_load(const char* json_path, int width, int height, int textures, int preview, int aabb, int bounces) {
	Json::Value jscene;
	std::string err;
	if (!parseJsonScene(json_path ? json_path : "", jscene, err)) { g_err = err; return PTX_E_INVALID; }
	delete g_logger;
	g_logger = new DialogLogger(NULL);
	if (!sceneState.load(jscene, g_logger)) {
		g_err = "scene load failed";
		for (size_t i = 0; i < g_logger->lines().size(); ++i)
			if (g_logger->lines()[i].find("[error]") == 0) g_err = g_logger->lines()[i];
		return PTX_E_INVALID;
	}
	if (width > 0 && height > 0) sceneState.overrideResolution(width, height);
	sceneState.togglePrevMode(preview > 0);
	if (aabb >= 0) sceneState.toggleAABBmode(aabb != 0);
	if (textures >= 0) sceneState.toggleTextures(textures != 0);
	if (bounces >= 0) sceneState.updateMaxRayBnc(bounces);
	g_scene = sceneState.clone();
	g_have_scene = true;
	return PTX_OK;
}

extern "C" int pth_scene_free(void) { sceneState.clean(); g_have_scene = false; return PTX_OK; }
extern "C" int pth_scene_is_loaded(void) { return sceneState.isLoaded() ? 1 : 0; }
extern "C" const PtxSceneDesc* pth_scene_desc(void) { return &hostSceneFlat().desc; }
extern "C" const float* pth_scene_tri_abc(void) { return hostSceneFlat().tri_abc.empty() ? NULL : hostSceneFlat().tri_abc.data(); }
extern "C" PtxScene* pth_scene_device(void) { return g_have_scene ? (PtxScene*)g_scene.dv_wobjects_ptr : NULL; }

extern "C" int pth_camera_get(PtxCamera* out) {
	if (!out || !sceneState.isLoaded()) { g_err = "no scene loaded"; return PTX_E_STATE; }
	Scene s = sceneState.clone();
	fillPtxCamera(s.camera, out);
	return PTX_OK;
}

extern "C" int pth_camera_control(const char* what, float value) {
	if (!what || !sceneState.isLoaded()) { g_err = "no scene loaded"; return PTX_E_STATE; }
	std::string w(what);
	if (w == "h_ang") sceneState.setCamHang(value);
	else if (w == "v_ang") sceneState.setCamVang(value);
	else if (w == "pos_x") sceneState.setCamPositionX(value);
	else if (w == "pos_y") sceneState.setCamPositionY(value);
	else if (w == "pos_z") sceneState.setCamPositionZ(value);
	else if (w == "reset") sceneState.resetCamera();
	else if (w == "bounces") sceneState.updateMaxRayBnc((int)value);
	else if (w == "aa_jitter") sceneState.updateAajitter(value);
	else if (w == "gamma") sceneState.updateGamma(value);
	else if (w == "preview") sceneState.togglePrevMode(value != 0.0f);
	else if (w == "aabb") sceneState.toggleAABBmode(value != 0.0f);
	else if (w == "textures") sceneState.toggleTextures(value != 0.0f);
	else { g_err = "unknown camera control: " + w; return PTX_E_INVALID; }
	return PTX_OK;
}

extern "C" int pth_log_lines(char* buf, int buf_len) {
	if (!buf || buf_len <= 0) return 0;
	std::string all;
	if (g_logger)
		for (size_t i = 0; i < g_logger->lines().size(); ++i) all += g_logger->lines()[i] + "\n";
	strncpy(buf, all.c_str(), (size_t)buf_len - 1);
	buf[buf_len - 1] = 0;
	return (int)all.size();
}

// distinct texture files of the loaded scene that failed to load: returns their number, writes them
// newline-separated into buf (if given)
extern "C" int pth_texture_failures(char* buf, int buf_len) {
	const std::vector<std::string>& f = failedTextureFiles();
	if (buf && buf_len > 0) {
		std::string all;
		for (size_t i = 0; i < f.size(); ++i) all += f[i] + "\n";
		strncpy(buf, all.c_str(), (size_t)buf_len - 1);
		buf[buf_len - 1] = 0;
	}
	return (int)f.size();
}

extern "C" PtxCtx* pth_context(void) { return kernelContext(); }

extern "C" int pth_kernel_init(void) {
	if (!sceneState.isLoaded()) { g_err = "no scene loaded"; return PTX_E_STATE; }
	g_scene = sceneState.clone();
	sceneState.cameraChanged(false, NULL);
	g_scene.camera.changed = false;
	kernelInit(g_scene);
	if (checkCudaError("kernelInit")) { g_err = ptx_last_error(); return PTX_E_HIP; }
	return PTX_OK;
}

// the body of the reference's render loop (qt/cudarunthread.cpp:21-41) for one iteration
static void refreshClone(int* iter) {
	Scene fresh = sceneState.clone();
	void* dv = g_scene.dv_wobjects_ptr;
	g_scene = fresh;
	if (!g_scene.dv_wobjects_ptr) g_scene.dv_wobjects_ptr = dv;
	sceneState.cameraChanged(false, iter);
}

extern "C" int pth_run_iteration(int iter, unsigned char* pbo_rgba8) {
	if (!g_have_scene) { g_err = "no scene loaded"; return PTX_E_STATE; }
	int i = iter;
	refreshClone(&i);
	runCUDA((uchar4*)pbo_rgba8, g_scene, i);
	if (checkCudaError("runCUDA")) { g_err = ptx_last_error(); return PTX_E_HIP; }
	return i; // the iteration actually rendered (1 if the camera changed)
}

extern "C" int pth_render_samples(int first_sample, int num_samples, int collect_stats) {
	if (!g_have_scene) { g_err = "no scene loaded"; return PTX_E_STATE; }
	refreshClone(NULL);
	int rc = renderSamples(g_scene, first_sample, num_samples, collect_stats != 0);
	if (rc != PTX_OK) g_err = ptx_last_error();
	return rc;
}

extern "C" int pth_kernel_cleanup(void) { kernelCleanUp(); return PTX_OK; }
extern "C" void pth_shutdown(void) {
	kernelCleanUp();
	sceneState.clean();
	g_have_scene = false;
	kernelShutdown();
}

extern "C" int pth_kd_build(const float* abc, int n, PtxKDNode** nodes_out, int32_t** leaf_tris_out, int* num_leaf_tris_out, int* max_depth_out) {
	if (n < 0 || (n > 0 && !abc) || !nodes_out || !leaf_tris_out || !num_leaf_tris_out || !max_depth_out) { g_err = "pth_kd_build: bad arguments"; return PTX_E_INVALID; }
	Triangle* tr = (Triangle*)calloc((size_t)(n > 0 ? n : 1), sizeof(Triangle));
	for (int i = 0; i < n; ++i) {
		const float* p = abc + (size_t)i * 9;
		tr[i].a = make_float3(p[0], p[1], p[2]);
		tr[i].b = make_float3(p[3], p[4], p[5]);
		tr[i].c = make_float3(p[6], p[7], p[8]);
		tr[i].e1 = tr[i].b - tr[i].a;
		tr[i].e2 = tr[i].c - tr[i].a;
	}
	int* idx = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
	for (int i = 0; i < n; ++i) idx[i] = i;
	int num_nodes = 0, max_depth = 0;
	KDNode* root = buildKDTree(tr, idx, n, 0, num_nodes, max_depth);
	KDNode* flat = (KDNode*)malloc(sizeof(KDNode) * (size_t)num_nodes);
	flatenKDTree(root, flat);
	size_t total = 0;
	for (int i = 0; i < num_nodes; ++i)
		if (flat[i].left_idx == -1 && flat[i].right_idx == -1) total += (size_t)flat[i].num_trgs;
	PtxKDNode* nodes = (PtxKDNode*)malloc(sizeof(PtxKDNode) * (size_t)num_nodes);
	int32_t* lt = (int32_t*)malloc(sizeof(int32_t) * (total > 0 ? total : 1));
	size_t pos = 0;
	for (int i = 0; i < num_nodes; ++i) {
		const KDNode& hn = flat[i];
		PtxKDNode& pn = nodes[i];
		pn.bmin[0] = hn.bbox.bounds[0].x; pn.bmin[1] = hn.bbox.bounds[0].y; pn.bmin[2] = hn.bbox.bounds[0].z;
		pn.bmax[0] = hn.bbox.bounds[1].x; pn.bmax[1] = hn.bbox.bounds[1].y; pn.bmax[2] = hn.bbox.bounds[1].z;
		pn.left = hn.left_idx; pn.right = hn.right_idx; pn.num_tris = hn.num_trgs; pn.first_tri = -1;
		if (hn.left_idx == -1 && hn.right_idx == -1) {
			pn.first_tri = (int32_t)pos;
			for (int k = 0; k < hn.num_trgs; ++k) lt[pos++] = hn.trg_idxs[k];
		}
	}
	free(flat);
	freeKDTree(root);
	free(tr);
	*nodes_out = nodes; *leaf_tris_out = lt; *num_leaf_tris_out = (int)total; *max_depth_out = max_depth;
	return num_nodes;
}

extern "C" void pth_free(void* p) { free(p); }
extern "C" unsigned int pth_wang_hash(unsigned int x) { return wang_hash(x); }
extern "C" unsigned int pth_wang_unhash(unsigned int h) { return wang_unhash(h); }
extern "C" float pth_fmedian(float* arr, int n) { return fmedian(arr, n); }

extern "C" void pth_transform_compose(const float* ops, int n, float* m16, float* normal16) {
	std::vector<TransformStep> steps((size_t)(n > 0 ? n : 0));
	for (int i = 0; i < n; ++i) {
		steps[(size_t)i].kind = (int)ops[5 * i];
		for (int k = 0; k < 4; ++k) steps[(size_t)i].v[k] = ops[5 * i + 1 + k];
	}
	const glmlite::mat4 m = composeTransform(steps.data(), n);
	const glmlite::mat4 nm = glmlite::transpose(glmlite::inverse(m)); // what loadTriangleMeshObj applies to normals
	memcpy(m16, &m.c[0].x, 16 * sizeof(float));
	memcpy(normal16, &nm.c[0].x, 16 * sizeof(float));
}

extern "C" void pth_transform_apply(const float* m16, const float* v3, int n, float* out4) {
	glmlite::mat4 m;
	memcpy(&m.c[0].x, m16, 16 * sizeof(float));
	for (int i = 0; i < n; ++i) {
		const glmlite::vec4 r = glmlite::mul(m, glmlite::vec4{v3[3 * i], v3[3 * i + 1], v3[3 * i + 2], 1.0f});
		out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
	}
}

extern "C" int pth_load_image_float(const char* path, int* w, int* h, int* channels, float** pixels_out) {
	std::vector<float> px;
	std::string err;
	if (!loadImageFloat(path, *w, *h, *channels, px, err)) { g_err = err; return PTX_E_INVALID; }
	float* o = (float*)malloc(px.size() * sizeof(float));
	memcpy(o, px.data(), px.size() * sizeof(float));
	*pixels_out = o;
	return PTX_OK;
}

// mirror_h != 0 reproduces what the viewer of the reference sees: its textured quad maps the
// buffer's x = 0 to the right edge of the window (reference: qt/previewglwidget.cpp:104-116).
extern "C" int pth_write_png(const char* path, int w, int h, int comp, const unsigned char* pixels, int mirror_h) {
	std::string err;
	std::vector<unsigned char> tmp;
	const unsigned char* src = pixels;
	if (mirror_h) {
		tmp.resize((size_t)w * h * comp);
		for (int y = 0; y < h; ++y)
			for (int x = 0; x < w; ++x)
				memcpy(&tmp[((size_t)y * w + x) * comp], &pixels[((size_t)y * w + (w - 1 - x)) * comp], (size_t)comp);
		src = tmp.data();
	}
	if (!writePNG(path, w, h, comp, src, err)) { g_err = err; return PTX_E_INVALID; }
	return PTX_OK;
}

extern "C" int pth_write_pfm(const char* path, int w, int h, const float* rgb) {
	std::string err;
	if (!writePFM(path, w, h, rgb, err)) { g_err = err; return PTX_E_INVALID; }
	return PTX_OK;
}

// ---- checkpoint.h ---------------------------------------------------------------------------------
#include "checkpoint.h"

extern "C" int pth_checkpoint_save(const char* path, int samples_done) {
	PtxCtx* ctx = kernelContext();
	if (!ctx || !path) { g_err = "pth_checkpoint_save: no context / path"; return PTX_E_STATE; }
	if (!sceneState.isLoaded()) { g_err = "pth_checkpoint_save: no scene"; return PTX_E_STATE; }
	Scene sc = sceneState.clone();
	AccumCheckpoint ck;
	ck.width = sc.camera.projection.width;
	ck.height = sc.camera.projection.height;
	ck.samples_done = samples_done;
	ck.seed = kernelSeed();
	ck.scene_hash = renderContentHash(hostSceneFlat().desc, sc.camera);
	ck.max_ray_bounces = sc.camera.max_ray_bounces;
	ck.texture_enabled = sc.camera.texture_enabled ? 1 : 0;
	ck.rgba.resize((size_t)ck.width * ck.height * 4);
	int rc = ptx_accum_download(ctx, ck.rgba.data());
	if (rc != PTX_OK) { g_err = ptx_last_error(); return rc; }
	std::string err;
	if (!saveCheckpoint(path, ck, err)) { g_err = err; return PTX_E_INVALID; }
	return PTX_OK;
}

extern "C" int pth_checkpoint_load(const char* path, int* samples_done) {
	PtxCtx* ctx = kernelContext();
	if (!ctx || !path || !samples_done) { g_err = "pth_checkpoint_load: no context / path"; return PTX_E_STATE; }
	if (!sceneState.isLoaded()) { g_err = "pth_checkpoint_load: no scene"; return PTX_E_STATE; }
	AccumCheckpoint ck;
	std::string err;
	if (!loadCheckpoint(path, ck, err)) { g_err = err; return PTX_E_INVALID; }
	Scene sc = sceneState.clone();
	if (ck.width != sc.camera.projection.width || ck.height != sc.camera.projection.height) {
		g_err = "pth_checkpoint_load: checkpoint resolution differs from the scene's";
		return PTX_E_INVALID;
	}
	if (ck.seed != kernelSeed()) { g_err = "pth_checkpoint_load: checkpoint was rendered with another seed"; return PTX_E_INVALID; }
	const std::string why = checkpointMismatch(ck, renderContentHash(hostSceneFlat().desc, sc.camera), sc.camera.max_ray_bounces, sc.camera.texture_enabled ? 1 : 0);
	if (!why.empty()) { g_err = "pth_checkpoint_load: checkpoint does not continue this render: " + why; return PTX_E_INVALID; }
	int rc = ptx_accum_upload(ctx, ck.rgba.data());
	if (rc != PTX_OK) { g_err = ptx_last_error(); return rc; }
	*samples_done = ck.samples_done;
	return PTX_OK;
}

extern "C" int pth_autosave_due(int iter, int every) { return autosaveDue(iter, every) ? 1 : 0; }

extern "C" int pth_autosave_name(const char* folder, const char* name, int iter, int day, int month, int year, char* out, int out_len) {
	if (!folder || !name || !out || out_len <= 0) return PTX_E_INVALID;
	const std::string s = day > 0 ? autosaveFileName(folder, name, iter, day, month, year) : autosaveFileName(folder, name, iter);
	if ((int)s.size() + 1 > out_len) return PTX_E_LIMIT;
	memcpy(out, s.c_str(), s.size() + 1);
	return PTX_OK;
}

extern "C" int pth_checkpoint_write_file(const char* path, int width, int height, int samples_done, uint64_t seed, const float* rgba) {
	if (!path || !rgba || width <= 0 || height <= 0) { g_err = "pth_checkpoint_write_file: bad arguments"; return PTX_E_INVALID; }
	AccumCheckpoint ck;
	ck.width = width; ck.height = height; ck.samples_done = samples_done; ck.seed = seed;
	ck.rgba.assign(rgba, rgba + (size_t)width * height * 4);
	std::string err;
	if (!saveCheckpoint(path, ck, err)) { g_err = err; return PTX_E_INVALID; }
	return PTX_OK;
}

extern "C" int pth_checkpoint_read_file(const char* path, int* width, int* height, int* samples_done, uint64_t* seed, float* rgba,
                                        int64_t rgba_capacity_floats) {
	if (!path || !width || !height || !samples_done || !seed) { g_err = "pth_checkpoint_read_file: bad arguments"; return PTX_E_INVALID; }
	AccumCheckpoint ck;
	std::string err;
	if (!loadCheckpoint(path, ck, err)) { g_err = err; return PTX_E_INVALID; }
	*width = ck.width; *height = ck.height; *samples_done = ck.samples_done; *seed = ck.seed;
	if (rgba) {
		if ((int64_t)ck.rgba.size() > rgba_capacity_floats) { g_err = "pth_checkpoint_read_file: buffer too small"; return PTX_E_LIMIT; }
		memcpy(rgba, ck.rgba.data(), ck.rgba.size() * sizeof(float));
	}
	return PTX_OK;
}
