// reference: src/scenestate.cpp:6-128 (QMutex -> std::mutex)
#include "scenestate.h"
#include "kernel.h"
#include "world_load.h"

SceneState sceneState;

SceneState::SceneState() : loaded(false) {}
SceneState::~SceneState() { this->clean(); }

bool SceneState::load(const Json::Value& jscene, DialogLogger* logger) {
	std::lock_guard<std::mutex> lk(mutex);
	freeWorldObjects();
	this->loaded = loadSceneWorldObjects(this->scene, jscene, logger);
	return this->loaded;
}

bool SceneState::isLoaded() { return this->loaded; }

bool SceneState::clean() {
	std::lock_guard<std::mutex> lk(mutex);
	freeWorldObjects();
	this->loaded = false;
	return true;
}

Scene SceneState::clone() {
	std::lock_guard<std::mutex> lk(mutex);
	return this->scene;
}

void SceneState::cameraChanged(bool flag, int* iter) {
	std::lock_guard<std::mutex> lk(mutex);
	if (iter && scene.camera.changed) *iter = 1;
	scene.camera.changed = flag;
}

void SceneState::resetCamera() { std::lock_guard<std::mutex> lk(mutex); scene.camera.reset(); }
void SceneState::setCamPositionX(float x) { std::lock_guard<std::mutex> lk(mutex); scene.camera.position.x = x; scene.camera.changed = true; }
void SceneState::setCamPositionY(float y) { std::lock_guard<std::mutex> lk(mutex); scene.camera.position.y = y; scene.camera.changed = true; }
void SceneState::setCamPositionZ(float z) { std::lock_guard<std::mutex> lk(mutex); scene.camera.position.z = z; scene.camera.changed = true; }
void SceneState::setCamVang(float ang) { std::lock_guard<std::mutex> lk(mutex); scene.camera.v_ang = ang; scene.camera.refresh(); }
void SceneState::setCamHang(float ang) { std::lock_guard<std::mutex> lk(mutex); scene.camera.h_ang = ang; scene.camera.refresh(); }
void SceneState::updateMaxRayBnc(int val) { std::lock_guard<std::mutex> lk(mutex); scene.camera.max_ray_bounces = val; scene.camera.changed = true; }
void SceneState::updateAajitter(float val) {
	if (val < 0.0f) return;
	std::lock_guard<std::mutex> lk(mutex);
	scene.camera.projection.aa_jitter = val;
	scene.camera.changed = true;
}
void SceneState::updateGamma(float gamma) {
	if (gamma < 0.0f) return;
	std::lock_guard<std::mutex> lk(mutex);
	scene.camera.projection.gamma_corr = gamma;
}
void SceneState::togglePrevMode(bool val) { std::lock_guard<std::mutex> lk(mutex); scene.camera.preview_mode = val; scene.camera.changed = true; }
void SceneState::toggleAABBmode(bool val) { std::lock_guard<std::mutex> lk(mutex); scene.camera.aabb_mode = val; scene.camera.changed = true; }
void SceneState::toggleTextures(bool val) { std::lock_guard<std::mutex> lk(mutex); scene.camera.texture_enabled = val; scene.camera.changed = true; }
void SceneState::overrideResolution(int w, int h) { std::lock_guard<std::mutex> lk(mutex); scene.camera.overrideResolution(w, h); }
