// Thread-safe scene state (reference: src/scenestate.h:9-47, a QObject guarded by a QMutex;
// here std::mutex, same method names and meanings).
#pragma once
#include <mutex>
#include "dialoglogger.h"
#include "world.h"

class SceneState {
public:
	SceneState();
	~SceneState();

	bool load(const Json::Value& jscene, DialogLogger* logger);
	bool isLoaded();
	bool clean();

	Scene clone();

	void cameraChanged(bool, int* = NULL);

	void resetCamera();
	void setCamPositionX(float);
	void setCamPositionY(float);
	void setCamPositionZ(float);
	void setCamVang(float);
	void setCamHang(float);
	void updateMaxRayBnc(int);
	void updateAajitter(float);
	void updateGamma(float);
	void togglePrevMode(bool);
	void toggleAABBmode(bool);
	void toggleTextures(bool);

	// headless addition: BASELINE configs force resolutions the scene files do not carry
	void overrideResolution(int width, int height);

private:
	Scene scene;
	std::mutex mutex;
	bool loaded;
};

extern SceneState sceneState;
