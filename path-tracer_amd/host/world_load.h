// Scene loading entry points (reference: src/world_load.h:10-62).
#pragma once
#include "world.h"
#include "glmlite.h"
#include <string>

class DialogLogger; // optional progress sink (see dialoglogger.h); may be NULL

#define MAX_OBJECTS_NUM   32
#define OBJECTS_TYPES_NUM 2

bool loadSceneWorldObjects(Scene&, const Json::Value&, DialogLogger*);
bool parseJsonScene(std::string, Json::Value&, std::string&);

void freeWorldObjects();

bool loadSphereObj(void*, WorldObject&);
bool loadTriangleMeshObj(void*, WorldObject&);

// kd tree functions
// reference copyKDTreeToCuda(KDNode*) uploaded a pointer tree the traversal never read;
// here it packs + uploads nothing by itself and only reports what ptx_scene_upload will take.
KDNode* copyKDTreeToCuda(KDNode*);
void freeKDTree(KDNode*);
void freeCudaKDTree(KDNode*);
KDNode* buildKDTree(Triangle*, int*, int, int, int&, int&);
void flatenKDTree(KDNode*, KDNode*);

typedef struct {
	WorldObjType type;
	void* worldObjectInfo;
} WorldObjectDescription;

typedef struct {
	WorldObjectDescription sources[MAX_OBJECTS_NUM];
	int num_objects;
	// dont reorder values, because they correspond enum types
	bool (*loadFuncMapping[OBJECTS_TYPES_NUM])(void*, WorldObject&);
} WorldObjectsSources;

/* object infos */
typedef struct {
	float radius;
	float3 position;
	Material material;
} SphereObjInfo;

typedef struct {
	char src_filename[256];
	glmlite::mat4 transform;
} TriangleMeshObjInfo;

extern WorldObjectsSources world_obj_sources;

// one entry of a mesh's "transform" list: kind 0 translate(v0,v1,v2), 1 rotate(v0 degrees about (v1,v2,v3)),
// 2 scale(v0,v1,v2); composeTransform applies them in order as M = M * T (reference: src/world_load.cpp:84-101)
typedef struct { int kind; float v[4]; } TransformStep;
glmlite::mat4 composeTransform(const TransformStep* steps, int n);

// ---- headless additions (no reference counterpart) -------------------------
// Directory that relative asset paths ("./scenes/...", ".\textures\...") are
// resolved against. The reference resolves them against the process CWD; the
// default here is the same ("").
void setAssetRoot(const std::string& dir);
const std::string& getAssetRoot();
// Host view of the loaded objects (valid until freeWorldObjects()).
const WorldObject* hostWorldObjects(int* num_objects);
// Triangle budget for the procedural stand-in used when lp560.obj is absent.
void setProceduralCarTriangles(int n);
