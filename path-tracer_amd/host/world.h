// Host data model of a scene (reference: src/world.h:10-127).
// Enum values, field names and meanings follow the reference so that the
// loader, the KD-tree build and the render entry points read the same way.
// What changes is where the data goes: instead of AoS structs copied verbatim
// to the GPU, the loader keeps these on the host and ptx_scene_upload()
// (include/ptx_hip.h) re-packs them into the SoA layout DESIGN.md describes.
#pragma once
#include "main.h"
#include "camera.h"
#include "vecmath.h"

/* World Object types, important: dont reorder values */
typedef enum { SphereObj = 0, TriangleMeshObj } WorldObjType;

/* Material types */
typedef enum { Luminescent = 0, Diffuse, Specular, Transparent } MaterialType;

typedef struct {
	MaterialType type;
	float3 color;         // mtl: Kd
	int cuda_texture_obj; // -1 = none, else index into the scene's texture table (name kept from the reference)
	float3 emittance;     // mtl: Ke
	float reflect_factor; // mtl: d
	float refract_index;  // mtl: Ni
	float sharpness;      // mtl: Ns / 1000, clamped to [0,1]
} Material;

typedef struct {
	float3 a, b, c, e1, e2;          // vertices and edges e1 = b - a, e2 = c - a
	float3 norm_a, norm_b, norm_c;   // vertex normals
	float2 tx_a, tx_b, tx_c;         // vertex texture coordinates
	int material_idx;                // index into the owning object's materials
} Triangle;

typedef struct {
	float3 bounds[2]; // [0] = min, [1] = max
} BBox;

typedef struct KDNode {
	BBox bbox;
	int* trg_idxs; // triangle indices stored in this node (owned, malloc'ed)
	int num_trgs;
	KDNode* left;
	KDNode* right;
	int idx;       // preorder index = slot in the flat array
	int left_idx;  // -1 = none
	int right_idx; // -1 = none
} KDNode;

typedef struct {
	Triangle* triangles; // host array (owned by the loader)
	int num_triangles;
	KDNode* kdroot;       // pointer tree (host)
	KDNode* flat_kd_root; // flat array indexed by KDNode::idx (host); trg_idxs point into the pointer tree's lists
	int num_nodes;
	int max_depth;
} MeshGeometryData;

typedef struct {
	float radius;
	float3 position;
} SphereGeometryData;

typedef struct {
	WorldObjType type;
	Material* materials; // host array (owned by the loader)
	int num_materials;
	void* geometry_data; // SphereGeometryData* or MeshGeometryData* (host)
} WorldObject;

typedef struct {
	// Reference: device array of WorldObject. Here: opaque handle to the scene
	// uploaded through the C ABI (a PtxScene*), owned by the loader until
	// freeWorldObjects(). Name kept so call sites read like the reference's.
	void* dv_wobjects_ptr;
	int num_wobjects;
	Camera camera;
} Scene;

typedef struct {
	float3 originPoint;
	float3 direction;
	float3 inv_direction;
	int sign[3];
} Ray;
