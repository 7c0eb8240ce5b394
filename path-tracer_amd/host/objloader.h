// Wavefront OBJ/MTL reader standing in for assimp in the scene loader.
//
// The reference imports meshes through assimp (reference: src/world_load.cpp:415-423,
// flags Triangulate | JoinIdenticalVertices | SortByPType | CalcTangentSpace; material
// keys read at :198-236, faces/normals/uvs at :289-364). assimp is a third-party
// dependency that is NOT in /root/reference (headers only, no source, version
// unknown; a Windows DLL is vendored), so its behaviour is restated from its
// documented OBJ semantics and from what the reference's own scenes need:
//   * triangles come out in FILE ORDER (one aiMesh per object/usemtl run, runs in
//     creation order, faces in order inside a run) -> primitive id = running index;
//   * polygons are fan-triangulated from their first vertex: (0,i,i+1);
//   * points and lines (faces with < 3 vertices) are dropped (SortByPType);
//   * material 0 is "DefaultMaterial" (Kd 0.6, d 1, Ni 1, Ns 0, Ke 0); MTL materials follow
//     in file order; absent keys keep those defaults;
//   * `mtllib X` that cannot be opened falls back to <obj basename>.mtl
//     (scenes/example/cube.obj names a non-existent materials.mtl; cube.mtl sits beside it);
//   * map_Ka -> ambient, map_Kd -> diffuse, map_Ke / map_emissive -> emissive texture;
//     `refl`, bump and the rest are ignored; backslashes in paths become '/'.
// Parity at this boundary is UNPINNED (no assimp here, and the reference has no tests).
#pragma once
#include <string>
#include <vector>

struct ObjMaterial {
	std::string name;
	float Kd[3];
	float Ke[3];
	float d;   // opacity
	float Ni;  // refraction index
	float Ns;  // shininess as written in the file
	std::string map_ambient, map_diffuse, map_emissive;
	ObjMaterial() : d(1.0f), Ni(1.0f), Ns(0.0f) {
		Kd[0] = Kd[1] = Kd[2] = 0.6f;
		Ke[0] = Ke[1] = Ke[2] = 0.0f;
	}
};

struct ObjTriangle {
	float p[3][3];   // positions (untransformed)
	float n[3][3];   // normals (valid if has_normals)
	float t[3][2];   // texture coordinates (valid if has_uv)
	int material;    // index into ObjScene::materials
	bool has_normals, has_uv;
};

struct ObjScene {
	std::vector<ObjMaterial> materials; // [0] = DefaultMaterial
	std::vector<ObjTriangle> triangles; // file order
	int num_meshes;                     // object/usemtl runs that produced faces
};

// Parses `path` (and its material library). Returns false and sets err on I/O failure.
bool loadObjFile(const std::string& path, ObjScene& out, std::string& err);
// Parses MTL text into materials appended to `out`.
void parseMtlText(const char* text, size_t len, std::vector<ObjMaterial>& out);
