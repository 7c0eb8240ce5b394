// Scene loading: JSON -> object sources -> host geometry + KD-trees -> ONE packed upload.
// Entry points, globals and error behaviour follow reference src/world_load.cpp:33-589.
//
// What differs by design:
//   * assimp is replaced by objloader.cpp (semantics documented there);
//   * stb_image is replaced by image_io.cpp;
//   * every cudaMalloc/cudaMemcpy of the reference (per material array, per triangle array, per
//     KD node, per texture) becomes a host-side array; when all objects are loaded the scene is
//     flattened into a PtxSceneDesc and handed to ptx_scene_upload() once.
#include "world_load.h"
#include "dialoglogger.h"
#include "errors.h"
#include "hostscene.h"
#include "image_io.h"
#include "jsonResolve.h"
#include "kernel.h"
#include "objloader.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>

void generateProceduralCar(ObjScene& out, int target_triangles); // procedural.cpp

static DialogLogger* reporter = NULL;
#define REPORT(t) if (reporter) reporter->t;

WorldObjectsSources world_obj_sources;

// host resources of the loaded scene (reference: dv_mem_ptrs / dv_textures / dv_mem_kdtrees_ptrs, :36-38)
// heap-allocated and never destroyed: the global sceneState's destructor (another translation unit)
// calls freeWorldObjects() during static destruction and must still find these alive
static std::vector<void*>& host_mem_ptrs = *new std::vector<void*>();
static std::vector<KDNode*>& host_kdtrees = *new std::vector<KDNode*>();
static std::vector<HostTexture>& host_textures = *new std::vector<HostTexture>();
static std::map<std::string, int>& texture_by_path = *new std::map<std::string, int>();
// distinct texture files the CURRENT scene names but that could not be loaded (bench.py checks them
// against the files the reference tree itself lacks, .MISSING_LARGE_BLOBS)
static std::vector<std::string>& failed_textures = *new std::vector<std::string>();
const std::vector<std::string>& failedTextureFiles() { return failed_textures; }
static std::vector<WorldObject>& host_wobjects = *new std::vector<WorldObject>();
static HostSceneFlat& flat_scene = *new HostSceneFlat();
static PtxScene* device_scene = NULL;

static std::string& asset_root = *new std::string();
static int procedural_car_triangles = 400000;
static bool device_upload_enabled = true;

void setAssetRoot(const std::string& dir) {
	asset_root = dir;
	if (!asset_root.empty() && asset_root[asset_root.size() - 1] != '/') asset_root += '/';
}
const std::string& getAssetRoot() { return asset_root; }
void setProceduralCarTriangles(int n) { procedural_car_triangles = n; }
void setDeviceUploadEnabled(bool on) { device_upload_enabled = on; }
const WorldObject* hostWorldObjects(int* num_objects) {
	if (num_objects) *num_objects = (int)host_wobjects.size();
	return host_wobjects.empty() ? NULL : host_wobjects.data();
}
const HostSceneFlat& hostSceneFlat() { return flat_scene; }

static std::string resolvePath(const std::string& p) {
	std::string s = p;
	for (size_t i = 0; i < s.size(); ++i)
		if (s[i] == '\\') s[i] = '/';
	if (!s.empty() && s[0] == '/') return s;
	if (s.size() >= 2 && s[0] == '.' && s[1] == '/') s = s.substr(2);
	return asset_root + s;
}

// One place where the five MTL-style numbers become a Material, for sphere objects (JSON "material" block) and
// for mesh materials (MTL records) alike. Semantics: reference src/world_load.cpp:62-69 and :264-269 —
// Kd -> color, Ke -> emittance, d -> reflect_factor, Ni -> refract_index, sharpness = clamp(Ns / 1000, 0, 1).
struct MaterialFields {
	MaterialType type;
	float3 Kd, Ke;
	float d, Ni, Ns;
};

static Material makeMaterial(const MaterialFields& f, int texture) {
	Material m;
	m.type = f.type;
	m.color = f.Kd;
	m.cuda_texture_obj = texture;
	m.emittance = f.Ke;
	m.reflect_factor = f.d;
	m.refract_index = f.Ni;
	m.sharpness = clampf(f.Ns / 1000.0f, 0.0f, 1.0f);
	return m;
}

// A mesh's "transform" list as plain records (kind, 4 numbers): kind 0 translate(x,y,z), 1 rotate(degrees, axis),
// 2 scale(x,y,z). Entries that are not single-member objects, and members with other names ("xrotate" ...), are
// skipped as the reference skips them (src/world_load.cpp:86-101).
static std::vector<TransformStep> readTransformList(const Json::Value& jlist) {
	std::vector<TransformStep> steps;
	if (!jlist.isArray()) return steps;
	for (unsigned t = 0; t < jlist.size(); ++t) {
		const Json::Value& entry = jlist[t];
		if (!entry.isObject() || entry.size() != 1) continue;
		const std::string key = entry.getMemberNames()[0];
		const Json::Value& arg = entry[key];
		TransformStep st = {-1, {0, 0, 0, 0}};
		if (key == "translate" || key == "scale") {
			const float3 v = resolveFloat3(arg);
			st.kind = key == "scale" ? 2 : 0;
			st.v[0] = v.x; st.v[1] = v.y; st.v[2] = v.z;
		} else if (key == "rotate") {
			const float3 axis = resolveFloat3(arg[1]);
			st.kind = 1;
			st.v[0] = resolveFloat(arg[0]);
			st.v[1] = axis.x; st.v[2] = axis.y; st.v[3] = axis.z;
		}
		if (st.kind >= 0) steps.push_back(st);
	}
	return steps;
}

// M = M * T for every step in list order (glm's post-multiplying translate / rotate / scale, :94-100).
glmlite::mat4 composeTransform(const TransformStep* steps, int n) {
	glmlite::mat4 m;
	for (int i = 0; i < n; ++i) {
		const float* v = steps[i].v;
		switch (steps[i].kind) {
			case 0: m = glmlite::translate(m, glmlite::vec3{v[0], v[1], v[2]}); break;
			case 1: m = glmlite::rotate(m, glmlite::radians(v[0]), glmlite::vec3{v[1], v[2], v[3]}); break;
			case 2: m = glmlite::scale(m, glmlite::vec3{v[0], v[1], v[2]}); break;
			default: break;
		}
	}
	return m;
}

static void* sphereSource(const Json::Value& jobj) {
	const Json::Value& jm = jobj["material"];
	MaterialFields f;
	f.type = resolveMatType(jm["type"]);
	f.Kd = resolveFloat3(jm["Kd"]);
	f.Ke = resolveFloat3(jm["Ke"]);
	f.d = resolveFloat(jm["d"]);
	f.Ni = resolveFloat(jm["Ni"]);
	f.Ns = resolveFloat(jm["Ns"]);
	SphereObjInfo* info = (SphereObjInfo*)malloc(sizeof(SphereObjInfo));
	info->material = makeMaterial(f, -1);
	info->position = resolveFloat3(jobj["position"]);
	info->radius = resolveFloat(jobj["radius"]);
	return info;
}

static void* meshSource(const Json::Value& jobj) {
	TriangleMeshObjInfo* info = (TriangleMeshObjInfo*)malloc(sizeof(TriangleMeshObjInfo));
	snprintf(info->src_filename, sizeof(info->src_filename), "%s", jobj["src"].asString().c_str());
	const std::vector<TransformStep> steps = readTransformList(jobj["transform"]);
	info->transform = composeTransform(steps.data(), (int)steps.size());
	return info;
}

// reference: src/world_load.cpp:41-115 — at most MAX_OBJECTS_NUM entries of "objects" are looked at, those
// with "render": false are passed over, any type other than sphere / mesh aborts the load.
void initWorldObjSources(Scene& scene, const Json::Value& jscene) {
	world_obj_sources.loadFuncMapping[SphereObj] = loadSphereObj;
	world_obj_sources.loadFuncMapping[TriangleMeshObj] = loadTriangleMeshObj;

	const Json::Value& jobjects = jscene["objects"];
	const int listed = jobjects.isArray() ? std::min((int)jobjects.size(), MAX_OBJECTS_NUM) : 0;
	int kept = 0;
	for (int i = 0; i < listed; ++i) {
		const Json::Value& jobj = jobjects[i];
		if (!jobj["render"].asBool()) continue;
		const std::string type = jobj["type"].asString();
		WorldObjectDescription& slot = world_obj_sources.sources[kept];
		if (type == "sphere") { slot.type = SphereObj; slot.worldObjectInfo = sphereSource(jobj); }
		else if (type == "mesh") { slot.type = TriangleMeshObj; slot.worldObjectInfo = meshSource(jobj); }
		else {
			world_obj_sources.num_objects = kept;
			freeWorldObjects();
			throw scene_file_error("Unknown object type, supported types: mesh/sphere");
		}
		++kept;
	}
	scene.num_wobjects = world_obj_sources.num_objects = kept;
}

static void freeWorldObjSources() {
	for (int i = 0; i < world_obj_sources.num_objects; ++i)
		if (world_obj_sources.sources[i].worldObjectInfo != NULL) free(world_obj_sources.sources[i].worldObjectInfo);
	for (int i = 0; i < MAX_OBJECTS_NUM; i++) world_obj_sources.sources[i].worldObjectInfo = NULL;
	world_obj_sources.num_objects = 0;
}

// reference: src/world_load.cpp:130-192 loadTexture. Returns the index into the scene's texture
// table (the reference returns a cudaTextureObject_t), -1 on failure. Rows are flipped and
// 1/2/3-channel images expanded to RGBA exactly as :139-156.
int loadTexture(const std::string src_path) {
	std::string path = resolvePath(src_path);
	std::map<std::string, int>::iterator it = texture_by_path.find(path);
	if (it != texture_by_path.end()) return it->second;

	int width = 0, height = 0, channels = 0;
	std::vector<float> img;
	std::string err;
	if (!loadImageFloat(path, width, height, channels, img, err)) return -1;

	// file row r becomes texture row height-1-r; grey -> (g, g, g, 1), grey+alpha -> (g, g, g, a), RGB -> alpha 1
	HostTexture tex;
	tex.width = width;
	tex.height = height;
	tex.rgba.resize((size_t)width * height * 4);
	static const int pick[4][4] = {{0, 0, 0, -1}, {0, 0, 0, 1}, {0, 1, 2, -1}, {0, 1, 2, 3}};
	const int* from = pick[channels - 1];
	for (int row = 0; row < height; ++row) {
		const float* src = &img[(size_t)channels * width * row];
		float* dst = &tex.rgba[(size_t)4 * width * (height - 1 - row)];
		for (int col = 0; col < width; ++col, src += channels, dst += 4)
			for (int k = 0; k < 4; ++k) dst[k] = from[k] < 0 ? 1.0f : src[from[k]];
	}
	host_textures.push_back(tex);
	int idx = (int)host_textures.size() - 1;
	texture_by_path[path] = idx;
	return idx;
}

// reference: src/world_load.cpp:194-278 loadMaterialsToHost
static Material* loadMaterialsToHost(const ObjScene& oscene) {
	int num_materials = (int)oscene.materials.size();
	if (num_materials == 0) return NULL;
	Material* mat_ptr = (Material*)malloc(num_materials * sizeof(Material));
	for (int i = 0; i < num_materials; ++i) {
		const ObjMaterial& material = oscene.materials[i];
		// name convention name.type, e.g. metal.spec, sun.lumi; no dot -> "default" -> Diffuse
		const std::string& sname = material.name;
		size_t dotpos = sname.find_first_of('.');
		std::string mat_type = "default";
		if (dotpos != std::string::npos && dotpos + 1 < sname.length()) mat_type = sname.substr(dotpos + 1);

		REPORT(info("  Loading material[%d]: name: %s, type: %s", i, sname.c_str(), mat_type.c_str()));

		int texture = -1;
		const std::string* texture_path = NULL;
		if (!material.map_ambient.empty()) texture_path = &material.map_ambient;
		else if (!material.map_diffuse.empty()) texture_path = &material.map_diffuse;
		else if (!material.map_emissive.empty()) texture_path = &material.map_emissive;
		if (texture_path) {
			REPORT(info("    Loading texture: %s", texture_path->c_str()));
			texture = loadTexture(*texture_path);
			if (texture < 0) {
				// printed once per file and process: lp560.mtl names 19 absent TGAs from 41 materials
				static std::map<std::string, int>* warned = new std::map<std::string, int>();
				if ((*warned)[*texture_path]++ == 0) printf("    Warning: texture load failed: %s\n", texture_path->c_str());
				if (std::find(failed_textures.begin(), failed_textures.end(), *texture_path) == failed_textures.end())
					failed_textures.push_back(*texture_path);
				REPORT(warning("    Warning: texture load failed: %s", texture_path->c_str()));
			}
		}

		// reference :235,269: assimp hands back 4x the MTL's Ns and the loader divides by 4 before
		// the /1000 clamp; net effect sharpness = clamp(Ns_file / 1000, 0, 1).
		MaterialFields f;
		f.type = resolveMatType(mat_type);
		f.Kd = make_float3(material.Kd[0], material.Kd[1], material.Kd[2]);
		f.Ke = make_float3(material.Ke[0], material.Ke[1], material.Ke[2]);
		f.d = material.d;
		f.Ni = material.Ni;
		f.Ns = material.Ns;
		mat_ptr[i] = makeMaterial(f, texture);
	}
	return mat_ptr;
}

// reference: src/world_load.cpp:280-382 loadTriangles
static Triangle* loadTriangles(const ObjScene& oscene, const glmlite::mat4& transform, const glmlite::mat4& inv_transform,
                               int& all_trs) {
	all_trs = (int)oscene.triangles.size();
	Triangle* tr = (Triangle*)malloc((all_trs > 0 ? all_trs : 1) * sizeof(Triangle));
	memset(tr, 0, (all_trs > 0 ? all_trs : 1) * sizeof(Triangle));
	for (int i = 0; i < all_trs; ++i) {
		const ObjTriangle& f = oscene.triangles[i];
		glmlite::vec4 g[3];
		for (int k = 0; k < 3; ++k) g[k] = glmlite::mul(transform, glmlite::vec4{f.p[k][0], f.p[k][1], f.p[k][2], 1.0f});
		float3 a = make_float3(g[0].x, g[0].y, g[0].z);
		float3 b = make_float3(g[1].x, g[1].y, g[1].z);
		float3 c = make_float3(g[2].x, g[2].y, g[2].z);
		tr[i].a = a; tr[i].b = b; tr[i].c = c;
		tr[i].e1 = b - a;
		tr[i].e2 = c - a;
		tr[i].material_idx = f.material;
		if (f.has_normals) {
			float3 n[3];
			for (int k = 0; k < 3; ++k) {
				// reference :339-345: inverse-transpose times (n, 1.0f), xyz taken, normalized
				glmlite::vec4 v = glmlite::mul(inv_transform, glmlite::vec4{f.n[k][0], f.n[k][1], f.n[k][2], 1.0f});
				n[k] = normalize(make_float3(v.x, v.y, v.z));
			}
			tr[i].norm_a = n[0]; tr[i].norm_b = n[1]; tr[i].norm_c = n[2];
		} else {
			// the reference leaves normals uninitialised when the file has none; use the face normal
			float3 fn = normalize(cross(tr[i].e1, tr[i].e2));
			tr[i].norm_a = tr[i].norm_b = tr[i].norm_c = fn;
		}
		if (f.has_uv) {
			tr[i].tx_a = make_float2(f.t[0][0], f.t[0][1]);
			tr[i].tx_b = make_float2(f.t[1][0], f.t[1][1]);
			tr[i].tx_c = make_float2(f.t[2][0], f.t[2][1]);
		}
	}
	return tr;
}

// reference: src/world_load.cpp:385-410 — a sphere object owns one material and a (radius, centre) record;
// both live in one host block here (the reference makes two device allocations).
bool loadSphereObj(void* objInfo, WorldObject& result) {
	const SphereObjInfo& src = *(const SphereObjInfo*)objInfo;
	REPORT(info("Loading Sphere Object: r=%f, pos=[%f,%f,%f]", src.radius, src.position.x, src.position.y, src.position.z));
	struct Block { Material material; SphereGeometryData geom; };
	Block* blk = (Block*)malloc(sizeof(Block));
	blk->material = src.material;
	blk->geom.radius = src.radius;
	blk->geom.position = src.position;
	host_mem_ptrs.push_back(blk);
	result.type = SphereObj;
	result.materials = &blk->material;
	result.num_materials = 1;
	result.geometry_data = &blk->geom;
	return true;
}

// reference: src/world_load.cpp:412-507
bool loadTriangleMeshObj(void* objInfo, WorldObject& result) {
	TriangleMeshObjInfo* info = (TriangleMeshObjInfo*)objInfo;
	REPORT(info("Loading Mesh Object: %s", info->src_filename));

	ObjScene oscene;
	std::string err;
	std::string path = resolvePath(info->src_filename);
	if (!loadObjFile(path, oscene, err)) {
		std::string fname(info->src_filename);
		bool is_car = fname.size() >= 9 && fname.substr(fname.size() - 9) == "lp560.obj";
		if (!is_car) {
			printf("  ERROR: Failed to load Mesh Object: %s\n", info->src_filename);
			throw scene_file_error("Failed to load file: " + fname);
		}
		// documented stand-in for the missing headline asset (procedural.cpp)
		oscene.materials.clear();
		oscene.triangles.clear();
		oscene.num_meshes = 0;
		ObjMaterial def;
		def.name = "DefaultMaterial";
		oscene.materials.push_back(def);
		std::string mtl = path.substr(0, path.size() - 3) + "mtl";
		FILE* mf = fopen(mtl.c_str(), "rb");
		if (mf) {
			std::vector<char> buf;
			fseek(mf, 0, SEEK_END);
			long n = ftell(mf);
			fseek(mf, 0, SEEK_SET);
			buf.resize((size_t)n + 1);
			size_t got = fread(buf.data(), 1, (size_t)n, mf);
			buf[got] = 0;
			fclose(mf);
			parseMtlText(buf.data(), got, oscene.materials);
		}
		generateProceduralCar(oscene, procedural_car_triangles);
		printf("  NOTE: %s is not available; using the procedural stand-in (%d triangles, %d materials)\n", info->src_filename,
		       (int)oscene.triangles.size(), (int)oscene.materials.size());
		REPORT(warning("  %s not available: procedural stand-in, %d triangles", info->src_filename, (int)oscene.triangles.size()));
	}

	MeshGeometryData* gptr = (MeshGeometryData*)malloc(sizeof(MeshGeometryData));
	Material* host_materials = loadMaterialsToHost(oscene);
	result.num_materials = (int)oscene.materials.size();
	result.materials = host_materials;
	host_mem_ptrs.push_back(host_materials);

	// inverse-transpose for normals (reference :461)
	glmlite::mat4 inv_transform = glmlite::transpose(glmlite::inverse(info->transform));

	int all_trgs_count = 0;
	gptr->triangles = loadTriangles(oscene, info->transform, inv_transform, all_trgs_count);
	gptr->num_triangles = all_trgs_count;
	host_mem_ptrs.push_back(gptr->triangles);
	int trgs_mem_size = (int)((size_t)all_trgs_count * sizeof(Triangle) / 1024);
	REPORT(info("  Loaded %d triangles in %d meshes. (Host memory taken: %d KB)", gptr->num_triangles, oscene.num_meshes, trgs_mem_size));

	REPORT(info("  Building KD-Tree for all triangles (%d).", all_trgs_count));
	int* trg_idxs = (int*)malloc((all_trgs_count > 0 ? all_trgs_count : 1) * sizeof(int));
	for (int idx = 0; idx < all_trgs_count; idx++) trg_idxs[idx] = idx;
	int num_nodes = 0, max_depth = 0;
	KDNode* host_kdtree = buildKDTree(gptr->triangles, trg_idxs, all_trgs_count, 0, num_nodes, max_depth);
	KDNode* host_flat_tree = (KDNode*)malloc(num_nodes * sizeof(KDNode));
	flatenKDTree(host_kdtree, host_flat_tree);
	int kd_tree_size = (int)((size_t)num_nodes * sizeof(KDNode) / 1024);
	printf("  Finished building KD-Tree (Num. nodes: %d, Max depth: %d, Mem. size: %d KB).\n", num_nodes, max_depth, kd_tree_size);
	REPORT(info("  Finished building KD-Tree (Num. nodes: %d, Max depth: %d).", num_nodes, max_depth));
	gptr->flat_kd_root = host_flat_tree;
	gptr->kdroot = copyKDTreeToCuda(host_kdtree);
	gptr->num_nodes = num_nodes;
	gptr->max_depth = max_depth;
	host_kdtrees.push_back(host_kdtree);
	host_mem_ptrs.push_back(host_flat_tree);
	host_mem_ptrs.push_back(gptr);

	result.type = TriangleMeshObj;
	result.geometry_data = gptr;
	return true;
}

// Flatten the loaded host objects into the C ABI's scene description.
static void flattenScene(const std::vector<WorldObject>& objs, HostSceneFlat& fs) {
	fs = HostSceneFlat();
	for (size_t i = 0; i < objs.size(); ++i) {
		const WorldObject& wo = objs[i];
		PtxObject po;
		memset(&po, 0, sizeof(po));
		po.type = wo.type == SphereObj ? PTX_OBJ_SPHERE : PTX_OBJ_MESH;
		po.first_material = (int)fs.materials.size();
		po.num_materials = wo.num_materials;
		for (int m = 0; m < wo.num_materials; ++m) {
			const Material& hm = wo.materials[m];
			PtxMaterial pm;
			memset(&pm, 0, sizeof(pm));
			pm.type = (int)hm.type;
			pm.color[0] = hm.color.x; pm.color[1] = hm.color.y; pm.color[2] = hm.color.z;
			pm.texture = hm.cuda_texture_obj;
			pm.emittance[0] = hm.emittance.x; pm.emittance[1] = hm.emittance.y; pm.emittance[2] = hm.emittance.z;
			pm.reflect_factor = hm.reflect_factor;
			pm.refract_index = hm.refract_index;
			pm.sharpness = hm.sharpness;
			fs.materials.push_back(pm);
		}
		if (wo.type == SphereObj) {
			const SphereGeometryData* g = (const SphereGeometryData*)wo.geometry_data;
			po.radius = g->radius;
			po.position[0] = g->position.x; po.position[1] = g->position.y; po.position[2] = g->position.z;
		} else {
			const MeshGeometryData* g = (const MeshGeometryData*)wo.geometry_data;
			po.first_triangle = (int)fs.triangles.size();
			po.num_triangles = g->num_triangles;
			for (int t = 0; t < g->num_triangles; ++t) {
				const Triangle& ht = g->triangles[t];
				PtxTriangle pt;
				pt.a[0] = ht.a.x; pt.a[1] = ht.a.y; pt.a[2] = ht.a.z;
				pt.e1[0] = ht.e1.x; pt.e1[1] = ht.e1.y; pt.e1[2] = ht.e1.z;
				pt.e2[0] = ht.e2.x; pt.e2[1] = ht.e2.y; pt.e2[2] = ht.e2.z;
				pt.na[0] = ht.norm_a.x; pt.na[1] = ht.norm_a.y; pt.na[2] = ht.norm_a.z;
				pt.nb[0] = ht.norm_b.x; pt.nb[1] = ht.norm_b.y; pt.nb[2] = ht.norm_b.z;
				pt.nc[0] = ht.norm_c.x; pt.nc[1] = ht.norm_c.y; pt.nc[2] = ht.norm_c.z;
				pt.ta[0] = ht.tx_a.x; pt.ta[1] = ht.tx_a.y;
				pt.tb[0] = ht.tx_b.x; pt.tb[1] = ht.tx_b.y;
				pt.tc[0] = ht.tx_c.x; pt.tc[1] = ht.tx_c.y;
				pt.material = ht.material_idx;
				fs.triangles.push_back(pt);
				const float abc[9] = {ht.a.x, ht.a.y, ht.a.z, ht.b.x, ht.b.y, ht.b.z, ht.c.x, ht.c.y, ht.c.z};
				fs.tri_abc.insert(fs.tri_abc.end(), abc, abc + 9);
			}
			po.first_node = (int)fs.nodes.size();
			po.num_nodes = g->num_nodes;
			po.first_leaf_tri = (int)fs.leaf_tris.size();
			po.max_depth = g->max_depth;
			for (int n = 0; n < g->num_nodes; ++n) {
				const KDNode& hn = g->flat_kd_root[n];
				PtxKDNode pn;
				pn.bmin[0] = hn.bbox.bounds[0].x; pn.bmin[1] = hn.bbox.bounds[0].y; pn.bmin[2] = hn.bbox.bounds[0].z;
				pn.bmax[0] = hn.bbox.bounds[1].x; pn.bmax[1] = hn.bbox.bounds[1].y; pn.bmax[2] = hn.bbox.bounds[1].z;
				pn.left = hn.left_idx;
				pn.right = hn.right_idx;
				pn.num_tris = hn.num_trgs;
				pn.first_tri = -1;
				if (hn.left_idx == -1 && hn.right_idx == -1) {
					pn.first_tri = (int)fs.leaf_tris.size() - po.first_leaf_tri;
					for (int k = 0; k < hn.num_trgs; ++k) fs.leaf_tris.push_back(hn.trg_idxs[k]);
				}
				fs.nodes.push_back(pn);
			}
			po.num_leaf_tris = (int)fs.leaf_tris.size() - po.first_leaf_tri;
		}
		fs.objects.push_back(po);
	}
	for (size_t t = 0; t < host_textures.size(); ++t) {
		PtxTexture pt;
		pt.rgba = host_textures[t].rgba.data();
		pt.width = host_textures[t].width;
		pt.height = host_textures[t].height;
		fs.textures.push_back(pt);
	}
	fs.fillDesc();
}

// reference: src/world_load.cpp:509-530 loadWorldObjects. Returns the device scene handle.
static void* loadWorldObjects() {
	int num_objects = world_obj_sources.num_objects;
	host_wobjects.assign((size_t)num_objects, WorldObject());
	for (int i = 0; i < num_objects; ++i) {
		REPORT(progress((int)(100.0f * (float)i / (float)num_objects)));
		WorldObjType type = world_obj_sources.sources[i].type;
		void* objInfo = world_obj_sources.sources[i].worldObjectInfo;
		if (!(*world_obj_sources.loadFuncMapping[type])(objInfo, host_wobjects[(size_t)i])) return NULL;
	}
	flattenScene(host_wobjects, flat_scene);
	if (!device_upload_enabled) return NULL;
	PtxCtx* ctx = kernelContext();
	if (!ctx) throw scene_file_error(std::string("No HIP device context: ") + ptx_last_error());
	if (ptx_scene_upload(ctx, &flat_scene.desc, &device_scene) != PTX_OK)
		throw scene_file_error(std::string("Scene upload failed: ") + ptx_last_error());
	return device_scene;
}

// reference: src/world_load.cpp:532-551
bool loadSceneWorldObjects(Scene& scene, const Json::Value& jscene, DialogLogger* logger) {
	printSep();
	try {
		reporter = logger;
		initWorldObjSources(scene, jscene);
		scene.camera.init(jscene["camera"]);
		scene.dv_wobjects_ptr = loadWorldObjects();
		REPORT(progress(100));
		reporter = NULL;
		return true;
	} catch (const scene_file_error& e) {
		freeWorldObjects();
		REPORT(error("Scene file error: %s", e.what()));
		fprintf(stderr, "Scene file error: %s\n", e.what());
		reporter = NULL;
		return false;
	}
}

// reference: src/world_load.cpp:553-574
void freeWorldObjects() {
	if (device_scene) {
		ptx_scene_free(kernelContext(), device_scene);
		device_scene = NULL;
	}
	for (size_t i = 0; i < host_kdtrees.size(); ++i) freeKDTree(host_kdtrees[i]);
	host_kdtrees.clear();
	for (size_t i = 0; i < host_mem_ptrs.size(); ++i)
		if (host_mem_ptrs[i]) free(host_mem_ptrs[i]);
	host_mem_ptrs.clear();
	host_textures.clear();
	texture_by_path.clear();
	failed_textures.clear();
	host_wobjects.clear();
	flat_scene = HostSceneFlat();
	freeWorldObjSources();
}

// reference: src/world_load.cpp:576-589
bool parseJsonScene(std::string fpath, Json::Value& root, std::string& errmsg) {
	Json::Reader reader;
	std::ifstream jstream(fpath, std::ifstream::binary);
	if (!jstream.is_open()) {
		errmsg = "Unable to open file: " + fpath;
		return false;
	}
	bool parsingSuccessful = reader.parse(jstream, root, false);
	if (!parsingSuccessful) errmsg = reader.getFormattedErrorMessages();
	jstream.close();
	return parsingSuccessful;
}
