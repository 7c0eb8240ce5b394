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

// reference: src/world_load.cpp:41-115
void initWorldObjSources(Scene& scene, const Json::Value& jscene) {
	world_obj_sources.loadFuncMapping[SphereObj] = loadSphereObj;
	world_obj_sources.loadFuncMapping[TriangleMeshObj] = loadTriangleMeshObj;

	int num_jobjects = jscene["objects"].isArray() ? (int)jscene["objects"].size() : 0;
	if (num_jobjects > MAX_OBJECTS_NUM) num_jobjects = MAX_OBJECTS_NUM;

	int obj_idx = 0;
	for (int i = 0; i < num_jobjects; i++) {
		const Json::Value& jobj = jscene["objects"][i];
		bool render = jobj["render"].asBool();
		if (!render) continue;
		std::string type = jobj["type"].asString();

		if (type == "sphere") {
			world_obj_sources.sources[obj_idx].type = SphereObj;
			Material msphere;
			msphere.color = resolveFloat3(jobj["material"]["Kd"]);
			msphere.cuda_texture_obj = -1;
			msphere.type = resolveMatType(jobj["material"]["type"]);
			msphere.emittance = resolveFloat3(jobj["material"]["Ke"]);
			msphere.reflect_factor = resolveFloat(jobj["material"]["d"]);
			msphere.refract_index = resolveFloat(jobj["material"]["Ni"]);
			msphere.sharpness = clampf(resolveFloat(jobj["material"]["Ns"]) / 1000.0f, 0.0f, 1.0f);

			SphereObjInfo* sinfo = (SphereObjInfo*)malloc(sizeof(SphereObjInfo));
			sinfo->material = msphere;
			sinfo->position = resolveFloat3(jobj["position"]);
			sinfo->radius = resolveFloat(jobj["radius"]);
			world_obj_sources.sources[obj_idx].worldObjectInfo = sinfo;
			obj_idx++;
		} else if (type == "mesh") {
			world_obj_sources.sources[obj_idx].type = TriangleMeshObj;
			TriangleMeshObjInfo* minfo = (TriangleMeshObjInfo*)malloc(sizeof(TriangleMeshObjInfo));
			std::string src = jobj["src"].asString();
			strncpy(minfo->src_filename, src.c_str(), sizeof(minfo->src_filename) - 1);
			minfo->src_filename[sizeof(minfo->src_filename) - 1] = 0;

			glmlite::mat4 transform;
			if (jobj["transform"].isArray()) {
				for (unsigned t = 0; t < jobj["transform"].size(); t++) {
					const Json::Value& jtrans = jobj["transform"][t];
					if (!jtrans.isObject() || jtrans.size() != 1) continue;
					std::string trans_type = jtrans.getMemberNames()[0];
					if (trans_type == "translate")
						transform = glmlite::translate(transform, resolveGlmVec3(jtrans[trans_type]));
					else if (trans_type == "rotate")
						transform = glmlite::rotate(transform, glmlite::radians(resolveFloat(jtrans[trans_type][0])),
						                            resolveGlmVec3(jtrans[trans_type][1]));
					else if (trans_type == "scale")
						transform = glmlite::scale(transform, resolveGlmVec3(jtrans[trans_type]));
				}
			}
			minfo->transform = transform;
			world_obj_sources.sources[obj_idx].worldObjectInfo = minfo;
			obj_idx++;
		} else {
			world_obj_sources.num_objects = obj_idx;
			freeWorldObjects();
			throw scene_file_error("Unknown object type, supported types: mesh/sphere");
		}
	}
	scene.num_wobjects = world_obj_sources.num_objects = obj_idx;
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

	HostTexture tex;
	tex.width = width;
	tex.height = height;
	tex.rgba.resize((size_t)width * height * 4);
	for (int row = 0; row < height; row++) {
		for (int col = 0; col < width; col++) {
			size_t index = (size_t)channels * ((size_t)row * width + col);
			float rgba[4] = {0.0f, 0.0f, 0.0f, 1.0f};
			for (int channel = 0; channel < channels; channel++) rgba[channel] = img[index + channel];
			size_t host_index = ((size_t)(height - row - 1) * width + col) * 4;
			float* o = &tex.rgba[host_index];
			if (channels < 2) { o[0] = rgba[0]; o[1] = rgba[0]; o[2] = rgba[0]; o[3] = rgba[3]; }
			else if (channels < 3) { o[0] = rgba[0]; o[1] = rgba[0]; o[2] = rgba[0]; o[3] = rgba[1]; }
			else { o[0] = rgba[0]; o[1] = rgba[1]; o[2] = rgba[2]; o[3] = rgba[3]; }
		}
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

		mat_ptr[i].cuda_texture_obj = -1;
		const std::string* texture_path = NULL;
		if (!material.map_ambient.empty()) texture_path = &material.map_ambient;
		else if (!material.map_diffuse.empty()) texture_path = &material.map_diffuse;
		else if (!material.map_emissive.empty()) texture_path = &material.map_emissive;
		if (texture_path) {
			REPORT(info("    Loading texture: %s", texture_path->c_str()));
			mat_ptr[i].cuda_texture_obj = loadTexture(*texture_path);
			if (mat_ptr[i].cuda_texture_obj < 0) {
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
		float Ns = material.Ns;
		mat_ptr[i].type = resolveMatType(mat_type);
		mat_ptr[i].color = make_float3(material.Kd[0], material.Kd[1], material.Kd[2]);
		mat_ptr[i].emittance = make_float3(material.Ke[0], material.Ke[1], material.Ke[2]);
		mat_ptr[i].reflect_factor = material.d;
		mat_ptr[i].refract_index = material.Ni;
		mat_ptr[i].sharpness = clampf(Ns / 1000.0f, 0.0f, 1.0f);
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

// reference: src/world_load.cpp:385-410
bool loadSphereObj(void* objInfo, WorldObject& result) {
	SphereObjInfo* info = (SphereObjInfo*)objInfo;
	REPORT(info("Loading Sphere Object: r=%f, pos=[%f,%f,%f]", info->radius, info->position.x, info->position.y, info->position.z));
	SphereGeometryData* gptr = (SphereGeometryData*)malloc(sizeof(SphereGeometryData));
	gptr->position = info->position;
	gptr->radius = info->radius;
	Material* mat = (Material*)malloc(sizeof(Material));
	*mat = info->material;
	host_mem_ptrs.push_back(mat);
	host_mem_ptrs.push_back(gptr);
	result.type = SphereObj;
	result.materials = mat;
	result.num_materials = 1;
	result.geometry_data = gptr;
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
