// See objloader.h for the semantics this reader defines.
#include "objloader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace {

bool readWholeFile(const std::string& path, std::vector<char>& buf) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf.resize((size_t)n + 1);
	size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
	fclose(f);
	buf[got] = 0;
	buf.resize(got + 1);
	return true;
}

inline const char* skipBlank(const char* p) { while (*p == ' ' || *p == '\t') ++p; return p; }
inline bool isEol(char c) { return c == '\n' || c == '\r' || c == 0; }

std::string restOfLine(const char* p) {
	p = skipBlank(p);
	const char* e = p;
	while (!isEol(*e)) ++e;
	while (e > p && (e[-1] == ' ' || e[-1] == '\t')) --e;
	return std::string(p, e);
}

std::string normalisePath(std::string s) {
	for (size_t i = 0; i < s.size(); ++i)
		if (s[i] == '\\') s[i] = '/';
	return s;
}

int readFloats(const char* p, float* out, int maxn) {
	int n = 0;
	while (n < maxn) {
		p = skipBlank(p);
		if (isEol(*p)) break;
		char* e = 0;
		float v = strtof(p, &e);
		if (e == p) break;
		out[n++] = v;
		p = e;
	}
	return n;
}

} // namespace

void parseMtlText(const char* text, size_t len, std::vector<ObjMaterial>& out) {
	const char* p = text;
	const char* end = text + len;
	ObjMaterial* cur = 0;
	while (p < end && *p) {
		p = skipBlank(p);
		const char* line = p;
		const char* kw_end = p;
		while (!isEol(*kw_end) && *kw_end != ' ' && *kw_end != '\t') ++kw_end;
		std::string kw(line, kw_end);
		if (kw == "newmtl") {
			out.push_back(ObjMaterial());
			cur = &out.back();
			cur->name = restOfLine(kw_end);
		} else if (cur) {
			float v[3];
			if (kw == "Kd") { if (readFloats(kw_end, v, 3) == 3) memcpy(cur->Kd, v, sizeof(v)); }
			else if (kw == "Ke") { if (readFloats(kw_end, v, 3) == 3) memcpy(cur->Ke, v, sizeof(v)); }
			else if (kw == "d") { if (readFloats(kw_end, v, 1) == 1) cur->d = v[0]; }
			else if (kw == "Tr") { if (readFloats(kw_end, v, 1) == 1) cur->d = 1.0f - v[0]; }
			else if (kw == "Ni") { if (readFloats(kw_end, v, 1) == 1) cur->Ni = v[0]; }
			else if (kw == "Ns") { if (readFloats(kw_end, v, 1) == 1) cur->Ns = v[0]; }
			else if (kw == "map_Ka") cur->map_ambient = normalisePath(restOfLine(kw_end));
			else if (kw == "map_Kd") cur->map_diffuse = normalisePath(restOfLine(kw_end));
			else if (kw == "map_Ke" || kw == "map_emissive") cur->map_emissive = normalisePath(restOfLine(kw_end));
		}
		while (p < end && *p && *p != '\n') ++p;
		if (p < end && *p == '\n') ++p;
	}
}

namespace {

struct FaceVert { int v, vt, vn; };

// parses "v", "v/vt", "v//vn", "v/vt/vn"; 0 = absent; negative = relative
const char* parseFaceVert(const char* p, FaceVert& fv, bool& ok) {
	fv.v = fv.vt = fv.vn = 0;
	char* e = 0;
	long v = strtol(p, &e, 10);
	if (e == p) { ok = false; return p; }
	fv.v = (int)v;
	p = e;
	if (*p == '/') {
		++p;
		if (*p != '/') {
			long t = strtol(p, &e, 10);
			if (e != p) { fv.vt = (int)t; p = e; }
		}
		if (*p == '/') {
			++p;
			long n = strtol(p, &e, 10);
			if (e != p) { fv.vn = (int)n; p = e; }
		}
	}
	ok = true;
	return p;
}

inline int resolveIndex(int idx, int count) {
	if (idx > 0) return idx - 1;
	if (idx < 0) return count + idx;
	return -1;
}

} // namespace

bool loadObjFile(const std::string& path, ObjScene& out, std::string& err) {
	std::vector<char> buf;
	if (!readWholeFile(path, buf)) { err = "Unable to open file: " + path; return false; }

	std::string dir;
	size_t slash = path.find_last_of("/\\");
	if (slash != std::string::npos) dir = path.substr(0, slash + 1);
	std::string base = (slash == std::string::npos) ? path : path.substr(slash + 1);
	size_t dot = base.find_last_of('.');
	std::string stem = (dot == std::string::npos) ? base : base.substr(0, dot);

	out.materials.clear();
	out.triangles.clear();
	out.num_meshes = 0;
	ObjMaterial def;
	def.name = "DefaultMaterial";
	out.materials.push_back(def);
	std::map<std::string, int> matIndex;
	matIndex[def.name] = 0;

	std::vector<float> pos, nrm, uv;
	int curMat = 0;
	bool runHasFaces = false; // current object/usemtl run produced a face yet?
	std::vector<FaceVert> face;

	const char* p = buf.data();
	while (*p) {
		p = skipBlank(p);
		if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
			float v[3] = {0, 0, 0};
			readFloats(p + 1, v, 3);
			pos.push_back(v[0]); pos.push_back(v[1]); pos.push_back(v[2]);
		} else if (p[0] == 'v' && p[1] == 'n') {
			float v[3] = {0, 0, 0};
			readFloats(p + 2, v, 3);
			nrm.push_back(v[0]); nrm.push_back(v[1]); nrm.push_back(v[2]);
		} else if (p[0] == 'v' && p[1] == 't') {
			float v[3] = {0, 0, 0};
			readFloats(p + 2, v, 3);
			uv.push_back(v[0]); uv.push_back(v[1]);
		} else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
			face.clear();
			const char* q = p + 1;
			for (;;) {
				q = skipBlank(q);
				if (isEol(*q)) break;
				FaceVert fv; bool ok;
				q = parseFaceVert(q, fv, ok);
				if (!ok) break;
				face.push_back(fv);
			}
			if (face.size() >= 3) {
				if (!runHasFaces) { runHasFaces = true; out.num_meshes++; }
				const int nv = (int)pos.size() / 3, nn = (int)nrm.size() / 3, nt = (int)uv.size() / 2;
				for (size_t i = 1; i + 1 < face.size(); ++i) {
					const FaceVert* fvs[3] = {&face[0], &face[i], &face[i + 1]};
					ObjTriangle tr;
					memset(&tr, 0, sizeof(tr));
					tr.material = curMat;
					tr.has_normals = true;
					tr.has_uv = true;
					bool valid = true;
					for (int k = 0; k < 3; ++k) {
						int vi = resolveIndex(fvs[k]->v, nv);
						if (vi < 0 || vi >= nv) { valid = false; break; }
						memcpy(tr.p[k], &pos[3 * vi], 3 * sizeof(float));
						int ni = resolveIndex(fvs[k]->vn, nn);
						if (ni >= 0 && ni < nn) memcpy(tr.n[k], &nrm[3 * ni], 3 * sizeof(float));
						else tr.has_normals = false;
						int ti = resolveIndex(fvs[k]->vt, nt);
						if (ti >= 0 && ti < nt) memcpy(tr.t[k], &uv[2 * ti], 2 * sizeof(float));
						else tr.has_uv = false;
					}
					if (valid) out.triangles.push_back(tr);
				}
			}
		} else if (!strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
			std::string name = restOfLine(p + 6);
			std::map<std::string, int>::iterator it = matIndex.find(name);
			int m = it == matIndex.end() ? 0 : it->second;
			if (m != curMat) { curMat = m; runHasFaces = false; }
		} else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || isEol(p[1]))) {
			runHasFaces = false;
		} else if (!strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
			std::string lib = normalisePath(restOfLine(p + 6));
			std::vector<char> mbuf;
			if (!readWholeFile(dir + lib, mbuf)) {
				if (!readWholeFile(dir + stem + ".mtl", mbuf))
					mbuf.clear();
			}
			if (!mbuf.empty()) {
				size_t first = out.materials.size();
				parseMtlText(mbuf.data(), mbuf.size() - 1, out.materials);
				for (size_t i = first; i < out.materials.size(); ++i)
					if (!matIndex.count(out.materials[i].name))
						matIndex[out.materials[i].name] = (int)i;
			}
		}
		while (*p && *p != '\n') ++p;
		if (*p == '\n') ++p;
	}
	return true;
}
