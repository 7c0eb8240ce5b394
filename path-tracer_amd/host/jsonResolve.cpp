// reference: src/jsonResolve.cpp:5-48 — 3-arrays, 1-arrays broadcast, anything else → 0;
// material type strings diff/spec/trans/lumi, default Diffuse.
#include "jsonResolve.h"

float3 resolveFloat3(const Json::Value& jval) {
	if (!jval.isArray()) return make_float3(0);
	if (jval.size() == 3) return make_float3(jval[0].asFloat(), jval[1].asFloat(), jval[2].asFloat());
	if (jval.size() == 1) return make_float3(jval[0].asFloat());
	return make_float3(0);
}

glmlite::vec3 resolveGlmVec3(const Json::Value& jval) {
	float3 v = resolveFloat3(jval);
	return glmlite::vec3{v.x, v.y, v.z};
}

float resolveFloat(const Json::Value& jval) {
	return jval.asFloat();
}

MaterialType resolveMatType(const Json::Value& jval) {
	return resolveMatType(jval.asString());
}

MaterialType resolveMatType(const std::string& t) {
	if (t == "diff") return Diffuse;
	if (t == "spec") return Specular;
	if (t == "trans") return Transparent;
	if (t == "lumi") return Luminescent;
	return Diffuse;
}
