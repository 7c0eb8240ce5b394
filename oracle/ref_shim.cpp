// extern "C" window onto the reference functions that oracle/Makefile's `ref` target compiles from
// /root/reference unchanged (utility.cpp, camera.cpp, jsonResolve.cpp, jsoncpp.cpp). TEST
// INFRASTRUCTURE ONLY. This file contains no reference code: it includes the reference's headers
// by path at build time and forwards plain-C arguments.
#include "camera.h"      // reference: src/camera.h
#include "jsonResolve.h" // reference: src/jsonResolve.h
#include "utility.h"     // reference: src/utility.h
#include "cutil_math.h"  // reference: external/include/cutil_math.h (float3 helpers of every hot-path function)
#include "cutility.cuh"  // reference: src/cutility.cuh (wang_hash; its __device__ printf helpers compile as host code)
#include <sstream>
#include <string.h>

extern "C" float ref_fmedian(float* arr, int size) { return fmedian(arr, size); }

// out[0..11] = position, direction, up, right; out[12..19] = pixel_size.x, pixel_size.y, half.x, half.y,
// screen_dist, aa_jitter, gamma_corr, max_ray_bounces; iout = width, height, num_pixels, preview, aabb, tex
extern "C" int ref_camera_init(const char* camera_json, float* out, int* iout) {
	Json::Reader reader;
	Json::Value root;
	std::istringstream is(camera_json);
	if (!reader.parse(is, root, false)) return -1;
	Camera cam;
	cam.init(root);
	float3 v[4] = {cam.position, cam.direction, cam.up, cam.right};
	for (int i = 0; i < 4; ++i) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
	out[12] = cam.projection.pixel_size.x; out[13] = cam.projection.pixel_size.y;
	out[14] = cam.projection.screen_halfsize.x; out[15] = cam.projection.screen_halfsize.y;
	out[16] = cam.projection.screen_dist; out[17] = cam.projection.aa_jitter; out[18] = cam.projection.gamma_corr;
	out[19] = (float)cam.max_ray_bounces;
	iout[0] = cam.projection.width; iout[1] = cam.projection.height; iout[2] = cam.projection.num_pixels;
	iout[3] = cam.preview_mode; iout[4] = cam.aabb_mode; iout[5] = cam.texture_enabled;
	return 0;
}

// camera after init + rotateH(h_deg) + rotateV(v_deg) + move(dx,dy,dz): out = position, direction, up, right
extern "C" int ref_camera_rotate(const char* camera_json, float h_deg, float v_deg, const float* move, float* out) {
	Json::Reader reader;
	Json::Value root;
	std::istringstream is(camera_json);
	if (!reader.parse(is, root, false)) return -1;
	Camera cam;
	cam.init(root);
	cam.rotateH(h_deg);
	cam.rotateV(v_deg);
	cam.move(make_float3(move[0], move[1], move[2]));
	float3 v[4] = {cam.position, cam.direction, cam.up, cam.right};
	for (int i = 0; i < 4; ++i) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
	return 0;
}

// resolveFloat3 / resolveFloat / resolveMatType on member `key` of a JSON object text
extern "C" int ref_resolve(const char* json_text, const char* key, float* f3out, float* fout, int* mat_type) {
	Json::Reader reader;
	Json::Value root;
	std::istringstream is(json_text);
	if (!reader.parse(is, root, false)) return -1;
	float3 v = resolveFloat3(root[key]);
	f3out[0] = v.x; f3out[1] = v.y; f3out[2] = v.z;
	*fout = root[key].isNumeric() ? resolveFloat(root[key]) : 0.0f;
	*mat_type = root[key].isString() ? (int)resolveMatType(root[key]) : -1;
	return 0;
}

// The float3 helpers the reference's hot path is built from (external/include/cutil_math.h:322-416), on one
// pair (a, b) and scalar s. out[0..19] = dot(a,b), length(a), cross(a,b).xyz, normalize(a).xyz,
// reflect(a,b).xyz, (a*s).xyz, (a/s).xyz, (a+b).x, (a-b).x, (a*b).x
extern "C" void ref_vec3_ops(const float* pa, const float* pb, float s, float* out) {
	const float3 a = make_float3(pa[0], pa[1], pa[2]), b = make_float3(pb[0], pb[1], pb[2]);
	const float3 c = cross(a, b), n = normalize(a), r = reflect(a, b), m = a * s, q = a / s, ad = a + b, sb = a - b, ml = a * b;
	out[0] = dot(a, b); out[1] = length(a);
	out[2] = c.x; out[3] = c.y; out[4] = c.z;
	out[5] = n.x; out[6] = n.y; out[7] = n.z;
	out[8] = r.x; out[9] = r.y; out[10] = r.z;
	out[11] = m.x; out[12] = m.y; out[13] = m.z;
	out[14] = q.x; out[15] = q.y; out[16] = q.z;
	out[17] = ad.x; out[18] = sb.x; out[19] = ml.x;
}

extern "C" unsigned int ref_wang_hash(unsigned int seed) { return wang_hash(seed); }
