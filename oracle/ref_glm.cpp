// extern "C" window onto the glm 0.9.8.5 operations the reference's scene loader applies to mesh vertices and
// normals (src/world_load.cpp:84-101 translate/rotate/scale with M = M*T, :310-341 mat4 x vec4, :461
// transpose(inverse(M))), from the reference's own external/include/glm compiled unchanged.
// TEST INFRASTRUCTURE ONLY. No reference code in this file.
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <string.h>

// ops: n records of 5 floats (kind, a, b, c, d): kind 0 translate(a,b,c); 1 rotate(radians(a), axis (b,c,d));
// 2 scale(a,b,c). Applied in order as the reference does. out16 = M (column-major), inv16 = transpose(inverse(M)).
extern "C" void ref_glm_transform(const float* ops, int n, float* out16, float* inv16) {
	glm::mat4 transform;
	for (int i = 0; i < n; ++i) {
		const float* o = ops + 5 * i;
		if (o[0] == 0.0f) transform = glm::translate(transform, glm::vec3(o[1], o[2], o[3]));
		else if (o[0] == 1.0f) transform = glm::rotate(transform, glm::radians(o[1]), glm::vec3(o[2], o[3], o[4]));
		else transform = glm::scale(transform, glm::vec3(o[1], o[2], o[3]));
	}
	glm::mat4 inv_transform = glm::transpose(glm::inverse(transform));
	memcpy(out16, &transform[0][0], 16 * sizeof(float));
	memcpy(inv16, &inv_transform[0][0], 16 * sizeof(float));
}

// out[4*i..] = M * vec4(v[3*i], v[3*i+1], v[3*i+2], 1.0f)
extern "C" void ref_glm_apply(const float* m16, const float* v3, int n, float* out4) {
	glm::mat4 m;
	memcpy(&m[0][0], m16, 16 * sizeof(float));
	for (int i = 0; i < n; ++i) {
		glm::vec4 r = m * glm::vec4(v3[3 * i], v3[3 * i + 1], v3[3 * i + 2], 1.0f);
		out4[4 * i] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
	}
}
