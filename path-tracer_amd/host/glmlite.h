// The handful of glm 0.9.8.5 operations the reference's scene loader and camera
// use (reference: src/world_load.cpp:84-103,310-345,461; src/camera.cpp:59-66),
// re-derived here with glm's evaluation order so mesh vertices, normals and the
// KD-tree built from them come out bit for bit the same (build with
// -ffp-contract=off). glm itself is not vendored.
//
// Formulas followed (reference: external/include/glm/...):
//   translate/rotate/scale  gtc/matrix_transform.inl:10-87   (post-multiply: M = M * T)
//   mat4 * vec4             detail/type_mat4x4.inl:526-537   ((c0*x + c1*y) + (c2*z + c3*w))
//   inverse(mat4)           detail/func_matrix.inl:297-356   (cofactors, 1/det, scale)
//   normalize(vec3)         detail/func_geometric.inl:94     (v * (1/sqrt(dot)))
//   radians                 detail/func_trigonometric.inl:16 (deg * float(0.0174532925199432957...))
#pragma once
#include <cmath>

namespace glmlite {

struct vec3 { float x, y, z; };
struct vec4 {
	float x, y, z, w;
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
};

static inline vec4 operator*(vec4 a, float s) { return vec4{a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline vec4 operator*(vec4 a, vec4 b) { return vec4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline vec4 operator+(vec4 a, vec4 b) { return vec4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline vec4 operator-(vec4 a, vec4 b) { return vec4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }

// column-major: c[i] is column i, c[i][j] is row j of column i (glm's m[i][j])
struct mat4 {
	vec4 c[4];
	mat4() {
		c[0] = vec4{1, 0, 0, 0}; c[1] = vec4{0, 1, 0, 0};
		c[2] = vec4{0, 0, 1, 0}; c[3] = vec4{0, 0, 0, 1};
	}
	vec4& operator[](int i) { return c[i]; }
	const vec4& operator[](int i) const { return c[i]; }
};

static inline float radians(float degrees) {
	return degrees * static_cast<float>(0.01745329251994329576923690768489);
}

static inline vec3 normalize(vec3 v) {
	float d = v.x * v.x + v.y * v.y + v.z * v.z;
	float inv = 1.0f / sqrtf(d);
	return vec3{v.x * inv, v.y * inv, v.z * inv};
}

static inline mat4 translate(const mat4& m, vec3 v) {
	mat4 r(m);
	r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
	return r;
}

static inline mat4 rotate(const mat4& m, float angle, vec3 v) {
	const float a = angle;
	const float c = cosf(a);
	const float s = sinf(a);
	vec3 axis = normalize(v);
	float omc = 1.0f - c;
	vec3 temp{omc * axis.x, omc * axis.y, omc * axis.z};

	float r00 = c + temp.x * axis.x;
	float r01 = temp.x * axis.y + s * axis.z;
	float r02 = temp.x * axis.z - s * axis.y;
	float r10 = temp.y * axis.x - s * axis.z;
	float r11 = c + temp.y * axis.y;
	float r12 = temp.y * axis.z + s * axis.x;
	float r20 = temp.z * axis.x + s * axis.y;
	float r21 = temp.z * axis.y - s * axis.x;
	float r22 = c + temp.z * axis.z;

	mat4 r;
	r[0] = m[0] * r00 + m[1] * r01 + m[2] * r02;
	r[1] = m[0] * r10 + m[1] * r11 + m[2] * r12;
	r[2] = m[0] * r20 + m[1] * r21 + m[2] * r22;
	r[3] = m[3];
	return r;
}

static inline mat4 scale(const mat4& m, vec3 v) {
	mat4 r;
	r[0] = m[0] * v.x;
	r[1] = m[1] * v.y;
	r[2] = m[2] * v.z;
	r[3] = m[3];
	return r;
}

static inline vec4 mul(const mat4& m, vec4 v) {
	vec4 mul0 = m[0] * v.x;
	vec4 mul1 = m[1] * v.y;
	vec4 add0 = mul0 + mul1;
	vec4 mul2 = m[2] * v.z;
	vec4 mul3 = m[3] * v.w;
	vec4 add1 = mul2 + mul3;
	return add0 + add1;
}

static inline mat4 transpose(const mat4& m) {
	mat4 r;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j)
			r[i][j] = m[j][i];
	return r;
}

static inline mat4 inverse(const mat4& m) {
	float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
	float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
	float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];

	float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
	float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
	float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];

	float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
	float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
	float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];

	float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
	float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
	float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];

	float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
	float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
	float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];

	float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
	float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
	float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

	vec4 Fac0{Coef00, Coef00, Coef02, Coef03};
	vec4 Fac1{Coef04, Coef04, Coef06, Coef07};
	vec4 Fac2{Coef08, Coef08, Coef10, Coef11};
	vec4 Fac3{Coef12, Coef12, Coef14, Coef15};
	vec4 Fac4{Coef16, Coef16, Coef18, Coef19};
	vec4 Fac5{Coef20, Coef20, Coef22, Coef23};

	vec4 Vec0{m[1][0], m[0][0], m[0][0], m[0][0]};
	vec4 Vec1{m[1][1], m[0][1], m[0][1], m[0][1]};
	vec4 Vec2{m[1][2], m[0][2], m[0][2], m[0][2]};
	vec4 Vec3{m[1][3], m[0][3], m[0][3], m[0][3]};

	vec4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
	vec4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
	vec4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
	vec4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;

	vec4 SignA{+1, -1, +1, -1};
	vec4 SignB{-1, +1, -1, +1};
	mat4 Inverse;
	Inverse[0] = Inv0 * SignA;
	Inverse[1] = Inv1 * SignB;
	Inverse[2] = Inv2 * SignA;
	Inverse[3] = Inv3 * SignB;

	vec4 Row0{Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]};
	vec4 Dot0 = m[0] * Row0;
	float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
	float OneOverDeterminant = 1.0f / Dot1;

	mat4 r;
	for (int i = 0; i < 4; ++i) r[i] = Inverse[i] * OneOverDeterminant;
	return r;
}

} // namespace glmlite
