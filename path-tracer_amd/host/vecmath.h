// Host-side float vector math for the MI355X path tracer.
//
// The reference leans on NVIDIA's cutil_math.h for float3 operators
// (reference: external/include/cutil_math.h:322-416). That header is
// NVIDIA-licensed and is not reused here; these are our own definitions with
// the SAME evaluation order, because the KD-tree build and the camera maths
// must be reproducible bit for bit (compile with -ffp-contract=off):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)    = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
//   a / s         = a * (1.0f / s)              (reciprocal, then multiply)
//   normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
//   reflect(i,n)  = i - (2.0f*n) * dot(n,i)
#pragma once
#include <cmath>

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };

static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float2 make_float2(float s) { return float2{s, s}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float3 make_float3(float s) { return float3{s, s, s}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float4 make_float4(float3 a, float w = 0.0f) { return float4{a.x, a.y, a.z, w}; }

static inline float3 operator+(float3 a, float3 b) { return float3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return float3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator-(float3 a) { return float3{-a.x, -a.y, -a.z}; }
static inline float3 operator*(float3 a, float3 b) { return float3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator*(float3 a, float s) { return float3{a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return float3{a.x * s, a.y * s, a.z * s}; }
static inline void operator+=(float3& a, float3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
static inline void operator-=(float3& a, float3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; }
static inline void operator*=(float3& a, float s) { a.x *= s; a.y *= s; a.z *= s; }
static inline float3 operator/(float3 a, float s) { float inv = 1.0f / s; return a * inv; }

static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
	return float3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(float3 v) { return sqrtf(dot(v, v)); }
static inline float3 normalize(float3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
static inline float3 reflect(float3 i, float3 n) { return i - 2.0f * n * dot(n, i); }
static inline float clampf(float f, float a, float b) {
	float m = f < b ? f : b;      // fminf(f,b) as "a<b?a:b"
	return a > m ? a : m;         // fmaxf(a,m) as "a>b?a:b"
}
