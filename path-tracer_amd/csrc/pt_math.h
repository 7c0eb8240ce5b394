// Device float maths for the gfx950 path tracer.
//
// Every expression here is written out operation by operation and the file is
// compiled with -ffp-contract=off: traversal decisions (which leaves a ray
// visits, which triangle wins) must be reproducible bit for bit against the CPU
// oracle, so no FMA contraction and no reassociation is allowed. Division and
// sqrt are IEEE correctly rounded (hipcc default for fp32).
//
// Evaluation order follows the reference's float3 helpers
// (reference: external/include/cutil_math.h:322-416): dot = (x*x' + y*y') + z*z',
// normalize = v * (1/sqrt(dot)), reflect = i - (2n)*dot(n,i).
//
// sin/cos/acos: the reference calls __sinf/__sincosf/acosf (reference:
// src/surfaces.cuh:24-29,64), whose bits depend on NVIDIA hardware. Here they are
// fixed polynomials (Cephes single-precision coefficients, public domain) evaluated
// in plain float, identical in the CPU oracle, so whole paths are reproducible.
// Max abs error ~2e-7 on the ranges used (theta in [0, 2pi], x in [0, 1]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 mk3(float s) { return mk3(s, s, s); }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV f3 cross3(f3 a, f3 b) {
	return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// IEEE correctly-rounded square root. NOTE: HIP's __fsqrt_rn() lowers to a bare v_sqrt_f32
// (about 1 ulp) on gfx950 with ROCm 7.2; __builtin_sqrtf gets the compiler's fix-up sequence.
PT_DEV float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_DEV float length3(f3 v) { return pt_sqrt(dot3(v, v)); }

// 1.0f / a in 5 instructions instead of the 10 of the IEEE division sequence: v_rcp_f32 (1 ulp) and
// two Newton steps in fma arithmetic. EQUAL to the correctly rounded quotient for every float with
// PT_RCP_LO <= |a| <= PT_RCP_HI: ptx_selftest_rcp compares all ~2^31 of them on the device
// (tests/test_gpu_parity.py::test_fast_reciprocal_is_exact). Callers fall back to 1.0f / a outside.
#define PT_RCP_LO 5.0e-8f                  /* = PT_TRI_EPSILON: smaller determinants are rejected before use */
#define PT_RCP_HI 8.5070591730234616e37f   /* 2^126: beyond it the quotient is subnormal */
PT_DEV float pt_rcp_newton(float a) {
	float r = __builtin_amdgcn_rcpf(a);
	float e = __builtin_fmaf(-a, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	e = __builtin_fmaf(-a, r, 1.0f);
	r = __builtin_fmaf(e, r, r);
	return r;
}
// The IEEE quotient 1.0f / a for |a| >= PT_RCP_LO (smaller |a|: unspecified, callers discard it). Determinants
// beyond 2^126 do not occur in practice; if one lane of the wave has one, the wave takes the division.
PT_DEV float pt_rcp_det(float a) {
	if (__ballot(__builtin_fabsf(a) > PT_RCP_HI) != 0ull) return 1.0f / a;
	return pt_rcp_newton(a);
}
// 1.0f / a for any a: the five-instruction form where it is proven equal (the wave takes the division if one of its lanes
// holds a value outside the checked range, a zero or a NaN: both paths give the same bits inside it, so which one a wave
// takes never shows in a result). Lengths and ray directions are inside the range in practice.
PT_DEV bool pt_rcp_in_range(float a) { const float m = __builtin_fabsf(a); return m >= PT_RCP_LO && m <= PT_RCP_HI; }
PT_DEV float pt_rcp(float a) {
	if (__ballot(!pt_rcp_in_range(a)) != 0ull) return 1.0f / a;
	return pt_rcp_newton(a);
}
PT_DEV f3 pt_rcp3(f3 a) {
	if (__ballot(!(pt_rcp_in_range(a.x) && pt_rcp_in_range(a.y) && pt_rcp_in_range(a.z))) != 0ull) return mk3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
	return mk3(pt_rcp_newton(a.x), pt_rcp_newton(a.y), pt_rcp_newton(a.z));
}
PT_DEV f3 normalize3(f3 v) {
	float invLen = pt_rcp(pt_sqrt(dot3(v, v)));   // == 1.0f / sqrt(dot(v, v)) bit for bit
	return v * invLen;
}
PT_DEV f3 reflect3(f3 i, f3 n) { return i - (2.0f * n) * dot3(n, i); }

// ---- fixed-polynomial trigonometry ---------------------------------------------------------
// sin and cos of theta for theta in [0, 2*pi + small]: quadrant k = round(theta * 2/pi),
// r = (theta - k*PIO2_HI) - k*PIO2_LO, |r| <= pi/4 (+eps), then Cephes sinf/cosf kernels.
PT_DEV void pt_sincos(float theta, float* s, float* c) {
	const float TWO_OVER_PI = 0.636619772367581343f;
	const float PIO2_HI = 1.5707962512969970703125f;      // float(pi/2)
	const float PIO2_LO = 7.5497894158615963e-08f;        // pi/2 - PIO2_HI
	float kf = __builtin_rintf(theta * TWO_OVER_PI);
	int k = (int)kf;
	float r = (theta - kf * PIO2_HI) - kf * PIO2_LO;
	float z = r * r;
	float sp = -1.9515295891e-4f;
	sp = sp * z + 8.3321608736e-3f;
	sp = sp * z + -1.6666654611e-1f;
	float sr = r + (r * z) * sp;
	float cp = 2.443315711809948e-5f;
	cp = cp * z + -1.388731625493765e-3f;
	cp = cp * z + 4.166664568298827e-2f;
	float cr = (1.0f - 0.5f * z) + (z * z) * cp;
	float ss, cc;
	switch (k & 3) {
		case 0: ss = sr; cc = cr; break;
		case 1: ss = cr; cc = -sr; break;
		case 2: ss = -sr; cc = -cr; break;
		default: ss = -cr; cc = sr; break;
	}
	*s = ss;
	*c = cc;
}

PT_DEV float pt_sin(float theta) {
	float s, c;
	pt_sincos(theta, &s, &c);
	return s;
}

// acos(x) for x in [0, 1] (values above 1 are clamped to 1, giving 0: the reference's
// acosf would return NaN there and its fmin() would then pick the other operand, which
// is the same choice min(max_cone, pi/2 - 0) makes because max_cone <= pi/2).
PT_DEV float pt_acos01(float x) {
	const float PIO2 = 1.5707963267948966f;
	if (x > 1.0f) x = 1.0f;
	if (x > 0.5f) {
		float z = 0.5f * (1.0f - x);
		float sq = pt_sqrt(z);
		float p = 4.2163199048e-2f;
		p = p * z + 2.4181311049e-2f;
		p = p * z + 4.5470025998e-2f;
		p = p * z + 7.4953002686e-2f;
		p = p * z + 1.6666752422e-1f;
		float as = sq + (sq * z) * p; // asin(sqrt(z))
		return 2.0f * as;
	} else {
		float z = x * x;
		float p = 4.2163199048e-2f;
		p = p * z + 2.4181311049e-2f;
		p = p * z + 4.5470025998e-2f;
		p = p * z + 7.4953002686e-2f;
		p = p * z + 1.6666752422e-1f;
		float as = x + (x * z) * p;   // asin(x)
		return PIO2 - as;
	}
}

// ---- counter-based RNG -----------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), keyed by the 64-bit master seed, counter =
// (pixel index, global sample index, dimension, 0) where dimension = bounce number for the
// bounce's BSDF uniforms and 0xFFFFFFFF for the sub-pixel jitter. Replaces the reference's
// per-launch curand_init(hash + threadId) XORWOW streams (reference: src/pathtracing.cu:98-104,
// src/kernel.cu:75-76): the stream of a (pixel, sample) no longer depends on launch shape or on
// which GPU renders it, which is what makes the spp split across GPUs exact.
struct u4 { uint32_t x, y, z, w; };
#ifndef PT_PHILOX_LOCAL_KEYS
#define PT_PHILOX_LOCAL_KEYS 1
#endif

PT_DEV u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
	const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if PT_PHILOX_LOCAL_KEYS
	// The key is wave-uniform (the master seed), so the twenty round keys are too - and the compiler, seeing them loop-invariant,
	// kept them in scalar registers across the kernels' main loops, where there is no room: they were spilled to vector lanes
	// and read back one by one (16 ... 27 v_readlane_b32 per call). Opaque here, they are re-derived per call by scalar adds.
	asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
	for (int i = 0; i < 10; ++i) {
		uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
		uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
		uint32_t n0 = hi1 ^ c1 ^ k0;
		uint32_t n1 = lo1;
		uint32_t n2 = hi0 ^ c3 ^ k1;
		uint32_t n3 = lo0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += W0; k1 += W1;
	}
	u4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
	return r;
}

// uniform in (0, 1], 24 bits, exactly representable (curand_uniform's range; reference:
// src/surfaces.cuh:24,26,56,123 consume it).
PT_DEV float u01(uint32_t x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }

#define PT_DIM_JITTER 0xFFFFFFFFu
