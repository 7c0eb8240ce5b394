// Which short sequences equal the correctly rounded square root for EVERY float in a range? (exhaustive, on the device)
// Candidates against __builtin_sqrtf (the compiler's v_sqrt_f32 + scale / neighbour fix-up, ~15 instructions):
//   A  y = rsq(x); s = x y; h = y / 2; s += (x - s s) h                (one residual step, fma arithmetic)
//   B  A with a second residual step
//   C  s = sqrt_hw(x); h = rcp(s) / 2; s += (x - s s) h
//   D  s = sqrt_hw(x); h = rsq(x) / 2; s += (x - s s) h
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/sqrt_probe.hip -o tools/micro/sqrt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
template <int K> __device__ float cand(float x) {
	if (K == 0) { float y = __builtin_amdgcn_rsqf(x); float s = x * y, h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
	if (K == 1) { float y = __builtin_amdgcn_rsqf(x); float s = x * y, h = 0.5f * y; float r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s); r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
	if (K == 2) { float s = __builtin_amdgcn_sqrtf(x); float h = 0.5f * __builtin_amdgcn_rcpf(s); float r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
	{ float s = __builtin_amdgcn_sqrtf(x); float h = 0.5f * __builtin_amdgcn_rsqf(x); float r = __builtin_fmaf(-s, s, x); return __builtin_fmaf(r, h, s); }
}
template <int K> __global__ void probe(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* out) {
	unsigned long long bad = 0, n = 0; uint32_t first = 0xffffffffu;
	for (unsigned long long b = lo_bits + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi_bits; b += (unsigned long long)gridDim.x * blockDim.x) {
		const float x = __uint_as_float((uint32_t)b);
		const float want = __builtin_sqrtf(x), got = cand<K>(x);
		++n;
		if (__float_as_uint(want) != __float_as_uint(got)) { ++bad; if ((uint32_t)b < first) first = (uint32_t)b; }
	}
	atomicAdd(&out[0], n); atomicAdd(&out[1], bad); if (first != 0xffffffffu) atomicMin(&out[2], (unsigned long long)first);
}
__global__ void ref_sample(const float* x, float* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = __builtin_sqrtf(x[i]); }
template <int K> void run(const char* name, float lo, float hi) {
	unsigned long long* d; (void)hipMalloc(&d, 24); unsigned long long h[3] = {0, 0, ~0ull};
	(void)hipMemcpy(d, h, 24, hipMemcpyHostToDevice);
	uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
	probe<K><<<4096, 256>>>(lb, hb, d);
	(void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
	float fb; uint32_t b = (uint32_t)h[2]; memcpy(&fb, &b, 4);
	printf("%s  [%g, %g]: %llu checked, %llu differ%s", name, lo, hi, h[0], h[1], h[1] ? "" : "\n");
	if (h[1]) printf(" (first at %g = 0x%08x)\n", fb, b);
	(void)hipFree(d);
}
int main() {
	// the device's __builtin_sqrtf against the host's correctly rounded sqrtf on 2^24 spread values
	{
		const int n = 1 << 24; float* hx = new float[n]; float* hy = new float[n];
		for (int i = 0; i < n; ++i) { uint32_t b = 0x00800000u + (uint32_t)(((unsigned long long)i * 0x7effffffull) >> 24) ; memcpy(&hx[i], &b, 4); }
		float *dx, *dy; (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
		ref_sample<<<n / 256, 256>>>(dx, dy, n); (void)hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost);
		long bad = 0; for (int i = 0; i < n; ++i) { float w = sqrtf(hx[i]); if (memcmp(&w, &hy[i], 4)) ++bad; }
		printf("device __builtin_sqrtf vs host sqrtf on %d values over all normal floats: %ld differ\n", n, bad);
	}
	const float ranges[][2] = {{1e-30f, 1e30f}, {1.17549435e-38f, 3.0e38f}};
	for (auto& r : ranges) {
		run<0>("A rsq, 1 step ", r[0], r[1]); run<1>("B rsq, 2 steps", r[0], r[1]); run<2>("C sqrt + rcp  ", r[0], r[1]); run<3>("D sqrt + rsq  ", r[0], r[1]);
	}
	return 0;
}
