// Render boundary over the C ABI (reference: src/kernel.cu:22-103).
#include "kernel.h"
#include "pathtracing.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

int hip_device_id = 0;

uchar4* pbo_dptr = NULL;
float4* device_accum_image = NULL;

static PtxCtx* g_ctx = NULL;
static uint64_t g_seed = 0;
// samples the one-sample-per-call loop may trace ahead (ptx_render_sample_ahead); PTX_LOOK_AHEAD in the environment, or
// kernelSetLookAhead; 1 = every call traces its own sample only, as the reference does
static int g_look_ahead = -1;
static int g_last_status = PTX_OK;
static bool g_render_ready = false;

PtxCtx* kernelContext() {
	if (!g_ctx) {
		if (ptx_ctx_create(hip_device_id, &g_ctx) != PTX_OK) {
			fprintf(stderr, "\n>  HIP RESULT FAIL: %s\n\n", ptx_last_error());
			g_ctx = NULL;
		}
	}
	return g_ctx;
}

void kernelShutdown() {
	if (g_ctx) { ptx_ctx_destroy(g_ctx); g_ctx = NULL; }
	g_render_ready = false;
}

void kernelSetSeed(uint64_t seed) { g_seed = seed; }

uchar4* kernelAllocPBO(size_t pixels) {
	PtxCtx* ctx = kernelContext();
	void* p = NULL;
	if (!ctx || ptx_host_alloc(ctx, pixels * sizeof(uchar4), &p) != PTX_OK) return NULL;
	return (uchar4*)p;
}
void kernelFreePBO(uchar4* pbo) { if (g_ctx && pbo) ptx_host_free(g_ctx, pbo); }
void kernelSetLookAhead(int samples) { g_look_ahead = samples < 1 ? 1 : samples; }
int kernelLookAhead() {
	if (g_look_ahead < 0) {
		const char* e = getenv("PTX_LOOK_AHEAD");
		const int v = e && *e ? atoi(e) : KERNEL_LOOK_AHEAD_DEFAULT;
		g_look_ahead = v < 1 ? 1 : v;
	}
	return g_look_ahead;
}
uint64_t kernelSeed() { return g_seed; }

void fillPtxCamera(const Camera& cam, PtxCamera* out) {
	memset(out, 0, sizeof(*out));
	const float3* src[4] = {&cam.position, &cam.direction, &cam.up, &cam.right};
	float* dst[4] = {out->position, out->direction, out->up, out->right};
	for (int i = 0; i < 4; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
	out->width = cam.projection.width;
	out->height = cam.projection.height;
	out->pixel_size = cam.projection.pixel_size.x;
	out->screen_half[0] = cam.projection.screen_halfsize.x;
	out->screen_half[1] = cam.projection.screen_halfsize.y;
	out->screen_dist = cam.projection.screen_dist;
	out->aa_jitter = cam.projection.aa_jitter;
	out->gamma_corr = cam.projection.gamma_corr;
	out->max_ray_bounces = cam.max_ray_bounces;
	out->preview_mode = cam.preview_mode ? 1 : 0;
	out->aabb_mode = cam.aabb_mode ? 1 : 0;
	out->texture_enabled = cam.texture_enabled ? 1 : 0;
}

// reference: src/pathtracing.cu:44-62 initPathTracing: the render-side buffers. The reference allocates the ray
// and pixel-midpoint buffers here and generates the primary rays; primaries are regenerated in-kernel, so what is
// left is the float accumulation buffer behind the C ABI (ptx_render_init allocates and clears it).
void initPathTracing(const Scene& scene) {
	PtxCtx* ctx = kernelContext();
	if (!ctx) { g_last_status = PTX_E_NO_DEVICE; return; }
	PtxCamera pc;
	fillPtxCamera(scene.camera, &pc);
	g_last_status = ptx_render_init(ctx, &pc, NULL);
	g_render_ready = g_last_status == PTX_OK;
	device_accum_image = (float4*)ptx_accum_device_ptr(ctx);
	if (!g_render_ready) fprintf(stderr, "\n>  HIP RESULT FAIL: %s\n\n", ptx_last_error());
}

// reference: src/pathtracing.cu:64-70 cleanUpPathTracing
void cleanUpPathTracing() {
	if (g_ctx) ptx_render_cleanup(g_ctx);
	device_accum_image = NULL;
	g_render_ready = false;
}

// reference: src/kernel.cu:22-28 kernelInit: accumulation buffer + initPathTracing (one call here: the buffer
// lives behind the same C ABI call)
void kernelInit(const Scene& scene) { initPathTracing(scene); }

// reference: src/kernel.cu:30-35
void kernelCleanUp() { cleanUpPathTracing(); }

static int syncCamera(PtxCtx* ctx, Scene& scene);

unsigned int wang_unhash(unsigned int h) {
	h ^= (h >> 15) ^ (h >> 30);                 // undo seed ^= seed >> 15
	h *= 0xfb699ca5u;                           // inverse of 0x27d4eb2d modulo 2^32
	h ^= (h >> 4) ^ (h >> 8) ^ (h >> 12) ^ (h >> 16) ^ (h >> 20) ^ (h >> 24) ^ (h >> 28);   // undo seed ^= seed >> 4
	h *= 0x38e38e39u;                           // inverse of 9 modulo 2^32
	const unsigned int hi = h >> 16;            // (x ^ 61) ^ (x >> 16): the top half of x is untouched (61 < 2^16)
	return (hi << 16) | ((h & 0xffffu) ^ 61u ^ hi);
}

// reference: src/pathtracing.cu:156-182 runPathTracing
void runPathTracing(Scene& scene, int iterHash, int jitterHash) {
	(void)jitterHash;
	const unsigned int iter = wang_unhash((unsigned int)iterHash);
	if (iter < 1u || iter > 0x7fffffffu) { g_last_status = PTX_E_INVALID; return; }
	renderOneSample(scene, (int)iter - 1);
}

// What one iteration of the reference's loop adds to the accumulation buffer: sample `sample`, exactly. The library may trace
// the samples that follow it in the same pass and keep their radiance parked (ptx_render_sample_ahead): the buffer after this
// call is bit for bit the one-sample-per-call buffer, only the following calls find their work done.
int renderOneSample(Scene& scene, int sample) {
	PtxCtx* ctx = kernelContext();
	if (!ctx || !g_render_ready) return g_last_status = PTX_E_STATE;
	int rc = syncCamera(ctx, scene);
	if (rc != PTX_OK) return g_last_status = rc;
	return g_last_status = ptx_render_sample_ahead(ctx, (const PtxScene*)scene.dv_wobjects_ptr, sample, g_seed, kernelLookAhead());
}

// reference: src/pathtracing.cu:166-172 - camera changed: regenerate primary rays (here: the kernels' camera block)
static int syncCamera(PtxCtx* ctx, Scene& scene) {
	if (!scene.camera.changed) return PTX_OK;
	PtxCamera pc;
	fillPtxCamera(scene.camera, &pc);
	int rc = ptx_render_set_camera(ctx, &pc);
	if (rc == PTX_OK) scene.camera.changed = false;
	return rc;
}

int renderSamples(Scene& scene, int first_sample, int num_samples, bool collect_stats) {
	PtxCtx* ctx = kernelContext();
	if (!ctx || !g_render_ready) return g_last_status = PTX_E_STATE;
	int rc = syncCamera(ctx, scene);
	if (rc != PTX_OK) return g_last_status = rc;
	return g_last_status = ptx_render_samples(ctx, (const PtxScene*)scene.dv_wobjects_ptr, first_sample, num_samples, g_seed,
	                                          collect_stats ? 1 : 0);
}

// reference: src/kernel.cu:59-91 kernelMain: iteration `iter` (1-based) = global sample iter-1
int kernelMain(uchar4* pbo, Scene& scene, int iter) {
	PtxCtx* ctx = kernelContext();
	if (!ctx || !g_render_ready) return g_last_status = PTX_E_STATE;
	if (iter < 1) return g_last_status = PTX_E_INVALID;
	if (iter == 1) {
		int rc = ptx_accum_clear(ctx); // :69-72
		if (rc != PTX_OK) return g_last_status = rc;
	}
	// :74-80 the reference's per-iteration hashes; runPathTracing recovers the iteration from them
	const int iterHash = (int)wang_hash((unsigned int)iter);
	const int jitterHash = (int)wang_hash(wang_hash((unsigned int)iter));
	runPathTracing(scene, iterHash, jitterHash);
	if (g_last_status != PTX_OK) return g_last_status;
	int rc;
	// :83-88 writeImageToPBO
	rc = ptx_tonemap(ctx, iter, scene.camera.projection.gamma_corr, (unsigned char*)pbo);
	return g_last_status = rc;
}

// reference: src/kernel.cu:94-103
void runCUDA(uchar4* pbo, Scene& scene, int iter) {
	if (pbo)
		kernelMain(pbo, scene, iter);
	else {
		printf("Failed to map pbo pointer.\n");
		g_last_status = PTX_E_INVALID;
	}
}

bool checkCudaError(const char* context) {
	if (g_last_status != PTX_OK) {
		fprintf(stderr, "HIP error in %s: %s\n", context, ptx_last_error());
		return true;
	}
	return false;
}
