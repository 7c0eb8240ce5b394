// Render boundary (reference: src/kernel.h:6-14). Same four functions, same call order
// (kernelInit once -> runCUDA per iteration -> kernelCleanUp), but headless: the "PBO" is a
// HOST uchar4 buffer the caller owns, and the accumulation buffer lives behind the C ABI.
#pragma once
#include "world.h"
#include "../../include/ptx_hip.h"

// reference globals (src/kernel.h:6-8): viewPBO_cuda is gone (no GL interop);
// pbo_dptr is a host buffer of width*height uchar4; device_accum_image is the device address
// of the float4 accumulation buffer (informational — owned by the C ABI context).
extern uchar4* pbo_dptr;
extern float4* device_accum_image;

// Returns PTX_OK or a negative PTX_E_* code (the reference returns cudaError_t).
int kernelMain(uchar4* pbo, Scene& scene, int iter);
void kernelInit(const Scene& scene);
void kernelCleanUp();
// reference: src/kernel.cu:94-103 runCUDA (name kept: it is the entry point callers know)
void runCUDA(uchar4* pbo, Scene& scene, int iter);
// reference: src/cudaUtility.cpp:30-37 checkCudaError — true if the last render call failed
bool checkCudaError(const char* context);

// ---- headless additions ---------------------------------------------------------------------
// The per-process device context (created on first use on device hip_device_id).
PtxCtx* kernelContext();
void kernelShutdown();
// Master seed of the counter-based RNG (default 0) and the batching of iterations: runCUDA renders
// exactly one sample per call like the reference; renderSamples() renders a range in one launch.
void kernelSetSeed(uint64_t seed);
uint64_t kernelSeed();
int renderSamples(Scene& scene, int first_sample, int num_samples, bool collect_stats);
// One iteration's sample (what runPathTracing renders): adds exactly sample `sample`; the library may trace up to
// kernelLookAhead() - 1 following samples in the same pass (ptx_render_sample_ahead). Default 16, PTX_LOOK_AHEAD in the
// environment; 1 = the reference's schedule, one pass per call.
#define KERNEL_LOOK_AHEAD_DEFAULT 16
int renderOneSample(Scene& scene, int sample);
void kernelSetLookAhead(int samples);
// The display buffer (reference: the GL pixel buffer behind pbo_dptr, mapped once: qt/previewglwidget.cpp:80-86): page-locked
// host memory of `pixels` uchar4, so that every iteration's tone-mapped image arrives at DMA speed; NULL without a device.
uchar4* kernelAllocPBO(size_t pixels);
void kernelFreePBO(uchar4* pbo);
int kernelLookAhead();
void fillPtxCamera(const Camera& cam, PtxCamera* out);
