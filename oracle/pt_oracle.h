/*
 * pt_oracle.h — CPU oracle for the path-tracing hot path. TEST INFRASTRUCTURE ONLY.
 *
 * A plain C++ restatement of the reference's algorithm (filiprak/path-tracer:
 * src/intersect.cuh, src/radiance.cuh, src/surfaces.cuh, src/pathtracing.cu,
 * src/kernel.cu, src/KDTree.cpp, src/utility.cpp), each function citing the
 * reference lines it follows. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the reported CPU
 * baseline; the product (libptx_hip.so and the host layer) never links or calls it.
 *
 * PARITY STATUS: the reference has no tests, golden vectors or fixtures, and its device
 * code cannot be compiled in this image without writing stand-ins for CUDA/cuRAND/Qt
 * headers (not allowed), so this restatement is pinned only where a reference source
 * file builds unchanged: utility.cpp (fquicksort/fmedian), camera.cpp + jsonResolve.cpp
 * + jsoncpp.cpp (Camera::init / rotate, JSON resolve) — see oracle/Makefile, target
 * _ref. Everything else is "parity unpinned": it follows the reference by reading.
 *
 * Floating point: compiled with -ffp-contract=off; float division and sqrt are IEEE.
 * sin/cos/acos are the fixed polynomials documented in pt_oracle.cpp (the reference's
 * __sinf/__sincosf/acosf bits depend on NVIDIA hardware); the RNG is Philox4x32-10
 * keyed (pixel, sample, dimension) instead of cuRAND XORWOW. Both choices are made
 * identically in the HIP kernels, so oracle and GPU agree bit for bit.
 *
 * Scene input is the same flat description the C ABI takes (include/ptx_hip.h
 * PtxSceneDesc), i.e. the reference's own data model: nodes carry THEIR OWN box and a
 * triangle index list, children boxes are read from the child nodes.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/ptx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/utility.cpp:33-41 */
float orc_fmedian(float* arr, int size);

/* reference: src/KDTree.cpp:84-150 buildKDTree + :152-163 flatenKDTree.
 * abc: n*9 floats, vertices a, b, c of every triangle in primitive order.
 * Outputs are malloc'ed by the oracle (release with orc_free): nodes in preorder-index order with
 * their own boxes, leaf_tris = the triangle lists of the LEAVES concatenated in node order
 * (PtxKDNode::first_tri/num_tris address it; inner nodes carry num_tris of their subtree and
 * first_tri = -1). Returns the number of nodes. */
int orc_kd_build(const float* abc, int n, PtxKDNode** nodes_out, int32_t** leaf_tris_out, int* num_leaf_tris_out,
                 int* max_depth_out);
void orc_free(void* p);

/* reference: src/intersect.cuh:322-350 (object = -1) / :260-319 (object >= 0) */
int orc_trace_rays(const PtxSceneDesc* scene, int n, const float* origins, const float* dirs, int object, PtxHit* hits);

/* reference: src/pathtracing.cu:73-94 (sample < 0) and :107-129 (sample >= 0) */
int orc_generate_rays(const PtxCamera* cam, int sample, uint64_t seed, float* origins, float* dirs);

/* reference: src/surfaces.cuh:37-137 */
int orc_bsdf_eval(int n, PtxBsdfCase* cases);

/* reference: src/radiance.cuh:15-87 for global sample `sample` of every pixel; rgb = w*h*3 floats */
int orc_radiance_sample(const PtxSceneDesc* scene, const PtxCamera* cam, int sample, uint64_t seed, float* rgb, int nthreads);

/* reference: src/pathtracing.cu:132-153 tracePaths + src/kernel.cu:59-81, num_samples iterations:
 * accum (w*h*4 floats, in/out) += radiance of samples [first_sample, first_sample+num_samples) in
 * sample order. stats may be NULL; counters are always collected. nthreads <= 0 -> hardware_concurrency. */
int orc_render(const PtxSceneDesc* scene, const PtxCamera* cam, int first_sample, int num_samples, uint64_t seed,
               float* accum_rgba, int nthreads, PtxStats* stats);

/* reference: src/kernel.cu:40-54 writeImageToPBO */
int orc_tonemap(const float* accum_rgba, int n_pixels, int iter, float gamma, unsigned char* rgba8);

/* reference: src/cutility.cuh:7-15 wang_hash */
uint32_t orc_wang_hash(uint32_t seed);
void orc_vec3_ops(const float* a, const float* b, float s, float* out20);

/* building blocks exposed for unit tests */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_u01(uint32_t x);
void orc_sincos(float theta, float* s, float* c);
float orc_acos01(float x);
int orc_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
