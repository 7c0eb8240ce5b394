/*
 * ptx_hip.h — C ABI of the MI355X (gfx950) path-tracing library, libptx_hip.so.
 *
 * This is the drop-in boundary for the hot path of filiprak/path-tracer:
 * primary-ray generation with jitter, KD-tree traversal + ray/triangle/sphere
 * intersection, the diffuse/specular/refractive bounce loop and the per-pixel
 * float accumulation. Plain C types only: no HIP, torch or C++ types cross it.
 * The C++ host layer (path-tracer_amd/host, reference entry-point names) and
 * the Python harness (ctypes) both bind exactly these symbols.
 *
 * Every function returns PTX_OK (0) or a negative PTX_E_* code and stores a
 * message retrievable with ptx_last_error() (thread-local). Nothing here falls
 * back to a CPU path: without a gfx950 device every compute entry point fails
 * with PTX_E_NO_DEVICE / PTX_E_HIP.
 *
 * Each declaration cites the reference interface (file:line under the
 * reference tree) it replaces.
 */
#ifndef PTX_HIP_H
#define PTX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_OK            0
#define PTX_E_INVALID    -1   /* bad argument / shape mismatch (checked on the host before any launch) */
#define PTX_E_NO_DEVICE  -2   /* no HIP device, or device index out of range */
#define PTX_E_HIP        -3   /* a HIP runtime call failed; ptx_last_error() has hipGetErrorString */
#define PTX_E_STATE      -4   /* call order violated (e.g. render before init) */
#define PTX_E_LIMIT      -5   /* a documented limit was exceeded (KD depth, object count, texture count) */

#define PTX_MAX_OBJECTS   32  /* reference: src/world_load.h:10 MAX_OBJECTS_NUM */
#define PTX_MAX_KD_DEPTH  60  /* reference: src/intersect.cuh:139 `int stack[64]` */

/* reference: src/world.h:10-14 WorldObjType, :17-23 MaterialType (same values) */
enum { PTX_OBJ_SPHERE = 0, PTX_OBJ_MESH = 1 };
enum { PTX_MAT_LUMINESCENT = 0, PTX_MAT_DIFFUSE = 1, PTX_MAT_SPECULAR = 2, PTX_MAT_TRANSPARENT = 3 };

typedef struct PtxCtx PtxCtx;     /* one per GPU: device, stream, render buffers */
typedef struct PtxScene PtxScene; /* device-resident scene in the packed SoA layout */

/* reference: src/camera.h:8-43 (the fields the kernels read) */
typedef struct PtxCamera {
	float position[3];
	float direction[3];
	float up[3];
	float right[3];
	int32_t width, height;
	float pixel_size;        /* projection.pixel_size.x */
	float screen_half[2];    /* projection.screen_halfsize */
	float screen_dist;
	float aa_jitter;
	float gamma_corr;
	int32_t max_ray_bounces;
	int32_t preview_mode;
	int32_t aabb_mode;
	int32_t texture_enabled;
} PtxCamera;

/* reference: src/world.h:28-43 Material */
typedef struct PtxMaterial {
	int32_t type;
	float color[3];
	int32_t texture;         /* -1 = none, else index into PtxSceneDesc::textures */
	float emittance[3];
	float reflect_factor;
	float refract_index;
	float sharpness;
	int32_t _pad;
} PtxMaterial;

/* reference: src/world_load.cpp:131-192 loadTexture — float RGBA texels, rows already flipped */
typedef struct PtxTexture {
	const float* rgba;       /* width*height*4 floats, row-major */
	int32_t width, height;
} PtxTexture;

/* reference: src/world.h:46-51 Triangle (b and c are not needed on the device) */
typedef struct PtxTriangle {
	float a[3], e1[3], e2[3];
	float na[3], nb[3], nc[3];
	float ta[2], tb[2], tc[2];
	int32_t material;        /* index into the owning object's materials */
} PtxTriangle;

/* reference: src/world.h:61-78 KDNode as laid out by flatenKDTree (src/KDTree.cpp:152-163) */
typedef struct PtxKDNode {
	float bmin[3], bmax[3];
	int32_t left, right;     /* indices inside the owning object's node range, -1 = none */
	int32_t first_tri;       /* offset into the owning object's leaf_tris range (leaves only) */
	int32_t num_tris;
} PtxKDNode;

/* reference: src/world.h:100-107 WorldObject + :81-96 geometry data */
typedef struct PtxObject {
	int32_t type;
	int32_t first_material, num_materials;   /* range in PtxSceneDesc::materials */
	float radius;                            /* sphere */
	float position[3];                       /* sphere */
	int32_t first_triangle, num_triangles;   /* mesh: range in ::triangles (original primitive order) */
	int32_t first_node, num_nodes;           /* mesh: range in ::nodes, node 0 of the range is the root */
	int32_t first_leaf_tri, num_leaf_tris;   /* mesh: range in ::leaf_tris */
	int32_t max_depth;                       /* mesh: deepest node level (root = 0) */
} PtxObject;

typedef struct PtxSceneDesc {
	int32_t num_objects;   const PtxObject* objects;
	int32_t num_materials; const PtxMaterial* materials;
	int32_t num_textures;  const PtxTexture* textures;
	int32_t num_triangles; const PtxTriangle* triangles;
	int32_t num_nodes;     const PtxKDNode* nodes;
	int32_t num_leaf_tris; const int32_t* leaf_tris;   /* triangle indices inside the owning object's range */
} PtxSceneDesc;

/* Result of one rayIntersectsScene-equivalent query (parity entry point). */
typedef struct PtxHit {
	int32_t hit;            /* 0/1: the reference's return value "some intersection test succeeded" */
	int32_t object;         /* index of the closest object, -1 if none (possible with hit = 1 when every success lies at an
	                         * infinite distance, e.g. unnormalised 1e18 directions; point/normal/bary are then zero) */
	int32_t prim;           /* ORIGINAL triangle index inside that object, -1 for spheres/miss */
	uint32_t t_bits;        /* bit pattern of length(ipoint - origin) of the closest hit */
	uint32_t leaf_hash;     /* FNV-1a over (object, leaf node idx) in visit order, all meshes */
	uint32_t n_inner;       /* inner-node steps */
	uint32_t n_box;         /* box tests incl. root tests */
	uint32_t n_tri;         /* triangle tests */
	uint32_t n_leaf;        /* leaves visited */
	float point[3];         /* ipoint */
	float normal[3];        /* shading normal */
	float bary[3];
} PtxHit;

/* Whole-render counters (device-side, reduced per wave). */
typedef struct PtxStats {
	uint64_t rays;          /* rayIntersectsScene-equivalent path segments */
	uint64_t paths;         /* (pixel, sample) paths started */
	uint64_t n_inner, n_box, n_tri, n_sphere, n_leaf;   /* filled by the *_stats render variant only */
	uint64_t hit_mesh, hit_sphere, hit_textured;        /* idem */
	uint64_t lane_steps_active, lane_steps_total;       /* idem: traversal steps done by lanes / (64 x the slowest lane's steps per wave round) */
	/* idem, restricted to multi-node meshes (what wf_traverse_kernel does): rays entering their root box,
	 * inner-node steps, box tests below the root, triangle tests, leaves */
	uint64_t trav_rays, trav_inner, trav_box, trav_tri, trav_leaf;
	/* kernel-tuning counters of the traversal kernels (stats variant; meaning may change between versions):
	 * per-lane kernel: diag0 = lanes holding a ray, diag1 = 64 per wave step (lane occupancy of the waves);
	 * cooperative kernel: ticks of the constant-rate shader clock, read by lane 0 of every wave and summed
	 * per phase: diag0 = commit + refill, diag1 = descent, diag2 = leaf phase, diag3 = the part of diag2
	 * spent in the unit steps of leaf mode 5 (tools/phase_clock.py prints the shares) */
	uint64_t diag0, diag1, diag2, diag3;
	/* what the traversal kernel EXECUTED, counting variant only (bench.py's roofline is built on these):
	 *   x_iters       wave iterations of the descent loop (unified kernel: of the item loop)
	 *   x_item_lanes  lanes that fetched and box-tested a 64-B record in them (node records; unified: + leaf halves)
	 *   x_half_lanes  cooperative kernel: lanes that fetched their leaf's four group boxes (128 B); unified: leaf-half records
	 *   x_wait_lanes  unified kernel: lanes waiting for their units, summed over iterations
	 *   x_unit_steps  steps of the unit phase (16 groups of eight triangles each), x_units the groups tested in them
	 *   x_unit_phases rounds with a leaf phase (unified: unit phases), x_turns commit / refill turns
	 * wide traversal (pt_wide.h, variant 14): x_iters = visit iterations, x_item_lanes = wide-node visits, x_half_lanes =
	 *   rays handed to the strict kernel (diag3 of them for a full stack), x_wait_lanes = stacked children dropped at the pop,
	 *   x_unit_steps / x_units as above, x_unit_phases = drains; diag0..2 = clock ticks in turns / node visits / unit steps;
	 *   lane_steps_active = rays handed to the strict kernel because their winning hit was not robust (pt_wide.h) */
	uint64_t x_iters, x_item_lanes, x_half_lanes, x_wait_lanes, x_unit_steps, x_units, x_unit_phases, x_turns;
	/* shading divergence in the segment kernel, counting variant only (the material-sorting question):
	 * waves in which at least one lane shades a hit, the lanes that do, the BSDF branches (diffuse / specular /
	 * transparent) those waves execute summed over waves, and lanes per material type 0..3 */
	uint64_t sh_waves, sh_lanes, sh_branches, sh_type_lanes[4];
	/* segment kernel, counting variant only: ticks of the shader clock read by lane 0 of every wave, summed per phase:
	 * [0] generate / resolve + shade + compact, [1] waiting at the workgroup barriers, [2] intersection of the cheap
	 * objects + state write, [3] queue appends (tools/segment_clock.py prints the shares) */
	uint64_t seg_clk[4];
} PtxStats;

/* BSDF known-answer entry (parity for surfaces.cuh). */
typedef struct PtxBsdfCase {
	float dir[3];           /* incoming ray direction */
	float normal[3];
	float point[3];
	int32_t mat_type;
	float reflect_factor, refract_index, sharpness;
	float u[4];             /* the uniforms, consumed in call order */
	/* outputs */
	float out_origin[3];
	float out_dir[3];
	float out_mask[3];      /* multiplier applied to the path mask by the BSDF itself (1 unless Transparent) */
} PtxBsdfCase;

/* ---- library / device ------------------------------------------------------------------- */
const char* ptx_last_error(void);
const char* ptx_version(void);
/* reference: src/main.cpp:22-41 detectCUDAdevice / src/cudaUtility.cpp:6-28 printCudaDevicesInfo */
int ptx_device_count(void);
int ptx_device_info(int device, char* name, int name_len, int* compute_units, int* clock_khz, uint64_t* total_mem);
int ptx_ctx_create(int device, PtxCtx** out);
int ptx_ctx_destroy(PtxCtx* ctx);

/* ---- scene --------------------------------------------------------------------------------
 * reference: src/world_load.cpp:509-530 loadWorldObjects + the per-object cudaMalloc/cudaMemcpy in
 * loadSphereObj (:385-410), loadTriangleMeshObj (:412-507), loadTexture (:131-192) and
 * flatenKDTree (src/KDTree.cpp:152-163). One packed upload replaces them all. */
int ptx_scene_upload(PtxCtx* ctx, const PtxSceneDesc* desc, PtxScene** out);
/* reference: src/world_load.cpp:553-574 freeWorldObjects */
int ptx_scene_free(PtxCtx* ctx, PtxScene* scene);
int64_t ptx_scene_device_bytes(const PtxScene* scene);
/* Diagnostics (no reference counterpart): how many multi-node meshes of the scene got the surface-area-heuristic wide tree
 * instead of the one collapsed from the reference's KD tree (ptx_ctx_tune "wide_build"; results do not depend on it). */
int ptx_scene_wide_sah_objects(const PtxScene* scene);

/* ---- render -------------------------------------------------------------------------------
 * reference: src/kernel.cu:22-28 kernelInit + src/pathtracing.cu:44-62 initPathTracing.
 * Allocates and zeroes the float4 accumulation buffer (width*height*16 B) unless the caller
 * hands in its own device buffer `external_accum` (e.g. a torch tensor used for the RCCL
 * all-reduce); primary rays are regenerated in-kernel, so no ray buffers exist. */
int ptx_render_init(PtxCtx* ctx, const PtxCamera* cam, void* external_accum);
/* reference: src/pathtracing.cu:166-172 (camera.changed → regenerate rays); resolution must not change */
int ptx_render_set_camera(PtxCtx* ctx, const PtxCamera* cam);
/* reference: src/kernel.cu:30-35 kernelCleanUp */
int ptx_render_cleanup(PtxCtx* ctx);
/* reference: src/kernel.cu:69-72 (iter == 1 → clear) */
int ptx_accum_clear(PtxCtx* ctx);
/* reference: src/kernel.cu:59-81 kernelMain → src/pathtracing.cu:156-182 runPathTracing, called
 * num_samples times: renders global samples [first_sample, first_sample + num_samples) of every
 * pixel (sample i ↔ reference iteration i+1) and adds them, in sample order, into the accumulation
 * buffer. Asynchronous on the context's stream. `collect_stats` != 0 runs the counting variant. */
int ptx_render_samples(PtxCtx* ctx, const PtxScene* scene, int first_sample, int num_samples,
                       uint64_t seed, int collect_stats);
/* reference: ONE call of kernelMain (src/kernel.cu:59-91) in the render thread's loop (qt/cudarunthread.cpp:21-41): adds
 * exactly global sample `sample` to the accumulation buffer, bit for bit what ptx_render_samples(ctx, scene, sample, 1, seed, 0)
 * adds. A one-sample pass cannot fill an MI355X (2.3 ms per sample of the headline frame, against 0.5 ms in a long pass), so
 * the call TRACES AHEAD: a sample outside the window it holds is traced together with the w - 1 that follow it, their radiance
 * stays parked per sample, and the calls for those samples only add their plane. w is 1 at first and doubles while the calls
 * continue the window (up to max_ahead, at most the pass size "wf_batch"; max_ahead <= 1: no look-ahead). A window ends with
 * anything that changes what a sample is or reuses the pass buffers: ptx_render_set_camera / _init / _cleanup,
 * ptx_render_samples, ptx_ctx_tune, ptx_scene_free. Rays traced ahead are counted (ptx_stats_get) when they are traced.
 * Memory: the path state of a pass of max_ahead samples. Asynchronous on the context's stream. */
int ptx_render_sample_ahead(PtxCtx* ctx, const PtxScene* scene, int sample, uint64_t seed, int max_ahead);
int ptx_render_sync(PtxCtx* ctx);
/* accumulation buffer access (headless replacement for the mapped PBO, src/kernel.h:6-8) */
void* ptx_accum_device_ptr(PtxCtx* ctx);
int ptx_accum_download(PtxCtx* ctx, float* rgba_host);
int ptx_accum_upload(PtxCtx* ctx, const float* rgba_host);
/* reference: src/kernel.cu:40-54 writeImageToPBO: uchar = 255*pow(clamp(acc/iter,0,1),gamma), w = 0;
 * written to a HOST rgba8 buffer of width*height*4 bytes, in buffer order (not mirrored). */
int ptx_tonemap(PtxCtx* ctx, int iter, float gamma, unsigned char* rgba8_host);
/* reference: the display buffer the render thread writes, a GL pixel buffer mapped once for the application's life
 * (qt/previewglwidget.cpp:80-86, src/kernel.h:6-8 pbo_dptr). Headless, the image is a host buffer and crosses PCIe in every
 * ptx_tonemap: page-locked memory from here takes that copy at DMA speed (1920x1080: 0.15 ms instead of 0.45 ms into pageable
 * memory). Any host pointer works for ptx_tonemap; this one is faster. Free with ptx_host_free before ptx_ctx_destroy. */
int ptx_host_alloc(PtxCtx* ctx, size_t bytes, void** out);
int ptx_host_free(PtxCtx* ctx, void* p);
int ptx_stats_get(PtxCtx* ctx, PtxStats* out);
int ptx_stats_reset(PtxCtx* ctx);
/* Diagnostics of the LAST wavefront pass (no reference counterpart): per bounce b in [0, max_bounces],
 * live_out[b] = paths still alive after bounce b's segment kernel, queued_out[b] = rays enqueued for
 * KD traversal of multi-node meshes. Arrays of max_bounces + 1 entries. */
int ptx_wavefront_counts(PtxCtx* ctx, int max_bounces, int32_t* live_out, int32_t* queued_out);
/* Per-kernel-class timing of ptx_render_samples (HIP events around every launch on the context's stream;
 * bench.py's roofline leg). enable != 0 starts collecting; ptx_profile_get waits for the stream, returns the
 * summed durations / launch counts since enabling and clears them. No reference counterpart. */
int ptx_profile_enable(PtxCtx* ctx, int enable);
int ptx_profile_get(PtxCtx* ctx, float* traverse_ms, int32_t* traverse_launches, float* segment_ms, int32_t* segment_launches);
/* Scheduling knobs of the render path (no reference counterpart). key =
 *   "trav_variant"    traversal kernel: 0 (default) = the wide traversal (pt_wide.h) + the strict kernel over its retry queue,
 *                     14 = the same with the wide kernel's own counting twin, 15 = the STRICT kernel alone: the reference's
 *                     node-by-node visiting order (src/intersect.cuh:119-212) and every triangle of every visited leaf -
 *                     and, in ptx_render_samples, segment kernels that test every triangle of the single-leaf meshes
 *                     instead of culling them by pair boxes: the STRICT SCHEDULE, nothing is skipped on the strength of a
 *                     box the reference does not have;
 *                     7 / 9 = the strict kernel pinned to one of its two leaf loops (15 picks by tree depth), 13 = the
 *                     GROUPED kernel: the reference's visiting order, but inside a leaf only the groups of eight triangles
 *                     whose box the ray meets (the default until r2 mid-round; what the counting pass runs)
 *   "wide_margin_ppm" margin of the wide traversal's distance culling, parts per million (default 10000 = 1 %)
 *   "wide_scap"       stack entries per lane of the wide traversal (default 10; rays that need more go to the strict kernel)
 *   "wide_build"      which 4-wide tree a multi-node mesh gets at the NEXT ptx_scene_upload: 0 (default) the cheaper, by modelled
 *                     cost, of 1 = the tree collapsed from the reference's KD tree and 2 = one built by the surface-area heuristic
 *   "seg_mode"        1 (default) segment kernel with the bounce loop inside the launch (pt_segloop.h), 0 = one bounce per launch
 *   "seg_blocks"      workgroups per CU of the segment launches
 *   "seg_compact"     [PTX_SEG_COMPACT] 1 (default) = 13-word LDS pool entries where they fit (bounce limit <= 7, <= 254 objects,
 *                     < 2^21 - 1 triangles): six workgroups per CU instead of five; 0 = always the 15-word entries
 *   "grp_mode", "grp_pad_ppm"  how leaf triangles are cut into groups of eight and how far their boxes are padded (parts per
 *                     million of the mesh diagonal, on top of 5e-5); read by the NEXT ptx_scene_upload
 *   "seg_lds_tables"  material / texture-descriptor tables of the segment kernel in LDS: 0 never, 1 (default) for the per-bounce kernel only
 *                     (the bounce-loop kernel would lose its fifth wave per SIMD to them), 2 for both
 *   "trav_blocks"     persistent workgroups per CU (the wide kernel launches at most six: what is resident);  "wf_batch"  samples per wavefront pass (1..256, default 128)
 *   "megakernel"      1 = one-lane-per-pixel kernel instead of the wavefront pipeline
 * WHAT CAN CHANGE A RESULT: nothing but "trav_variant" 0 / 14 / 13 (with "wide_margin_ppm" and "grp_pad_ppm"), and only on
 * rays at grazing incidence where the reference's triangle test accepts a hit out of rounding noise (|det| ~ 1e-7 |e1||e2|:
 * a distance or barycentrics that float64 arithmetic contradicts) on a triangle whose group box the ray does not meet, or
 * whose boxes lie more than the margin behind a well-conditioned hit: measured 7 rays in 19 654 555 adversarial grazing
 * rays on the headline mesh (the same seven for 0 and 13; tests/golden/grazing_hard_rays.npz), 0 pixels of the headline
 * frame at 1000 spp (tests/test_gpu_wide.py). "trav_variant" 15 / 7 / 9, ptx_trace_rays and the
 * "megakernel" path reproduce the reference on those too. Every other knob only reschedules work; the parity tests run the variants
 * against each other. Defaults come from the environment (PTX_TRAV_VARIANT, PTX_WIDE_MARGIN_PPM, PTX_WIDE_SCAP,
 * PTX_WIDE_BUILD, PTX_SEG_MODE, PTX_SEG_BLOCKS, PTX_GRP_MODE, PTX_GRP_PAD_PPM, PTX_SEG_LDS_TABLES, PTX_TRAV_BLOCKS, PTX_WF_BATCH, PTX_MEGAKERNEL) at
 * ptx_ctx_create; value < 0 restores that default. */
int ptx_ctx_tune(PtxCtx* ctx, const char* key, int value);
/* HIP-event timer on the context's stream (for bench.py's roofline leg). */
int ptx_timer_begin(PtxCtx* ctx);
int ptx_timer_end(PtxCtx* ctx, float* elapsed_ms);

/* ---- parity entry points (same device functions as the render kernel) -------------------------
 * reference: src/intersect.cuh:322-350 rayIntersectsScene on caller-supplied rays.
 * origins/dirs: n*3 floats on the host. `object` = -1 intersects the whole scene, else only that
 * object (rayIntersectsObject, src/intersect.cuh:260-319). */
int ptx_trace_rays(PtxCtx* ctx, const PtxScene* scene, int n, const float* origins, const float* dirs,
                   int object, PtxHit* hits_host);
/* reference: src/intersect.cuh:260-319 rayIntersectsObject for ONE multi-node mesh, like ptx_trace_rays
 * with that `object`, but the rays run through the PRODUCTION traversal kernel of the wavefront
 * pipeline (root box test, queue, the kernel variant ptx_ctx_tune("trav_variant") selects), so that
 * kernel can be checked ray by ray. Fills hit, object, prim, t_bits and bary; the counters, leaf_hash,
 * point and normal stay zero. Fails with PTX_E_INVALID for spheres and single-leaf meshes (they have no
 * traversal queue). */
int ptx_trace_rays_queued(PtxCtx* ctx, const PtxScene* scene, int n, const float* origins, const float* dirs,
                          int object, PtxHit* hits_host);
/* reference: src/pathtracing.cu:73-94 generatePrimaryRays (+ :107-129 jitterPrimaryRays when
 * sample >= 0). Writes width*height*3 floats each, pixel index = x + y*width. */
int ptx_generate_rays(PtxCtx* ctx, const PtxCamera* cam, int sample, uint64_t seed,
                      float* origins_host, float* dirs_host);
/* reference: src/surfaces.cuh:37-137 on explicit inputs */
int ptx_bsdf_eval(PtxCtx* ctx, int n, PtxBsdfCase* cases_host);
/* reference: src/radiance.cuh:15-87 gatherRadiance for ONE sample of every pixel, not accumulated:
 * writes width*height*3 floats (rgb radiance of global sample `sample`). */
int ptx_radiance_sample(PtxCtx* ctx, const PtxScene* scene, int sample, uint64_t seed, float* rgb_host);

/* Device self-test (no reference counterpart): the triangle test's 5-instruction reciprocal is compared with the
 * IEEE quotient 1.0f / a for EVERY float with lo <= |a| <= hi; reports how many were checked, how many differ
 * and the bit pattern of the first that does (0xffffffff if none). */
int ptx_selftest_rcp(PtxCtx* ctx, float lo, float hi, uint64_t* checked, uint64_t* mismatches, uint32_t* first_bad_bits);

/* Device self-test: the float3 helpers every hot-path function is built from (reference:
 * external/include/cutil_math.h:322-416 dot, cross, length, normalize, reflect, operator* / + - ) on n pairs
 * (a, b) and scalars s, all on the host, 3 / 3 / 1 floats per case; out: 20 floats per case = dot(a,b),
 * length(a), cross(a,b).xyz, normalize(a).xyz, reflect(a,b).xyz, (a*s).xyz, (a/s).xyz, (a+b).x, (a-b).x,
 * (a*b).x. tests/ compares them bit for bit with the reference's own header compiled on the host (oracle/_ref). */
int ptx_selftest_vec3(PtxCtx* ctx, int n, const float* a_host, const float* b_host, const float* s_host, float* out_host);

#ifdef __cplusplus
}
#endif
#endif /* PTX_HIP_H */
