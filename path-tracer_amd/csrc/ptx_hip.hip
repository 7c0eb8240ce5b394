// libptx_hip.so — C ABI (include/ptx_hip.h) over the gfx950 kernels in pt_kernels.h.
// Host side of the library: argument validation, the packed scene upload, launches.
// There is no CPU fallback anywhere in this file: without a HIP device every compute
// entry point returns an error.
#include "pt_wide.h"
#include "pt_segloop.h"
#include <cstdlib>
#include <algorithm>
#include <cmath>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}

#define HIP_TRY(expr)                                                                               \
	do {                                                                                            \
		hipError_t e_ = (expr);                                                                     \
		if (e_ != hipSuccess)                                                                       \
			return fail(PTX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

struct PtxCtx;
static void prof_mark(PtxCtx* c, int cls);
static void prof_clear(PtxCtx* c);

struct PtxCtx {
	int device;
	int cus;              // hipDeviceProp_t::multiProcessorCount: sizes the persistent launches
	hipStream_t stream;
	bool has_render;
	DevCamera cam;
	float4* accum;
	bool owns_accum;
	PtxStats* d_stats;
	uchar4* d_pbo;        // tonemap output staging (device)
	size_t pbo_pixels;
	hipEvent_t ev0, ev1;
	bool timer_open;
	// wavefront pipeline buffers (allocated on first use, grown on demand); wf.s0..s3 and wf_b[0..3] are the two
	// sets of compacted path state (pt_wavefront.h), used in turn from bounce to bounce
	WfParams wf;
	size_t wf_paths, wf_rad;   // capacities (paths per state set, rad entries)
	int wf_queues, wf_sets;    // queues and state sets allocated (wf.s0..s3 hold wf_sets sets of wf_paths records each, wf.queue[k] as many segments)
	int wf_batch;              // samples per wavefront pass
	bool force_megakernel;
	int trav_variant, trav_blocks_per_cu;   // tuning knobs (env PTX_TRAV_VARIANT / PTX_TRAV_BLOCKS)
	int wide_scap, grp_mode, seg_lds_tables, wide_margin_ppm, seg_mode, seg_blocks_per_cu, wide_build, seg_compact;
	int grp_pad_ppm;                        // group-box padding used by the next ptx_scene_upload (env PTX_GRP_PAD_PPM)
	bool profiling;
	std::vector<hipEvent_t>* prof_ev[2];    // [0] traverse, [1] segment: start/stop event pairs
	// look-ahead window of ptx_render_sample_ahead: samples [la_first, la_first + la_count) of (la_scene, la_seed, camera
	// generation la_gen) are traced and their radiance is parked in wf.rad, one plane per sample; la_window = the size
	// asked for when the window was rendered (doubles while calls continue it, back to 1 on anything else)
	int la_first, la_count, la_window;
	uint64_t la_seed, la_gen, cam_gen;
	const PtxScene* la_scene;
};

// whatever overwrites wf.rad, or changes what a sample is (camera, scheduling knobs, scene), ends the window
static void la_drop(PtxCtx* c) { c->la_count = 0; c->la_window = 0; c->la_scene = NULL; }

struct PtxScene {
	int device;
	DevScene dev;
	std::vector<void*> allocs;
	int64_t bytes;
	int num_objects;
	int max_depth;
	int num_dev_triangles;
	int num_big;                       // meshes with inner nodes
	int wide_sah_objects;              // of them: meshes whose wide tree is the surface-area-heuristic candidate
	int big_object[PT_MAX_BIG];        // their object indices, by queue slot
};

extern "C" const char* ptx_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* ptx_version(void) { return "ptx_hip 0.1 (gfx950)"; }

extern "C" int ptx_device_count(void) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) {
		fail(PTX_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
		return 0;
	}
	return n;
}

extern "C" int ptx_device_info(int device, char* name, int name_len, int* compute_units, int* clock_khz,
                               uint64_t* total_mem) {
	int n = ptx_device_count();
	if (device < 0 || device >= n) return fail(PTX_E_NO_DEVICE, "device %d out of range (%d devices)", device, n);
	hipDeviceProp_t p;
	HIP_TRY(hipGetDeviceProperties(&p, device));
	if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
	if (compute_units) *compute_units = p.multiProcessorCount;
	if (clock_khz) *clock_khz = p.clockRate;
	if (total_mem) *total_mem = (uint64_t)p.totalGlobalMem;
	return PTX_OK;
}

static int env_int(const char* name, int dflt) {
	const char* e = getenv(name);
	return e ? atoi(e) : dflt;
}

#define PT_GRP_PAD_PPM_DEFAULT 100
// queue chunk a wave claims at a time and descent threshold of the default traversal kernel (pt_wavefront.h)
#ifndef WF_CHUNK
#define WF_CHUNK 128
#endif
#ifndef WFW_CHUNK
#define WFW_CHUNK 256   // the wide kernel's own chunk: 128 / 192 / 256 / 512 / 1024 -> 7.02 / 6.88 / 6.84 / 6.97 / 7.28 ms (a claim is a returning atomic in a turn)
#endif
#ifndef WFW_TURN_MIN
#define WFW_TURN_MIN 24   // re-swept after the node visit lost a fifth of its instructions: 12 / 16 / 20 / 24 / 32 -> 7.34 / 7.16 / 7.10 / 7.06 / 7.11 ms
#endif
#ifndef WFW_DRAIN_MIN
#define WFW_DRAIN_MIN 16  // with TURN_MIN 24: 16 / 24 / 32 -> 7.03 / 7.04 / 7.06 ms (48: 7.21)
#endif
#ifndef WF_ATHRESH
#define WF_ATHRESH 28
#endif

struct Tuning {
	int wf_batch, trav_variant, trav_blocks_per_cu, grp_pad_ppm, wide_scap, grp_mode, seg_lds_tables, wide_margin_ppm, seg_mode, seg_blocks_per_cu, wide_build, seg_compact;
	bool force_megakernel;
};

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// defaults: the environment, else the values measured best on the bench workload
static Tuning tuning_defaults() {
	Tuning t;
	// samples per pass: large queues amortise the ramp/tail of the persistent traversal launches
	// (8 -> 64: 1274 -> 1601 Mrays/s with the first cooperative kernel; 64 -> 128 -> 256: 3278 -> 3377 -> 3426 now);
	// costs 148 B of state per path (two sets of 64 B, queue entry, parked radiance): 39 GB at 1080p x 128 samples,
	// capped at 2^28 paths per pass
	t.wf_batch = clampi(env_int("PTX_WF_BATCH", 128), 1, 256);
	t.force_megakernel = env_int("PTX_MEGAKERNEL", 0) != 0;
	t.trav_variant = env_int("PTX_TRAV_VARIANT", 0);
	if (t.trav_variant != 7 && t.trav_variant != 9 && t.trav_variant != 13 && t.trav_variant != 14 && t.trav_variant != 15) t.trav_variant = 0;
	t.trav_blocks_per_cu = clampi(env_int("PTX_TRAV_BLOCKS", 16), 1, 64);
	// padding of the group boxes (pt_scene.h) relative to the mesh diagonal, in parts per million; read at scene upload
	t.grp_pad_ppm = clampi(env_int("PTX_GRP_PAD_PPM", PT_GRP_PAD_PPM_DEFAULT), 0, 1000000);
	// wide traversal (pt_wide.h): A/B configuration of the kernel and the most stack entries a lane may hold
	t.grp_mode = clampi(env_int("PTX_GRP_MODE", 1), 0, 1);   // how a leaf's triangles are cut into groups of eight; read at scene upload
	t.seg_lds_tables = clampi(env_int("PTX_SEG_LDS_TABLES", 1), 0, 2);
	// segment kernel: 1 = bounce loop inside the launch (pt_segloop.h), 0 = one bounce per launch (wf_segment_kernel)
	t.seg_mode = clampi(env_int("PTX_SEG_MODE", 1), 0, 1);
	t.seg_blocks_per_cu = clampi(env_int("PTX_SEG_BLOCKS", 16), 1, 64);
	t.seg_compact = clampi(env_int("PTX_SEG_COMPACT", 1), 0, 1);
	// margin of the wide traversal's distance culling, relative to the best t (and absolute), in parts per million (pt_wide.h)
	t.wide_margin_ppm = clampi(env_int("PTX_WIDE_MARGIN_PPM", 10000), 0, 1000000);
	t.wide_scap = clampi(env_int("PTX_WIDE_SCAP", WFW_SCAP_MAX), 1, 32);
	t.wide_build = clampi(env_int("PTX_WIDE_BUILD", 0), 0, 2);   // which wide tree a mesh gets: read at scene upload
	return t;
}

static void tune_defaults(PtxCtx* c) {
	const Tuning t = tuning_defaults();
	c->wf_batch = t.wf_batch;
	c->force_megakernel = t.force_megakernel;
	c->trav_variant = t.trav_variant;
	c->trav_blocks_per_cu = t.trav_blocks_per_cu;
	c->grp_pad_ppm = t.grp_pad_ppm;
	c->grp_mode = t.grp_mode;
	c->wide_margin_ppm = t.wide_margin_ppm;
	c->seg_lds_tables = t.seg_lds_tables;
	c->seg_mode = t.seg_mode;
	c->seg_blocks_per_cu = t.seg_blocks_per_cu;
	c->seg_compact = t.seg_compact;
	c->wide_scap = t.wide_scap;
	c->wide_build = t.wide_build;
}

extern "C" int ptx_ctx_tune(PtxCtx* ctx, const char* key, int value) {
	if (!ctx || !key) return fail(PTX_E_INVALID, "ptx_ctx_tune: NULL argument");
	const Tuning d = tuning_defaults();
	la_drop(ctx);   // some knobs can change what a sample is on adversarial rays (trav_variant, wide_margin_ppm)
	if (!strcmp(key, "trav_variant")) {
		if (value >= 0 && value != 0 && value != 7 && value != 9 && value != 13 && value != 14 && value != 15)
			return fail(PTX_E_INVALID, "ptx_ctx_tune: trav_variant %d is not in this build (0, 7, 9, 13, 14, 15)", value);
		ctx->trav_variant = value < 0 ? d.trav_variant : value;
	}
	else if (!strcmp(key, "trav_blocks")) ctx->trav_blocks_per_cu = value < 0 ? d.trav_blocks_per_cu : clampi(value, 1, 64);
	else if (!strcmp(key, "wf_batch")) ctx->wf_batch = value < 0 ? d.wf_batch : clampi(value, 1, 256);
	else if (!strcmp(key, "megakernel")) ctx->force_megakernel = value < 0 ? d.force_megakernel : value != 0;
	else if (!strcmp(key, "grp_pad_ppm")) ctx->grp_pad_ppm = value < 0 ? d.grp_pad_ppm : clampi(value, 0, 1000000);
	else if (!strcmp(key, "grp_mode")) ctx->grp_mode = value < 0 ? d.grp_mode : clampi(value, 0, 1);
	else if (!strcmp(key, "seg_lds_tables")) ctx->seg_lds_tables = value < 0 ? d.seg_lds_tables : clampi(value, 0, 2);
	else if (!strcmp(key, "seg_mode")) ctx->seg_mode = value < 0 ? d.seg_mode : clampi(value, 0, 1);
	else if (!strcmp(key, "seg_blocks")) ctx->seg_blocks_per_cu = value < 0 ? d.seg_blocks_per_cu : clampi(value, 1, 64);
	else if (!strcmp(key, "seg_compact")) ctx->seg_compact = value < 0 ? d.seg_compact : clampi(value, 0, 1);
	else if (!strcmp(key, "wide_margin_ppm")) ctx->wide_margin_ppm = value < 0 ? d.wide_margin_ppm : clampi(value, 0, 1000000);
	else if (!strcmp(key, "wide_scap")) ctx->wide_scap = value < 0 ? d.wide_scap : clampi(value, 1, 32);
	else if (!strcmp(key, "wide_build")) ctx->wide_build = value < 0 ? d.wide_build : clampi(value, 0, 2);
	else return fail(PTX_E_INVALID, "ptx_ctx_tune: unknown key '%s'", key);
	return PTX_OK;
}

extern "C" int ptx_ctx_create(int device, PtxCtx** out) {
	if (!out) return fail(PTX_E_INVALID, "ptx_ctx_create: out is NULL");
	*out = NULL;
	int n = ptx_device_count();
	if (n <= 0) return fail(PTX_E_NO_DEVICE, "no HIP device available (the product has no CPU path)");
	if (device < 0 || device >= n) return fail(PTX_E_NO_DEVICE, "device %d out of range (%d devices)", device, n);
	HIP_TRY(hipSetDevice(device));
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	PtxCtx* c = new PtxCtx();
	memset(c, 0, sizeof(*c));
	c->device = device;
	c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete c; return fail(PTX_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
	e = hipMalloc((void**)&c->d_stats, sizeof(PtxStats));
	if (e != hipSuccess) { hipStreamDestroy(c->stream); delete c; return fail(PTX_E_HIP, "hipMalloc stats: %s", hipGetErrorString(e)); }
	hipMemsetAsync(c->d_stats, 0, sizeof(PtxStats), c->stream);
	hipEventCreate(&c->ev0);
	hipEventCreate(&c->ev1);
	tune_defaults(c);
	*out = c;
	return PTX_OK;
}

extern "C" int ptx_render_cleanup(PtxCtx* ctx);

static void wf_free(PtxCtx* c) {
	la_drop(c);
	void* ptrs[] = {c->wf.s0, c->wf.s1, c->wf.s2, c->wf.s3, c->wf.rad, c->wf.counters, c->wf.retry};
	for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i)
		if (ptrs[i]) hipFree(ptrs[i]);
	for (int k = 0; k < PT_MAX_BIG; ++k)
		if (c->wf.queue[k]) hipFree(c->wf.queue[k]);
	memset(&c->wf, 0, sizeof(c->wf));
	c->wf_paths = c->wf_rad = 0;
	c->wf_queues = c->wf_sets = 0;
}

// Path-state sets, queues and radiance buffers: `sets` state sets of `paths` records (64 B each, four planes), `queues`
// queues of sets x paths entries, `rad` radiance slots.
static int wf_ensure(PtxCtx* c, size_t paths, size_t rad, int queues, int sets) {
	if (paths <= c->wf_paths && rad <= c->wf_rad && queues <= c->wf_queues && sets <= c->wf_sets) return PTX_OK;
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (paths < c->wf_paths) paths = c->wf_paths;
	if (rad < c->wf_rad) rad = c->wf_rad;
	if (queues < c->wf_queues) queues = c->wf_queues;
	if (sets < c->wf_sets) sets = c->wf_sets;
	wf_free(c);
	const size_t recs = paths * (size_t)sets;
	// the segment loop kernel indexes (bounce, slot) and (sample, pixel) in 32 bits; the caller halves the pass on any failure here
	if (recs >= ((size_t)1 << 32) || rad >= ((size_t)1 << 32) / sizeof(Rad3))
		return fail(PTX_E_LIMIT, "a pass of %zu paths x %d state sets passes the 32-bit record index", paths, sets);
	HIP_TRY(hipMalloc((void**)&c->wf.s0, recs * sizeof(float4)));
	HIP_TRY(hipMalloc((void**)&c->wf.s1, recs * sizeof(float4)));
	HIP_TRY(hipMalloc((void**)&c->wf.s2, recs * sizeof(float4)));
	HIP_TRY(hipMalloc((void**)&c->wf.s3, recs * sizeof(float4)));
	HIP_TRY(hipMalloc((void**)&c->wf.rad, rad * sizeof(Rad3)));
	for (int k = 0; k < queues; ++k) HIP_TRY(hipMalloc((void**)&c->wf.queue[k], recs * sizeof(int)));
	HIP_TRY(hipMalloc((void**)&c->wf.counters, WF_COUNTER_INTS * sizeof(int))); // counts + traversal cursors, queues and retry queues
	if (queues > 0) HIP_TRY(hipMalloc((void**)&c->wf.retry, paths * sizeof(int)));
	c->wf_paths = paths;
	c->wf_rad = rad;
	c->wf_queues = queues;
	c->wf_sets = sets;
	return PTX_OK;
}

extern "C" int ptx_ctx_destroy(PtxCtx* ctx) {
	if (!ctx) return PTX_OK;
	hipSetDevice(ctx->device);
	hipStreamSynchronize(ctx->stream);
	ptx_render_cleanup(ctx);
	wf_free(ctx);
	prof_clear(ctx);
	delete ctx->prof_ev[0];
	delete ctx->prof_ev[1];
	if (ctx->d_stats) hipFree(ctx->d_stats);
	if (ctx->d_pbo) hipFree(ctx->d_pbo);
	hipEventDestroy(ctx->ev0);
	hipEventDestroy(ctx->ev1);
	hipStreamDestroy(ctx->stream);
	delete ctx;
	return PTX_OK;
}

// ---- scene upload -----------------------------------------------------------------------------

static inline float as_float(int32_t v) { float f; memcpy(&f, &v, 4); return f; }
static inline int32_t as_int(float f) { int32_t v; memcpy(&v, &f, 4); return v; }

template <typename T>
static int upload_vec(PtxCtx* ctx, PtxScene* s, const std::vector<T>& v, const T** out) {
	*out = NULL;
	size_t bytes = v.size() * sizeof(T);
	if (bytes == 0) bytes = sizeof(T); // keep pointers non-null and loads in-bounds for empty scenes
	void* d = NULL;
	HIP_TRY(hipMalloc(&d, bytes));
	s->allocs.push_back(d);
	s->bytes += (int64_t)bytes;
	if (!v.empty()) HIP_TRY(hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	else HIP_TRY(hipMemsetAsync(d, 0, bytes, ctx->stream));
	*out = (const T*)d;
	return PTX_OK;
}

static void free_scene_allocs(PtxScene* s) {
	for (size_t i = 0; i < s->allocs.size(); ++i) hipFree(s->allocs[i]);
	s->allocs.clear();
}

// Four boxes in 48 bytes (pt_scene.h, grp_qbox / wide nodes): the union U of the used boxes as origin and
// step = extent / 255 per axis, then per box six bytes (lo.xyz rounded down, hi.xyz rounded up, in steps from U's
// origin). The decoded box always contains the given one; an unused slot is (255, 255, 255, 0, 0, 0): never touched.
// Byte order: box-major (box g = bytes 6g .. 6g+5: lo.xyz, hi.xyz; the grouped kernel's grp_qbox) or, for the wide nodes,
// AXIS-major (word 2k = the four lo codes of axis k, word 2k+1 the four hi codes, box g in byte g): a ray then picks its
// near and far faces per axis by swapping two words on the sign of its direction, once per node instead of once per box.
static void quantise_boxes4(const float lo[4][3], const float hi[4][3], const bool used[4], float4 out[3], bool axis_major = false) {
	double ulo[3] = {1e300, 1e300, 1e300}, uhi[3] = {-1e300, -1e300, -1e300};
	for (int g = 0; g < 4; ++g) {
		if (!used[g]) continue;
		for (int k = 0; k < 3; ++k) { if (lo[g][k] < ulo[k]) ulo[k] = lo[g][k]; if (hi[g][k] > uhi[k]) uhi[k] = hi[g][k]; }
	}
	float org[3] = {0, 0, 0}, step[3] = {0, 0, 0};
	unsigned char q[24];
	for (int g = 0; g < 4; ++g) { q[6 * g] = q[6 * g + 1] = q[6 * g + 2] = 255; q[6 * g + 3] = q[6 * g + 4] = q[6 * g + 5] = 0; }
	if (used[0] || used[1] || used[2] || used[3]) {
		for (int k = 0; k < 3; ++k) {
			org[k] = (float)ulo[k];
			if ((double)org[k] > ulo[k]) org[k] = nextafterf(org[k], -INFINITY);
			step[k] = (float)((uhi[k] - (double)org[k]) / 255.0);
			// the largest code must reach the union's upper face in float arithmetic as the device decodes it
			while (!((double)org[k] + 255.0 * (double)step[k] >= uhi[k])) step[k] = nextafterf(step[k], INFINITY);
		}
		for (int g = 0; g < 4; ++g) {
			if (!used[g]) continue;
			for (int k = 0; k < 3; ++k) {
				int ql = 0, qh = 255;
				if (step[k] > 0.0f) {
					ql = (int)floor(((double)lo[g][k] - (double)org[k]) / (double)step[k]);
					qh = (int)ceil(((double)hi[g][k] - (double)org[k]) / (double)step[k]);
				}
				ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
				qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
				while (ql > 0 && (double)org[k] + (double)ql * (double)step[k] > (double)lo[g][k]) --ql;
				while (qh < 255 && (double)org[k] + (double)qh * (double)step[k] < (double)hi[g][k]) ++qh;
				q[6 * g + k] = (unsigned char)ql;
				q[6 * g + 3 + k] = (unsigned char)qh;
			}
		}
	}
	if (axis_major) {
		unsigned char t[24];
		for (int g = 0; g < 4; ++g) for (int k = 0; k < 3; ++k) { t[8 * k + g] = q[6 * g + k]; t[8 * k + 4 + g] = q[6 * g + 3 + k]; }
		memcpy(q, t, 24);
	}
	uint32_t w[6];
	memcpy(w, q, 24);
	out[0] = make_float4(org[0], org[1], org[2], step[0]);
	out[1] = make_float4(step[1], step[2], as_float((int32_t)w[0]), as_float((int32_t)w[1]));
	out[2] = make_float4(as_float((int32_t)w[2]), as_float((int32_t)w[3]), as_float((int32_t)w[4]), as_float((int32_t)w[5]));
}


// ---- wide tree (pt_scene.h, wide_nodes; traversed by pt_wide.h) -------------------------------------------------
// A 4-wide bounding-volume hierarchy over the GROUPS of eight triangles, collapsed from the reference's binary
// KD tree: a wide node's children are found by opening, largest surface first, the inner KD nodes / leaves among
// them until four are reached. Boxes are unions of group boxes (which hold their triangles with the usual
// padding), stored in the 8-bit form of grp_qbox. Leaves of more than four groups get intermediate nodes over
// contiguous quarters of their (sorted) group list.
struct WideItem {
	int kind;        // 0 = KD node (scene-global index), 1 = run of `n` consecutive groups from global group `a`, 2 = node `a` of the SAH tree
	int a, n;
	float lo[3], hi[3];
};

struct WideBuilder {
	const std::vector<float4>& nodes;       // packed KD records of the whole scene
	const std::vector<float4>& grp_boxes;
	std::vector<float4>& out;               // 4 float4 per wide node
	std::vector<float> kd_box;              // per KD node of the object: 6 floats, filled bottom-up
	int base = 0;                           // first KD node of the object
	int max_stack = 0;                      // most stack entries a traversal can hold at once (one per node with two or more inner children)
	double cost = 0.0;                      // sum of the surfaces of all emitted nodes' child boxes: ~ box tests per random ray x root surface
	// second candidate: a binary tree over the object's groups built top-down by the surface-area heuristic (16 bins per axis
	// on the centres of the group boxes); leaves are single groups
	struct BinNode { float lo[3], hi[3]; int left, right, group; };
	std::vector<BinNode> bin;

	static float area(const WideItem& it) {
		const float dx = it.hi[0] - it.lo[0], dy = it.hi[1] - it.lo[1], dz = it.hi[2] - it.lo[2];
		return dx * dy + dy * dz + dz * dx;
	}
	void group_box(int g, float lo[3], float hi[3]) const {
		const float4 l = grp_boxes[(size_t)g * 2], h = grp_boxes[(size_t)g * 2 + 1];
		lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
	}
	WideItem run_item(int g0, int n) const {
		WideItem it; it.kind = 1; it.a = g0; it.n = n;
		for (int k = 0; k < 3; ++k) { it.lo[k] = INFINITY; it.hi[k] = -INFINITY; }
		for (int g = g0; g < g0 + n; ++g) {
			float lo[3], hi[3];
			group_box(g, lo, hi);
			for (int k = 0; k < 3; ++k) { if (lo[k] < it.lo[k]) it.lo[k] = lo[k]; if (hi[k] > it.hi[k]) it.hi[k] = hi[k]; }
		}
		return it;
	}
	// number of groups of a KD leaf record (0 for an empty leaf)
	static int leaf_groups(const float4* rec) { const int n = as_int(rec[3].w); return n > 0 ? (n + 7) / 8 : 0; }
	bool kd_is_leaf(int ni) const { const float4* r = &nodes[(size_t)ni * 4]; return as_int(r[3].x) == -1 && as_int(r[3].y) == -1; }

	// boxes of the KD nodes as unions of the group boxes beneath (empty subtrees: lo > hi)
	void fill_boxes(int root, int num_nodes) {
		kd_box.assign((size_t)num_nodes * 6, 0.0f);
		std::vector<std::pair<int, int> > todo;   // (node, phase)
		todo.push_back(std::make_pair(root, 0));
		while (!todo.empty()) {
			const int ni = todo.back().first, phase = todo.back().second;
			todo.pop_back();
			const float4* r = &nodes[(size_t)ni * 4];
			float* b = &kd_box[(size_t)(ni - base) * 6];
			if (kd_is_leaf(ni)) {
				const int ng = leaf_groups(r);
				for (int k = 0; k < 3; ++k) { b[k] = INFINITY; b[3 + k] = -INFINITY; }
				if (ng > 0) { const WideItem it = run_item(4 * as_int(r[0].x), ng); for (int k = 0; k < 3; ++k) { b[k] = it.lo[k]; b[3 + k] = it.hi[k]; } }
				continue;
			}
			const int ch[2] = {as_int(r[3].x), as_int(r[3].y)};
			if (phase == 0) {
				todo.push_back(std::make_pair(ni, 1));
				for (int c = 0; c < 2; ++c) if (ch[c] >= 0) todo.push_back(std::make_pair(ch[c], 0));
			} else {
				for (int k = 0; k < 3; ++k) { b[k] = INFINITY; b[3 + k] = -INFINITY; }
				for (int c = 0; c < 2; ++c) if (ch[c] >= 0) {
					const float* cb = &kd_box[(size_t)(ch[c] - base) * 6];
					for (int k = 0; k < 3; ++k) { if (cb[k] < b[k]) b[k] = cb[k]; if (cb[3 + k] > b[3 + k]) b[3 + k] = cb[3 + k]; }
				}
			}
		}
	}
	bool kd_empty(int ni) const { const float* b = &kd_box[(size_t)(ni - base) * 6]; return !(b[0] <= b[3]); }
	// a KD node as an item: leaves become their run of groups, chains of single children are followed; a subtree
	// without triangles becomes the empty run (n = 0), which children() drops
	WideItem kd_item(int ni) const {
		for (;;) {
			if (ni < 0 || kd_empty(ni)) return run_item(0, 0);
			const float4* r = &nodes[(size_t)ni * 4];
			if (kd_is_leaf(ni)) return run_item(4 * as_int(r[0].x), leaf_groups(r));
			const int l = as_int(r[3].x), rr = as_int(r[3].y);
			const bool hl = l >= 0 && !kd_empty(l), hr = rr >= 0 && !kd_empty(rr);
			if (hl && hr) break;
			ni = hl ? l : rr;
		}
		WideItem it; it.kind = 0; it.a = ni; it.n = 0;
		const float* b = &kd_box[(size_t)(ni - base) * 6];
		for (int k = 0; k < 3; ++k) { it.lo[k] = b[k]; it.hi[k] = b[3 + k]; }
		return it;
	}
	// every group of the object, in leaf order
	void object_groups(int root, std::vector<int>& ids) const {
		std::vector<int> todo(1, root);
		while (!todo.empty()) {
			const int ni = todo.back();
			todo.pop_back();
			if (ni < 0) continue;
			const float4* r = &nodes[(size_t)ni * 4];
			if (kd_is_leaf(ni)) { const int g0 = 4 * as_int(r[0].x), ng = leaf_groups(r); for (int g = 0; g < ng; ++g) ids.push_back(g0 + g); continue; }
			todo.push_back(as_int(r[3].y));
			todo.push_back(as_int(r[3].x));
		}
	}
	static float box_area(const float lo[3], const float hi[3]) {
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return dx * dy + dy * dz + dz * dx;
	}
	int build_sah(std::vector<int>& ids, int first, int last) {   // ids[first, last) -> index of the subtree's node in `bin`
		BinNode nd;
		nd.left = nd.right = nd.group = -1;
		float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
		for (int k = 0; k < 3; ++k) { nd.lo[k] = INFINITY; nd.hi[k] = -INFINITY; }
		for (int i = first; i < last; ++i) {
			float lo[3], hi[3];
			group_box(ids[(size_t)i], lo, hi);
			for (int k = 0; k < 3; ++k) {
				if (lo[k] < nd.lo[k]) nd.lo[k] = lo[k];
				if (hi[k] > nd.hi[k]) nd.hi[k] = hi[k];
				const float c = 0.5f * (lo[k] + hi[k]);
				if (c < clo[k]) clo[k] = c;
				if (c > chi[k]) chi[k] = c;
			}
		}
		const int me = (int)bin.size();
		if (last - first == 1) { nd.group = ids[(size_t)first]; bin.push_back(nd); return me; }
		bin.push_back(nd);
		const int NB = 16;
		int best_axis = -1, best_cut = 0;
		double best = 1e300;
		for (int ax = 0; ax < 3; ++ax) {
			const float ext = chi[ax] - clo[ax];
			if (!(ext > 0.0f)) continue;
			int cnt[NB];
			float blo[NB][3], bhi[NB][3];
			for (int b = 0; b < NB; ++b) { cnt[b] = 0; for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; } }
			for (int i = first; i < last; ++i) {
				float lo[3], hi[3];
				group_box(ids[(size_t)i], lo, hi);
				int b = (int)((0.5f * (lo[ax] + hi[ax]) - clo[ax]) / ext * (float)NB);
				b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
				cnt[b]++;
				for (int k = 0; k < 3; ++k) { if (lo[k] < blo[b][k]) blo[b][k] = lo[k]; if (hi[k] > bhi[b][k]) bhi[b][k] = hi[k]; }
			}
			// sweep from the right, then from the left
			float rarea[NB];
			int rcnt[NB];
			float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
			int c = 0;
			for (int b = NB - 1; b >= 1; --b) {
				for (int k = 0; k < 3; ++k) { if (blo[b][k] < lo[k]) lo[k] = blo[b][k]; if (bhi[b][k] > hi[k]) hi[k] = bhi[b][k]; }
				c += cnt[b];
				rcnt[b] = c;
				rarea[b] = c ? box_area(lo, hi) : 0.0f;
			}
			for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
			c = 0;
			for (int b = 0; b < NB - 1; ++b) {   // cut between bin b and b + 1
				for (int k = 0; k < 3; ++k) { if (blo[b][k] < lo[k]) lo[k] = blo[b][k]; if (bhi[b][k] > hi[k]) hi[k] = bhi[b][k]; }
				c += cnt[b];
				if (c == 0 || rcnt[b + 1] == 0) continue;
				const double sah = (double)box_area(lo, hi) * c + (double)rarea[b + 1] * rcnt[b + 1];
				if (sah < best) { best = sah; best_axis = ax; best_cut = b; }
			}
		}
		int mid;
		if (best_axis < 0) mid = first + (last - first) / 2;   // all centres coincide: halve the list
		else {
			const float ext = chi[best_axis] - clo[best_axis];
			const int ax = best_axis, cut = best_cut;
			const float c0 = clo[ax];
			std::vector<int>::iterator it = std::stable_partition(ids.begin() + first, ids.begin() + last, [&](int g) {
				float lo[3], hi[3];
				group_box(g, lo, hi);
				int b = (int)((0.5f * (lo[ax] + hi[ax]) - c0) / ext * (float)NB);
				b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
				return b <= cut;
			});
			mid = (int)(it - ids.begin());
			if (mid == first || mid == last) mid = first + (last - first) / 2;
		}
		const int l = build_sah(ids, first, mid);
		const int r = build_sah(ids, mid, last);
		bin[(size_t)me].left = l;
		bin[(size_t)me].right = r;
		return me;
	}
	WideItem bin_item(int bi) const {
		const BinNode& nd = bin[(size_t)bi];
		if (nd.group >= 0) return run_item(nd.group, 1);
		WideItem it; it.kind = 2; it.a = bi; it.n = 0;
		for (int k = 0; k < 3; ++k) { it.lo[k] = nd.lo[k]; it.hi[k] = nd.hi[k]; }
		return it;
	}
	static bool is_empty(const WideItem& it) { return it.kind == 1 && it.n == 0; }
	static bool is_group(const WideItem& it) { return it.kind == 1 && it.n == 1; }
	// the children of an item that is not a single group (2..4 of them)
	void children(const WideItem& it, std::vector<WideItem>& ch) const {
		ch.clear();
		if (it.kind == 2) {
			ch.push_back(bin_item(bin[(size_t)it.a].left));
			ch.push_back(bin_item(bin[(size_t)it.a].right));
		} else if (it.kind == 0) {
			const float4* r = &nodes[(size_t)it.a * 4];
			const WideItem l = kd_item(as_int(r[3].x)), rr = kd_item(as_int(r[3].y));
			if (!is_empty(l)) ch.push_back(l);
			if (!is_empty(rr)) ch.push_back(rr);
		} else if (it.n <= 4) {
			for (int g = 0; g < it.n; ++g) ch.push_back(run_item(it.a + g, 1));
		} else {
			const int parts = 4;
			int at = it.a;
			for (int p = 0; p < parts; ++p) {
				const int len = (it.n - (at - it.a) + (parts - p) - 1) / (parts - p);
				ch.push_back(run_item(at, len));
				at += len;
			}
		}
	}
	// Writes wide node `idx` over `list` (0..4 items) and everything beneath. Children that are nodes come FIRST in the
	// record and get CONSECUTIVE indices (link of slot s = link of slot 0 + s): the traversal stacks one entry per
	// node, (first child's index, the slots still to visit), so `depth` = stack entries a ray can hold below here.
	void emit(int idx, std::vector<WideItem> list, int depth) {
		std::vector<WideItem> ch;
		for (;;) {
			int pick = -1;
			for (size_t i = 0; i < list.size(); ++i) {
				if (is_group(list[i])) continue;
				children(list[i], ch);
				if (list.size() - 1 + ch.size() > 4) continue;
				if (pick < 0 || area(list[i]) > area(list[(size_t)pick])) pick = (int)i;
			}
			if (pick < 0) break;
			children(list[(size_t)pick], ch);
			list.erase(list.begin() + pick);
			list.insert(list.end(), ch.begin(), ch.end());
		}
		std::stable_partition(list.begin(), list.end(), [](const WideItem& it) { return !is_group(it); });
		int inner = 0;
		for (size_t i = 0; i < list.size(); ++i) if (!is_group(list[i])) ++inner;
		const int first = (int)(out.size() / 4);
		out.resize(out.size() + (size_t)inner * 4, make_float4(0, 0, 0, 0));
		const int below = depth + (inner > 1 ? 1 : 0);
		if (below > max_stack) max_stack = below;
		float lo[4][3], hi[4][3];
		bool used[4] = {false, false, false, false};
		int32_t ref[4] = {-1, -1, -1, -1};
		for (size_t i = 0; i < list.size(); ++i) {
			used[i] = true;
			cost += (double)area(list[i]);
			for (int k = 0; k < 3; ++k) { lo[i][k] = list[i].lo[k]; hi[i][k] = list[i].hi[k]; }
			if (is_group(list[i])) ref[i] = (int32_t)(0x80000000u | (uint32_t)list[i].a);
			else { ref[i] = first + (int)i; children(list[i], ch); emit(first + (int)i, ch, below); }
		}
		float4 q[3];
		quantise_boxes4(lo, hi, used, q, true);
		out[(size_t)idx * 4 + 0] = q[0]; out[(size_t)idx * 4 + 1] = q[1]; out[(size_t)idx * 4 + 2] = q[2];
		out[(size_t)idx * 4 + 3] = make_float4(as_float(ref[0]), as_float(ref[1]), as_float(ref[2]), as_float(ref[3]));
	}
	int emit_root(const std::vector<WideItem>& list) {
		const int idx = (int)(out.size() / 4);
		out.resize(out.size() + 4, make_float4(0, 0, 0, 0));
		emit(idx, list, 0);
		return idx;
	}
};


extern "C" int ptx_scene_upload(PtxCtx* ctx, const PtxSceneDesc* d, PtxScene** out) {
	if (!ctx || !d || !out) return fail(PTX_E_INVALID, "ptx_scene_upload: NULL argument");
	*out = NULL;
	if (d->num_objects < 0 || d->num_objects > PTX_MAX_OBJECTS)
		return fail(PTX_E_LIMIT, "num_objects %d outside [0, %d]", d->num_objects, PTX_MAX_OBJECTS);
	if (d->num_materials < 0 || d->num_textures < 0 || d->num_triangles < 0 || d->num_nodes < 0 || d->num_leaf_tris < 0)
		return fail(PTX_E_INVALID, "negative element count in scene description");
	if ((d->num_objects && !d->objects) || (d->num_materials && !d->materials) || (d->num_textures && !d->textures) ||
	    (d->num_triangles && !d->triangles) || (d->num_nodes && !d->nodes) || (d->num_leaf_tris && !d->leaf_tris))
		return fail(PTX_E_INVALID, "NULL array with non-zero count in scene description");
	for (int t = 0; t < d->num_textures; ++t)
		if (!d->textures[t].rgba || d->textures[t].width <= 0 || d->textures[t].height <= 0)
			return fail(PTX_E_INVALID, "texture %d is empty", t);
	for (int m = 0; m < d->num_materials; ++m) {
		const PtxMaterial& pm = d->materials[m];
		if (pm.type < 0 || pm.type > 3) return fail(PTX_E_INVALID, "material %d: type %d", m, pm.type);
		if (pm.texture < -1 || pm.texture >= d->num_textures) return fail(PTX_E_INVALID, "material %d: texture %d out of range", m, pm.texture);
	}

	std::vector<DevObject> objs((size_t)d->num_objects);
	std::vector<float4> nodes, geom, shade, mats;
	nodes.reserve((size_t)d->num_nodes * 4);
	std::vector<float4> pairs;   // tri_pair blocks (pt_scene.h)
	std::vector<float4> grp_tris, grp_boxes;   // tri_grp / grp_box (pt_scene.h)
	std::vector<float4> cheap_pairs;           // pair records of the single-leaf meshes (pt_scene.h)
	geom.reserve((size_t)d->num_leaf_tris * 3);
	shade.reserve((size_t)d->num_leaf_tris * 4);
	int scene_max_depth = 0;
	int num_big = 0, big_object[PT_MAX_BIG + 1];

	for (int m = 0; m < d->num_materials; ++m) {
		const PtxMaterial& pm = d->materials[m];
		mats.push_back(make_float4(as_float(pm.type), pm.color[0], pm.color[1], pm.color[2]));
		mats.push_back(make_float4(as_float(pm.texture), pm.emittance[0], pm.emittance[1], pm.emittance[2]));
		mats.push_back(make_float4(pm.reflect_factor, pm.refract_index, pm.sharpness, 0.0f));
	}

	for (int oi = 0; oi < d->num_objects; ++oi) {
		const PtxObject& o = d->objects[oi];
		DevObject& dv = objs[(size_t)oi];
		memset(&dv, 0, sizeof(dv));
		if (o.type != PTX_OBJ_SPHERE && o.type != PTX_OBJ_MESH) return fail(PTX_E_INVALID, "object %d: type %d", oi, o.type);
		if (o.num_materials < 1 || o.first_material < 0 || o.first_material + o.num_materials > d->num_materials)
			return fail(PTX_E_INVALID, "object %d: material range [%d,+%d) outside %d materials", oi, o.first_material, o.num_materials, d->num_materials);
		dv.type = o.type;
		dv.mat_first = o.first_material;
		dv.root_node = -1;
		dv.big_slot = -1;
		if (o.type == PTX_OBJ_SPHERE) {
			dv.radius = o.radius;
			dv.px = o.position[0]; dv.py = o.position[1]; dv.pz = o.position[2];
			continue;
		}
		if (o.num_triangles < 0 || o.first_triangle < 0 || o.first_triangle + o.num_triangles > d->num_triangles)
			return fail(PTX_E_INVALID, "object %d: triangle range outside scene", oi);
		if (o.num_nodes < 0 || o.first_node < 0 || o.first_node + o.num_nodes > d->num_nodes)
			return fail(PTX_E_INVALID, "object %d: node range outside scene", oi);
		if (o.num_leaf_tris < 0 || o.first_leaf_tri < 0 || o.first_leaf_tri + o.num_leaf_tris > d->num_leaf_tris)
			return fail(PTX_E_INVALID, "object %d: leaf triangle range outside scene", oi);
		dv.num_triangles = o.num_triangles;
		if (o.num_nodes == 0 || o.num_triangles == 0) continue; // empty mesh: never hit

		const PtxKDNode* on = d->nodes + o.first_node;
		const int base = (int)(nodes.size() / 4);
		dv.first_node = base;
		dv.root_node = base;
		for (int k = 0; k < 3; ++k) { dv.bmin[k] = on[0].bmin[k]; dv.bmax[k] = on[0].bmax[k]; }
		float grp_pad = 0.0f;   // padding of the group boxes (pt_scene.h): absolute + relative to the mesh size
		{
			double dg = 0.0;
			for (int k = 0; k < 3; ++k) dg += ((double)on[0].bmax[k] - on[0].bmin[k]) * ((double)on[0].bmax[k] - on[0].bmin[k]);
			grp_pad = 5e-5f + 1e-6f * (float)ctx->grp_pad_ppm * (float)sqrt(dg);
			if (!(grp_pad < 1e30f)) grp_pad = 1e30f;
		}

		// Walk the tree from the root: checks that it IS a tree (every node reached once, so the
		// device loop terminates) and measures its depth for the LDS stack.
		std::vector<unsigned char> seen((size_t)o.num_nodes, 0);
		std::vector<std::pair<int, int> > todo; // (node, depth)
		todo.push_back(std::make_pair(0, 0));
		int depth_max = 0;
		nodes.resize(nodes.size() + (size_t)o.num_nodes * 4, make_float4(0, 0, 0, 0));
		// nodes never reached stay as zero records with left=right=0 ... make them harmless leaves
		for (int k = 0; k < o.num_nodes; ++k)
			nodes[(size_t)(base + k) * 4 + 3] = make_float4(as_float(-1), as_float(-1), as_float(0), as_float(0));
		while (!todo.empty()) {
			const int ni = todo.back().first, dep = todo.back().second;
			todo.pop_back();
			if (ni < 0 || ni >= o.num_nodes) return fail(PTX_E_INVALID, "object %d: child index %d outside node range", oi, ni);
			if (seen[(size_t)ni]) return fail(PTX_E_INVALID, "object %d: node %d reached twice (not a tree)", oi, ni);
			seen[(size_t)ni] = 1;
			if (dep > depth_max) depth_max = dep;
			if (dep > PTX_MAX_KD_DEPTH) return fail(PTX_E_LIMIT, "object %d: KD depth exceeds %d", oi, PTX_MAX_KD_DEPTH);
			const PtxKDNode& nd = on[ni];
			float4* rec = &nodes[(size_t)(base + ni) * 4];
			const bool leaf = nd.left == -1 && nd.right == -1;
			if (leaf) {
				if (nd.num_tris < 0 || nd.first_tri < 0 || nd.first_tri + nd.num_tris > o.num_leaf_tris)
					return fail(PTX_E_INVALID, "object %d node %d: leaf triangle list outside range", oi, ni);
				const int first_dev = (int)(geom.size() / 3);
				for (int k = 0; k < nd.num_tris; ++k) {
					const int ti = d->leaf_tris[o.first_leaf_tri + nd.first_tri + k];
					if (ti < 0 || ti >= o.num_triangles) return fail(PTX_E_INVALID, "object %d node %d: triangle index %d", oi, ni, ti);
					const PtxTriangle& t = d->triangles[o.first_triangle + ti];
					if (t.material < 0 || t.material >= o.num_materials)
						return fail(PTX_E_INVALID, "object %d triangle %d: material %d outside [0,%d)", oi, ti, t.material, o.num_materials);
					geom.push_back(make_float4(t.a[0], t.a[1], t.a[2], t.e1[0]));
					geom.push_back(make_float4(t.e1[1], t.e1[2], t.e2[0], t.e2[1]));
					geom.push_back(make_float4(t.e2[2], as_float(ti), as_float(ni), 0.0f));
					shade.push_back(make_float4(t.na[0], t.na[1], t.na[2], t.ta[0]));
					shade.push_back(make_float4(t.nb[0], t.nb[1], t.nb[2], t.ta[1]));
					shade.push_back(make_float4(t.nc[0], t.nc[1], t.nc[2], t.tb[0]));
					shade.push_back(make_float4(t.tb[1], t.tc[0], t.tc[1], as_float(o.first_material + t.material)));
				}
				// every leaf run starts on a multiple of 32 (zero records in between: never indexed, never hit),
				// so that run and tri_pair block share one number: first = 32 * block
				while ((geom.size() / 3) % 32 != 0) {
					for (int z = 0; z < 3; ++z) geom.push_back(make_float4(0, 0, 0, 0));
					for (int z = 0; z < 4; ++z) shade.push_back(make_float4(0, 0, 0, 0));
				}
				rec[3] = make_float4(as_float(-1), as_float(-1), as_float(first_dev), as_float(nd.num_tris));
				// the same run again as pair-interleaved blocks of 32 slots for the cooperative kernel
				const int block_first = (int)(pairs.size() / PT_PAIR_BLOCK_F4);
				if (block_first * 32 != first_dev) return fail(PTX_E_INVALID, "internal: leaf run %d not aligned with block %d", first_dev, block_first);
				const int nblocks = (nd.num_tris + 31) / 32;
				pairs.resize(pairs.size() + (size_t)nblocks * PT_PAIR_BLOCK_F4, make_float4(0, 0, 0, 0));
				for (int k = 0; k < nd.num_tris; ++k) {
					const PtxTriangle& t = d->triangles[o.first_triangle + d->leaf_tris[o.first_leaf_tri + nd.first_tri + k]];
					const float comp[9] = {t.a[0], t.a[1], t.a[2], t.e1[0], t.e1[1], t.e1[2], t.e2[0], t.e2[1], t.e2[2]};
					const int b = k / 32, r = k % 32, pr = r % 16, hi = r / 16;
					for (int c = 0; c < 9; ++c) {
						float4& q = pairs[(size_t)(block_first + b) * PT_PAIR_BLOCK_F4 + (size_t)(c / 2) * 16 + pr];
						(&q.x)[(c % 2) * 2 + hi] = comp[c];
					}
				}
				rec[0] = make_float4(as_float(block_first), 0.0f, 0.0f, 0.0f);
				if (ni == 0) {
					// a single-leaf mesh is intersected inline by the segment kernel: a padded box for every PAIR of consecutive
					// triangles of its run (pt_scene.h, cheap_pairs), so that a lane tests only the pairs whose box its ray touches
					const int pair_base = (int)(cheap_pairs.size() / 2);
					const int npairs = (nd.num_tris + 1) / 2;
					for (int j = 0; j < npairs; ++j) {
						float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
						for (int h = 0; h < 2 && 2 * j + h < nd.num_tris; ++h) {
							const PtxTriangle& t = d->triangles[o.first_triangle + d->leaf_tris[o.first_leaf_tri + nd.first_tri + 2 * j + h]];
							for (int c = 0; c < 3; ++c) {
								const float v[3] = {t.a[c], t.a[c] + t.e1[c], t.a[c] + t.e2[c]};
								for (int z = 0; z < 3; ++z) {
									if (v[z] - grp_pad < lo[c]) lo[c] = v[z] - grp_pad;
									if (v[z] + grp_pad > hi[c]) hi[c] = v[z] + grp_pad;
								}
							}
						}
						cheap_pairs.push_back(make_float4(lo[0], lo[1], lo[2], 0.0f));
						cheap_pairs.push_back(make_float4(hi[0], hi[1], hi[2], 0.0f));
					}
					rec[0].y = as_float(pair_base);
				}
				// and as boxed groups of eight, sorted along the longest axis of the leaf's box (pt_scene.h)
				{
					const float inf = __builtin_huge_valf();
					grp_tris.resize((size_t)(block_first + nblocks) * 4 * PT_GRP_BLOCK_F4, make_float4(0, 0, 0, 0));
					grp_boxes.resize((size_t)(block_first + nblocks) * 4 * 2, make_float4(0, 0, 0, 0));
					for (int g = 0; g < nblocks * 4; ++g) {
						grp_boxes[(size_t)(block_first * 4 + g) * 2] = make_float4(inf, inf, inf, 0.0f);
						grp_boxes[(size_t)(block_first * 4 + g) * 2 + 1] = make_float4(-inf, -inf, -inf, 0.0f);
					}
					// Order of the run's triangles in the group arrays; groups are consecutive eights of it.
					//   grp_mode 0: sorted along the longest axis of the leaf's box (strips)
					//   grp_mode 1: recursive bisection - the (sub)set is sorted by centroid along the longest axis of its
					//               centroid box and cut at a multiple of eight near its middle: squarer groups, smaller boxes
					std::vector<std::pair<double, int> > order((size_t)nd.num_tris);
					{
						std::vector<double> cen((size_t)nd.num_tris * 3);
						for (int k = 0; k < nd.num_tris; ++k) {
							const PtxTriangle& t = d->triangles[o.first_triangle + d->leaf_tris[o.first_leaf_tri + nd.first_tri + k]];
							for (int c = 0; c < 3; ++c) cen[(size_t)k * 3 + c] = 3.0 * (double)t.a[c] + (double)t.e1[c] + (double)t.e2[c];
							order[(size_t)k] = std::make_pair(0.0, k);
						}
						int axis0 = 0;
						for (int k = 1; k < 3; ++k) if (nd.bmax[k] - nd.bmin[k] > nd.bmax[axis0] - nd.bmin[axis0]) axis0 = k;
						std::vector<std::pair<int, int> > parts;   // [begin, end) ranges still to order
						parts.push_back(std::make_pair(0, nd.num_tris));
						while (!parts.empty()) {
							const int b = parts.back().first, e = parts.back().second;
							parts.pop_back();
							if (e - b <= 1) continue;
							int axis = axis0;
							if (ctx->grp_mode == 1) {
								double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
								for (int i = b; i < e; ++i)
									for (int c = 0; c < 3; ++c) {
										const double v = cen[(size_t)order[(size_t)i].second * 3 + c];
										if (v < lo[c]) lo[c] = v;
										if (v > hi[c]) hi[c] = v;
									}
								axis = 0;
								for (int c = 1; c < 3; ++c) if (hi[c] - lo[c] > hi[axis] - lo[axis]) axis = c;
							}
							for (int i = b; i < e; ++i) order[(size_t)i].first = cen[(size_t)order[(size_t)i].second * 3 + axis];
							// equal keys keep run order (the pairs compare by key, then by run index)
							std::sort(order.begin() + b, order.begin() + e);
							if (ctx->grp_mode != 1) continue;
							const int groups = (e - b + 7) / 8;
							if (groups <= 1) continue;
							const int mid = b + 8 * ((groups + 1) / 2);   // b is a multiple of eight, so every cut is one
							parts.push_back(std::make_pair(b, mid));
							parts.push_back(std::make_pair(mid, e));
						}
					}
					for (int j = 0; j < nd.num_tris; ++j) {
						const int k = order[(size_t)j].second;
						const PtxTriangle& t = d->triangles[o.first_triangle + d->leaf_tris[o.first_leaf_tri + nd.first_tri + k]];
						const float comp[9] = {t.a[0], t.a[1], t.a[2], t.e1[0], t.e1[1], t.e1[2], t.e2[0], t.e2[1], t.e2[2]};
						const size_t gi = (size_t)block_first * 4 + (size_t)(j / 8);
						const int r = j % 8, ln = r % 4, hi = r / 4;
						for (int c = 0; c < 9; ++c) {
							float4& q = grp_tris[gi * PT_GRP_BLOCK_F4 + (size_t)(c / 2) * 4 + ln];
							(&q.x)[(c % 2) * 2 + hi] = comp[c];
						}
						(&grp_tris[gi * PT_GRP_BLOCK_F4 + 16 + ln].x)[2 + hi] = as_float(first_dev + k);   // device index of the triangle: run order
						float4& lo = grp_boxes[gi * 2];
						float4& up = grp_boxes[gi * 2 + 1];
						for (int c = 0; c < 3; ++c) {
							const float v[3] = {t.a[c], t.a[c] + t.e1[c], t.a[c] + t.e2[c]};
							for (int z = 0; z < 3; ++z) {
								if (v[z] - grp_pad < (&lo.x)[c]) (&lo.x)[c] = v[z] - grp_pad;
								if (v[z] + grp_pad > (&up.x)[c]) (&up.x)[c] = v[z] + grp_pad;
							}
						}
					}
				}
			} else {
				float l[6] = {0, 0, 0, 0, 0, 0}, r[6] = {0, 0, 0, 0, 0, 0};
				if (nd.left != -1) {
					if (nd.left < 0 || nd.left >= o.num_nodes) return fail(PTX_E_INVALID, "object %d node %d: left child %d", oi, ni, nd.left);
					for (int k = 0; k < 3; ++k) { l[k] = on[nd.left].bmin[k]; l[3 + k] = on[nd.left].bmax[k]; }
					todo.push_back(std::make_pair(nd.left, dep + 1));
				}
				if (nd.right != -1) {
					if (nd.right < 0 || nd.right >= o.num_nodes) return fail(PTX_E_INVALID, "object %d node %d: right child %d", oi, ni, nd.right);
					for (int k = 0; k < 3; ++k) { r[k] = on[nd.right].bmin[k]; r[3 + k] = on[nd.right].bmax[k]; }
					todo.push_back(std::make_pair(nd.right, dep + 1));
				}
				rec[0] = make_float4(l[0], r[0], l[1], r[1]);
				rec[1] = make_float4(l[2], r[2], l[3], r[3]);
				rec[2] = make_float4(l[4], r[4], l[5], r[5]);
				// .x/.y: child node indices; .z/.w: the same links in packed form, filled in below
				rec[3] = make_float4(as_float(nd.left == -1 ? -1 : base + nd.left), as_float(nd.right == -1 ? -1 : base + nd.right),
				                     as_float(-1), as_float(-1));
			}
		}
		// Packed links (pt_scene.h): a child that is a leaf with 1..63 triangles is named by its triangle run,
		// 0x80000000 | block << 6 | count, so the cooperative traversal never fetches the leaf's node record.
		for (int k = 0; k < o.num_nodes; ++k) {
			float4& q3 = nodes[(size_t)(base + k) * 4 + 3];
			const int32_t links[2] = {as_int(q3.x), as_int(q3.y)};
			if (links[0] == -1 && links[1] == -1) continue; // leaf
			int32_t enc[2];
			for (int c = 0; c < 2; ++c) {
				enc[c] = links[c];
				if (links[c] < 0) continue;
				const float4 c3 = nodes[(size_t)links[c] * 4 + 3];
				const int32_t cf = as_int(c3.z), cc = as_int(c3.w);
				if (as_int(c3.x) == -1 && as_int(c3.y) == -1 && cc >= 1 && cc <= 63 && (cf >> 5) < (1 << 24))
					enc[c] = (int32_t)(0x80000000u | ((uint32_t)(cf >> 5) << 6) | (uint32_t)cc);
			}
			q3.z = as_float(enc[0]);
			q3.w = as_float(enc[1]);
		}
		if (depth_max > scene_max_depth) scene_max_depth = depth_max;
		if (!(on[0].left == -1 && on[0].right == -1)) {
			if (num_big < PT_MAX_BIG) { dv.big_slot = num_big; big_object[num_big] = oi; }
			num_big++; // more than PT_MAX_BIG multi-node meshes: the scene renders through the megakernel
		}
	}

	// Leaf-half records (pt_scene.h): the group boxes of every 32-triangle block once more, two per record
	// in the layout of an inner node's child boxes, appended to the node array so that the unified traversal
	// kernel fetches and tests them with the code path of an inner step. Record half_base + 2 * block + h
	// holds groups 2h and 2h + 1 of that block.
	const size_t half_base = nodes.size() / 4;
	{
		const size_t nblocks_all = grp_boxes.size() / 8;
		nodes.resize(nodes.size() + nblocks_all * 2 * 4, make_float4(0, 0, 0, 0));
		for (size_t b = 0; b < nblocks_all; ++b)
			for (int h = 0; h < 2; ++h) {
				const float4 llo = grp_boxes[(4 * b + 2 * h) * 2], lhi = grp_boxes[(4 * b + 2 * h) * 2 + 1];
				const float4 rlo = grp_boxes[(4 * b + 2 * h + 1) * 2], rhi = grp_boxes[(4 * b + 2 * h + 1) * 2 + 1];
				float4* rec = &nodes[(half_base + 2 * b + (size_t)h) * 4];
				rec[0] = make_float4(llo.x, rlo.x, llo.y, rlo.y);
				rec[1] = make_float4(llo.z, rlo.z, lhi.x, rhi.x);
				rec[2] = make_float4(lhi.y, rhi.y, lhi.z, rhi.z);
				rec[3] = make_float4(0, 0, 0, 0);
			}
	}
	// Quantised group boxes (pt_scene.h, grp_qbox): per 32-triangle block the union of its (up to) four group boxes
	// and each box in 8-bit steps of it, rounded outward.
	std::vector<float4> qboxes(grp_boxes.size() / 8 * 3, make_float4(0, 0, 0, 0));
	for (size_t b = 0; b < grp_boxes.size() / 8; ++b) {
		float lo[4][3], hi[4][3];
		bool used[4];
		for (int g = 0; g < 4; ++g) {
			const float4 l = grp_boxes[(4 * b + g) * 2], h = grp_boxes[(4 * b + g) * 2 + 1];
			used[g] = l.x <= h.x;   // unused slots hold (+inf, -inf)
			lo[g][0] = l.x; lo[g][1] = l.y; lo[g][2] = l.z; hi[g][0] = h.x; hi[g][1] = h.y; hi[g][2] = h.z;
		}
		quantise_boxes4(lo, hi, used, &qboxes[3 * b]);
	}
	if (qboxes.empty()) qboxes.assign(3, make_float4(0, 0, 0, 0));
	// Wide trees (pt_scene.h, wide_nodes): one per multi-node mesh, roots per object
	std::vector<float4> wide;
	std::vector<int32_t> wide_root((size_t)(d->num_objects > 0 ? d->num_objects : 1), -1);
	int wide_stack = 1, wide_sah_objects = 0;
	for (int oi = 0; oi < d->num_objects; ++oi) {
		const DevObject& dv = objs[(size_t)oi];
		if (dv.type != PTX_OBJ_MESH || dv.root_node < 0) continue;
		// two candidates, each built into its own array: the tree collapsed from the reference's median-split KD tree, and one
		// built over the same groups by the surface-area heuristic. The modelled cost (sum of child-box surfaces = expected box
		// tests of a random ray) picks; the collapsed tree keeps a 10 % bonus: on the evenly tessellated car the SAH tree has 4 %
		// fewer node visits but 5 % more triangle units and is 1 % slower, on unevenly tessellated meshes it wins by a wide margin
		// (tools/uneven_bench.py). ptx_ctx_tune "wide_build": 0 = choose (default), 1 = collapsed, 2 = surface-area heuristic.
		std::vector<float4> cand[2];
		int cand_root[2] = {-1, -1}, cand_stack[2] = {0, 0};
		double cand_cost[2] = {1e300, 1e300};
		for (int which = 0; which < 2; ++which) {
			if ((ctx->wide_build == 1 && which == 1) || (ctx->wide_build == 2 && which == 0)) continue;
			WideBuilder wb{nodes, grp_boxes, cand[which]};
			wb.base = dv.first_node;
			std::vector<WideItem> top;
			if (which == 0) {
				wb.fill_boxes(dv.root_node, d->objects[oi].num_nodes);
				const WideItem root = wb.kd_item(dv.root_node);
				if (WideBuilder::is_group(root)) top.push_back(root);
				else if (!WideBuilder::is_empty(root)) wb.children(root, top);
			} else {
				std::vector<int> ids;
				wb.object_groups(dv.root_node, ids);
				if (ids.size() == 1) top.push_back(wb.run_item(ids[0], 1));
				else if (ids.size() > 1) {
					const WideItem root = wb.bin_item(wb.build_sah(ids, 0, (int)ids.size()));
					wb.children(root, top);
				}
			}
			cand_root[which] = wb.emit_root(top);
			cand_stack[which] = wb.max_stack;
			cand_cost[which] = wb.cost;
		}
		int use = cand_root[0] < 0 ? 1 : (cand_root[1] < 0 ? 0 : (cand_cost[1] < 0.9 * cand_cost[0] ? 1 : 0));
		// rays that need more stack entries than a lane has (WFW_SCAP_MAX) go to the strict kernel: rare on either tree of the test
		// meshes (0 and 1400 of 10^8 traversals), but a degenerate chain would send most rays there: then prefer the tree that fits
		if (cand_root[0] >= 0 && cand_root[1] >= 0 && cand_stack[use] > 3 * WFW_SCAP_MAX && cand_stack[use ^ 1] <= 3 * WFW_SCAP_MAX) use ^= 1;
		if (getenv("PTX_WIDE_BUILD_VERBOSE"))
			fprintf(stderr, "object %d: wide tree cost collapsed %.4g (stack %d), surface-area heuristic %.4g (stack %d) -> %s\n", oi, cand_cost[0], cand_stack[0],
			        cand_cost[1], cand_stack[1], use ? "surface-area heuristic" : "collapsed");
		const int offset = (int)(wide.size() / 4);
		for (size_t n = 0; n < cand[use].size() / 4; ++n) {   // links to nodes are indices into the candidate's own array
			float4& links = cand[use][n * 4 + 3];
			float* lf = &links.x;
			for (int c = 0; c < 4; ++c) { const int32_t r = as_int(lf[c]); if (r >= 0) lf[c] = as_float(r + offset); }
		}
		wide.insert(wide.end(), cand[use].begin(), cand[use].end());
		wide_root[(size_t)oi] = cand_root[use] + offset;
		if (cand_stack[use] > wide_stack) wide_stack = cand_stack[use];
		wide_sah_objects += use;
	}
	if (wide.empty()) wide.assign(4, make_float4(0, 0, 0, 0));
	if (wide.size() / 4 >= ((size_t)1 << 24)) return fail(PTX_E_LIMIT, "scene has %zu wide nodes; the limit is 2^24", wide.size() / 4);
	// the wide traversal addresses node records and triangle groups by 32-bit BYTE offsets from their bases (one shift instead of a
	// sign extension and two 64-bit operations per fetch): 4 GB of groups = 13.4 million groups = 107 million triangles
	if (grp_tris.size() * sizeof(float4) >= ((size_t)1 << 32))
		return fail(PTX_E_LIMIT, "scene has %zu triangle groups; the limit is 2^32 / 320", grp_tris.size() / PT_GRP_BLOCK_F4);
	if (grp_boxes.size() / 2 >= ((size_t)1 << 26)) return fail(PTX_E_LIMIT, "scene has %zu triangle groups; the limit is 2^26", grp_boxes.size() / 2);
	// the traversal kernels address node / leaf-half records with 32-bit byte offsets from the array base
	if (nodes.size() / 4 >= ((size_t)1 << 26)) return fail(PTX_E_LIMIT, "scene has %zu KD node and leaf-half records; the limit is 2^26", nodes.size() / 4);
	HIP_TRY(hipSetDevice(ctx->device));
	PtxScene* s = new PtxScene();
	s->device = ctx->device;
	s->bytes = 0;
	s->num_objects = d->num_objects;
	s->max_depth = scene_max_depth;
	s->num_dev_triangles = (int)(geom.size() / 3);
	s->num_big = num_big;
	for (int k = 0; k < num_big && k < PT_MAX_BIG; ++k) s->big_object[k] = big_object[k];
	memset(&s->dev, 0, sizeof(s->dev));
	int rc;
#define UP(vec, field) \
	if ((rc = upload_vec(ctx, s, vec, &s->dev.field)) != PTX_OK) { free_scene_allocs(s); delete s; return rc; }
	UP(objs, objects)
	UP(nodes, nodes)
	UP(geom, tri_geom)
	{
		const size_t nt = geom.size() / 3;
		std::vector<float4> soa(nt * 3);
		for (size_t i = 0; i < nt; ++i) {
			const float4 g0 = geom[3 * i], g1 = geom[3 * i + 1], g2 = geom[3 * i + 2];
			soa[i] = make_float4(g0.x, g0.y, g0.z, g2.y);
			soa[nt + i] = make_float4(g0.w, g1.x, g1.y, 0.0f);
			soa[2 * nt + i] = make_float4(g1.z, g1.w, g2.x, 0.0f);
		}
		UP(soa, tri_soa)
		s->dev.num_tris = (int)nt;
	}
	if (pairs.empty()) pairs.push_back(make_float4(0, 0, 0, 0));
	UP(pairs, tri_pair)
	if (grp_tris.empty()) grp_tris.push_back(make_float4(0, 0, 0, 0));
	if (grp_boxes.empty()) grp_boxes.push_back(make_float4(0, 0, 0, 0));
	UP(grp_tris, tri_grp)
	UP(grp_boxes, grp_box)
	UP(qboxes, grp_qbox)
	if (cheap_pairs.empty()) cheap_pairs.assign(2, make_float4(0, 0, 0, 0));
	UP(cheap_pairs, cheap_pairs)
	UP(wide, wide_nodes)
	UP(wide_root, wide_root)
	UP(shade, tri_shade)
	UP(mats, materials)
#undef UP
	std::vector<DevTexture> texs((size_t)d->num_textures);
	for (int t = 0; t < d->num_textures; ++t) {
		const PtxTexture& pt = d->textures[t];
		size_t bytes = (size_t)pt.width * (size_t)pt.height * sizeof(float4);
		void* dt = NULL;
		hipError_t e = hipMalloc(&dt, bytes);
		if (e != hipSuccess) { free_scene_allocs(s); delete s; return fail(PTX_E_HIP, "hipMalloc texture %d (%zu B): %s", t, bytes, hipGetErrorString(e)); }
		s->allocs.push_back(dt);
		s->bytes += (int64_t)bytes;
		e = hipMemcpyAsync(dt, pt.rgba, bytes, hipMemcpyHostToDevice, ctx->stream);
		if (e != hipSuccess) { free_scene_allocs(s); delete s; return fail(PTX_E_HIP, "hipMemcpy texture %d: %s", t, hipGetErrorString(e)); }
		texs[(size_t)t].texels = (const float4*)dt;
		texs[(size_t)t].width = pt.width;
		texs[(size_t)t].height = pt.height;
	}
	if ((rc = upload_vec(ctx, s, texs, &s->dev.textures)) != PTX_OK) { free_scene_allocs(s); delete s; return rc; }
	s->dev.num_objects = d->num_objects;
	s->dev.stack_entries = scene_max_depth < 1 ? 1 : scene_max_depth;
	s->dev.half_base = (int32_t)half_base;
	s->dev.wide_stack = wide_stack;
	s->wide_sah_objects = wide_sah_objects;
	s->dev.num_materials = d->num_materials;
	s->dev.num_textures = d->num_textures;
	hipError_t e = hipStreamSynchronize(ctx->stream); // host vectors die at return
	if (e != hipSuccess) { free_scene_allocs(s); delete s; return fail(PTX_E_HIP, "scene upload: %s", hipGetErrorString(e)); }
	*out = s;
	return PTX_OK;
}

extern "C" int ptx_scene_free(PtxCtx* ctx, PtxScene* scene) {
	if (!scene) return PTX_OK;
	if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); if (ctx->la_scene == scene) la_drop(ctx); }
	free_scene_allocs(scene);
	delete scene;
	return PTX_OK;
}

extern "C" int64_t ptx_scene_device_bytes(const PtxScene* scene) { return scene ? scene->bytes : 0; }
extern "C" int ptx_scene_wide_sah_objects(const PtxScene* scene) { return scene ? scene->wide_sah_objects : 0; }

// ---- render -----------------------------------------------------------------------------------

static int fill_camera(const PtxCamera* c, DevCamera* d) {
	if (c->width <= 0 || c->height <= 0) return fail(PTX_E_INVALID, "camera: resolution %dx%d", c->width, c->height);
	if ((int64_t)c->width * (int64_t)c->height > (int64_t)1 << 28) return fail(PTX_E_LIMIT, "camera: more than 2^28 pixels");
	if (c->max_ray_bounces < 0) return fail(PTX_E_INVALID, "camera: max_ray_bounces %d", c->max_ray_bounces);
	for (int k = 0; k < 3; ++k) {
		d->pos[k] = c->position[k]; d->dir[k] = c->direction[k]; d->up[k] = c->up[k]; d->right[k] = c->right[k];
	}
	d->width = c->width; d->height = c->height;
	d->pixel_size = c->pixel_size;
	d->half_x = c->screen_half[0]; d->half_y = c->screen_half[1];
	d->screen_dist = c->screen_dist;
	d->aa_jitter = c->aa_jitter;
	d->max_bounces = c->max_ray_bounces;
	d->preview_mode = c->preview_mode; d->aabb_mode = c->aabb_mode; d->texture_enabled = c->texture_enabled;
	return PTX_OK;
}

extern "C" int ptx_render_init(PtxCtx* ctx, const PtxCamera* cam, void* external_accum) {
	if (!ctx || !cam) return fail(PTX_E_INVALID, "ptx_render_init: NULL argument");
	HIP_TRY(hipSetDevice(ctx->device));
	if (ctx->has_render) ptx_render_cleanup(ctx);
	la_drop(ctx);
	ctx->cam_gen++;
	int rc = fill_camera(cam, &ctx->cam);
	if (rc != PTX_OK) return rc;
	const size_t bytes = (size_t)cam->width * (size_t)cam->height * sizeof(float4);
	if (external_accum) {
		ctx->accum = (float4*)external_accum;
		ctx->owns_accum = false;
	} else {
		HIP_TRY(hipMalloc((void**)&ctx->accum, bytes));
		ctx->owns_accum = true;
		HIP_TRY(hipMemsetAsync(ctx->accum, 0, bytes, ctx->stream));
	}
	ctx->has_render = true;
	return PTX_OK;
}

extern "C" int ptx_render_set_camera(PtxCtx* ctx, const PtxCamera* cam) {
	if (!ctx || !cam) return fail(PTX_E_INVALID, "ptx_render_set_camera: NULL argument");
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_render_set_camera before ptx_render_init");
	if (cam->width != ctx->cam.width || cam->height != ctx->cam.height)
		return fail(PTX_E_INVALID, "resolution change %dx%d -> %dx%d needs ptx_render_init", ctx->cam.width, ctx->cam.height, cam->width, cam->height);
	la_drop(ctx);   // samples traced ahead belong to the old camera
	ctx->cam_gen++;
	return fill_camera(cam, &ctx->cam);
}

extern "C" int ptx_render_cleanup(PtxCtx* ctx) {
	if (!ctx || !ctx->has_render) return PTX_OK;
	hipSetDevice(ctx->device);
	hipStreamSynchronize(ctx->stream);
	if (ctx->owns_accum && ctx->accum) hipFree(ctx->accum);
	wf_free(ctx);
	ctx->accum = NULL;
	ctx->owns_accum = false;
	ctx->has_render = false;
	return PTX_OK;
}

extern "C" int ptx_accum_clear(PtxCtx* ctx) {
	if (!ctx || !ctx->has_render) return fail(PTX_E_STATE, "ptx_accum_clear before ptx_render_init");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMemsetAsync(ctx->accum, 0, (size_t)ctx->cam.width * ctx->cam.height * sizeof(float4), ctx->stream));
	return PTX_OK;
}

// Division of n < 2^28 by a launch-uniform d in [2, 2^28] as mulhi(n, m) >> s: with L = ceil(log2 d) and k = 31 + L,
// m = floor(2^k / d) + 1 < 2^32 and the error n (m d - 2^k) <= n d < 2^(28 + L) < 2^k, so the quotient is exact
// (wf_magic_check compares it with n / d for the launch's extreme n in ptx_render_samples). d = 1 has no such pair: the kernel tests for it.
static void wf_magic_div(uint32_t d, uint32_t* m, uint32_t* s) {
	if (d <= 1) { *m = 0; *s = 0; return; }
	uint32_t L = 0;
	while (((uint64_t)1 << L) < d) ++L;
	const uint32_t k = 31 + L;
	*m = (uint32_t)((((uint64_t)1 << k) / d) + 1);
	*s = k - 32;
}

static bool wf_magic_check(uint32_t n, uint32_t d, uint32_t m, uint32_t sh) {
	if (d <= 1) return true;
	return (uint32_t)((((uint64_t)n * m) >> 32) >> sh) == n / d;
}

static size_t stack_lds_bytes(const PtxScene* s) { return (size_t)s->dev.stack_entries * PT_BLOCK * sizeof(int); }

static int check_scene(const PtxCtx* ctx, const PtxScene* scene, const char* who) {
	if (!ctx || !scene) return fail(PTX_E_INVALID, "%s: NULL argument", who);
	if (scene->device != ctx->device) return fail(PTX_E_INVALID, "%s: scene lives on device %d, context on %d", who, scene->device, ctx->device);
	if (stack_lds_bytes(scene) > 64 * 1024) return fail(PTX_E_LIMIT, "%s: traversal stack of %d levels does not fit LDS", who, scene->dev.stack_entries);
	return PTX_OK;
}

// One traversal launch of the wavefront pipeline: the rays queued for multi-node mesh `obj` (queue k).
//   trav_variant 0 (default)  the wide traversal (pt_wide.h), then the STRICT kernel over its retry queue
//                14           the same, counted by the wide kernel's own counting twin
//                15           the STRICT kernel alone: wf_traverse_coop_kernel in the reference's visiting order, testing EVERY triangle
//                             of every leaf it visits (leaf mode 4, or 3 when the tree's stack leaves no LDS for 4's staging area):
//                             the reference's test set, so also the hits its triangle test accepts out of rounding noise
//                7, 9         the strict kernel in leaf mode 3 / 4 by name (so that both are tested on ordinary scenes)
//                             (15, 7 and 9 are the STRICT SCHEDULE: ptx_render_samples then also runs the segment kernels that test every
//                             triangle of the single-leaf meshes instead of culling by their pair boxes, PT_F_ALLTRIS)
//                13           the reference's visiting order with GROUPED leaves (leaf mode 9: a leaf's triangles in groups of eight
//                             with a padded box each, only touched groups tested): 1.75 x as fast as 15, and like the wide
//                             traversal blind to a noise hit on a triangle whose group box the ray misses (7 of 19.7 million
//                             adversarial rays, tools/grazing_rate.py; until late round 3 this was what 15 selected and what the
//                             retry queue ran). Still the counting twin: its counters are the reference's by construction
// (Rounds 1-2 carried thirteen more A/B variants - one ray per lane, leaf modes 0-2 and 5-8, the unified item loop, probe
// builds of the wide kernel; their measurements are in DESIGN.md 3.2, their code is in the history.)
static void launch_traversal(PtxCtx* ctx, const PtxScene* scene, const WfParams& wp, int obj, int k, int collect_stats) {
	const int cus = ctx->cus;
	const size_t tb = (size_t)cus * (size_t)ctx->trav_blocks_per_cu; // persistent waves, each claims queue chunks
	const size_t lds = stack_lds_bytes(scene);
	// <stats, min waves/SIMD for the register allocator, queue chunk, descent threshold, leaf mode>
#define COOP(F, O, C, A, T) hipLaunchKernelGGL((wf_traverse_coop_kernel<F, O, C, A, T>), dim3((unsigned)tb), dim3(PT_BLOCK), lds + ((T) == 4 ? WF_STAGE_BYTES : (T) >= 5 ? WF_GRP_BYTES : 0), ctx->stream, scene->dev, wp, obj, k)
	const int scap = scene->dev.wide_stack < ctx->wide_scap ? scene->dev.wide_stack : ctx->wide_scap;
	const size_t wide_lds = (size_t)scap * PT_BLOCK * WFW_STACK_ENTRY_BYTES + WFW_BYTES;
	// <stats, waves/SIMD for the register allocator, queue chunk, lanes that must wait before a turn, units that must gather before a drain>
	// the wide kernel's grid: no more workgroups than are resident at once (six per CU: registers and LDS), so that none starts into a
	// drained queue behind another (16 / 12 / 6 per CU: 6.85 / 6.83 / 6.78 ms); trav_blocks below six still lowers it
	const size_t tbw = (size_t)cus * (size_t)(ctx->trav_blocks_per_cu < 6 ? ctx->trav_blocks_per_cu : 6);
#define WIDE(F) hipLaunchKernelGGL((wf_traverse_wide_kernel<F, 6, WFW_CHUNK, WFW_TURN_MIN, WFW_DRAIN_MIN>), dim3((unsigned)tbw), dim3(PT_BLOCK), wide_lds, ctx->stream, scene->dev, wp, obj, k, scap, 1.0f + 1e-6f * (float)ctx->wide_margin_ppm, 1e-6f * (float)ctx->wide_margin_ppm)
	const bool stage_fits = lds + WF_STAGE_BYTES <= 64 * 1024;
	const bool grp_fits = lds + WF_GRP_BYTES <= 64 * 1024;
	// the reference-order kernel: strict in the best leaf loop the tree's stack leaves LDS for, or the leaf mode asked for by name
	auto exact = [&](const WfParams& wp, size_t tb, bool stats) {
		// strict (4, else 3) unless the grouped kernel is asked for by name, or counters are collected (same counters, faster)
		const int mode = ctx->trav_variant == 7 ? 3 : ctx->trav_variant == 9 ? 4 : (ctx->trav_variant == 13 || stats) ? 9 : 4;
		if (stats) {
			if (mode == 9 && grp_fits) COOP(PT_F_STATS, 4, WF_CHUNK, WF_ATHRESH, 9);
			else if (mode != 3 && stage_fits) COOP(PT_F_STATS, 4, 64, 32, 4);
			else COOP(PT_F_STATS, 4, 64, 32, 3);
		} else {
			if (mode == 9 && grp_fits) COOP(0, 4, WF_CHUNK, WF_ATHRESH, 9);
			else if (mode != 3 && stage_fits) COOP(0, 5, 64, 32, 4);
			else COOP(0, 4, 64, 32, 3);
		}
	};
	// the rays the wide traversal could not decide (equal t from two leaves, stack overflow, fragile winner): few (bench
	// scene: about 250 of 40 million), so one workgroup per CU; a full persistent grid costs 0.3 ms
	auto exact_retry = [&]() {
		WfParams wr = wp;
		wr.use_retry = 1;
		exact(wr, (size_t)cus, false);
	};
	const bool wide = ctx->trav_variant == 0 || ctx->trav_variant == 14;
	if (collect_stats) {
		if (ctx->trav_variant == 14) { WIDE(PT_F_STATS); exact_retry(); }
		else exact(wp, tb, true);   // the reference's counters: the grouped reference-order kernel's counting twin
	} else if (wide) { WIDE(0); exact_retry(); }
	else exact(wp, tb, false);
#undef WIDE
#undef COOP
}

// one-lane-per-pixel megakernel instead of the wavefront pipeline: the AABB debug view, PTX_MEGAKERNEL, bounce limits and mesh
// counts beyond the wavefront kernels' tables
static bool uses_megakernel(const PtxCtx* ctx, const PtxScene* scene) {
	return ctx->cam.aabb_mode || ctx->force_megakernel || ctx->cam.max_bounces > PT_WF_MAX_BOUNCES || scene->num_big > PT_MAX_BIG;
}

// The render pass behind ptx_render_samples and ptx_render_sample_ahead. `park` != NULL (wavefront pipeline only): ONE pass of at
// most num_samples samples is traced and NOT accumulated - its radiance stays parked in wf.rad, plane s = sample first_sample + s -
// and *park = the number of samples it held (fewer than asked for if the path state of that many does not fit); `reserve` = the
// pass size to allocate for at once if the device has the room (the window's ramp 1, 2, 4 ... would otherwise free and allocate
// the state buffers at every step: 120 ms over the first loop of the headline frame).
static int render_samples_impl(PtxCtx* ctx, const PtxScene* scene, int first_sample, int num_samples,
                               uint64_t seed, int collect_stats, int* park, int reserve = 0) {
	int rc = check_scene(ctx, scene, "ptx_render_samples");
	if (rc != PTX_OK) return rc;
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_render_samples before ptx_render_init");
	if (first_sample < 0 || num_samples < 0 || (int64_t)first_sample + num_samples > 0x7fffffff)
		return fail(PTX_E_INVALID, "sample range [%d,+%d)", first_sample, num_samples);
	if (park) *park = 0;
	if (num_samples == 0) return PTX_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	la_drop(ctx);   // this pass overwrites the parked radiance of any window traced ahead
	const uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
	if (uses_megakernel(ctx, scene)) {
		if (park) return fail(PTX_E_STATE, "render_samples_impl: no parked pass through the megakernel");
		// one-lane-per-pixel megakernel: kept for the AABB debug view (its mask is a float product
		// taken in object order) and as an A/B reference (PTX_MEGAKERNEL=1)
		const int tiles_x = (ctx->cam.width + 15) / 16, tiles_y = (ctx->cam.height + 15) / 16;
		const dim3 grid((unsigned)(tiles_x * tiles_y)), block(PT_BLOCK);
		const size_t lds = stack_lds_bytes(scene);
		if (ctx->cam.aabb_mode)
			hipLaunchKernelGGL(render_kernel<PT_F_DEBUGMASK>, grid, block, lds, ctx->stream, scene->dev, ctx->cam, ctx->accum, first_sample, num_samples, k0, k1, ctx->d_stats);
		else if (collect_stats)
			hipLaunchKernelGGL(render_kernel<PT_F_STATS>, grid, block, lds, ctx->stream, scene->dev, ctx->cam, ctx->accum, first_sample, num_samples, k0, k1, ctx->d_stats);
		else
			hipLaunchKernelGGL(render_kernel<0>, grid, block, lds, ctx->stream, scene->dev, ctx->cam, ctx->accum, first_sample, num_samples, k0, k1, ctx->d_stats);
		HIP_TRY(hipGetLastError());
		return PTX_OK;
	}

	// ---- wavefront pipeline (pt_wavefront.h) ----------------------------------------------------
	const int W = ctx->cam.width, Hh = ctx->cam.height;
	const int tiles_x = (W + 7) / 8, tiles_y = (Hh + 7) / 8;
	const size_t npt = (size_t)tiles_x * tiles_y * 64, npix = (size_t)W * Hh;
	// path ids and their decomposition (multiply-high division, wf_magic_div) are proven for fewer than 2^28 paths per pass
	if (npt > ((size_t)1 << 28)) return fail(PTX_E_LIMIT, "ptx_render_samples: %zu padded pixels per sample exceed the 2^28 paths of a pass", npt);
	int batch = ctx->wf_batch;
	while (batch > 1 && npt * (size_t)batch > ((size_t)1 << 28)) batch >>= 1;
	if (batch > num_samples) batch = num_samples;
	const int cus = ctx->cus;
	const int maxb = ctx->cam.max_bounces;
	// state sets: one per bounce that has segments when the bounce loop runs inside the launch (pt_segloop.h), else two in turn
	// (its per-wave run table needs (bounces x lists) LDS entries: scenes with dozens of multi-node meshes AND deep bounce limits keep the per-bounce kernel)
	const bool seg_loop = ctx->seg_mode == 1 && WSL_DYN_BYTES(maxb, 1 + scene->num_big) <= 16 * 1024;
	const int nsets = seg_loop ? (maxb > 1 ? maxb : 1) : 2;
	// 64 B of state per path and set, a queue entry per set, 12 B of parked radiance: if the device cannot hold the preferred
	// batch (other tenants, huge frames, deep bounce limits), halve it rather than fail; the result does not depend on the batch size
	while (reserve > 1 && npt * (size_t)reserve > ((size_t)1 << 28)) reserve >>= 1;
	for (;;) {
		const int room = reserve > batch ? reserve : batch;
		rc = wf_ensure(ctx, (npt * (size_t)room + WF_ROUNDS * PT_BLOCK), npix * (size_t)room, scene->num_big, nsets);
		if (rc == PTX_OK || room == 1) break;
		wf_free(ctx);
		(void)hipGetLastError();   // clear the sticky out-of-memory status
		if (reserve > batch) reserve = 0; else batch >>= 1;
	}
	if (rc != PTX_OK) return rc;
	if (park) {   // one pass, whatever it holds
		if (num_samples > batch) num_samples = batch;
		*park = num_samples;
	}
	const size_t stride = ctx->wf_paths;
	for (int s0 = first_sample; s0 < first_sample + num_samples; s0 += batch) {
		const int b = (first_sample + num_samples - s0) < batch ? (first_sample + num_samples - s0) : batch;
		WfParams wp = ctx->wf;
		wp.stats = ctx->d_stats;
		wp.first_sample = s0;
		wp.batch = b;
		wp.npt = (int)npt;
		wp.tiles_x = tiles_x;
		wf_magic_div((uint32_t)npt, &wp.npt_m, &wp.npt_s);
		wf_magic_div((uint32_t)tiles_x, &wp.tx_m, &wp.tx_s);
		wp.npix = (int)npix;
		wp.n_total = (int)(npt * (size_t)b);
		if (!wf_magic_check((uint32_t)wp.n_total - 1u, (uint32_t)npt, wp.npt_m, wp.npt_s) || !wf_magic_check((uint32_t)npt - 1u, (uint32_t)npt, wp.npt_m, wp.npt_s) ||
		    !wf_magic_check((uint32_t)(npt >> 6) - 1u, (uint32_t)tiles_x, wp.tx_m, wp.tx_s) || !wf_magic_check((uint32_t)tiles_x, (uint32_t)tiles_x, wp.tx_m, wp.tx_s))
			return fail(PTX_E_LIMIT, "ptx_render_samples: multiply-high division check failed (npt %zu, tiles_x %d)", npt, tiles_x);
		wp.k0 = k0; wp.k1 = k1;
		wp.num_big = scene->num_big;
		wp.max_bounces = maxb;
		wp.set_stride = stride;
		wp.queue_stride = stride;
		HIP_TRY(hipMemsetAsync(wp.counters, 0, WF_COUNTER_INTS * sizeof(int), ctx->stream));
		for (int bounce = 0; bounce <= maxb; ++bounce) {
			wp.bounce = bounce;
			wp.is_last = bounce == maxb ? 1 : 0;
			size_t blocks = ((size_t)wp.n_total + WF_ROUNDS * PT_BLOCK - 1) / (WF_ROUNDS * PT_BLOCK);
			if (blocks > (size_t)cus * 16) blocks = (size_t)cus * 16;
			// the bounce-loop kernel's waves are persistent (every wave ends with a tail of partly filled steps): as many workgroups as the chip holds
			if (seg_loop && blocks > (size_t)cus * (size_t)ctx->seg_blocks_per_cu) blocks = (size_t)cus * (size_t)ctx->seg_blocks_per_cu;
			// material / texture-descriptor tables in LDS: the per-bounce kernel gains 2.5 % from them; the bounce-loop kernel loses its fifth
			// wave per SIMD to them (39.4 KB per workgroup instead of 30.7: 6.58 vs 5.70 ms per launch), so there they need seg_lds_tables = 2
			// the strict schedule (trav_variant 15 / 7 / 9): no triangle is skipped on the strength of a box that is not the reference's own
			const bool strict = ctx->trav_variant == 15 || ctx->trav_variant == 7 || ctx->trav_variant == 9;
			const bool lds_tables = (seg_loop ? ctx->seg_lds_tables >= 2 : ctx->seg_lds_tables >= 1) && scene->dev.num_materials <= WF_LDS_MATS && scene->dev.num_textures <= WF_LDS_TEXS;
			// the set this bounce's new rays are written to (and its traversals update): set `bounce` of the bounce-loop kernel,
			// else the two sets in turn
			const size_t out_set = seg_loop ? (size_t)bounce : (size_t)(bounce & 1);
			prof_mark(ctx, 1);
			if (seg_loop) {
				// the kernel addresses every set and queue segment itself: base pointers
				wp.s0 = ctx->wf.s0; wp.s1 = ctx->wf.s1; wp.s2 = ctx->wf.s2; wp.s3 = ctx->wf.s3;
				for (int k = 0; k < scene->num_big; ++k) wp.queue[k] = ctx->wf.queue[k];
				const size_t dyn = WSL_DYN_BYTES(maxb, 1 + scene->num_big);
				// 13-word pool entries (six workgroups per CU) where bounce, object and triangle fit one word (pt_segloop.h)
				const bool compact = ctx->seg_compact != 0 && maxb <= WSL_COMPACT_MAX_BOUNCES && scene->num_objects <= WSL_COMPACT_MAX_OBJECTS &&
				                     scene->dev.num_tris <= WSL_COMPACT_MAX_TRIS && dyn <= 256;
				if (collect_stats)
					hipLaunchKernelGGL((wf_segment_loop_kernel<PT_F_STATS, 0, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), dyn, ctx->stream, scene->dev, ctx->cam, wp);
				else if (strict)   // every triangle of the single-leaf meshes too (PT_F_ALLTRIS)
					hipLaunchKernelGGL((wf_segment_loop_kernel<PT_F_ALLTRIS, 0, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), dyn, ctx->stream, scene->dev, ctx->cam, wp);
				else if (lds_tables)
					hipLaunchKernelGGL((wf_segment_loop_kernel<0, 1, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), dyn, ctx->stream, scene->dev, ctx->cam, wp);
				else if (compact)
					hipLaunchKernelGGL((wf_segment_loop_kernel<0, 0, 1>), dim3((unsigned)blocks), dim3(PT_BLOCK), dyn, ctx->stream, scene->dev, ctx->cam, wp);
				else
					hipLaunchKernelGGL((wf_segment_loop_kernel<0, 0, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), dyn, ctx->stream, scene->dev, ctx->cam, wp);
			} else {
				const size_t in_set = (size_t)((bounce & 1) ^ 1);
				wp.s0 = ctx->wf.s0 + out_set * stride; wp.s1 = ctx->wf.s1 + out_set * stride; wp.s2 = ctx->wf.s2 + out_set * stride; wp.s3 = ctx->wf.s3 + out_set * stride;
				wp.i0 = ctx->wf.s0 + in_set * stride; wp.i1 = ctx->wf.s1 + in_set * stride; wp.i2 = ctx->wf.s2 + in_set * stride; wp.i3 = ctx->wf.s3 + in_set * stride;
				if (collect_stats)
					hipLaunchKernelGGL((wf_segment_kernel<PT_F_STATS, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), 0, ctx->stream, scene->dev, ctx->cam, wp);
				else if (strict)
					hipLaunchKernelGGL((wf_segment_kernel<PT_F_ALLTRIS, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), 0, ctx->stream, scene->dev, ctx->cam, wp);
				else if (lds_tables)   // material / texture-descriptor tables in LDS when they fit (pt_wavefront.h)
					hipLaunchKernelGGL((wf_segment_kernel<0, 1>), dim3((unsigned)blocks), dim3(PT_BLOCK), 0, ctx->stream, scene->dev, ctx->cam, wp);
				else
					hipLaunchKernelGGL((wf_segment_kernel<0, 0>), dim3((unsigned)blocks), dim3(PT_BLOCK), 0, ctx->stream, scene->dev, ctx->cam, wp);
			}
			prof_mark(ctx, 1);
			HIP_TRY(hipGetLastError());
			if (wp.is_last) break;
			// the traversal kernels see ONE set and ONE queue segment: this bounce's
			wp.s0 = ctx->wf.s0 + out_set * stride; wp.s1 = ctx->wf.s1 + out_set * stride; wp.s2 = ctx->wf.s2 + out_set * stride; wp.s3 = ctx->wf.s3 + out_set * stride;
			for (int k = 0; k < scene->num_big; ++k) wp.queue[k] = ctx->wf.queue[k] + (seg_loop ? out_set * stride : 0);
			for (int k = 0; k < scene->num_big; ++k) {
				prof_mark(ctx, 0);
				launch_traversal(ctx, scene, wp, scene->big_object[k], k, collect_stats);
				prof_mark(ctx, 0);
				HIP_TRY(hipGetLastError());
			}
		}
		if (park) break;   // the caller adds the planes one by one
		hipLaunchKernelGGL(wf_accumulate_kernel, dim3((unsigned)((npix + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream,
		                   ctx->accum, (const Rad3*)wp.rad, (int)npix, b);
		HIP_TRY(hipGetLastError());
	}
	return PTX_OK;
}

extern "C" int ptx_render_samples(PtxCtx* ctx, const PtxScene* scene, int first_sample, int num_samples,
                                  uint64_t seed, int collect_stats) {
	return render_samples_impl(ctx, scene, first_sample, num_samples, seed, collect_stats, NULL);
}

// The reference renders ONE sample per call (kernelMain, src/kernel.cu:59-91, in the loop of qt/cudarunthread.cpp:21-41). A
// one-sample pass cannot fill this chip: the headline frame has 2 M paths per sample for 0.4 M lanes, and every launch of the
// pass lasts as long as its longest ray (2.3 ms per sample, against 0.48 ms per sample in a 128-sample pass). So this entry point
// ADDS exactly one sample to the accumulation buffer, like ptx_render_samples(ctx, scene, sample, 1, seed, 0) and bit for bit
// (radiance is parked per (sample, pixel) and added in sample order in either case), but TRACES AHEAD: when `sample` is not in
// the window it holds, it traces [sample, sample + w) in one pass, parks the radiance and adds plane 0; the following calls
// find their sample in the window and only add its plane (one 14-us kernel). w starts at 1 and doubles while the calls continue
// the window (sample = its end, same scene / seed / camera), up to max_ahead: a moving camera pays for one sample per frame, a
// standing one gets passes of max_ahead. Anything that changes what a sample is, or overwrites the parked radiance, ends the
// window: ptx_render_set_camera / _init / _cleanup, ptx_render_samples, ptx_ctx_tune, ptx_scene_free.
extern "C" int ptx_render_sample_ahead(PtxCtx* ctx, const PtxScene* scene, int sample, uint64_t seed, int max_ahead) {
	int rc = check_scene(ctx, scene, "ptx_render_sample_ahead");
	if (rc != PTX_OK) return rc;
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_render_sample_ahead before ptx_render_init");
	if (sample < 0 || sample == 0x7fffffff) return fail(PTX_E_INVALID, "ptx_render_sample_ahead: sample %d", sample);
	if (max_ahead <= 1 || uses_megakernel(ctx, scene)) return render_samples_impl(ctx, scene, sample, 1, seed, 0, NULL);
	HIP_TRY(hipSetDevice(ctx->device));
	const bool same = ctx->la_scene == scene && ctx->la_seed == seed && ctx->la_gen == ctx->cam_gen;
	const bool held = same && ctx->la_count > 0 && sample >= ctx->la_first && sample - ctx->la_first < ctx->la_count;
	if (!held) {
		int w = 1;
		if (same && ctx->la_window > 0 && sample == ctx->la_first + ctx->la_count) w = ctx->la_window * 2;
		if (w > max_ahead) w = max_ahead;
		if (w > ctx->wf_batch) w = ctx->wf_batch;
		if (w > 0x7fffffff - sample) w = 0x7fffffff - sample;
		int got = 0;
		rc = render_samples_impl(ctx, scene, sample, w, seed, 0, &got, max_ahead < ctx->wf_batch ? max_ahead : ctx->wf_batch);   // drops the old window first
		if (rc != PTX_OK) return rc;
		ctx->la_scene = scene; ctx->la_seed = seed; ctx->la_gen = ctx->cam_gen;
		ctx->la_first = sample; ctx->la_count = got; ctx->la_window = w;
	}
	const size_t npix = (size_t)ctx->cam.width * ctx->cam.height;
	hipLaunchKernelGGL(wf_accumulate_kernel, dim3((unsigned)((npix + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream,
	                   ctx->accum, (const Rad3*)ctx->wf.rad + (size_t)(sample - ctx->la_first) * npix, (int)npix, 1);
	HIP_TRY(hipGetLastError());
	return PTX_OK;
}

extern "C" int ptx_wavefront_counts(PtxCtx* ctx, int max_bounces, int32_t* live_out, int32_t* queued_out) {
	if (!ctx || !live_out || !queued_out || max_bounces < 0 || max_bounces > PT_WF_MAX_BOUNCES) return fail(PTX_E_INVALID, "ptx_wavefront_counts: bad arguments");
	if (!ctx->wf.counters) return fail(PTX_E_STATE, "ptx_wavefront_counts: no wavefront pass has run");
	HIP_TRY(hipSetDevice(ctx->device));
	std::vector<int> h((size_t)(max_bounces + 1) * (1 + PT_MAX_BIG));
	HIP_TRY(hipMemcpyAsync(h.data(), ctx->wf.counters, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	for (int b = 0; b <= max_bounces; ++b) {
		live_out[b] = h[(size_t)b * (1 + PT_MAX_BIG)];
		int q = 0;
		for (int k = 0; k < PT_MAX_BIG; ++k) q += h[(size_t)b * (1 + PT_MAX_BIG) + 1 + k];
		queued_out[b] = q;
	}
	return PTX_OK;
}

static void prof_mark(PtxCtx* c, int cls) {
	if (!c->profiling) return;
	if (!c->prof_ev[cls]) c->prof_ev[cls] = new std::vector<hipEvent_t>();
	hipEvent_t e;
	if (hipEventCreate(&e) != hipSuccess) return;
	hipEventRecord(e, c->stream);
	c->prof_ev[cls]->push_back(e);
}

static void prof_clear(PtxCtx* c) {
	for (int k = 0; k < 2; ++k)
		if (c->prof_ev[k]) {
			for (size_t i = 0; i < c->prof_ev[k]->size(); ++i) hipEventDestroy((*c->prof_ev[k])[i]);
			c->prof_ev[k]->clear();
		}
}

extern "C" int ptx_profile_enable(PtxCtx* ctx, int enable) {
	if (!ctx) return fail(PTX_E_INVALID, "ptx_profile_enable: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	prof_clear(ctx);
	ctx->profiling = enable != 0;
	return PTX_OK;
}

extern "C" int ptx_profile_get(PtxCtx* ctx, float* traverse_ms, int32_t* traverse_launches, float* segment_ms, int32_t* segment_launches) {
	if (!ctx || !traverse_ms || !traverse_launches || !segment_ms || !segment_launches) return fail(PTX_E_INVALID, "ptx_profile_get: NULL argument");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	float* ms[2] = {traverse_ms, segment_ms};
	int32_t* cnt[2] = {traverse_launches, segment_launches};
	for (int k = 0; k < 2; ++k) {
		*ms[k] = 0.0f;
		*cnt[k] = 0;
		if (!ctx->prof_ev[k]) continue;
		const std::vector<hipEvent_t>& v = *ctx->prof_ev[k];
		for (size_t i = 0; i + 1 < v.size(); i += 2) {
			float e = 0.0f;
			HIP_TRY(hipEventElapsedTime(&e, v[i], v[i + 1]));
			*ms[k] += e;
			(*cnt[k])++;
		}
	}
	prof_clear(ctx);
	return PTX_OK;
}

extern "C" int ptx_render_sync(PtxCtx* ctx) {
	if (!ctx) return fail(PTX_E_INVALID, "ptx_render_sync: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" void* ptx_accum_device_ptr(PtxCtx* ctx) { return ctx ? (void*)ctx->accum : NULL; }

extern "C" int ptx_accum_download(PtxCtx* ctx, float* rgba_host) {
	if (!ctx || !rgba_host) return fail(PTX_E_INVALID, "ptx_accum_download: NULL argument");
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_accum_download before ptx_render_init");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMemcpyAsync(rgba_host, ctx->accum, (size_t)ctx->cam.width * ctx->cam.height * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_accum_upload(PtxCtx* ctx, const float* rgba_host) {
	if (!ctx || !rgba_host) return fail(PTX_E_INVALID, "ptx_accum_upload: NULL argument");
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_accum_upload before ptx_render_init");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMemcpyAsync(ctx->accum, rgba_host, (size_t)ctx->cam.width * ctx->cam.height * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_tonemap(PtxCtx* ctx, int iter, float gamma, unsigned char* rgba8_host) {
	if (!ctx || !rgba8_host) return fail(PTX_E_INVALID, "ptx_tonemap: NULL argument");
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_tonemap before ptx_render_init");
	if (iter < 1) return fail(PTX_E_INVALID, "ptx_tonemap: iter %d < 1", iter);
	HIP_TRY(hipSetDevice(ctx->device));
	const size_t n = (size_t)ctx->cam.width * ctx->cam.height;
	if (ctx->pbo_pixels < n) {
		if (ctx->d_pbo) { HIP_TRY(hipStreamSynchronize(ctx->stream)); hipFree(ctx->d_pbo); ctx->d_pbo = NULL; ctx->pbo_pixels = 0; }
		HIP_TRY(hipMalloc((void**)&ctx->d_pbo, n * sizeof(uchar4)));
		ctx->pbo_pixels = n;
	}
	hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream,
	                   ctx->d_pbo, gamma, (int)n, iter, (const float4*)ctx->accum);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(rgba8_host, ctx->d_pbo, n * sizeof(uchar4), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

// Page-locked host memory for the tone-mapped image (the reference maps a GL pixel buffer once, qt/previewglwidget.cpp:80-86:
// its kernel writes the display buffer in place; here the image crosses PCIe every iteration). Into pageable memory the copy
// of the 1920x1080 image takes 0.45 ms of a 1.2-ms iteration, into memory from here 0.15 ms.
extern "C" int ptx_host_alloc(PtxCtx* ctx, size_t bytes, void** out) {
	if (!ctx || !out || bytes == 0) return fail(PTX_E_INVALID, "ptx_host_alloc: bad argument");
	*out = NULL;
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
	return PTX_OK;
}

extern "C" int ptx_host_free(PtxCtx* ctx, void* p) {
	if (!p) return PTX_OK;
	if (!ctx) return fail(PTX_E_INVALID, "ptx_host_free: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->stream));   // a tone-map copy into it may be in flight
	HIP_TRY(hipHostFree(p));
	return PTX_OK;
}

extern "C" int ptx_stats_get(PtxCtx* ctx, PtxStats* out) {
	if (!ctx || !out) return fail(PTX_E_INVALID, "ptx_stats_get: NULL argument");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMemcpyAsync(out, ctx->d_stats, sizeof(PtxStats), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_stats_reset(PtxCtx* ctx) {
	if (!ctx) return fail(PTX_E_INVALID, "ptx_stats_reset: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMemsetAsync(ctx->d_stats, 0, sizeof(PtxStats), ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_timer_begin(PtxCtx* ctx) {
	if (!ctx) return fail(PTX_E_INVALID, "ptx_timer_begin: NULL context");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
	ctx->timer_open = true;
	return PTX_OK;
}

extern "C" int ptx_timer_end(PtxCtx* ctx, float* elapsed_ms) {
	if (!ctx || !elapsed_ms) return fail(PTX_E_INVALID, "ptx_timer_end: NULL argument");
	if (!ctx->timer_open) return fail(PTX_E_STATE, "ptx_timer_end without ptx_timer_begin");
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
	HIP_TRY(hipEventSynchronize(ctx->ev1));
	HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
	ctx->timer_open = false;
	return PTX_OK;
}

// ---- parity entry points ------------------------------------------------------------------------

struct DevBuf {
	void* p;
	DevBuf() : p(NULL) {}
	~DevBuf() { if (p) hipFree(p); }
};

extern "C" int ptx_trace_rays(PtxCtx* ctx, const PtxScene* scene, int n, const float* origins, const float* dirs,
                              int object, PtxHit* hits_host) {
	int rc = check_scene(ctx, scene, "ptx_trace_rays");
	if (rc != PTX_OK) return rc;
	if (n < 0 || (n > 0 && (!origins || !dirs || !hits_host))) return fail(PTX_E_INVALID, "ptx_trace_rays: bad ray arrays");
	if (object < -1 || object >= scene->num_objects) return fail(PTX_E_INVALID, "ptx_trace_rays: object %d outside [-1,%d)", object, scene->num_objects);
	if (n == 0) return PTX_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	DevBuf d_o, d_d, d_h;
	const size_t rb = (size_t)n * 3 * sizeof(float);
	HIP_TRY(hipMalloc(&d_o.p, rb));
	HIP_TRY(hipMalloc(&d_d.p, rb));
	HIP_TRY(hipMalloc(&d_h.p, (size_t)n * sizeof(PtxHit)));
	HIP_TRY(hipMemcpyAsync(d_o.p, origins, rb, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemcpyAsync(d_d.p, dirs, rb, hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(trace_kernel, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), stack_lds_bytes(scene), ctx->stream,
	                   scene->dev, n, (const float*)d_o.p, (const float*)d_d.p, object, (PtxHit*)d_h.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(hits_host, d_h.p, (size_t)n * sizeof(PtxHit), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_trace_rays_queued(PtxCtx* ctx, const PtxScene* scene, int n, const float* origins, const float* dirs,
                                     int object, PtxHit* hits_host) {
	int rc = check_scene(ctx, scene, "ptx_trace_rays_queued");
	if (rc != PTX_OK) return rc;
	if (n < 0 || (n > 0 && (!origins || !dirs || !hits_host))) return fail(PTX_E_INVALID, "ptx_trace_rays_queued: bad ray arrays");
	int slot = -1;
	for (int k = 0; k < scene->num_big && k < PT_MAX_BIG; ++k) if (scene->big_object[k] == object) slot = k;
	if (slot < 0) return fail(PTX_E_INVALID, "ptx_trace_rays_queued: object %d is not a multi-node mesh with a traversal queue", object);
	if (n == 0) return PTX_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	DevBuf d_o, d_d, d_h, d_s, d_q, d_c, d_r;
	const size_t rb = (size_t)n * 3 * sizeof(float);
	const size_t ncnt = WF_COUNTER_INTS;
	HIP_TRY(hipMalloc(&d_o.p, rb));
	HIP_TRY(hipMalloc(&d_d.p, rb));
	HIP_TRY(hipMalloc(&d_h.p, (size_t)n * sizeof(PtxHit)));
	HIP_TRY(hipMalloc(&d_s.p, (size_t)n * 4 * sizeof(float4)));
	HIP_TRY(hipMalloc(&d_q.p, (size_t)n * sizeof(int)));
	HIP_TRY(hipMalloc(&d_c.p, ncnt * sizeof(int)));
	HIP_TRY(hipMalloc(&d_r.p, (size_t)n * sizeof(int)));
	HIP_TRY(hipMemcpyAsync(d_o.p, origins, rb, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemcpyAsync(d_d.p, dirs, rb, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemsetAsync(d_c.p, 0, ncnt * sizeof(int), ctx->stream));
	WfParams wp;
	memset(&wp, 0, sizeof(wp));
	wp.s0 = (float4*)d_s.p; wp.s1 = wp.s0 + n; wp.s2 = wp.s1 + n; wp.s3 = wp.s2 + n;
	wp.queue[slot] = (int*)d_q.p;
	wp.counters = (int*)d_c.p;
	wp.retry = (int*)d_r.p;
	wp.stats = ctx->d_stats;
	wp.n_total = n;
	wp.num_big = scene->num_big;
	const dim3 grid((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), block(PT_BLOCK);
	hipLaunchKernelGGL(wfq_fill_kernel, grid, block, 0, ctx->stream, scene->dev, wp, n, (const float*)d_o.p, (const float*)d_d.p, object, slot);
	HIP_TRY(hipGetLastError());
	launch_traversal(ctx, scene, wp, object, slot, 0);
	HIP_TRY(hipGetLastError());
	hipLaunchKernelGGL(wfq_result_kernel, grid, block, 0, ctx->stream, scene->dev, wp, n, object, (PtxHit*)d_h.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(hits_host, d_h.p, (size_t)n * sizeof(PtxHit), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_generate_rays(PtxCtx* ctx, const PtxCamera* cam, int sample, uint64_t seed, float* origins_host,
                                 float* dirs_host) {
	if (!ctx || !cam || !origins_host || !dirs_host) return fail(PTX_E_INVALID, "ptx_generate_rays: NULL argument");
	DevCamera dc;
	int rc = fill_camera(cam, &dc);
	if (rc != PTX_OK) return rc;
	HIP_TRY(hipSetDevice(ctx->device));
	const size_t n = (size_t)cam->width * cam->height;
	DevBuf d_o, d_d;
	HIP_TRY(hipMalloc(&d_o.p, n * 3 * sizeof(float)));
	HIP_TRY(hipMalloc(&d_d.p, n * 3 * sizeof(float)));
	hipLaunchKernelGGL(genrays_kernel, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, dc, sample,
	                   (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), (float*)d_o.p, (float*)d_d.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(origins_host, d_o.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipMemcpyAsync(dirs_host, d_d.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_bsdf_eval(PtxCtx* ctx, int n, PtxBsdfCase* cases_host) {
	if (!ctx || n < 0 || (n > 0 && !cases_host)) return fail(PTX_E_INVALID, "ptx_bsdf_eval: bad arguments");
	if (n == 0) return PTX_OK;
	for (int i = 0; i < n; ++i)
		if (cases_host[i].mat_type < 0 || cases_host[i].mat_type > 3) return fail(PTX_E_INVALID, "ptx_bsdf_eval: case %d material type %d", i, cases_host[i].mat_type);
	HIP_TRY(hipSetDevice(ctx->device));
	DevBuf d_c;
	HIP_TRY(hipMalloc(&d_c.p, (size_t)n * sizeof(PtxBsdfCase)));
	HIP_TRY(hipMemcpyAsync(d_c.p, cases_host, (size_t)n * sizeof(PtxBsdfCase), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(bsdf_kernel, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, n, (PtxBsdfCase*)d_c.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(cases_host, d_c.p, (size_t)n * sizeof(PtxBsdfCase), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_selftest_rcp(PtxCtx* ctx, float lo, float hi, uint64_t* checked, uint64_t* mismatches, uint32_t* first_bad_bits) {
	if (!ctx || !checked || !mismatches || !first_bad_bits) return fail(PTX_E_INVALID, "ptx_selftest_rcp: NULL argument");
	HIP_TRY(hipSetDevice(ctx->device));
	DevBuf d;
	HIP_TRY(hipMalloc(&d.p, 3 * sizeof(unsigned long long)));
	unsigned long long h[3] = {0, 0, ~0ull};
	HIP_TRY(hipMemcpyAsync(d.p, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(rcp_selftest_kernel, dim3(256 * 16), dim3(PT_BLOCK), 0, ctx->stream, lo, hi, (unsigned long long*)d.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	*checked = h[0];
	*mismatches = h[1];
	*first_bad_bits = (uint32_t)h[2];
	return PTX_OK;
}

extern "C" int ptx_selftest_vec3(PtxCtx* ctx, int n, const float* a_host, const float* b_host, const float* s_host, float* out_host) {
	if (!ctx || n < 0 || (n > 0 && (!a_host || !b_host || !s_host || !out_host))) return fail(PTX_E_INVALID, "ptx_selftest_vec3: bad arguments");
	if (n == 0) return PTX_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	DevBuf da, db, ds, dout;
	HIP_TRY(hipMalloc(&da.p, (size_t)n * 3 * sizeof(float)));
	HIP_TRY(hipMalloc(&db.p, (size_t)n * 3 * sizeof(float)));
	HIP_TRY(hipMalloc(&ds.p, (size_t)n * sizeof(float)));
	HIP_TRY(hipMalloc(&dout.p, (size_t)n * 20 * sizeof(float)));
	HIP_TRY(hipMemcpyAsync(da.p, a_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemcpyAsync(db.p, b_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(hipMemcpyAsync(ds.p, s_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(vec3_selftest_kernel, dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, ctx->stream, n,
	                   (const float*)da.p, (const float*)db.p, (const float*)ds.p, (float*)dout.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out_host, dout.p, (size_t)n * 20 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return PTX_OK;
}

extern "C" int ptx_radiance_sample(PtxCtx* ctx, const PtxScene* scene, int sample, uint64_t seed, float* rgb_host) {
	int rc = check_scene(ctx, scene, "ptx_radiance_sample");
	if (rc != PTX_OK) return rc;
	if (!ctx->has_render) return fail(PTX_E_STATE, "ptx_radiance_sample before ptx_render_init");
	if (!rgb_host || sample < 0) return fail(PTX_E_INVALID, "ptx_radiance_sample: bad arguments");
	HIP_TRY(hipSetDevice(ctx->device));
	const size_t n = (size_t)ctx->cam.width * ctx->cam.height;
	DevBuf tmp;
	HIP_TRY(hipMalloc(&tmp.p, n * sizeof(float4)));
	HIP_TRY(hipMemsetAsync(tmp.p, 0, n * sizeof(float4), ctx->stream));
	float4* saved = ctx->accum;
	ctx->accum = (float4*)tmp.p;
	rc = ptx_render_samples(ctx, scene, sample, 1, seed, 0);
	ctx->accum = saved;
	if (rc != PTX_OK) return rc;
	std::vector<float> host(n * 4);
	HIP_TRY(hipMemcpyAsync(host.data(), tmp.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	for (size_t i = 0; i < n; ++i) {
		rgb_host[3 * i] = host[4 * i]; rgb_host[3 * i + 1] = host[4 * i + 1]; rgb_host[3 * i + 2] = host[4 * i + 2];
	}
	return PTX_OK;
}
