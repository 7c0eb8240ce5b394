/*
 * host_capi.h — plain-C window onto the C++ host layer (scene JSON -> loader -> KD build ->
 * render calls with the reference's entry-point names), for the ctypes harness and bench.py.
 * These forward to the functions in world_load.h / scenestate.h / kernel.h; they add no logic.
 */
#ifndef PT_HOST_CAPI_H
#define PT_HOST_CAPI_H
#include "../../include/ptx_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

const char* pth_last_error(void);
void pth_set_asset_root(const char* dir);
void pth_set_device(int device);
void pth_set_device_upload(int enabled);            /* 0: host-side load only (CPU tests) */
void pth_set_procedural_car_triangles(int n);
void pth_set_seed(uint64_t seed);
/* samples pth_run_iteration (= runCUDA) may trace ahead of the one it adds (ptx_render_sample_ahead); 1 = none; default 16 */
void pth_set_look_ahead(int samples);
int pth_look_ahead(void);

/* parseJsonScene + sceneState.load (reference: qt/sceneloaderthread.cpp:11-23); width/height > 0
 * force the resolution, textures/preview/aabb/bounces < 0 keep the Camera::init defaults except
 * that preview_mode defaults to OFF here (the GUI's "normal" radio, qt/mainwindow.cpp:244-250). */
int pth_scene_load(const char* json_path, int width, int height, int textures, int preview, int aabb, int bounces);
int pth_scene_free(void);
int pth_scene_is_loaded(void);
const PtxSceneDesc* pth_scene_desc(void);            /* what ptx_scene_upload received */
const float* pth_scene_tri_abc(void);                /* 9 floats per triangle (a, b, c) */
PtxScene* pth_scene_device(void);
int pth_camera_get(PtxCamera* out);
int pth_camera_control(const char* what, float value); /* "h_ang","v_ang","pos_x","pos_y","pos_z","reset","bounces","aa_jitter","gamma","preview","aabb","textures" */
int pth_log_lines(char* buf, int buf_len);
unsigned int pth_wang_hash(unsigned int x);
unsigned int pth_wang_unhash(unsigned int h);
int pth_texture_failures(char* buf, int buf_len);

/* kernel.h */
PtxCtx* pth_context(void);
int pth_kernel_init(void);                             /* kernelInit(sceneState.clone()) */
int pth_run_iteration(int iter, unsigned char* pbo_rgba8); /* runCUDA(pbo, scene, iter) */
int pth_render_samples(int first_sample, int num_samples, int collect_stats);
int pth_kernel_cleanup(void);
void pth_shutdown(void);

/* KD build on caller triangles (buildKDTree + flatenKDTree); outputs malloc'ed, release with pth_free */
int pth_kd_build(const float* abc, int n, PtxKDNode** nodes_out, int32_t** leaf_tris_out, int* num_leaf_tris_out, int* max_depth_out);
void pth_free(void* p);
float pth_fmedian(float* arr, int n);
/* mesh transforms as the loader composes and applies them (world_load.cpp composeTransform, glmlite.h); ops = n records
 * of 5 floats (kind 0 translate / 1 rotate degrees+axis / 2 scale, then 4 numbers); m16 column-major, normal16 = transpose(inverse(M)) */
void pth_transform_compose(const float* ops, int n, float* m16, float* normal16);
void pth_transform_apply(const float* m16, const float* v3, int n, float* out4);

/* image helpers */
int pth_load_image_float(const char* path, int* w, int* h, int* channels, float** pixels_out);
int pth_write_png(const char* path, int w, int h, int comp, const unsigned char* pixels, int mirror_h);
int pth_write_pfm(const char* path, int w, int h, const float* rgb);


/* checkpoint.h: float accumulation checkpoint / resume of the context's buffer, autosave naming */
int pth_checkpoint_save(const char* path, int samples_done);          /* downloads the accumulation buffer */
int pth_checkpoint_load(const char* path, int* samples_done);         /* checks resolution + seed, uploads */
/* the file format itself, host memory only (rgba: width*height*4 floats) */
int pth_checkpoint_write_file(const char* path, int width, int height, int samples_done, uint64_t seed, const float* rgba);
int pth_checkpoint_read_file(const char* path, int* width, int* height, int* samples_done, uint64_t* seed, float* rgba, int64_t rgba_capacity_floats);
int pth_autosave_due(int iter, int every);
int pth_autosave_name(const char* folder, const char* name, int iter, int day, int month, int year, char* out, int out_len);

#ifdef __cplusplus
}
#endif
#endif
