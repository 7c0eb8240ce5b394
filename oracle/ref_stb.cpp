// extern "C" window onto the reference's own image decoder, external/include/stb/stb_image.h (2.16), compiled
// unchanged exactly as src/world_load.cpp:8-10 compiles it (one C++ translation unit with
// STB_IMAGE_IMPLEMENTATION; x86-64 => its SSE2 kernels are on, as in the reference's host build).
// TEST INFRASTRUCTURE ONLY. No reference code in this file: the header is included by path at build time.
#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>

// reference call: src/world_load.cpp:134  stbi_loadf(path, &w, &h, &channels, 0)
extern "C" float* ref_stbi_loadf(const char* path, int* w, int* h, int* channels) { return stbi_loadf(path, w, h, channels, 0); }
extern "C" unsigned char* ref_stbi_load8(const char* path, int* w, int* h, int* channels) { return stbi_load(path, w, h, channels, 0); }
extern "C" void ref_stbi_free(void* p) { stbi_image_free(p); }
extern "C" const char* ref_stbi_failure(void) { return stbi_failure_reason(); }
