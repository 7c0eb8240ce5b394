// Common host declarations (mirrors the role of reference src/main.h:1-23:
// alignMem(), dv_ptr, printSep; the CUDA device picker becomes a HIP one).
#pragma once
#include <string>
#include "json/json.h"

#define alignMem(n) __attribute__((aligned(n)))

typedef void* dv_ptr;

void printSep();
void printProgramConfig();
// reference: src/main.cpp:22-41 detectCUDAdevice(); here it asks the C ABI for gfx950 devices.
int detectHIPdevice();

extern int hip_device_id;
