// Program-level helpers (reference: src/main.cpp:13-41 printSep / printProgramConfig / detectCUDAdevice).
#include "main.h"
#include "config.h"
#include "kernel.h"
#include "world_load.h"

#include <cstdio>

void printSep() { printf("----------------------------------------------------------------\n"); }

void printProgramConfig() {
	printSep();
	printf("MI355X path tracer (HIP, gfx950)\n");
	printf("  MAX_NUM_RAY_BOUNCES: %d\n  MAX_KD_LEAF_TRGS: %d\n  MAX_OBJECTS_NUM: %d\n", MAX_NUM_RAY_BOUNCES, MAX_KD_LEAF_TRGS, MAX_OBJECTS_NUM);
	printSep();
}

// reference picks device 0 and raises the CUDA stack limit for the recursive traversal;
// the traversal stack here lives in LDS, so only the device choice remains.
int detectHIPdevice() {
	int n = ptx_device_count();
	if (n <= 0) {
		printf("No HIP devices detected: %s\n", ptx_last_error());
		return -1;
	}
	for (int d = 0; d < n; ++d) {
		char name[256];
		int cus = 0, khz = 0;
		uint64_t mem = 0;
		if (ptx_device_info(d, name, sizeof(name), &cus, &khz, &mem) == PTX_OK)
			printf("  device %d: %s, %d CUs, %.0f MHz, %.1f GiB\n", d, name, cus, khz / 1000.0, mem / 1073741824.0);
	}
	if (hip_device_id >= n) hip_device_id = 0;
	return hip_device_id;
}
