// Path-tracing driver one level below the render boundary (reference: src/pathtracing.cuh:14-21). Same three
// host entry points with the reference's signatures, so src/kernel.cu's call sites read unchanged:
//     kernelInit      -> initPathTracing(scene)
//     kernelMain      -> runPathTracing(scene, wang_hash(iter), wang_hash(wang_hash(iter)))
//     kernelCleanUp   -> cleanUpPathTracing()
// In the reference the two hashes seed one cuRAND stream per thread and launch. Here random numbers are
// counter-based, keyed (master seed, pixel, GLOBAL SAMPLE INDEX, bounce), so what runPathTracing needs from
// its arguments is the iteration number: wang_hash is a bijection on 32 bits and is inverted
// (wang_unhash), iteration i renders sample i - 1. jitterHash is accepted for signature parity; the jitter's
// random numbers come from the same counter stream (dimension 0xFFFFFFFF).
#pragma once
#include "world.h"
#include <stdint.h>

// reference: src/cutility.cuh:7-15 (Thomas Wang's 32-bit integer hash, burtleburtle.net/bob/hash/integer.html)
inline unsigned int wang_hash(unsigned int seed) {
	seed = (seed ^ 61u) ^ (seed >> 16);
	seed *= 9u;
	seed = seed ^ (seed >> 4);
	seed *= 0x27d4eb2du;
	seed = seed ^ (seed >> 15);
	return seed;
}
// its inverse: wang_unhash(wang_hash(x)) == x for every 32-bit x
unsigned int wang_unhash(unsigned int h);

void initPathTracing(const Scene& scene);
void cleanUpPathTracing();
// Renders ONE sample of every pixel into the accumulation buffer (reference: src/pathtracing.cu:156-182:
// [regenerate primaries if the camera changed,] jitter, tracePaths). Status through checkCudaError().
void runPathTracing(Scene& scene, int iterHash, int jitterHash);
