// Host stand-in for <hip/hip_runtime.h>: lets g++ compile the device headers of smarts_amd/csrc (the walk /
// interpolation code of smx_roadmap.h) as plain C++ so that AddressSanitizer / UBSan can watch them run.
// Test infrastructure only (tests/native/host_walk.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
using std::max;
using std::min;
// (glibc declares sincos itself)

// One-lane "teams" (smx_scan.h with TEAM = 1): a shuffle returns the lane's own value.
struct ShimDim3 { unsigned x, y, z; };
static const ShimDim3 threadIdx = {0, 0, 0};
template <class T> inline T __shfl_xor(T v, int, int = 64) { return v; }
template <class T> inline T __shfl(T v, int, int = 64) { return v; }
template <class T> inline T __shfl_up(T v, unsigned, int = 64) { return v; }
