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
