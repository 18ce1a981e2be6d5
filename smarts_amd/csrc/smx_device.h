// smx_device.h — device-side tables and numeric helpers shared by the kernels.
// gfx950 only.  Built with -ffp-contract=off: the integer / flag outputs of this path
// (lane ids, events, done) hang on floating-point comparisons, so products and sums
// must round exactly like the reference's Python floats (no FMA contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/smx.h"

#define SMX_PI 3.141592653589793
#define SMX_TWO_PI 6.283185307179586
#define SMX_HALF_PI 1.5707963267948966

// Device copy of smx_map_tables: same fields, device pointers — plus the route table of the agents' missions
// (smx_set_missions), which rides with the map so that a route filter stays three words.
struct MapDev : smx_map_tables {
  const int16_t* route_pos;      // [slots][n_roads]: position of a road in the slot's route, -1 = not on it; may be null
  const uint8_t* route_lane_ok;  // [slots][n_lanes]: may a path of the slot's route continue onto this lane (lanepoints.py:666-683)
  MapDev() = default;
  MapDev(const smx_map_tables& t) : smx_map_tables(t), route_pos(nullptr), route_lane_ok(nullptr) {}
};

// Developer build (-DSMX_DEBUG_BOUNDS): every table index is checked; the first violation's site
// code and value are recorded instead of faulting.
#ifdef SMX_DEBUG_BOUNDS
__device__ int smx_dbg_site = 0;
__device__ long long smx_dbg_value = 0;
__device__ int smx_dbg_aux[8] = {0, 0, 0, 0, 0, 0, 0, 0};
__device__ double smx_dbg_f[64];
#define SMX_BCHK(site, idx, n)                                                  \
  (((idx) < 0 || (long long)(idx) >= (long long)(n))                            \
       ? (atomicCAS(&smx_dbg_site, 0, (site)) == 0 ? (smx_dbg_value = (long long)(idx), 0) : 0) \
       : (idx))
#else
#define SMX_BCHK(site, idx, n) (idx)
#endif

// Developer build (-DSMX_DEBUG_TIMING): per-phase wave-clock accumulation.
#ifdef SMX_DEBUG_TIMING
__device__ unsigned long long smx_prof[128];
#define SMX_TSTAMP(var) unsigned long long var = wall_clock64()
// (one workgroup in 64 reports: every wavefront adding to the same word serialises the chip's atomics and the
//  stamps then measure themselves; slot + 64 counts the reports)
#define SMX_TACC(slot, t0, t1) \
  do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&smx_prof[slot], (t1) - (t0)); atomicAdd(&smx_prof[(slot) + 64], 1ull); } } while (0)
#define SMX_TACC_ALL(slot, t0, t1) \
  do { if (threadIdx.x == 0) { atomicAdd(&smx_prof[slot], (t1) - (t0)); atomicAdd(&smx_prof[(slot) + 64], 1ull); } } while (0)
#define SMX_COUNT(slot, cond) \
  do { if (cond) atomicAdd(&smx_prof[slot], 1ull); } while (0)
// a wavefront's whole span in a kernel, every wavefront its own word (no atomics: 8 192 of them on one address take
// longer than the kernel): the kernel ends with its slowest wavefront
#define SMX_SPAN_KERNELS 8
#define SMX_SPAN_WAVES 16384
__device__ unsigned int smx_span[SMX_SPAN_KERNELS * SMX_SPAN_WAVES];
#define SMX_TSPAN(slot, t0, t1)                                                                       \
  do {                                                                                                \
    const unsigned w__ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                         \
    if ((threadIdx.x & 63) == 0 && w__ < SMX_SPAN_WAVES) smx_span[(slot) * SMX_SPAN_WAVES + w__] = (unsigned int)((t1) - (t0)); \
  } while (0)
#else
#define SMX_TSTAMP(var)
#define SMX_TACC(slot, t0, t1)
#define SMX_TACC_ALL(slot, t0, t1)
#define SMX_COUNT(slot, cond)
#define SMX_TSPAN(slot, t0, t1)
#endif

// ---------------------------------------------------------------------------------
// angle helpers (reference smarts/core/utils/math.py, smarts/core/coordinates.py)
// ---------------------------------------------------------------------------------
// Python / numpy float modulo for a positive divisor.
__device__ __forceinline__ double py_mod(double a, double b) {
  // |a| < b is the common case on this path (angles): the remainder is then `a` itself, or
  // `a + b` for negative `a` — exactly what fmod-then-adjust yields, without the slow fmod.
  if (a > -b && a < b) return a < 0.0 ? a + b : a;  // one branch and a select (the same values as two tests)
  double m = fmod(a, b);
  if (m != 0.0 && m < 0.0) m += b;
  return m;
}

// Heading.__new__ (coordinates.py:175-184): wrap to (-pi, pi].
__device__ __forceinline__ double wrap_heading(double x) {
  double v = py_mod(x, SMX_TWO_PI);
  if (v > SMX_PI) v -= SMX_TWO_PI;
  return v;
}

// Heading.relative_to (coordinates.py:227-239).
__device__ __forceinline__ double heading_relative_to(double h, double other) {
  return wrap_heading(wrap_heading(h - other));
}

// min_angles_difference_signed (math.py:447-449).
__device__ __forceinline__ double min_angles_difference_signed(double first, double second) {
  return py_mod((first - second) + SMX_PI, SMX_TWO_PI) - SMX_PI;
}

// vec_to_radians (math.py:256-277).
__device__ __forceinline__ double vec_to_radians(double x, double y) {
  double r = atan2(fabs(y), fabs(x));
  if (x < 0.0) {
    if (y < 0.0) return py_mod(r + 0.5 * SMX_PI, SMX_TWO_PI);
    return py_mod(0.5 * SMX_PI - r, SMX_TWO_PI);
  } else if (y < 0.0) {
    return py_mod(1.5 * SMX_PI - r, SMX_TWO_PI);
  }
  return py_mod(r - 0.5 * SMX_PI, SMX_TWO_PI);
}

// radians_to_vec (math.py:247-253).
__device__ __forceinline__ void radians_to_vec(double radians, double& vx, double& vy) {
  double angle = py_mod(radians + SMX_PI * 0.5, SMX_TWO_PI);
  vx = cos(angle);
  vy = sin(angle);
}

// lerp (math.py:206-216).
__device__ __forceinline__ double lerp_ref(double a, double b, double p) { return a * (1.0 - p) + b * p; }

__device__ __forceinline__ double clip_ref(double v, double lo, double hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// signed_dist_to_line (math.py:163-185): point p, line through lp with direction d.
__device__ __forceinline__ double signed_dist_to_line(double px, double py, double lx, double ly, double dx,
                                                      double dy) {
  double p2x = lx + dx, p2y = ly + dy;
  double u = fabs(dy * px - dx * py + p2x * ly - p2y * lx);
  double d = u / sqrt(dx * dx + dy * dy);
  double nx = -dy, ny = dx;
  double dot = (px - lx) * nx + (py - ly) * ny;
  double sgn = dot > 0.0 ? 1.0 : (dot < 0.0 ? -1.0 : 0.0);
  return d * sgn;
}

// ---------------------------------------------------------------------------------
// point <-> lane centre line (sumolib.geomhelper twins: math.py:293-433)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double euclid(double ax, double ay, double bx, double by) {
  double dx = ax - bx, dy = ay - by;
  return sqrt(dx * dx + dy * dy);
}

// distance_point_to_line(point, p1, p2, perpendicular=False) (math.py:393-411)
__device__ __forceinline__ double dist_point_segment(double px, double py, double x1, double y1, double x2,
                                                     double y2) {
  double d = euclid(x1, y1, x2, y2);
  double u = ((px - x1) * (x2 - x1)) + ((py - y1) * (y2 - y1));
  double offset;
  if (d == 0.0 || u < 0.0 || u > d * d) {
    offset = (u < 0.0) ? 0.0 : d;
  } else {
    offset = u / d;
  }
  if (offset == 0.0) return euclid(px, py, x1, y1);
  double uu = offset / d;
  double ix = x1 + uu * (x2 - x1), iy = y1 + uu * (y2 - y1);
  return euclid(px, py, ix, iy);
}
