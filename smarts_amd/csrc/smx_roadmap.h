// smx_roadmap.h — device-side road-map queries: nearest lane, closest lanepoints,
// waypoint paths.  Batched restatement of
//   SumoRoadNetwork.nearest_lanes / road_with_point      sumo_road_network.py:675-709
//   LanePoints.closest_lanepoints / closest_linked_*      lanepoints.py:526-644
//   LanePoints.paths_starting_at_lanepoint                lanepoints.py:646-692
//   SumoRoadNetwork.waypoint_paths / _equally_spaced_path sumo_road_network.py:815-882, 1312-1437
// over flat tables (smx_map_tables) with uniform grids instead of KD-/R-trees.
#pragma once
#include "smx_device.h"

#define SMX_INF 1.0e300

// ---------------------------------------------------------------------------------
// nearest lane(s)
// ---------------------------------------------------------------------------------
struct LaneHit {
  int lane;      // argmin-distance lane (ties: lowest table index), -1 if none within radius
  double dist;   // its centre-line distance
  bool on_road;  // road_with_point(): some lane with dist < 0.5*width + 0.1 within road radius
};

// All lanes whose centre polyline is closer than `radius` to (px, py); the nearest one is
// what RoadMap.nearest_lane returns (road_map.py:91-96).  The segment grid only prunes: a
// lane is within `radius` iff one of its segments is, and that segment's bounding box then
// meets the query square, so it is listed in a visited cell.
__device__ inline LaneHit nearest_lane_scan(const MapDev& m, double px, double py, double radius) {
  LaneHit hit;
  hit.lane = -1;
  hit.dist = SMX_INF;
  hit.on_road = false;
  const double road_radius = fmax(5.0, 2.0 * m.default_lane_width);  // sumo_road_network.py:705
  int cx0 = (int)floor((px - radius - m.sg_x0) / m.sg_cell);
  int cx1 = (int)floor((px + radius - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((py - radius - m.sg_y0) / m.sg_cell);
  int cy1 = (int)floor((py + radius - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  for (int cy = cy0; cy <= cy1; ++cy) {
    for (int cx = cx0; cx <= cx1; ++cx) {
      int c = cy * m.sg_nx + cx;
      int a = m.sg_off[c], b = m.sg_off[c + 1];
      for (int k = a; k < b; ++k) {
        int s = m.sg_idx[k];
        int v = m.seg_v0[s];
        int lane = m.seg_lane[s];
        double d = dist_point_segment(px, py, m.shape_x[v], m.shape_y[v], m.shape_x[v + 1], m.shape_y[v + 1]);
        if (d < radius) {
          if (d < hit.dist || (d == hit.dist && lane < hit.lane)) {
            hit.dist = d;
            hit.lane = lane;
          }
          if (d < road_radius && d < 0.5 * m.lane_width[lane] + 1e-1) hit.on_road = true;
        }
      }
    }
  }
  return hit;
}

// ---------------------------------------------------------------------------------
// lanepoint nearest-neighbour queries on the uniform grid
// ---------------------------------------------------------------------------------
// Visit every lanepoint in ring `r` (Chebyshev) around cell (cx, cy).
template <class F>
__device__ __forceinline__ void lp_ring_visit(const MapDev& m, int cx, int cy, int r, F&& f) {
  int y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
  for (int y = max(y0, 0); y <= min(y1, m.lpg_ny - 1); ++y) {
    bool edge_row = (y == y0) || (y == y1);
    int step = edge_row ? 1 : max(2 * r, 1);
    for (int x = x0; x <= x1; x += step) {
      if (x < 0 || x >= m.lpg_nx) continue;
      int c = y * m.lpg_nx + x;
      int a = m.lpg_off[c], b = m.lpg_off[c + 1];
      for (int k = a; k < b; ++k) f(m.lpg_idx[k]);
    }
  }
}

__device__ __forceinline__ int lp_max_ring(const MapDev& m, int cx, int cy) {
  int rx = max(abs(cx), abs(cx - (m.lpg_nx - 1)));
  int ry = max(abs(cy), abs(cy - (m.lpg_ny - 1)));
  return max(rx, ry);
}

// Ring r is complete => every unvisited lanepoint is farther than r*cell from the query.
__device__ __forceinline__ bool ring_covers(const MapDev& m, int r, double d2) {
  if (r < 1) return false;
  double reach = (double)r * m.lpg_cell - 1e-6;
  return d2 <= reach * reach;
}

// closest_lanepoints([pose], within_radius, maximum_count=10)[0] (lanepoints.py:526-623):
// the 10 nearest lanepoints, those beyond within_radius dropped (the nearest always kept),
// winner = min of dist^2 + |heading difference|.  Ties resolve to the nearer point, then to the
// lower table index.  within_radius < 0 means None.
__device__ inline int closest_lanepoint(const MapDev& m, double px, double py, double heading,
                                        double within_radius) {
  const int K = 10;
  double bd[K];
  int bi[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    bd[i] = SMX_INF;
    bi[i] = -1;
  }
  int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  int keff = min(K, m.n_lanepoints);
  int rmax = lp_max_ring(m, cx, cy);
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit(m, cx, cy, r, [&](int i) {
      double dx = m.lp_x[i] - px, dy = m.lp_y[i] - py;
      double d2 = dx * dx + dy * dy;
      if (d2 < bd[K - 1] || (d2 == bd[K - 1] && i < bi[K - 1])) {
        // insert into the sorted top-K (registers; fully unrolled bubble)
        double cd = d2;
        int ci = i;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          bool lt = (cd < bd[k]) || (cd == bd[k] && ci < bi[k]);
          double td = lt ? bd[k] : cd;
          int ti = lt ? bi[k] : ci;
          bd[k] = lt ? cd : bd[k];
          bi[k] = lt ? ci : bi[k];
          cd = td;
          ci = ti;
        }
      }
    });
    if (bi[keff - 1] >= 0 && ring_covers(m, r, bd[keff - 1])) break;
  }
  double r2 = within_radius * within_radius;
  int best = -1;
  double best_score = SMX_INF;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (bi[k] < 0) continue;
    if (within_radius >= 0.0 && k > 0 && !(bd[k] <= r2)) continue;
    double score = bd[k] + fabs(heading_relative_to(heading, m.lp_heading[bi[k]]));
    if (score < best_score) {
      best_score = score;
      best = bi[k];
    }
  }
  return best;
}

// closest_linked_lanepoint_on_lane_to_point (lanepoints.py:629-636) when by_road == false,
// closest_linked_lanepoint_on_road (lanepoints.py:638-644) when by_road == true.
__device__ inline int closest_lanepoint_filtered(const MapDev& m, double px, double py, int key, bool by_road,
                                                 double* out_d2 = nullptr) {
  int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  int rmax = lp_max_ring(m, cx, cy);
  double bd = SMX_INF;
  int bi = -1;
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit(m, cx, cy, r, [&](int i) {
      int lane = m.lp_lane[i];
      int k = by_road ? m.lane_road[lane] : lane;
      if (k != key) return;
      double dx = m.lp_x[i] - px, dy = m.lp_y[i] - py;
      double d2 = dx * dx + dy * dy;
      if (d2 < bd || (d2 == bd && i < bi)) {
        bd = d2;
        bi = i;
      }
    });
    if (bi >= 0 && ring_covers(m, r, bd)) break;
  }
  if (out_d2) *out_d2 = bd;
  return bi;
}

// ---------------------------------------------------------------------------------
// lanepoint paths (lanepoints.py:646-692) and equally spaced waypoints
// (sumo_road_network.py:1312-1437)
// ---------------------------------------------------------------------------------
// Route filter: the road ids of _resolve_in_junction (at most the junction road and the
// road it leads to), or none.
struct RouteFilter {
  int n;       // 0 = no filter
  int road[2];
  __device__ __forceinline__ bool has(int r) const { return (n > 0 && road[0] == r) || (n > 1 && road[1] == r); }
  __device__ __forceinline__ int last() const { return road[n - 1]; }
};

// lanepoints.py:666-683: may the walk continue onto lanepoint `nx`?
__device__ __forceinline__ bool next_allowed(const MapDev& m, const RouteFilter& f, int nx) {
  if (f.n == 0) return true;
  int lane = m.lp_lane[nx];
  int road = m.lane_road[lane];
  if (!f.has(road)) return false;
  if (road != f.last()) {
    bool any = false;
    for (int k = m.lane_out_off[lane]; k < m.lane_out_off[lane + 1]; ++k)
      any = any || f.has(m.lane_road[m.lane_out_idx[k]]);
    if (!any) return false;
  }
  return true;
}

// Branch bookkeeping for depth-first enumeration in the reference's order: paths are ordered
// lexicographically by the choice taken at each branching lanepoint.  4 bits per level.
struct BranchState {
  unsigned long long choice;  // chosen child at level b
  unsigned long long count;   // number of allowed children at level b
  int nb;                     // levels currently fixed
  __device__ __forceinline__ void reset() {
    choice = 0ull;
    count = 0ull;
    nb = 0;
  }
  __device__ __forceinline__ int get_choice(int b) const { return (int)((choice >> (4 * b)) & 15ull); }
  __device__ __forceinline__ int get_count(int b) const { return (int)((count >> (4 * b)) & 15ull); }
  __device__ __forceinline__ void set(int b, int ch, int cnt) {
    unsigned long long mask = ~(15ull << (4 * b));
    choice = (choice & mask) | ((unsigned long long)ch << (4 * b));
    count = (count & mask) | ((unsigned long long)cnt << (4 * b));
  }
  // advance to the next leaf; false when the enumeration is exhausted
  __device__ __forceinline__ bool advance() {
    while (nb > 0 && get_choice(nb - 1) + 1 >= get_count(nb - 1)) --nb;
    if (nb == 0) return false;
    set(nb - 1, get_choice(nb - 1) + 1, get_count(nb - 1));
    return true;
  }
};

// One hop of the walk: returns the next lanepoint or -1 when the path cannot grow.
__device__ __forceinline__ int walk_next(const MapDev& m, const RouteFilter& f, BranchState& bs, int& level, int lp) {
  int a = m.lp_next_off[lp], b = m.lp_next_off[lp + 1];
  int n = b - a;
  if (n == 0) return -1;
  if (n == 1) {
    int nx = m.lp_next_idx[a];
    return next_allowed(m, f, nx) ? nx : -1;
  }
  int allowed = 0;
  for (int k = a; k < b; ++k) allowed += next_allowed(m, f, m.lp_next_idx[k]) ? 1 : 0;
  if (allowed == 0) return -1;
  int want = 0;
  if (allowed > 1) {
    if (level < bs.nb) {
      want = bs.get_choice(level);
    } else if (level < 16) {
      bs.set(level, 0, min(allowed, 15));
      bs.nb = level + 1;
    }
    ++level;
  }
  int seen = 0;
  for (int k = a; k < b; ++k) {
    int nx = m.lp_next_idx[k];
    if (!next_allowed(m, f, nx)) continue;
    if (seen == want) return nx;
    ++seen;
  }
  return -1;
}

// Running heading unwrap (math.py:537-550), one element at a time.
struct Unwrap {
  double prev;  // previous raw heading
  double corr;  // cumulative correction
  __device__ __forceinline__ void start(double h0) {
    prev = h0;
    corr = 0.0;
  }
  __device__ __forceinline__ double push(double h) {
    double dd = h - prev;
    double ddmod = py_mod(dd + SMX_PI, SMX_TWO_PI) - SMX_PI;
    if (ddmod == -SMX_PI && dd > 0.0) ddmod = SMX_PI;
    double ph = ddmod - dd;
    if (fabs(dd) < SMX_PI) ph = 0.0;
    corr += ph;
    prev = h;
    return h + corr;
  }
};

struct WaypointOut {
  double x, y, heading, width, speed;
  int lane;
};

// Equally spaced waypoints of ONE lanepoint path (sumo_road_network.py:1312-1437), streamed.
//   start     first lanepoint of the path
//   lookahead number of hops requested
//   bs        branch choices selecting this path (updated with newly met branchings)
//   (px, py)  the query point (vehicle position)
//   max_emit  emit(i, wp) is called for i < min(max_emit, #waypoints)
// Returns the number of waypoints of the path (= number of lanepoints on it).
//
// The reference keeps, as interpolation knots, the first lanepoint (moved to the projection of
// the query point on its heading line), every non-inferred lanepoint strictly inside the path,
// and the last lanepoint.  Pass 1 walks the path for its knot arclength D; pass 2 walks it again
// and emits the waypoints t_i = i * D / (n - 1) by np.interp's rule (knot j = last knot with
// cum[j] <= t; exact knot value when t == cum[j]) while lane_id / lane_index follow the
// "last knot strictly passed" rule of :1404-1417.
template <class Emit>
__device__ inline int equally_spaced_path(const MapDev& m, const RouteFilter& f, BranchState& bs, int start,
                                          int lookahead, double px, double py, int max_emit, Emit&& emit) {
  // ---- knot 0
  const double l0x = m.lp_x[start], l0y = m.lp_y[start];
  const double hx = m.lp_dirx[start], hy = m.lp_diry[start];
  const double proj = (px - l0x) * hx + (py - l0y) * hy;
  const double k0x = l0x + proj * hx, k0y = l0y + proj * hy;

  // ---- pass 1: path length in lanepoints and knot arclength
  int n = 1;
  double D = 0.0;
  {
    int level = 0;
    int lp = start;
    double lastx = k0x, lasty = k0y;
    int cur = start;
    bool cur_is_knot = true;
    for (int hop = 0; hop < lookahead; ++hop) {
      int nx = walk_next(m, f, bs, level, lp);
      if (nx < 0) break;
      lp = nx;
      ++n;
      cur = nx;
      cur_is_knot = !m.lp_inferred[nx];
      if (cur_is_knot) {
        double qx = m.lp_x[nx], qy = m.lp_y[nx];
        double ex = qx - lastx, ey = qy - lasty;
        D += sqrt(ex * ex + ey * ey);
        lastx = qx;
        lasty = qy;
      }
    }
    if (!cur_is_knot) {
      double ex = m.lp_x[cur] - lastx, ey = m.lp_y[cur] - lasty;
      D += sqrt(ex * ex + ey * ey);
    }
  }
  const int lane0 = m.lp_lane[start];
  if (n == 1) {
    // :1379-1390 (a one-point path): the lanepoint itself, not the projection
    if (max_emit > 0) {
      WaypointOut w;
      w.x = l0x;
      w.y = l0y;
      w.heading = m.lp_heading[start];
      w.width = m.lane_width[lane0];
      w.speed = m.lane_speed[lane0];
      w.lane = lane0;
      emit(0, w);
    }
    return 1;
  }

  // ---- pass 2: emit
  const int n_emit = min(n, max_emit);
  const double step = D / (double)(n - 1);  // np.linspace(0, D, n)
  int i = 0;                                // next waypoint to emit
  double t = 0.0;
  // current knot j
  double jx = k0x, jy = k0y, jh = m.lp_heading[start], jcum = 0.0;
  int jlane = lane0;
  int strict_lane = lane0;  // lane of the last knot with cum strictly below jcum (knot 0 if none)
  Unwrap uw;
  uw.start(jh);
  int level = 0;
  int lp = start;
  for (int hop = 1; hop < n && i < n_emit; ++hop) {
    int nx = walk_next(m, f, bs, level, lp);
    lp = nx;
    bool knot = (!m.lp_inferred[nx]) || (hop == n - 1);
    if (!knot) continue;
    // next knot j+1
    double qx = m.lp_x[nx], qy = m.lp_y[nx];
    double ex = qx - jx, ey = qy - jy;
    double qcum = jcum + sqrt(ex * ex + ey * ey);
    double qh = uw.push(m.lp_heading[nx]);
    int qlane = m.lp_lane[nx];
    // waypoints with jcum <= t < qcum interpolate on [j, j+1]
    while (i < n_emit && t < qcum) {
      WaypointOut w;
      int dl;
      if (t == jcum) {
        w.x = jx;
        w.y = jy;
        w.heading = jh;
        w.width = m.lane_width[jlane];
        w.speed = m.lane_speed[jlane];
        dl = strict_lane;
      } else {
        double den = qcum - jcum, dt_ = t - jcum;
        w.x = ((qx - jx) / den) * dt_ + jx;
        w.y = ((qy - jy) / den) * dt_ + jy;
        w.heading = ((qh - jh) / den) * dt_ + jh;
        double wj = m.lane_width[jlane], sj = m.lane_speed[jlane];
        w.width = ((m.lane_width[qlane] - wj) / den) * dt_ + wj;
        w.speed = ((m.lane_speed[qlane] - sj) / den) * dt_ + sj;
        dl = jlane;
      }
      w.heading = wrap_heading(w.heading);
      w.lane = dl;
      emit(i, w);
      ++i;
      t = (i == n - 1) ? D : (double)i * step;
    }
    if (qcum > jcum) strict_lane = jlane;
    jx = qx;
    jy = qy;
    jh = qh;
    jcum = qcum;
    jlane = qlane;
  }
  // waypoints at (or beyond) the last knot
  while (i < n_emit) {
    WaypointOut w;
    w.x = jx;
    w.y = jy;
    w.heading = wrap_heading(jh);
    w.width = m.lane_width[jlane];
    w.speed = m.lane_speed[jlane];
    w.lane = (t > jcum) ? jlane : strict_lane;
    emit(i, w);
    ++i;
    t = (i == n - 1) ? D : (double)i * step;
  }
  return n;
}

// Resolution of the road whose lanes seed the paths (sumo_road_network.py:815-882) for an agent
// with an empty route (EndlessGoal): junction lanes pin the route to [junction road, next road].
struct PathSeed {
  int road;        // road whose lanes are enumerated
  RouteFilter f;   // route filter for the walks
};

__device__ inline PathSeed resolve_path_seed(const MapDev& m, double px, double py, double heading,
                                             double within_radius, bool has_route_object) {
  PathSeed s;
  s.f.n = 0;
  s.road = -1;
  if (has_route_object) {
    // _resolve_in_junction (:842-860)
    int lp = closest_lanepoint(m, px, py, heading, -1.0);
    if (lp >= 0) {
      int lane = m.lp_lane[lp];
      int road = m.lane_road[lane];
      if (m.road_is_junction[road]) {
        s.f.n = 1;
        s.f.road[0] = road;
        int nr = m.road_out_road[road];
        if (nr >= 0) {
          s.f.n = 2;
          s.f.road[1] = nr;
        }
        // _waypoint_paths_along_route (:862-882): nearest lanepoint over the route roads
        double bd = SMX_INF;
        int best = -1;
        for (int k = 0; k < s.f.n; ++k) {
          double d2;
          int c = closest_lanepoint_filtered(m, px, py, s.f.road[k], true, &d2);
          // reference compares np.linalg.norm distances; sqrt is monotone and the first
          // minimum wins
          double d = sqrt(d2);
          if (c >= 0 && d < bd) {
            bd = d;
            best = c;
          }
        }
        s.road = best >= 0 ? m.lane_road[m.lp_lane[best]] : -1;
        return s;
      }
    }
  }
  int lp = closest_lanepoint(m, px, py, heading, within_radius);
  s.road = lp >= 0 ? m.lane_road[m.lp_lane[lp]] : -1;
  return s;
}
