// smx_roadmap.h — device-side road-map queries: nearest lane, closest lanepoints,
// waypoint paths.  Batched restatement of
//   SumoRoadNetwork.nearest_lanes / road_with_point      sumo_road_network.py:675-709
//   LanePoints.closest_lanepoints / closest_linked_*      lanepoints.py:526-644
//   LanePoints.paths_starting_at_lanepoint                lanepoints.py:646-692
//   SumoRoadNetwork.waypoint_paths / _equally_spaced_path sumo_road_network.py:815-882, 1312-1437
// over packed tables (smx_map_tables) with uniform grids instead of KD-/R-trees.
//
// Memory behaviour is the design driver: one thread's tick used to be ~5000 *dependent*
// cache misses.  Here a lanepoint is one 64-byte record, grid cells hold their members by
// value, a path walk visits only the shape points it passes (knot skip-links), the five
// nearest-lane queries of a tick share one cell sweep, and the lanepoint searches of a tick
// are two sweeps whose results are reused by the next tick's controller.
#pragma once
#include "smx_device.h"

#define SMX_INF 1.0e300

__device__ __forceinline__ smx_lp_rec load_lp(const MapDev& m, int i, int site = 1) {
  return m.lp_rec[SMX_BCHK(site, i, m.n_lanepoints)];
}

// ---------------------------------------------------------------------------------
// nearest lane(s): centre + 4 corners of a vehicle in one sweep
// ---------------------------------------------------------------------------------
struct RoadFacts {
  int lane;         // nearest lane to the centre within `radius` (ties: lowest table index), -1 none
  double dist;      // its centre-line distance
  bool on_road;     // road_with_point(centre) is not None
  int corner_mask;  // bit q: road_with_point(corner q) is not None
};

// All lanes whose centre polyline is closer than `radius` to the centre; the nearest one is what
// RoadMap.nearest_lane returns (road_map.py:91-96).  The segment grid only prunes: a lane is
// within reach iff one of its segments is, and that segment's bounding box then meets the query
// square, so it is listed in a visited cell.  Corners are at most half a vehicle diagonal from
// the centre and only ask for segments closer than half a lane width, so the same sweep
// answers them (n_corners = 0 skips them).
__device__ inline RoadFacts road_facts_scan(const MapDev& m, double px, double py, double radius, int n_corners,
                                            const double* cx, const double* cy) {
  RoadFacts out;
  out.lane = -1;
  out.dist = SMX_INF;
  out.on_road = false;
  out.corner_mask = 0;
  const double road_radius = fmax(5.0, 2.0 * m.default_lane_width);  // sumo_road_network.py:705
  int cx0 = (int)floor((px - radius - m.sg_x0) / m.sg_cell);
  int cx1 = (int)floor((px + radius - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((py - radius - m.sg_y0) / m.sg_cell);
  int cy1 = (int)floor((py + radius - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  for (int gy = cy0; gy <= cy1; ++gy) {
    const int row = gy * m.sg_nx;
    // cells of one grid row are contiguous in the member array
    const int a = m.sg_off[SMX_BCHK(2, row + cx0, m.sg_nx * m.sg_ny + 1)], b = m.sg_off[SMX_BCHK(3, row + cx1 + 1, m.sg_nx * m.sg_ny + 1)];
    for (int k = a; k < b; ++k) {
      const smx_seg_rec s = m.sg_rec[SMX_BCHK(4, k, m.sg_off[m.sg_nx * m.sg_ny])];
      // cheap exact-safe prefilter: the distance to the segment's bounding box bounds the
      // distance to the segment from below.  A segment can only matter to the centre if it
      // can beat (or tie) the current nearest lane or pass the road_with_point threshold, and
      // to a corner only through that threshold.
      {
        const double bx0 = fmin(s.x1, s.x2), bx1 = fmax(s.x1, s.x2), by0 = fmin(s.y1, s.y2), by1 = fmax(s.y1, s.y2);
        const double gx = fmax(fmax(bx0 - px, px - bx1), 0.0), gy2 = fmax(fmax(by0 - py, py - by1), 0.0);
        const double lb2 = gx * gx + gy2 * gy2;
        const double keep_c = fmin(fmax(out.dist, s.thr), radius) + 1e-6;
        // corners sit within half a vehicle diagonal (< 2.0 m for every supported chassis) of the centre
        const double keep_q = s.thr + 2.0 + 1e-6;
        const double keep = n_corners > 0 ? fmax(keep_c, keep_q) : keep_c;
        if (lb2 > keep * keep) continue;
      }
      // distance_point_to_line(point, p1, p2) (math.py:393-411), shared segment length
      const double ex = s.x1 - s.x2, ey = s.y1 - s.y2;
      const double d = sqrt(ex * ex + ey * ey);
      const double dd = d * d;
      const double sx = s.x2 - s.x1, sy = s.y2 - s.y1;
#pragma unroll
      for (int q = -1; q < 4; ++q) {
        if (q >= n_corners) break;
        const double qx = q < 0 ? px : cx[q], qy = q < 0 ? py : cy[q];
        const double u = ((qx - s.x1) * sx) + ((qy - s.y1) * sy);
        double offset;
        if (d == 0.0 || u < 0.0 || u > dd) {
          offset = (u < 0.0) ? 0.0 : d;
        } else {
          offset = u / d;
        }
        double dist;
        if (offset == 0.0) {
          const double fx = qx - s.x1, fy = qy - s.y1;
          dist = sqrt(fx * fx + fy * fy);
        } else {
          const double uu = offset / d;
          const double ix = s.x1 + uu * sx, iy = s.y1 + uu * sy;
          const double fx = qx - ix, fy = qy - iy;
          dist = sqrt(fx * fx + fy * fy);
        }
        if (q < 0) {
          if (dist < radius) {
            if (dist < out.dist || (dist == out.dist && s.lane < out.lane)) {
              out.dist = dist;
              out.lane = s.lane;
            }
            if (dist < road_radius && dist < s.thr) out.on_road = true;
          }
        } else {
          if (dist < road_radius && dist < s.thr) out.corner_mask |= (1 << q);
        }
      }
    }
  }
  return out;
}

// ---------------------------------------------------------------------------------
// lanepoint nearest-neighbour queries on the uniform grid
// ---------------------------------------------------------------------------------
// Visit every lanepoint of Chebyshev ring `r` around cell (cx, cy).
template <class F>
__device__ __forceinline__ void lp_ring_visit(const MapDev& m, int cx, int cy, int r, F&& f) {
  const int y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
  for (int y = max(y0, 0); y <= min(y1, m.lpg_ny - 1); ++y) {
    const int row = y * m.lpg_nx;
    if (y == y0 || y == y1) {
      // full edge row: its cells are contiguous
      const int xa = max(x0, 0), xb = min(x1, m.lpg_nx - 1);
      if (xa > xb) continue;
      const int a = m.lpg_off[SMX_BCHK(5, row + xa, m.lpg_nx * m.lpg_ny + 1)], b = m.lpg_off[SMX_BCHK(6, row + xb + 1, m.lpg_nx * m.lpg_ny + 1)];
      for (int k = a; k < b; ++k) f(m.lpg_pts[SMX_BCHK(7, k, m.n_lanepoints)]);
    } else {
      if (x0 >= 0 && x0 < m.lpg_nx) {
        const int a = m.lpg_off[SMX_BCHK(8, row + x0, m.lpg_nx * m.lpg_ny + 1)], b = m.lpg_off[row + x0 + 1];
        for (int k = a; k < b; ++k) f(m.lpg_pts[SMX_BCHK(9, k, m.n_lanepoints)]);
      }
      if (x1 != x0 && x1 >= 0 && x1 < m.lpg_nx) {
        const int a = m.lpg_off[SMX_BCHK(10, row + x1, m.lpg_nx * m.lpg_ny + 1)], b = m.lpg_off[row + x1 + 1];
        for (int k = a; k < b; ++k) f(m.lpg_pts[SMX_BCHK(11, k, m.n_lanepoints)]);
      }
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

// The 10 nearest lanepoints (maximum_count = 10, lanepoints.py:592-623), sorted by
// (distance^2, table index).
struct Top10 {
  double d2[10];
  int idx[10];
};

__device__ inline void nearest10(const MapDev& m, double px, double py, Top10& t) {
  const int K = 10;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    t.d2[i] = SMX_INF;
    t.idx[i] = -1;
  }
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int keff = min(K, m.n_lanepoints);
  const int rmax = lp_max_ring(m, cx, cy);
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit(m, cx, cy, r, [&](const smx_pt_rec& p) {
      double dx = p.x - px, dy = p.y - py;
      double d2 = dx * dx + dy * dy;
      if (d2 < t.d2[K - 1] || (d2 == t.d2[K - 1] && p.idx < t.idx[K - 1])) {
        double cd = d2;
        int ci = p.idx;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          bool lt = (cd < t.d2[k]) || (cd == t.d2[k] && ci < t.idx[k]);
          double td = lt ? t.d2[k] : cd;
          int ti = lt ? t.idx[k] : ci;
          t.d2[k] = lt ? cd : t.d2[k];
          t.idx[k] = lt ? ci : t.idx[k];
          cd = td;
          ci = ti;
        }
      }
    });
    // keff is 10 except on maps with fewer lanepoints
    bool full = true;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k == keff - 1) full = t.idx[k] >= 0 && ring_covers(m, r, t.d2[k]);
    if (full) break;
  }
}

// closest_lanepoints([pose], within_radius)[0] from the 10 nearest (lanepoints.py:526-590):
// those beyond within_radius dropped (the nearest always kept), winner = min of
// dist^2 + |heading difference| (first minimum in distance order).  within_radius < 0 = None.
// The heading term of every candidate, computed once for all the queries of a tick: the ten
// lanepoint headings are loaded back to back (one scattered load per candidate behind the
// candidate's own test would be ten load latencies in a row, per query).
struct Top10Scores {
  double rel[10];  // |heading difference| of candidate k
};

__device__ inline Top10Scores top10_heading_terms(const MapDev& m, const Top10& t, double heading) {
  Top10Scores sc;
  double h[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) h[k] = m.lp_rec[SMX_BCHK(12, t.idx[k] < 0 ? 0 : t.idx[k], m.n_lanepoints)].heading;
#pragma unroll
  for (int k = 0; k < 10; ++k) sc.rel[k] = fabs(heading_relative_to(heading, h[k]));
  return sc;
}

__device__ inline int pick_closest(const Top10& t, const Top10Scores& sc, double within_radius) {
  const double r2 = within_radius * within_radius;
  int best = -1;
  double best_score = SMX_INF;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    if (t.idx[k] < 0) continue;
    if (within_radius >= 0.0 && k > 0 && !(t.d2[k] <= r2)) continue;
    double score = t.d2[k] + sc.rel[k];
    if (score < best_score) {
      best_score = score;
      best = t.idx[k];
    }
  }
  return best;
}

__device__ inline int pick_closest(const MapDev& m, const Top10& t, double heading, double within_radius) {
  return pick_closest(t, top10_heading_terms(m, t, heading), within_radius);
}

// closest_linked_lanepoint_on_lane_to_point (lanepoints.py:629-636) for up to 4 lanes at once
// (by_road == false), or closest_linked_lanepoint_on_road (:638-644) for up to 4 roads.
__device__ inline void closest_filtered4(const MapDev& m, double px, double py, const int* keys, int nkeys,
                                         bool by_road, int* out_idx, double* out_d2) {
  double bd[4] = {SMX_INF, SMX_INF, SMX_INF, SMX_INF};
  int bi[4] = {-1, -1, -1, -1};
  const int k0 = keys[0], k1 = nkeys > 1 ? keys[1] : -9, k2 = nkeys > 2 ? keys[2] : -9, k3 = nkeys > 3 ? keys[3] : -9;
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int rmax = lp_max_ring(m, cx, cy);
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit(m, cx, cy, r, [&](const smx_pt_rec& p) {
      const int key = by_road ? m.lane_road[SMX_BCHK(13, p.lane, m.n_lanes)] : p.lane;
      double dx = p.x - px, dy = p.y - py;
      double d2 = dx * dx + dy * dy;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kq = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
        if (key == kq && (d2 < bd[q] || (d2 == bd[q] && p.idx < bi[q]))) {
          bd[q] = d2;
          bi[q] = p.idx;
        }
      }
    });
    bool all = true;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nkeys) all = all && bi[q] >= 0 && ring_covers(m, r, bd[q]);
    if (all) break;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    out_idx[q] = bi[q];
    if (out_d2) out_d2[q] = bd[q];
  }
}

// ---------------------------------------------------------------------------------
// lanepoint paths (lanepoints.py:646-692)
// ---------------------------------------------------------------------------------
// Missions of the agent slots (smx_set_missions), device side.  Null pointers: every mission is endless.
struct MissionsDev {
  const int32_t* route_last;  // [slots]: last road of the route (lanepoints.py:674), -1 = endless mission (empty route)
  const double* goal;         // [slots][3]: PositionalGoal x, y, radius
};

// Route filter: the road ids of _resolve_in_junction (at most the junction road and the
// road it leads to), the roads of a mission's fixed route, or none.
#define SMX_ROUTE_FIXED 3
struct RouteFilter {
  int n;  // 0 = no filter; 1, 2 = the list road[]; SMX_ROUTE_FIXED = a fixed route: road[0] = the agent slot (row of
          // MapDev::route_pos / route_lane_ok), road[1] = the route's last road
  int road[2];
  __device__ __forceinline__ bool has(const MapDev& m, int r) const {
    if (n == SMX_ROUTE_FIXED) return m.route_pos[road[0] * m.n_roads + r] >= 0;
    return (n > 0 && road[0] == r) || (n > 1 && road[1] == r);
  }
  __device__ __forceinline__ int last() const { return (n == SMX_ROUTE_FIXED || n == 2) ? road[1] : road[0]; }  // (n = 1, 2: road[n - 1])
  __device__ __forceinline__ void none() {
    n = 0;
    road[0] = road[1] = -1;
  }
  // the filter of agent slot `slot`'s mission; false (and no filter) if that mission is endless
  __device__ __forceinline__ bool fixed_route(const MissionsDev& ms, int slot, int n_roads) {
    none();
    if (ms.route_last == nullptr) return false;
    const int last_road = ms.route_last[slot];
    if (last_road < 0) return false;
    n = SMX_ROUTE_FIXED;
    road[0] = slot;
    road[1] = last_road;
    (void)n_roads;
    return true;
  }
};

// lanepoints.py:666-683: may a path continue onto a lanepoint of `lane`?  (A pure function of
// the lane, so a walk is checked where the lane changes, not per hop.)
// For a fixed route the answer per (slot, lane) is tabulated by smx_set_missions (the same rule, evaluated once
// on the host: host_lane_allowed in smx_kernels.hip).
__device__ __forceinline__ bool lane_allowed(const MapDev& m, const RouteFilter& f, int lane) {
  if (f.n == 0) return true;
  if (f.n == SMX_ROUTE_FIXED) return m.route_lane_ok[f.road[0] * m.n_lanes + lane] != 0;
  const int road = m.lane_road[SMX_BCHK(14, lane, m.n_lanes)];
  if (!f.has(m, road)) return false;
  if (road != f.last()) {
    bool any = false;
    for (int k = m.lane_out_off[lane]; k < m.lane_out_off[lane + 1]; ++k)
      any = any || f.has(m, m.lane_road[m.lane_out_idx[k]]);
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

// j-th successor (j >= 0) down a chain of interpolated lanepoints starting at `first`.
__device__ __forceinline__ int chain_at(const MapDev& m, int first, int j, bool consecutive) {
  if (consecutive) return first + j;
  int cur = first;
  for (int k = 0; k < j; ++k) cur = m.lp_rec[cur].next0;
  return cur;
}

// Knot walker: enumerates, for one lanepoint path, the lanepoints that become interpolation
// knots — every non-inferred lanepoint after the start, and the last lanepoint — without
// touching the interpolated points in between.
struct KnotWalk {
  int remaining;  // hops still allowed
  int n;          // lanepoints on the path so far (start included)
  int level;      // branching level (BranchState index)
  smx_lp_rec cur; // record of the lanepoint the walk stands on
  bool start;     // still on the start point (its own lane has not been checked by the filter)
  int cur_idx;    // index of `cur`

  __device__ __forceinline__ void begin(const MapDev& m, int start_lp, int lookahead) {
    remaining = lookahead;
    n = 1;
    level = 0;
    cur = load_lp(m, start_lp, 40);
    cur_idx = start_lp;
    start = true;
  }

  __device__ __forceinline__ int first_of_run(const MapDev& m, int first) const {
#ifdef SMX_WALK_CARRIED_NEXT0  // the round-1 form under investigation (tests/native/host_walk.cpp)
    return first >= 0 ? first : cur.next0;
#else
    return first >= 0 ? first : m.lp_rec[cur_idx].next0;
#endif
  }

  // Advance to the next knot.  Returns its lanepoint index (its record is left in `rec`),
  // or -1 when the path has ended.  `last` tells whether it is the path's last lanepoint.
  __device__ inline int next(const MapDev& m, const RouteFilter& f, BranchState& bs, smx_lp_rec& rec, bool& last) {
    if (remaining <= 0 || cur.n_next == 0) return -1;
    int first, knot, hops, first_lane;
    if (cur.n_next == 1) {
      // `first` (= next0 of the record the walk stands on) is only needed when the path ends
      // inside the interpolated run; it is re-read there (first_of_run) instead of being carried
      // in the by-value record across the walk.
      first = -1;
      knot = cur.knot_next;
      hops = cur.knot_hops;
      first_lane = -1;  // same lane as `cur` when interpolated, else the knot's own lane
    } else {
      // one pass over the successors: count the allowed ones, remember the first allowed and — where this level's
      // choice is already fixed (an earlier leaf of the enumeration) — the chosen one
      const int a = cur.next_off, b = a + cur.n_next;
      const bool fixed = level < bs.nb;
      const int want = fixed ? bs.get_choice(level) : 0;  // meaningful only if several are allowed
      int allowed = 0;
      smx_succ_rec s0, sw;
      s0.idx = sw.idx = -1;
      s0.knot = sw.knot = -1;
      s0.hops = sw.hops = 0;
      s0.lane = sw.lane = -1;
      for (int k = a; k < b; ++k) {
        const smx_succ_rec sr = m.succ_rec[SMX_BCHK(15, k, m.n_succ)];
        if (!lane_allowed(m, f, sr.lane)) continue;
        if (allowed == 0) s0 = sr;
        if (allowed == want) sw = sr;
        ++allowed;
      }
      if (allowed == 0) return -1;
      smx_succ_rec take = s0;
      if (allowed > 1) {
        if (fixed) {
          take = sw;
        } else if (level < 16) {
          bs.set(level, 0, min(allowed, 15));
          bs.nb = level + 1;
        }
        ++level;
      }
      first = take.idx;
      knot = take.knot;
      hops = take.hops;
      first_lane = take.lane;
      if (first < 0) return -1;
    }
    const bool consecutive = (cur.flags & 1) != 0;
    if (hops > 1 && cur.n_next == 1 && start && !lane_allowed(m, f, cur.lane)) return -1;
    start = false;
    if (hops <= remaining) {
      rec = load_lp(m, knot, 41);
      // arriving on the knot is a hop onto its lane
      const bool ok = (hops > 1 || cur.n_next == 1 || first_lane == rec.lane) ? lane_allowed(m, f, rec.lane) : true;
      if (ok) {
        remaining -= hops;
        n += hops;
        cur = rec;
        cur_idx = knot;
        last = (remaining == 0) || (rec.n_next == 0);
        if (!last) {
          // the path also ends here if no successor can be taken; found out by the next call
        }
        return knot;
      }
      if (hops == 1) return -1;
      // the knot's lane is closed: the path stops on the interpolated point before it
      const int fin = chain_at(m, first_of_run(m, first), hops - 2, consecutive);
      rec = load_lp(m, fin, 42);
      n += hops - 1;
      remaining = 0;
      cur = rec;
      last = true;
      return fin;
    }
    // the path stops inside the interpolated run
    const int fin = chain_at(m, first_of_run(m, first), remaining - 1, consecutive);
#ifdef SMX_DEBUG_BOUNDS
    if (fin < 0 || fin >= m.n_lanepoints) {
      smx_dbg_aux[0] = first;
      smx_dbg_aux[1] = remaining;
      smx_dbg_aux[2] = hops;
      smx_dbg_aux[3] = cur.n_next;
      smx_dbg_aux[4] = cur.lane;
      smx_dbg_aux[5] = cur.next0;
      smx_dbg_aux[6] = cur_idx;
      smx_dbg_aux[7] = ((const volatile smx_lp_rec*)m.lp_rec)[cur_idx].next0;
    }
#endif
    rec = load_lp(m, fin, 43);
    n += remaining;
    remaining = 0;
    cur = rec;
    last = true;
    return fin;
  }
};

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

#define SMX_MAX_KNOTS 36  // lookahead <= 34: start + at most one knot per hop + the last point

// Equally spaced waypoints of ONE lanepoint path (sumo_road_network.py:1312-1437).
//   start     first lanepoint of the path
//   lookahead number of hops requested
//   bs        branch choices selecting this path (updated with newly met branchings)
//   (px, py)  the query point (vehicle position)
//   knots     per-thread scratch for the path's knot indices ([SMX_MAX_KNOTS], stride `kstride`)
//   max_emit  emit(i, wp) is called for i < min(max_emit, #waypoints)
// Returns the number of waypoints of the path (= number of lanepoints on it).
//
// The reference keeps, as interpolation knots, the first lanepoint (moved to the projection of
// the query point on its heading line), every non-inferred lanepoint strictly inside the path,
// and the last lanepoint.  Pass 1 walks the knots (dependent loads, one record per knot) for the
// path length n and the knot arclength D, remembering the knot indices; pass 2 re-reads those
// records (independent loads) and emits the waypoints t_i = i * D / (n - 1) by np.interp's rule
// (knot j = last knot with cum[j] <= t; exact knot value when t == cum[j]) while lane_id /
// lane_index follow the "last knot strictly passed" rule of :1404-1417.
// Pass 2: the waypoints of a path whose knots are known — `fetch(k)` gives the lanepoint index of knot
// k = 0..nk-1 in path order (r0 = the start lanepoint's record, n = lanepoints on the path, D = the knot
// arclength).  emit(i, wp) is called for i < min(n, max_emit).
template <class Fetch, class Emit>
__device__ inline void interpolate_knots(const MapDev& m, const smx_lp_rec& r0, int nk, int n, double D, double px,
                                         double py, int max_emit, Fetch&& fetch, Emit&& emit) {
  const int lane0 = r0.lane;
  if (n == 1) {
    // :1379-1390 (a one-point path): the lanepoint itself, not the projection
    if (max_emit > 0) {
      WaypointOut o;
      o.x = r0.x;
      o.y = r0.y;
      o.heading = r0.heading;
      o.width = m.lane_width[SMX_BCHK(17, lane0, m.n_lanes)];
      o.speed = m.lane_speed[lane0];
      o.lane = lane0;
      emit(0, o);
    }
    return;
  }
  const int n_emit = min(n, max_emit);
  if (n_emit <= 0) return;
  // ---- knot 0: projection of the query point on the first lanepoint's heading line
  const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
  const double k0x = r0.x + proj * r0.dirx, k0y = r0.y + proj * r0.diry;
  const double step = D / (double)(n - 1);  // np.linspace(0, D, n)
  int i = 0;                                // next waypoint to emit
  double t = 0.0;
  double jx = k0x, jy = k0y, jh = r0.heading, jcum = 0.0;
  int jlane = lane0;
  int strict_lane = lane0;  // lane of the last knot with cum strictly below jcum (knot 0 if none)
  Unwrap uw;
  uw.start(jh);
  smx_lp_rec q = (nk > 0) ? load_lp(m, fetch(0), 44) : r0;
  // width / speed limit of knot j's lane, carried from knot to knot: most knots share their lane,
  // and a table look-up per knot would put a dependent load in front of every segment
  double wj = m.lane_width[SMX_BCHK(18, jlane, m.n_lanes)], sj = m.lane_speed[jlane];
  for (int k = 0; k < nk && i < n_emit; ++k) {
    const smx_lp_rec cur = q;
    if (k + 1 < nk) q = load_lp(m, fetch(k + 1), 45);  // prefetch the next knot
    const double qx = cur.x, qy = cur.y;
    const double ex = qx - jx, ey = qy - jy;
    const double qcum = jcum + sqrt(ex * ex + ey * ey);
    const double qh = uw.push(cur.heading);
    const int qlane = cur.lane;
    // waypoints with jcum <= t < qcum interpolate on [j, j+1]; np.interp's slope
    // (dy[j+1] - dy[j]) / (dx[j+1] - dx[j]) is the same for every waypoint of the segment
    double wq = wj, sq = sj;
    if (qlane != jlane) {
      wq = m.lane_width[SMX_BCHK(18, qlane, m.n_lanes)];
      sq = m.lane_speed[qlane];
    }
    if (i < n_emit && t < qcum) {
      const double den = qcum - jcum;
      const double sx = (qx - jx) / den, sy = (qy - jy) / den, sh = (qh - jh) / den;
      double sw = 0.0, ss = 0.0;
      if (qlane != jlane) {
        sw = (wq - wj) / den;
        ss = (sq - sj) / den;
      }
      do {
        // np.interp returns the knot's own value when t sits on it; the expression below does too
        // (dt_ = 0, the slopes are finite: den > 0 here), so only the lane rule needs the test
        WaypointOut o;
        const double dt_ = t - jcum;
        o.x = sx * dt_ + jx;
        o.y = sy * dt_ + jy;
        o.heading = sh * dt_ + jh;
        o.width = sw * dt_ + wj;
        o.speed = ss * dt_ + sj;
        const int dl = (t == jcum) ? strict_lane : jlane;
        o.heading = wrap_heading(o.heading);
        o.lane = dl;
        emit(i, o);
        ++i;
        t = (i == n - 1) ? D : (double)i * step;
      } while (i < n_emit && t < qcum);
    }
    if (qcum > jcum) strict_lane = jlane;
    jx = qx;
    jy = qy;
    jh = qh;
    jcum = qcum;
    jlane = qlane;
    wj = wq;
    sj = sq;
  }
  // waypoints at (or beyond) the last knot
  while (i < n_emit) {
    WaypointOut o;
    o.x = jx;
    o.y = jy;
    o.heading = wrap_heading(jh);
    o.width = wj;
    o.speed = sj;
    o.lane = (t > jcum) ? jlane : strict_lane;
    emit(i, o);
    ++i;
    t = (i == n - 1) ? D : (double)i * step;
  }
}

// The same pass 2 for a path whose first KP knots are already in registers: x / y / heading / lane / lane
// width / lane speed limit of knot k in element k of the arrays, that part of the knot loop unrolled so that
// every element is a register.  A wavefront whose lanes each interpolate their own path meets a knot in one
// lane or another at almost every step, and a record (or lane table) load at that point stalls all of them;
// here those loads have been issued, together, before the loop starts.  Knots beyond KP (seldom needed: the
// loop ends with the last waypoint kept) are fetched one by one as interpolate_knots does.  Same
// expressions, same bits.
template <int KP, class Fetch, class Emit>
__device__ __forceinline__ void interpolate_knots_preloaded(const MapDev& m, const smx_lp_rec& r0, double w0, double s0,
                                                            int nk, int n, double D, double px, double py, int max_emit,
                                                            const double (&kx)[KP], const double (&ky)[KP],
                                                            const double (&kh)[KP], const int (&kl)[KP],
                                                            const double (&kw)[KP], const double (&ks)[KP],
                                                            Fetch&& fetch, Emit&& emit) {
  const int lane0 = r0.lane;
  if (n == 1) {
    if (max_emit > 0) {
      WaypointOut o;
      o.x = r0.x;
      o.y = r0.y;
      o.heading = r0.heading;
      o.width = w0;
      o.speed = s0;
      o.lane = lane0;
      emit(0, o);
    }
    return;
  }
  const int n_emit = min(n, max_emit);
  if (n_emit <= 0) return;
  const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
  const double k0x = r0.x + proj * r0.dirx, k0y = r0.y + proj * r0.diry;
  const double step = D / (double)(n - 1);
  int i = 0;
  double t = 0.0;
  double jx = k0x, jy = k0y, jh = r0.heading, jcum = 0.0;
  int jlane = lane0;
  int strict_lane = lane0;
  Unwrap uw;
  uw.start(jh);
  double wj = w0, sj = s0;
  // one knot: the waypoints of the interval it closes, then it becomes knot j
  auto knot = [&](double qx, double qy, double qhead, int qlane, double wq, double sq) {
    const double ex = qx - jx, ey = qy - jy;
    const double qcum = jcum + sqrt(ex * ex + ey * ey);
    const double qh = uw.push(qhead);
    if (i < n_emit && t < qcum) {
      const double den = qcum - jcum;
      const double sx = (qx - jx) / den, sy = (qy - jy) / den, sh = (qh - jh) / den;
      double sw = 0.0, ss = 0.0;
      if (qlane != jlane) {
        sw = (wq - wj) / den;
        ss = (sq - sj) / den;
      }
      do {
        WaypointOut o;
        const double dt_ = t - jcum;
        o.x = sx * dt_ + jx;
        o.y = sy * dt_ + jy;
        o.heading = sh * dt_ + jh;
        o.width = sw * dt_ + wj;
        o.speed = ss * dt_ + sj;
        const int dl = (t == jcum) ? strict_lane : jlane;
        o.heading = wrap_heading(o.heading);
        o.lane = dl;
        emit(i, o);
        ++i;
        t = (i == n - 1) ? D : (double)i * step;
      } while (i < n_emit && t < qcum);
    }
    if (qcum > jcum) strict_lane = jlane;
    jx = qx;
    jy = qy;
    jh = qh;
    jcum = qcum;
    jlane = qlane;
    wj = wq;
    sj = sq;
  };
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (k < nk && i < n_emit) {
      const bool change = kl[k] != jlane;
      knot(kx[k], ky[k], kh[k], kl[k], change ? kw[k] : wj, change ? ks[k] : sj);
    }
  }
  for (int k = KP; k < nk && i < n_emit; ++k) {
    const smx_lp_rec cur = load_lp(m, fetch(k), 45);
    double wq = wj, sq = sj;
    if (cur.lane != jlane) {
      wq = m.lane_width[SMX_BCHK(18, cur.lane, m.n_lanes)];
      sq = m.lane_speed[cur.lane];
    }
    knot(cur.x, cur.y, cur.heading, cur.lane, wq, sq);
  }
  while (i < n_emit) {
    WaypointOut o;
    o.x = jx;
    o.y = jy;
    o.heading = wrap_heading(jh);
    o.width = wj;
    o.speed = sj;
    o.lane = (t > jcum) ? jlane : strict_lane;
    emit(i, o);
    ++i;
    t = (i == n - 1) ? D : (double)i * step;
  }
}

template <int MAXK = SMX_MAX_KNOTS, class Emit>
__device__ inline int equally_spaced_path(const MapDev& m, const RouteFilter& f, BranchState& bs, int start,
                                          int lookahead, double px, double py, int* knots, int kstride,
                                          int max_emit, Emit&& emit) {
  KnotWalk w;
  w.begin(m, start, lookahead);
  const smx_lp_rec r0 = w.cur;
  // ---- knot 0: projection of the query point on the first lanepoint's heading line
  const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
  const double k0x = r0.x + proj * r0.dirx, k0y = r0.y + proj * r0.diry;

  // ---- pass 1
  SMX_TSTAMP(te0);
  int nk = 0;
  double D = 0.0;
  {
    double lastx = k0x, lasty = k0y;
    smx_lp_rec rec;
    bool last = false;
    for (;;) {
      int idx = w.next(m, f, bs, rec, last);
      if (idx < 0) break;
      if (nk < MAXK) knots[nk * kstride] = idx;
      (void)SMX_BCHK(16, nk, MAXK);
      ++nk;
      double ex = rec.x - lastx, ey = rec.y - lasty;
      D += sqrt(ex * ex + ey * ey);
      lastx = rec.x;
      lasty = rec.y;
    }
  }
  const int n = w.n;
  SMX_TSTAMP(te1);
  SMX_TACC(4, te0, te1);
  // ---- pass 2: emit
  interpolate_knots(m, r0, nk, n, D, px, py, max_emit, [&](int k) { return knots[k * kstride]; }, emit);
  SMX_TSTAMP(te2);
  SMX_TACC(5, te1, te2);
  return n;
}

// ---------------------------------------------------------------------------------
// Knot lists: the chain walk of a path, split from its interpolation.
// ---------------------------------------------------------------------------------
// Knot lists: what the chain walk (one lane per path, dependent loads) leaves in device memory for the
// kernels that need the path's knots afterwards — they re-read the records with independent loads.
// Layout [entry][path] (path = vehicle * 4 + seed lane), so a wavefront's store of entry k is contiguous.
#define SMX_WPK_CAP 20  // entries kept per path: entry 0 = the start lanepoint, then the knots in path order
struct KnotLists {
  int32_t* idx;   // [SMX_WPK_CAP + 1][paths]
  double* D;      // [paths] arclength over all knots, from the projection of the query point
  int16_t* n;     // [paths] lanepoints on the path; 0 = the seed lane starts no path
  int16_t* nk;    // [paths] knots after the start
  uint8_t* cnt;   // [paths] paths that start on this seed lane (1 unless the walk meets a branching)
  // The controller's lookahead-16 path (lane_following_controller.py:96-98) runs along the same lanepoints: its
  // knots are this list's knots less than 16 hops down plus the lanepoint 16 hops down.
  uint8_t* nk16;  // [paths] knots of that path (the last one included)
  int32_t* end16; // [paths] its last knot when that is not a knot of this list (an interpolated lanepoint), else -1
  int32_t* key;   // [3][paths] what the list was walked for: start lanepoint, route filter roads (-1 none)
};

struct PathWalk {
  int n;      // lanepoints on the path
  int nk;     // knots after entry 0
  double D;   // arclength over all knots
};

// Pass 1 of equally_spaced_path: sink(k, lanepoint index, hops from the start) is called for every knot
// k = 1..nk in path order.
template <class Sink>
__device__ inline PathWalk walk_knots(const MapDev& m, const RouteFilter& f, BranchState& bs, int start, int lookahead,
                                      double px, double py, Sink&& sink) {
  KnotWalk w;
  w.begin(m, start, lookahead);
  const smx_lp_rec r0 = w.cur;
  const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
  PathWalk out;
  out.nk = 0;
  out.D = 0.0;
  double lastx = r0.x + proj * r0.dirx, lasty = r0.y + proj * r0.diry;
  smx_lp_rec rec;
  bool last = false;
  for (;;) {
    const int idx = w.next(m, f, bs, rec, last);
    if (idx < 0) break;
    ++out.nk;
    const double ex = rec.x - lastx, ey = rec.y - lasty;
    out.D += sqrt(ex * ex + ey * ey);
    lastx = rec.x;
    lasty = rec.y;
    sink(out.nk, idx, w.n - 1);
  }
  out.n = w.n;
  return out;
}

// ---------------------------------------------------------------------------------
// path seeds: the road whose lanes start the paths, the route filter, and the start lanepoint
// on each of its lanes (sumo_road_network.py:815-882, Lane._waypoint_paths_at :429-446)
// ---------------------------------------------------------------------------------
#define SMX_SEED_LANES 4
struct PathSeeds {
  int road;                    // -1: none found
  RouteFilter f;
  int n_lanes;                 // lanes of `road`
  int start[SMX_SEED_LANES];   // start lanepoint per lane (first SMX_SEED_LANES lanes)
};

// _waypoint_paths_along_route's seed (sumo_road_network.py:862-876): closest_linked_lanepoint_on_road for every
// road of the route, then the first minimum of np.linalg.norm(position - point) in route order.  One sweep:
// candidates ordered by (sqrt(d2), position of the road in the route, d2, lanepoint index) — within a road
// that is the KD-tree's minimum of d2 (index ties, DESIGN.md deviation 1), across roads the first minimum.
struct RouteBest {
  double d, d2;
  int pos, idx;
  __device__ __forceinline__ void none() {
    d = d2 = SMX_INF;
    pos = idx = 0x7fffffff;
  }
  __device__ __forceinline__ bool worse_than(double od, int opos, double od2, int oidx) const {
    if (od != d) return od < d;
    if (opos != pos) return opos < pos;
    if (od2 != d2) return od2 < d2;
    return oidx < idx;
  }
  __device__ __forceinline__ void offer(const MapDev& m, const RouteFilter& f, const smx_pt_rec& p, double px, double py) {
    const int rp = m.route_pos[f.road[0] * m.n_roads + m.lane_road[p.lane]];
    if (rp < 0) return;
    const double dx = p.x - px, dy = p.y - py;
    const double q2 = dx * dx + dy * dy;
    const double q = sqrt(q2);
    if (worse_than(q, rp, q2, p.idx)) {
      d = q;
      d2 = q2;
      pos = rp;
      idx = p.idx;
    }
  }
};

__device__ inline int closest_on_route(const MapDev& m, const RouteFilter& f, double px, double py) {
  RouteBest b;
  b.none();
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int rmax = lp_max_ring(m, cx, cy);
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit(m, cx, cy, r, [&](const smx_pt_rec& p) { b.offer(m, f, p, px, py); });
    if (b.idx != 0x7fffffff && ring_covers(m, r, b.d2)) break;
  }
  return b.idx == 0x7fffffff ? -1 : b.idx;
}

// has_route_object: the agent carries a (possibly empty) Route — the controller and the
// waypoints sensor do; TripMeterSensor's constructor query does not.  `ms`, `slot`: the missions
// table and the agent's slot (a fixed route replaces the in-junction rule: sumo_road_network.py:822-829).
__device__ inline PathSeeds compute_path_seeds(const MapDev& m, double px, double py, double heading,
                                               double within_radius, bool has_route_object,
                                               const MissionsDev* ms = nullptr, int slot = 0) {
  PathSeeds s;
  s.f.none();
  s.road = -1;
  s.n_lanes = 0;
#pragma unroll
  for (int q = 0; q < SMX_SEED_LANES; ++q) s.start[q] = -1;
  Top10 t;
  nearest10(m, px, py, t);
  bool routed = false;
  if (has_route_object && ms != nullptr && s.f.fixed_route(*ms, slot, m.n_roads)) {
    const int best = closest_on_route(m, s.f, px, py);
    s.road = best >= 0 ? m.lane_road[m.lp_rec[best].lane] : -1;
    routed = true;
  } else if (has_route_object) {
    // _resolve_in_junction (:842-860)
    int lp = pick_closest(m, t, heading, -1.0);
    if (lp >= 0) {
      int road = m.lane_road[SMX_BCHK(19, m.lp_rec[lp].lane, m.n_lanes)];
      if (m.road_is_junction[SMX_BCHK(20, road, m.n_roads)]) {
        s.f.n = 1;
        s.f.road[0] = road;
        int nr = m.road_out_road[road];
        if (nr >= 0) {
          s.f.n = 2;
          s.f.road[1] = nr;
        }
        // _waypoint_paths_along_route (:862-882): nearest lanepoint over the route roads; the
        // reference compares np.linalg.norm distances, first minimum wins
        int idx4[4];
        double d24[4];
        closest_filtered4(m, px, py, s.f.road, s.f.n, true, idx4, d24);
        double bd = SMX_INF;
        int best = -1;
        for (int k = 0; k < s.f.n; ++k) {
          double d = sqrt(d24[k]);
          if (idx4[k] >= 0 && d < bd) {
            bd = d;
            best = idx4[k];
          }
        }
        s.road = best >= 0 ? m.lane_road[m.lp_rec[best].lane] : -1;
        routed = true;
      }
    }
  }
  if (!routed) {
    int lp = pick_closest(m, t, heading, within_radius);
    s.road = lp >= 0 ? m.lane_road[m.lp_rec[lp].lane] : -1;
  }
  if (s.road >= 0) {
    const int la = m.road_lane_off[SMX_BCHK(21, s.road, m.n_roads)], lb = m.road_lane_off[s.road + 1];
    s.n_lanes = lb - la;
    int keys[4] = {-9, -9, -9, -9};
    const int nk = min(s.n_lanes, SMX_SEED_LANES);
    for (int q = 0; q < nk; ++q) keys[q] = m.road_lanes[la + q];
    closest_filtered4(m, px, py, keys, nk, false, s.start, nullptr);
  }
  return s;
}

// Start lanepoint on lane number `li` (position within the road) of the seed road.
__device__ __forceinline__ int seed_start(const MapDev& m, const PathSeeds& s, int li, double px, double py) {
  if (li < SMX_SEED_LANES) {
    // a tree of selects on the index's bits keeps the array in registers (an indexed read, and this compiler's rewrite of
    // a chain of equality selects into one, put the whole PathSeeds into scratch memory: 40 B a lane in every walk kernel)
    const int lo = (li & 1) ? s.start[1] : s.start[0], hi = (li & 1) ? s.start[3] : s.start[2];
    return (li & 2) ? hi : lo;
  }
  int key[4] = {m.road_lanes[m.road_lane_off[s.road] + li], -9, -9, -9};
  int idx[4];
  closest_filtered4(m, px, py, key, 1, false, idx, nullptr);
  return idx[0];
}
