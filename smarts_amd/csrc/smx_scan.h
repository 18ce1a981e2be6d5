// smx_scan.h — team-cooperative map scans: SMX_TEAM adjacent lanes of a wavefront serve ONE
// vehicle.  The sweeps of the segment grid (nearest lane / road_with_point) and of the lanepoint
// grid (10 nearest lanepoints, nearest lanepoint per lane) are embarrassingly parallel over grid
// members; each lane takes every SMX_TEAM-th member and the team combines with xor-shuffles
// (wave64 cross-lane ops, no LDS).  Results are identical to the one-thread forms in
// smx_roadmap.h: all reductions are exact minima with the same (value, table index) tie order.
#pragma once
#include "smx_roadmap.h"

#ifndef SMX_TEAM
#define SMX_TEAM 8  // lanes per vehicle on small batches (one wavefront's latency); large batches: SMX_TEAM_LARGE
#endif
#ifndef SMX_TEAM_LARGE
#define SMX_TEAM_LARGE 4  // fewer lanes repeat the per-vehicle uniform work (cell ranges, merges) at 131 k vehicles
#endif

template <int TEAM>
__device__ __forceinline__ int team_rank() { return threadIdx.x & (TEAM - 1); }

// lexicographic (value, index) minimum across the team; every lane receives the result
template <int TEAM>
__device__ __forceinline__ void team_min_pair(double& d, int& idx) {
#pragma unroll
  for (int msk = TEAM / 2; msk >= 1; msk >>= 1) {
    double od = __shfl_xor(d, msk, TEAM);
    int oi = __shfl_xor(idx, msk, TEAM);
    if (od < d || (od == d && oi < idx)) {
      d = od;
      idx = oi;
    }
  }
}

template <int TEAM>
__device__ __forceinline__ int team_or(int v) {
#pragma unroll
  for (int msk = TEAM / 2; msk >= 1; msk >>= 1) v |= __shfl_xor(v, msk, TEAM);
  return v;
}

// ---------------------------------------------------------------------------------
// road facts (see road_facts_scan): centre + 4 corners, one sweep, segments strided over the team
// ---------------------------------------------------------------------------------
// `dist_bound`: an upper bound of the nearest lane's distance known to the caller (last tick's nearest lane cannot
// have moved farther away than the vehicle has driven), or SMX_INF: segments whose bounding box lies beyond it and
// beyond the road_with_point thresholds are dropped before any distance is taken.  `cell_radius` (<= radius): how
// far from the centre the grid cells are enumerated; the caller passes a value that covers `dist_bound` and every
// segment a corner test can see (its threshold + half a vehicle diagonal), or `radius` itself.
template <int TEAM>
__device__ inline RoadFacts team_road_facts(const MapDev& m, double px, double py, double radius, int n_corners,
                                            const double* cx, const double* cy, double dist_bound = SMX_INF,
                                            double cell_radius = -1.0) {
  RoadFacts out;
  out.lane = -1;
  out.dist = SMX_INF;
  out.on_road = false;
  out.corner_mask = 0;
  int lane_key = 0x7fffffff;  // lane id as tie key (INT_MAX = none)
  int on_road = 0;
  const int r = team_rank<TEAM>();
  const double road_radius = fmax(5.0, 2.0 * m.default_lane_width);  // sumo_road_network.py:705
  const double reach = (cell_radius >= 0.0 && cell_radius < radius) ? cell_radius : radius;
  int cx0 = (int)floor((px - reach - m.sg_x0) / m.sg_cell);
  int cx1 = (int)floor((px + reach - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((py - reach - m.sg_y0) / m.sg_cell);
  int cy1 = (int)floor((py + reach - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  if (cx0 <= cx1) {
    for (int gy = cy0; gy <= cy1; ++gy) {
      const int row = gy * m.sg_nx;
      const int a = m.sg_off[row + cx0], b = m.sg_off[row + cx1 + 1];
      for (int k = a + r; k < b; k += TEAM) {
        const smx_seg_rec s = m.sg_rec[k];
        {
          const double bx0 = fmin(s.x1, s.x2), bx1 = fmax(s.x1, s.x2), by0 = fmin(s.y1, s.y2), by1 = fmax(s.y1, s.y2);
          const double gx = fmax(fmax(bx0 - px, px - bx1), 0.0), gy2 = fmax(fmax(by0 - py, py - by1), 0.0);
          const double lb2 = gx * gx + gy2 * gy2;
          const double keep_c = fmin(fmax(fmin(out.dist, dist_bound), s.thr), radius) + 1e-6;
          const double keep_q = s.thr + 2.0 + 1e-6;
          const double keep = n_corners > 0 ? fmax(keep_c, keep_q) : keep_c;
          if (lb2 > keep * keep) continue;
        }
        const double ex = s.x1 - s.x2, ey = s.y1 - s.y2;
        const double d = sqrt(ex * ex + ey * ey);
        const double dd = d * d;
        const double sx = s.x2 - s.x1, sy = s.y2 - s.y1;
#pragma unroll
        for (int q = -1; q < 4; ++q) {
          if (q >= n_corners) break;
          const double qx = q < 0 ? px : (q == 0 ? cx[0] : (q == 1 ? cx[1] : (q == 2 ? cx[2] : cx[3])));
          const double qy = q < 0 ? py : (q == 0 ? cy[0] : (q == 1 ? cy[1] : (q == 2 ? cy[2] : cy[3])));
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
              if (dist < out.dist || (dist == out.dist && s.lane < lane_key)) {
                out.dist = dist;
                lane_key = s.lane;
              }
              if (dist < road_radius && dist < s.thr) on_road = 1;
            }
          } else {
            if (dist < road_radius && dist < s.thr) out.corner_mask |= (1 << q);
          }
        }
      }
    }
  }
  team_min_pair<TEAM>(out.dist, lane_key);
  out.lane = lane_key == 0x7fffffff ? -1 : lane_key;
  out.on_road = team_or<TEAM>(on_road) != 0;
  out.corner_mask = team_or<TEAM>(out.corner_mask);
  return out;
}

// What the facts half of the scan keeps from tick to tick for one vehicle: the pose it last ran at and the distance
// of the nearest lane it found there (valid: there was one).
struct FactsCarry {
  bool valid;
  double qx, qy, prev_dist;
};

// team_road_facts with radius = max(pose_radius, 2 x default lane width), seeded from `c`: last tick's nearest lane
// lies at most (its distance then + the way driven since) away, which bounds the nearest distance now; the corners
// see segments up to the widest road_with_point threshold (`thr_max`) + half a vehicle diagonal (< 2 m) away.
template <int TEAM>
__device__ inline RoadFacts team_road_facts_seeded(const MapDev& m, double px, double py, double pose_radius, int n_corners,
                                                   const double* cx, const double* cy, const FactsCarry& c, double thr_max) {
  const double radius = fmax(pose_radius, 2.0 * m.default_lane_width);
  double dist_bound = SMX_INF, cell_radius = -1.0;
  if (c.valid && c.qx == c.qx && c.qy == c.qy && c.prev_dist < radius) {
    dist_bound = (c.prev_dist + euclid(px, py, c.qx, c.qy)) * (1.0 + 1e-12) + 1e-9;
    cell_radius = fmax(dist_bound, thr_max + (n_corners > 0 ? 2.0 : 0.0)) + 2e-6;
  }
  return team_road_facts<TEAM>(m, px, py, radius, n_corners, cx, cy, dist_bound, cell_radius);
}

// ---------------------------------------------------------------------------------
// lanepoint grid, members strided over the team
// ---------------------------------------------------------------------------------
template <int TEAM, class F>
__device__ __forceinline__ void lp_ring_visit_team(const MapDev& m, int cx, int cy, int r, int rank, F&& f) {
  const int y0 = cy - r, y1 = cy + r, x0 = cx - r, x1 = cx + r;
  for (int y = max(y0, 0); y <= min(y1, m.lpg_ny - 1); ++y) {
    const int row = y * m.lpg_nx;
    if (y == y0 || y == y1) {
      const int xa = max(x0, 0), xb = min(x1, m.lpg_nx - 1);
      if (xa > xb) continue;
      const int a = m.lpg_off[row + xa], b = m.lpg_off[row + xb + 1];
      for (int k = a + rank; k < b; k += TEAM) f(m.lpg_pts[k]);
    } else {
      if (x0 >= 0 && x0 < m.lpg_nx) {
        const int a = m.lpg_off[row + x0], b = m.lpg_off[row + x0 + 1];
        for (int k = a + rank; k < b; k += TEAM) f(m.lpg_pts[k]);
      }
      if (x1 != x0 && x1 >= 0 && x1 < m.lpg_nx) {
        const int a = m.lpg_off[row + x1], b = m.lpg_off[row + x1 + 1];
        for (int k = a + rank; k < b; k += TEAM) f(m.lpg_pts[k]);
      }
    }
  }
}

// Rings 0..R around (cx, cy) in one go: the (2R+1) rows of the block are (2R+1) contiguous member
// ranges, whose 2 (2R+1) offsets are loaded together before any member is touched.  Ring by ring
// the same cells cost a dependent offset load per row segment (13 for R = 2) in front of their
// members — with one wavefront per SIMD nothing hides that latency.  The order of the visit does
// not matter to the callers (minima with an index tie-break).
template <int TEAM, int R, class F>
__device__ __forceinline__ void lp_block_visit_team(const MapDev& m, int cx, int cy, int rank, F&& f) {
  constexpr int ROWS = 2 * R + 1;
  int a[ROWS], b[ROWS];
  const int xa = max(cx - R, 0), xb = min(cx + R, m.lpg_nx - 1);
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int y = cy - R + i;
    const bool in = y >= 0 && y < m.lpg_ny && xa <= xb;
    const int row = (in ? y : 0) * m.lpg_nx;
    const int va = m.lpg_off[in ? row + xa : 0], vb = m.lpg_off[in ? row + xb + 1 : 0];
    a[i] = va;
    b[i] = in ? vb : va;
  }
  // step j takes this lane's j-th member of EVERY row: the ROWS records are loaded back to back
  // (independent addresses) and only then looked at, instead of one load -> wait -> use per member
  for (int j = 0;; ++j) {
    smx_pt_rec rec[ROWS];
    bool ok[ROWS];
    bool any = false;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int k = a[i] + rank + j * TEAM;
      ok[i] = k < b[i];
      any = any || ok[i];
      rec[i] = m.lpg_pts[ok[i] ? k : 0];
    }
    if (!any) break;
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
      if (ok[i]) f(rec[i]);
  }
}

// The 10 nearest lanepoints (see nearest10).  Each lane keeps the best 10 of its share; the team
// merges by repeatedly taking the smallest head.  Every lane ends with the same Top10.
template <int TEAM>
__device__ inline void team_nearest10(const MapDev& m, double px, double py, Top10& res) {
  const int K = 10;
  double ld[K];
  int li[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    ld[i] = SMX_INF;
    li[i] = 0x7fffffff;
  }
  const int rank = team_rank<TEAM>();
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int keff = min(K, m.n_lanepoints);
  const int rmax = lp_max_ring(m, cx, cy);
  // `bound`: the merged 10th smallest distance so far (SMX_INF before the first merge).  A point
  // beyond it cannot be one of the 10 nearest, so it is not worth the insertion chain.
  double bound = SMX_INF;
  auto take = [&](const smx_pt_rec& p) {
    double dx = p.x - px, dy = p.y - py;
    double d2 = dx * dx + dy * dy;
    if (d2 <= bound && (d2 < ld[K - 1] || (d2 == ld[K - 1] && p.idx < li[K - 1]))) {
      double cd = d2;
      int ci = p.idx;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        bool lt = (cd < ld[k]) || (cd == ld[k] && ci < li[k]);
        double td = lt ? ld[k] : cd;
        int ti = lt ? li[k] : ci;
        ld[k] = lt ? cd : ld[k];
        li[k] = lt ? ci : li[k];
        cd = td;
        ci = ti;
      }
    }
  };
  int r = 0;
  for (;;) {
    if (r == 0) {
      // rings 0-2 before the first merge: with 4 m cells the 10th neighbour is usually inside ring 2
      lp_block_visit_team<TEAM, 2>(m, cx, cy, rank, take);
      r = 3;
    } else {
      lp_ring_visit_team<TEAM>(m, cx, cy, r, rank, take);
      ++r;
    }
    // merge (on copies: the local lists keep growing if another ring is needed)
    double hd[K];
    int hi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      hd[i] = ld[i];
      hi[i] = li[i];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      double wd = hd[0];
      int wi = hi[0];
      team_min_pair<TEAM>(wd, wi);
      res.d2[j] = wd;
      res.idx[j] = (wi == 0x7fffffff) ? -1 : wi;
      const bool mine = (hi[0] == wi) && (wi != 0x7fffffff);
#pragma unroll
      for (int k = 0; k + 1 < K; ++k) {
        hd[k] = mine ? hd[k + 1] : hd[k];
        hi[k] = mine ? hi[k + 1] : hi[k];
      }
      if (mine) {
        hd[K - 1] = SMX_INF;
        hi[K - 1] = 0x7fffffff;
      }
    }
    bound = fmin(bound, res.d2[K - 1]);  // SMX_INF while fewer than 10 points have been seen
    const int last = r - 1;  // last completed ring
    bool full = false;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k == keff - 1) full = res.idx[k] >= 0 && ring_covers(m, last, res.d2[k]);
    if (full || last >= rmax) break;
  }
}

// ---------------------------------------------------------------------------------
// Seeded searches.  A vehicle moves less than two metres per tick, so last tick's answers bound this tick's:
//   * the 10 lanepoints that were nearest at the previous pose q all lie within R10 = d10(q) + |p - q| of the new
//     pose p, so the 10th nearest distance at p is at most R10 and no point beyond R10 can be one of the 10;
//   * the lanepoint that was nearest on lane L still is a lanepoint of L: its distance to p bounds L's new nearest.
// Every grid cell that meets the square of half width R around p is visited once, R the largest of those bounds:
// the visit is exact (no ring certification), a handful of cells instead of the 5 x 5 block, and serves both the
// ten-nearest query and the nearest lanepoint per lane of the road the paths started on last tick (a guess the
// caller checks against the road this tick's ten nearest choose).  Results are those of team_nearest10 /
// team_closest_filtered4 bit for bit: the same (d2, index) minima.
// ---------------------------------------------------------------------------------
#define SMX_SEEDED_SPAN 6  // cells per side the seeded visit handles (R up to ~10 m with 4 m cells); larger: unseeded search
struct LaneGuess {
  int road;     // the road whose lanes were tracked (-1: none)
  int nk;       // lanes tracked (the road's first min(n_lanes, 4))
  int idx[4];   // nearest lanepoint per tracked lane (-1 none)
};

template <int TEAM>
__device__ inline bool team_nearest10_seeded(const MapDev& m, double px, double py, double r10, double reach, int k0, int k1,
                                             int k2, int k3, int nkeys, Top10& res, int* lane_idx) {
  const int K = 10;
  int cx0 = (int)floor((px - reach - m.lpg_x0) / m.lpg_cell), cx1 = (int)floor((px + reach - m.lpg_x0) / m.lpg_cell);
  int cy0 = (int)floor((py - reach - m.lpg_y0) / m.lpg_cell), cy1 = (int)floor((py + reach - m.lpg_y0) / m.lpg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.lpg_nx - 1);
  cy1 = min(cy1, m.lpg_ny - 1);
  if (cx0 > cx1 || cy0 > cy1 || cy1 - cy0 >= SMX_SEEDED_SPAN || cx1 - cx0 >= SMX_SEEDED_SPAN) return false;
  double ld[K];
  int li[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    ld[i] = SMX_INF;
    li[i] = 0x7fffffff;
  }
  double bd[4] = {SMX_INF, SMX_INF, SMX_INF, SMX_INF};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  const double bound = r10 * r10;
  const int rank = team_rank<TEAM>();
  // the rows of the square are contiguous member ranges: their offsets are loaded together, then the members are
  // taken as ONE sequence (member n of the concatenated rows), strided over the team, four loads in flight
  int ra[SMX_SEEDED_SPAN], rn[SMX_SEEDED_SPAN];  // first member and running member count of every row
  int total = 0;
#pragma unroll
  for (int i = 0; i < SMX_SEEDED_SPAN; ++i) {
    const int y = cy0 + i;
    const bool in = y <= cy1;
    const int row = (in ? y : cy0) * m.lpg_nx;
    const int va = m.lpg_off[row + cx0], vb = m.lpg_off[row + cx1 + 1];
    ra[i] = va;
    total += in ? vb - va : 0;
    rn[i] = total;
  }
  auto member = [&](int n) {  // table index of member n of the concatenated rows
    int k = ra[0] + n;
#pragma unroll
    for (int i = 1; i < SMX_SEEDED_SPAN; ++i)
      if (n >= rn[i - 1]) k = ra[i] + (n - rn[i - 1]);
    return k;
  };
  auto take = [&](const smx_pt_rec& p) {
    const double dx = p.x - px, dy = p.y - py;
    const double d2 = dx * dx + dy * dy;
    if (d2 <= bound && (d2 < ld[K - 1] || (d2 == ld[K - 1] && p.idx < li[K - 1]))) {
      double cd = d2;
      int ci = p.idx;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool lt = (cd < ld[k]) || (cd == ld[k] && ci < li[k]);
        const double td = lt ? ld[k] : cd;
        const int ti = lt ? li[k] : ci;
        ld[k] = lt ? cd : ld[k];
        li[k] = lt ? ci : li[k];
        cd = td;
        ci = ti;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
      if (q < nkeys && p.lane == kq && (d2 < bd[q] || (d2 == bd[q] && p.idx < bi[q]))) {
        bd[q] = d2;
        bi[q] = p.idx;
      }
    }
  };
  for (int n0 = rank; n0 < total; n0 += 4 * TEAM) {
    smx_pt_rec rec[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + u * TEAM;
      rec[u] = m.lpg_pts[member(n < total ? n : 0)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (n0 + u * TEAM < total) take(rec[u]);
  }
  // merge the team's lists (as team_nearest10 does); one lane: the list is the result
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double wd = ld[0];
    int wi = li[0];
    team_min_pair<TEAM>(wd, wi);
    res.d2[j] = wd;
    res.idx[j] = (wi == 0x7fffffff) ? -1 : wi;
    const bool mine = (li[0] == wi) && (wi != 0x7fffffff);
#pragma unroll
    for (int k = 0; k + 1 < K; ++k) {
      ld[k] = mine ? ld[k + 1] : ld[k];
      li[k] = mine ? li[k + 1] : li[k];
    }
    if (mine) {
      ld[K - 1] = SMX_INF;
      li[K - 1] = 0x7fffffff;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double d = bd[q];
    int i = bi[q];
    if (q < nkeys) team_min_pair<TEAM>(d, i);
    lane_idx[q] = (q < nkeys && i != 0x7fffffff) ? i : -1;
  }
  return res.idx[K - 1] >= 0;  // ten points were found inside the bound, as the bound promises (else: unseeded search)
}

// What the seeds half of the scan keeps from tick to tick for one vehicle.
struct SeedsCarry {
  bool valid;
  double qx, qy, d10;   // pose of the last search, d2 of its 10th nearest lanepoint (< 0 or NaN: none)
  double d1;            // d2 of its nearest lanepoint (< 0 or NaN: none)
  int prev_road, prev_lanes, prev_start[SMX_SEED_LANES];  // last tick's path seeds (seed_cache)
};

// The ten nearest lanepoints at (px, py), seeded from `c` when it holds a usable bound, and the nearest lanepoint on
// each lane of last tick's seed road (`guess`, road -1 when not available).
template <int TEAM>
__device__ inline void team_nearest10_carried(const MapDev& m, double px, double py, const SeedsCarry& c, Top10& t,
                                              LaneGuess& guess) {
  guess.road = -1;
  guess.nk = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) guess.idx[q] = -1;
  bool seeded = false;
  if (c.valid && c.d10 >= 0.0 && c.d10 < 1.0e290 && c.qx == c.qx && c.qy == c.qy) {
    const double r10 = (sqrt(c.d10) + euclid(px, py, c.qx, c.qy)) * (1.0 + 1e-12) + 1e-9;
    double reach = r10;
    int k0 = -9, k1 = -9, k2 = -9, k3 = -9, nk = 0;
    if (c.prev_road >= 0 && c.prev_lanes >= 1) {
      nk = min(c.prev_lanes, SMX_SEED_LANES);
      const int la = m.road_lane_off[c.prev_road];
      k0 = m.road_lanes[la];
      k1 = nk > 1 ? m.road_lanes[la + 1] : -9;
      k2 = nk > 2 ? m.road_lanes[la + 2] : -9;
      k3 = nk > 3 ? m.road_lanes[la + 3] : -9;
      bool all = true;
#pragma unroll
      for (int q = 0; q < SMX_SEED_LANES; ++q) {
        const int st = c.prev_start[q];
        if (q < nk) {
          if (st >= 0) {
            const smx_lp_rec* r = m.lp_rec + st;
            reach = fmax(reach, euclid(px, py, r->x, r->y) * (1.0 + 1e-12) + 1e-9);
          } else {
            all = false;
          }
        }
      }
      if (!all) nk = 0;
    }
    seeded = team_nearest10_seeded<TEAM>(m, px, py, r10, reach, k0, k1, k2, k3, nk, t, guess.idx);
    if (seeded && nk > 0) {
      bool all = true;
#pragma unroll
      for (int q = 0; q < SMX_SEED_LANES; ++q) all = all && (q >= nk || guess.idx[q] >= 0);
      if (all) {
        guess.road = c.prev_road;
        guess.nk = nk;
      }
    }
  }
  if (!seeded) team_nearest10<TEAM>(m, px, py, t);
}

// Nearest lanepoint per key (lane or road), up to 4 keys at once (see closest_filtered4).
template <int TEAM>
__device__ inline void team_closest_filtered4(const MapDev& m, double px, double py, int k0, int k1, int k2, int k3,
                                              int nkeys, bool by_road, int* out_idx, double* out_d2) {
  double bd[4] = {SMX_INF, SMX_INF, SMX_INF, SMX_INF};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  const int rank = team_rank<TEAM>();
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int rmax = lp_max_ring(m, cx, cy);
  double rd[4];
  int ri[4];
  auto take = [&](const smx_pt_rec& p) {
    const int key = by_road ? m.lane_road[p.lane] : p.lane;
    double dx = p.x - px, dy = p.y - py;
    double d2 = dx * dx + dy * dy;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
      if (q < nkeys && key == kq && (d2 < bd[q] || (d2 == bd[q] && p.idx < bi[q]))) {
        bd[q] = d2;
        bi[q] = p.idx;
      }
    }
  };
  // rings 0-2 at once (the lanes of a road lie up to ~3 lane widths away: ring 1 seldom certifies
  // them all), then ring by ring; a certified minimum is the global one whichever ring certifies it
  for (int r = 2; r <= max(rmax, 2); ++r) {
    if (r == 2)
      lp_block_visit_team<TEAM, 2>(m, cx, cy, rank, take);
    else
      lp_ring_visit_team<TEAM>(m, cx, cy, r, rank, take);
    bool all = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rd[q] = bd[q];
      ri[q] = bi[q];
      if (q < nkeys) {
        team_min_pair<TEAM>(rd[q], ri[q]);
        all = all && ri[q] != 0x7fffffff && ring_covers(m, r, rd[q]);
      }
    }
    if (all) break;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double d = bd[q];
    int i = bi[q];
    if (q < nkeys) team_min_pair<TEAM>(d, i);
    out_idx[q] = (q < nkeys && i != 0x7fffffff) ? i : -1;
    if (out_d2) out_d2[q] = d;
  }
}

// top10_heading_terms over the team: lane r evaluates candidates r, r + TEAM, ...; every lane
// receives all ten by shuffles.
template <int TEAM>
__device__ inline Top10Scores team_top10_heading_terms(const MapDev& m, const Top10& t, double heading) {
  const int rank = team_rank<TEAM>();
  constexpr int PER = (10 + TEAM - 1) / TEAM;
  double mine[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int k = rank + j * TEAM;
    int idx = 0;
#pragma unroll
    for (int q = 0; q < 10; ++q)
      if (q == k) idx = t.idx[q];
    const double h = m.lp_rec[idx < 0 ? 0 : idx].heading;
    mine[j] = fabs(heading_relative_to(heading, h));
  }
  Top10Scores sc;
#pragma unroll
  for (int k = 0; k < 10; ++k) sc.rel[k] = __shfl(mine[k / TEAM], k % TEAM, TEAM);
  return sc;
}

// closest_on_route (smx_roadmap.h), the ring cells strided over the team
template <int TEAM>
__device__ inline int team_closest_on_route(const MapDev& m, const RouteFilter& f, double px, double py) {
  RouteBest b;
  b.none();
  const int rank = team_rank<TEAM>();
  const int cx = (int)floor((px - m.lpg_x0) / m.lpg_cell);
  const int cy = (int)floor((py - m.lpg_y0) / m.lpg_cell);
  const int rmax = lp_max_ring(m, cx, cy);
  RouteBest all;
  all.none();
  for (int r = 0; r <= rmax; ++r) {
    lp_ring_visit_team<TEAM>(m, cx, cy, r, rank, [&](const smx_pt_rec& p) { b.offer(m, f, p, px, py); });
    all = b;
#pragma unroll
    for (int msk = TEAM / 2; msk >= 1; msk >>= 1) {
      const double od = __shfl_xor(all.d, msk, TEAM), od2 = __shfl_xor(all.d2, msk, TEAM);
      const int opos = __shfl_xor(all.pos, msk, TEAM), oidx = __shfl_xor(all.idx, msk, TEAM);
      if (all.worse_than(od, opos, od2, oidx)) {
        all.d = od;
        all.d2 = od2;
        all.pos = opos;
        all.idx = oidx;
      }
    }
    if (all.idx != 0x7fffffff && ring_covers(m, r, all.d2)) break;
  }
  return all.idx == 0x7fffffff ? -1 : all.idx;
}

// compute_path_seeds, team form (see smx_roadmap.h for the semantics).  ROUTED: some slot has a fixed route
// (the instance without that code serves every batch until smx_set_missions is called).
template <int TEAM, bool ROUTED>
__device__ inline PathSeeds team_compute_path_seeds(const MapDev& m, double px, double py, double heading,
                                                    double within_radius, bool has_route_object, const Top10& t,
                                                    const Top10Scores& sc, const MissionsDev& ms, int slot,
                                                    const LaneGuess* guess = nullptr) {
  PathSeeds s;
  s.f.none();
  s.road = -1;
  s.n_lanes = 0;
#pragma unroll
  for (int q = 0; q < SMX_SEED_LANES; ++q) s.start[q] = -1;
  bool routed = false;
  SMX_TSTAMP(tp0);
  if (ROUTED && has_route_object && s.f.fixed_route(ms, slot, m.n_roads)) {
    const int best = team_closest_on_route<TEAM>(m, s.f, px, py);
    s.road = best >= 0 ? m.lane_road[m.lp_rec[best].lane] : -1;
    routed = true;
  } else if (has_route_object) {
    int lp = pick_closest(t, sc, -1.0);
    if (lp >= 0) {
      int road = m.lane_road[m.lp_rec[lp].lane];
      if (m.road_is_junction[road]) {
        s.f.n = 1;
        s.f.road[0] = road;
        int nr = m.road_out_road[road];
        if (nr >= 0) {
          s.f.n = 2;
          s.f.road[1] = nr;
        }
        int idx4[4];
        double d24[4];
        team_closest_filtered4<TEAM>(m, px, py, s.f.road[0], s.f.road[1], -9, -9, s.f.n, true, idx4, d24);
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
    int lp = pick_closest(t, sc, within_radius);
    s.road = lp >= 0 ? m.lane_road[m.lp_rec[lp].lane] : -1;
  }
  SMX_TSTAMP(tp1);
  SMX_TACC(21, tp0, tp1);
  if (s.road >= 0) {
    const int la = m.road_lane_off[s.road], lb = m.road_lane_off[s.road + 1];
    s.n_lanes = lb - la;
    const int nk = min(s.n_lanes, SMX_SEED_LANES);
    const int k0 = m.road_lanes[la], k1 = nk > 1 ? m.road_lanes[la + 1] : -9, k2 = nk > 2 ? m.road_lanes[la + 2] : -9,
              k3 = nk > 3 ? m.road_lanes[la + 3] : -9;
    SMX_TSTAMP(tp2);
    SMX_TACC(22, tp1, tp2);
    if (guess != nullptr && guess->road == s.road && guess->nk == nk) {
      // the seeded visit has already found the nearest lanepoint on each of this road's lanes
#pragma unroll
      for (int q = 0; q < SMX_SEED_LANES; ++q) s.start[q] = q < nk ? guess->idx[q] : -1;
    } else {
      team_closest_filtered4<TEAM>(m, px, py, k0, k1, k2, k3, nk, false, s.start, nullptr);
    }
    SMX_TSTAMP(tp3);
    SMX_TACC(23, tp2, tp3);
  }
  return s;
}


// ---------------------------------------------------------------------------------
// lane heading at the centre-line point closest to (px, py):
//   Lane.center_pose_at_point(point).heading  (road_map.py:390-396)
//   = offset_along_lane (sumo_road_network.py:476-491, utils/math.py:370-390)
//   + vector_at_offset (road_map.py:377-388) over from_lane_coord (utils/math.py:319-345)
//   + Pose(fast_quaternion_from_angle(vec_to_radians(v))).heading (coordinates.py:394-403)
// Team form of lane_heading_at_point (smx_kernels.hip): the lane's segments are strided over the
// team; the running arclength of the reference's loops comes from smx_shape_rec.cum (summed on the
// host vertex by vertex, the same additions), so every lane can evaluate its segments on its own.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool is_close_ref(double a, double b) {
  return fabs(a - b) <= fmax(1e-09 * fmax(fabs(a), fabs(b)), 0.0);
}

__device__ __forceinline__ void position_at_offset(double x1, double y1, double x2, double y2, double dist,
                                                   double offset, double& ox, double& oy) {
  // utils/math.py:300-316; `dist` is the segment length
  if (is_close_ref(offset, 0.0)) {
    ox = x1;
    oy = y1;
    return;
  }
  if (is_close_ref(dist, offset)) {
    ox = x2;
    oy = y2;
    return;
  }
  ox = x1 + (x2 - x1) * (offset / dist);
  oy = y1 + (y2 - y1) * (offset / dist);
}

// position_at_shape_offset (utils/math.py:319-331): the first segment v with cum[v] + len[v] > offset
// `vfrom` (v0 <= vfrom): a vertex with cum[vfrom] <= offset.  No segment before it can be the answer
// (cum[v + 1] = cum[v] + len[v] <= cum[vfrom] <= offset for v < vfrom), so the search starts there.
template <int TEAM>
__device__ inline void team_position_at_shape_offset(const MapDev& m, int v0, int v1, double offset, double& ox,
                                                     double& oy, int vfrom) {
  const int r = team_rank<TEAM>();
  int hit = 0x7fffffff;
  (void)v0;
  for (int v = vfrom + r; v + 1 < v1; v += TEAM) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.cum + a.len > offset) {
      hit = v;
      break;  // cum is non-decreasing along the lane
    }
  }
#pragma unroll
  for (int msk = TEAM / 2; msk >= 1; msk >>= 1) hit = min(hit, __shfl_xor(hit, msk, TEAM));
  if (hit == 0x7fffffff) {
    const smx_shape_rec z = m.shape_rec[v1 - 1];
    ox = z.x;
    oy = z.y;
    return;
  }
  const smx_shape_rec a = m.shape_rec[hit], b = m.shape_rec[hit + 1];
  position_at_offset(a.x, a.y, b.x, b.y, a.len, offset - a.cum, ox, oy);
}

// `dist_hint`: an upper bound of the point's distance to this lane's centre line known to the caller (the
// nearest-lane sweep has just measured it), or SMX_INF: segments whose bounding box lies farther cannot hold
// the minimum and are skipped from the first one on, not only once the sweep has found a good candidate.
template <int TEAM>
__device__ inline double team_lane_heading_at_point_linear(const MapDev& m, int lane, double px, double py, double dist_hint) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  const int r = team_rank<TEAM>();
  // ---- offset_along_lane: a vertex that equals the point wins (first such vertex) ...
  int vertex_hit = 0x7fffffff;
  // ... else the first segment at minimum distance
  double min_dist = SMX_INF, min_offset = -1.0;
  int min_v = 0x7fffffff;
  for (int v = v0 + r; v < v1; v += TEAM) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.x == px && a.y == py) vertex_hit = min(vertex_hit, v);
    if (v + 1 >= v1) continue;
    const smx_shape_rec b = m.shape_rec[v + 1];
    {
      // a segment whose bounding box is farther than this lane's best so far can neither be the
      // team's minimum nor tie it
      const double gx = fmax(fmax(fmin(a.x, b.x) - px, px - fmax(a.x, b.x)), 0.0);
      const double gy = fmax(fmax(fmin(a.y, b.y) - py, py - fmax(a.y, b.y)), 0.0);
      // (the hint comes from another evaluation of the same distance: 1e-6 covers their rounding many times over)
      const double keep = fmin(min_dist, dist_hint) + 1e-6;
      if (gx * gx + gy * gy > keep * keep) continue;
    }
    const double d = a.len;
    const double u = ((px - a.x) * (b.x - a.x)) + ((py - a.y) * (b.y - a.y));
    const double poff = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
    double fx, fy;
    position_at_offset(a.x, a.y, b.x, b.y, d, poff, fx, fy);
    const double dist = euclid(px, py, fx, fy);
    if (dist < min_dist) {
      min_dist = dist;
      min_offset = poff + a.cum;
      min_v = v;
    }
  }
#pragma unroll
  for (int msk = TEAM / 2; msk >= 1; msk >>= 1) vertex_hit = min(vertex_hit, __shfl_xor(vertex_hit, msk, TEAM));
  int v_near = v0;  // the vertex the offset was measured from
  double offset;
  if (vertex_hit != 0x7fffffff) {
    offset = m.shape_rec[vertex_hit].cum;
    v_near = vertex_hit;
  } else {
    double bd = min_dist;
    int bv = min_v;
    team_min_pair<TEAM>(bd, bv);  // smallest distance, then the earliest segment
    const int owner = (bv == 0x7fffffff) ? 0 : ((bv - v0) & (TEAM - 1));
    offset = __shfl(min_offset, owner, TEAM);
    if (bv != 0x7fffffff) v_near = bv;
  }
  // ---- vector_at_offset
  const double L = m.lane_length[lane];
  double s_off, e_off;
  if (offset >= L) {
    s_off = L - 1.0;
    e_off = L;
  } else {
    s_off = offset;
    e_off = offset + 1.0;
  }
  s_off = fmax(s_off, 0.0);
  double p1x, p1y, p2x, p2y;
  // both offsets lie at or beyond the vertex the offset was measured from, unless the end of the lane pulled
  // them back (offset >= L): the lane's vertices before it need not be looked at again
  const int vfrom = (m.shape_rec[v_near].cum <= s_off) ? v_near : v0;
  team_position_at_shape_offset<TEAM>(m, v0, v1, s_off, p1x, p1y, vfrom);
  team_position_at_shape_offset<TEAM>(m, v0, v1, e_off, p2x, p2y, vfrom);
  const double ang = vec_to_radians(p2x - p1x, p2y - p1y);
  const double half = ang * 0.5;
  const double qz = sin(half), qw = cos(half);
  return wrap_heading(atan2(2.0 * (0.0 * 0.0 + qw * qz), qw * qw + 0.0 * 0.0 - 0.0 * 0.0 - qz * qz));
}


// The same, with the candidate segments taken from the segment grid instead of a walk over the lane's vertices
// (a lane of scenarios/loop has up to 49 of them).  `dist_hint` is the lane's distance as the nearest-lane sweep has
// just measured it (another evaluation of the same quantity, so 1e-6 covers their rounding many times over): only
// segments whose bounding box lies within it can hold the minimum of offset_along_lane — or a vertex equal to the
// point — and every such segment is listed in a grid cell that meets the square of that half width around the
// point.  Candidates come in cell order, so the reference's "first segment at the minimum" (a strict < in vertex
// order) is the lexicographic minimum of (distance, vertex index); a segment listed in several cells is evaluated
// more than once, with the same result.
template <int TEAM>
__device__ inline double team_lane_heading_at_point(const MapDev& m, int lane, double px, double py, double dist_hint) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  const int r = team_rank<TEAM>();
  int vertex_hit = 0x7fffffff;
  double min_dist = SMX_INF, min_offset = -1.0;
  int min_v = 0x7fffffff;
  {
    const double reach = dist_hint + 1e-6;
    int cx0 = (int)floor((px - reach - m.sg_x0) / m.sg_cell), cx1 = (int)floor((px + reach - m.sg_x0) / m.sg_cell);
    int cy0 = (int)floor((py - reach - m.sg_y0) / m.sg_cell), cy1 = (int)floor((py + reach - m.sg_y0) / m.sg_cell);
    cx0 = max(cx0, 0);
    cy0 = max(cy0, 0);
    cx1 = min(cx1, m.sg_nx - 1);
    cy1 = min(cy1, m.sg_ny - 1);
    if (cx0 <= cx1) {
      for (int gy = cy0; gy <= cy1; ++gy) {
        const int row = gy * m.sg_nx;
        const int ka = m.sg_off[row + cx0], kb = m.sg_off[row + cx1 + 1];
        for (int k = ka + r; k < kb; k += TEAM) {
          const smx_seg_rec s = m.sg_rec[k];
          if (s.lane != lane) continue;
          if (s.x1 == px && s.y1 == py) vertex_hit = min(vertex_hit, s.v0);
          if (s.x2 == px && s.y2 == py) vertex_hit = min(vertex_hit, s.v0 + 1);
          {
            const double gx = fmax(fmax(fmin(s.x1, s.x2) - px, px - fmax(s.x1, s.x2)), 0.0);
            const double gy2 = fmax(fmax(fmin(s.y1, s.y2) - py, py - fmax(s.y1, s.y2)), 0.0);
            const double keep = fmin(min_dist, dist_hint) + 1e-6;
            if (gx * gx + gy2 * gy2 > keep * keep) continue;
          }
          const smx_shape_rec a = m.shape_rec[s.v0];  // (x, y) = (s.x1, s.y1); its length and arclength are wanted
          const double d = a.len;
          const double u = ((px - s.x1) * (s.x2 - s.x1)) + ((py - s.y1) * (s.y2 - s.y1));
          const double poff = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
          double fx, fy;
          position_at_offset(s.x1, s.y1, s.x2, s.y2, d, poff, fx, fy);
          const double dist = euclid(px, py, fx, fy);
          if (dist < min_dist || (dist == min_dist && s.v0 < min_v)) {
            min_dist = dist;
            min_offset = poff + a.cum;
            min_v = s.v0;
          }
        }
      }
    }
  }
#pragma unroll
  for (int msk = TEAM / 2; msk >= 1; msk >>= 1) vertex_hit = min(vertex_hit, __shfl_xor(vertex_hit, msk, TEAM));
  int v_near = v0;  // the vertex the offset was measured from
  double offset;
  if (vertex_hit != 0x7fffffff) {
    offset = m.shape_rec[vertex_hit].cum;
    v_near = vertex_hit;
  } else {
    double bd = min_dist;
    int bv = min_v;
    team_min_pair<TEAM>(bd, bv);  // smallest distance, then the earliest segment
    // a lane that holds the winning (distance, segment) pair hands its offset to the team
    int owner = (min_dist == bd && min_v == bv) ? r : TEAM;
#pragma unroll
    for (int msk = TEAM / 2; msk >= 1; msk >>= 1) owner = min(owner, __shfl_xor(owner, msk, TEAM));
    offset = __shfl(min_offset, owner < TEAM ? owner : 0, TEAM);
    if (bv != 0x7fffffff) v_near = bv;
  }
  // ---- vector_at_offset
  const double L = m.lane_length[lane];
  double s_off, e_off;
  if (offset >= L) {
    s_off = L - 1.0;
    e_off = L;
  } else {
    s_off = offset;
    e_off = offset + 1.0;
  }
  s_off = fmax(s_off, 0.0);
  double p1x, p1y, p2x, p2y;
  const int vfrom = (m.shape_rec[v_near].cum <= s_off) ? v_near : v0;
  team_position_at_shape_offset<TEAM>(m, v0, v1, s_off, p1x, p1y, vfrom);
  team_position_at_shape_offset<TEAM>(m, v0, v1, e_off, p2x, p2y, vfrom);
  const double ang = vec_to_radians(p2x - p1x, p2y - p1y);
  const double half = ang * 0.5;
  const double qz = sin(half), qw = cos(half);
  return wrap_heading(atan2(2.0 * (0.0 * 0.0 + qw * qz), qw * qw + 0.0 * 0.0 - 0.0 * 0.0 - qz * qz));
}


// ---------------------------------------------------------------------------------
// One lane per vehicle (large batches).  With the seeded bounds a vehicle looks at a dozen grid records and needs
// the full distance arithmetic for three to six of them; a team of lanes then mostly repeats the per-vehicle part
// (SQ counters, round 3: 4 lanes x 4.0 k instructions against 13.5 k for one lane that walks the records and
// evaluates on the spot — 64 vehicles of a wavefront meet their few hits at different records, so the wavefront
// pays the evaluation at nearly every record).  The one-lane form therefore runs in two passes: pass 1 applies the
// cheap bounding-box test to every record and keeps the survivors' indices in a per-lane list in LDS; pass 2 runs
// the distance arithmetic down the lists, whose lengths are alike across the wavefront.  Same minima, same bits as
// team_road_facts / team_lane_heading_at_point; lists that overflow (`false`) send the vehicle to those.
// ---------------------------------------------------------------------------------
#define SMX_FACTS_CAND 24  // survivors kept per vehicle (3 lanes x 1-2 segments typically)

// Two positions along a lane's centre line (position_at_shape_offset, utils/math.py:319-331, for offsets s <= e),
// searching from vertex `vfrom` (cum[vfrom] <= s): five vertex records are loaded together, which covers the common
// case (both offsets within four segments of vfrom); otherwise the walk of team_position_at_shape_offset.
__device__ inline void lane_positions_at_offsets(const MapDev& m, int v0, int v1, int vfrom, double s_off, double e_off,
                                                 double& p1x, double& p1y, double& p2x, double& p2y) {
  // (named scalars, not an array of records: the compiler kept the array in scratch memory — 176 B a lane — and
  // turned the selects below into selected scratch addresses, a memory round trip in front of each position)
  const smx_shape_rec q0 = m.shape_rec[min(vfrom + 0, v1 - 1)], q1 = m.shape_rec[min(vfrom + 1, v1 - 1)],
                      q2 = m.shape_rec[min(vfrom + 2, v1 - 1)], q3 = m.shape_rec[min(vfrom + 3, v1 - 1)],
                      q4 = m.shape_rec[min(vfrom + 4, v1 - 1)];
  const double x0 = q0.x, x1 = q1.x, x2 = q2.x, x3 = q3.x, x4 = q4.x;
  const double y0 = q0.y, y1 = q1.y, y2 = q2.y, y3 = q3.y, y4 = q4.y;
  const double l0 = q0.len, l1 = q1.len, l2 = q2.len, l3 = q3.len;
  const double c0 = q0.cum, c1 = q1.cum, c2 = q2.cum, c3 = q3.cum;
  // The first segment (ascending) whose end lies beyond the offset: evaluated in descending order, every hit
  // overwriting the later ones — no index into the records, which the compiler would serve from a table in scratch
  // memory (it did: a store and a dependent load in front of each position).
  bool fs = false, fe = false;
  auto seg_at = [&](bool seg, double ax, double ay, double bx, double by, double al, double ac) {
    if (seg && ac + al > s_off) {
      position_at_offset(ax, ay, bx, by, al, s_off - ac, p1x, p1y);
      fs = true;
    }
    if (seg && ac + al > e_off) {
      position_at_offset(ax, ay, bx, by, al, e_off - ac, p2x, p2y);
      fe = true;
    }
  };
  seg_at(vfrom + 4 < v1, x3, y3, x4, y4, l3, c3);
  seg_at(vfrom + 3 < v1, x2, y2, x3, y3, l2, c2);
  seg_at(vfrom + 2 < v1, x1, y1, x2, y2, l1, c1);
  seg_at(vfrom + 1 < v1, x0, y0, x1, y1, l0, c0);
  if (!fs) team_position_at_shape_offset<1>(m, v0, v1, s_off, p1x, p1y, vfrom);
  if (!fe) team_position_at_shape_offset<1>(m, v0, v1, e_off, p2x, p2y, vfrom);
}

// `cand`: this lane's column of an LDS array [SMX_FACTS_CAND][stride] of record indices.
// Loads come in batches of four records (the kernel runs two wavefronts per SIMD and every dependent load is a
// microsecond under load): the rows' offsets together, then the grid records, then the survivors.
#define SMX_FACTS_SPAN 4  // grid rows the seeded visit handles (a reach of ~4 m over 8 m cells meets two, seldom three)
__device__ inline bool facts_one_lane(const MapDev& m, double px, double py, double pose_radius, int n_corners,
                                      const double* cx, const double* cy, const FactsCarry& c, double thr_max, int* cand,
                                      int stride, bool want_heading, RoadFacts& out, double& lane_heading) {
  SMX_TSTAMP(tf0);
  const double radius = fmax(pose_radius, 2.0 * m.default_lane_width);
  if (!(c.valid && c.qx == c.qx && c.qy == c.qy && c.prev_dist < radius)) return false;
  const double dist_bound = (c.prev_dist + euclid(px, py, c.qx, c.qy)) * (1.0 + 1e-12) + 1e-9;
  if (!(dist_bound < radius)) return false;
  const double reach = fmax(dist_bound, thr_max + (n_corners > 0 ? 2.0 : 0.0)) + 4e-6;
  const double road_radius = fmax(5.0, 2.0 * m.default_lane_width);  // sumo_road_network.py:705
  int cx0 = (int)floor((px - reach - m.sg_x0) / m.sg_cell), cx1 = (int)floor((px + reach - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((py - reach - m.sg_y0) / m.sg_cell), cy1 = (int)floor((py + reach - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  if (cx0 > cx1 || cy0 > cy1 || cy1 - cy0 >= SMX_FACTS_SPAN) return false;
  // ---- pass 1: records whose bounding box can matter to the centre (nearest lane within the bound, or its
  // road_with_point threshold) or to a corner (threshold + half a vehicle diagonal)
  int ra[SMX_FACTS_SPAN], rn[SMX_FACTS_SPAN];
  int total = 0;
#pragma unroll
  for (int i = 0; i < SMX_FACTS_SPAN; ++i) {
    const int y = cy0 + i;
    const bool in = y <= cy1;
    const int row = (in ? y : cy0) * m.sg_nx;
    const int va = m.sg_off[row + cx0], vb = m.sg_off[row + cx1 + 1];
    ra[i] = va;
    total += in ? vb - va : 0;
    rn[i] = total;
  }
  auto member = [&](int q) {
    int k = ra[0] + q;
#pragma unroll
    for (int i = 1; i < SMX_FACTS_SPAN; ++i)
      if (q >= rn[i - 1]) k = ra[i] + (q - rn[i - 1]);
    return k;
  };
  int n = 0;
  bool overflow = false;
  for (int q0 = 0; q0 < total; q0 += 4) {
    double x1[4], y1[4], x2[4], y2[4], thr[4];
    int kk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kk[u] = member(q0 + u < total ? q0 + u : 0);
      const smx_seg_rec* r = m.sg_rec + kk[u];
      x1[u] = r->x1;
      y1[u] = r->y1;
      x2[u] = r->x2;
      y2[u] = r->y2;
      thr[u] = r->thr;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (q0 + u < total) {
        const double gx = fmax(fmax(fmin(x1[u], x2[u]) - px, px - fmax(x1[u], x2[u])), 0.0);
        const double gy2 = fmax(fmax(fmin(y1[u], y2[u]) - py, py - fmax(y1[u], y2[u])), 0.0);
        const double keep_c = fmin(fmax(dist_bound, thr[u]), radius) + 3e-6;
        const double keep = n_corners > 0 ? fmax(keep_c, thr[u] + 2.0 + 1e-6) : keep_c;
        if (!(gx * gx + gy2 * gy2 > keep * keep)) {
          if (n < SMX_FACTS_CAND)
            cand[n * stride] = kk[u];
          else
            overflow = true;
          ++n;
        }
      }
    }
  }
  if (overflow) return false;
  SMX_TSTAMP(tf1);
  SMX_TACC(40, tf0, tf1);
  // ---- pass 2: the distances (distance_point_to_line, math.py:393-411), as team_road_facts takes them
  out.lane = -1;
  out.dist = SMX_INF;
  out.on_road = false;
  out.corner_mask = 0;
  int lane_key = 0x7fffffff;
#if defined(SMX_ABLATE)  // developer timing variants: pieces switched off (results are then wrong)
  if ((SMX_ABLATE) & (1 << 24)) {
    out.lane = n > 0 ? m.sg_rec[cand[0]].lane : -1;
    out.dist = (double)n;
    lane_heading = 0.0;
    return true;
  }
#endif
  for (int i0 = 0; i0 < n; i0 += 4) {
    double x1[4], y1[4], x2[4], y2[4], thr[4];
    int ln[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const smx_seg_rec* r = m.sg_rec + cand[(i0 + u < n ? i0 + u : i0) * stride];
      x1[u] = r->x1;
      y1[u] = r->y1;
      x2[u] = r->x2;
      y2[u] = r->y2;
      thr[u] = r->thr;
      ln[u] = r->lane;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u < n) {
        const double sx1 = x1[u], sy1 = y1[u], sx2 = x2[u], sy2 = y2[u];
        const double ex = sx1 - sx2, ey = sy1 - sy2;
        const double d = sqrt(ex * ex + ey * ey);
        const double dd = d * d;
        const double sx = sx2 - sx1, sy = sy2 - sy1;
        auto dist_to = [&](double qx, double qy) {
          const double uq = ((qx - sx1) * sx) + ((qy - sy1) * sy);
          double offset;
          if (d == 0.0 || uq < 0.0 || uq > dd) {
            offset = (uq < 0.0) ? 0.0 : d;
          } else {
            offset = uq / d;
          }
          if (offset == 0.0) {
            const double fx = qx - sx1, fy = qy - sy1;
            return sqrt(fx * fx + fy * fy);
          }
          const double uu = offset / d;
          const double ix = sx1 + uu * sx, iy = sy1 + uu * sy;
          const double fx = qx - ix, fy = qy - iy;
          return sqrt(fx * fx + fy * fy);
        };
        {
          const double dist = dist_to(px, py);
          if (dist < radius) {
            if (dist < out.dist || (dist == out.dist && ln[u] < lane_key)) {
              out.dist = dist;
              lane_key = ln[u];
            }
            if (dist < road_radius && dist < thr[u]) out.on_road = true;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // a corner that already lies on a road has nothing more to learn
          if (q < n_corners && !(out.corner_mask & (1 << q))) {
            const double dist = dist_to(q == 0 ? cx[0] : (q == 1 ? cx[1] : (q == 2 ? cx[2] : cx[3])),
                                        q == 0 ? cy[0] : (q == 1 ? cy[1] : (q == 2 ? cy[2] : cy[3])));
            if (dist < road_radius && dist < thr[u]) out.corner_mask |= (1 << q);
          }
        }
      }
    }
  }
  out.lane = lane_key == 0x7fffffff ? -1 : lane_key;
  lane_heading = 0.0;
  SMX_TSTAMP(tf2);
  SMX_TACC(41, tf1, tf2);
#if defined(SMX_ABLATE)
  if ((SMX_ABLATE) & (1 << 25)) return true;
#endif
  if (!want_heading || out.lane < 0 || m.lane_in_junction[out.lane]) return true;
  // ---- the lane heading at the nearest point of the nearest lane (team_lane_heading_at_point): its candidate
  // segments — bounding box within the lane's distance — are among the survivors of pass 1
  const int lane = out.lane;
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  const double lane_len = m.lane_length[lane];
  int vertex_hit = 0x7fffffff;
  double hit_cum = 0.0;
  double min_dist = SMX_INF, offset = -1.0, near_cum = 0.0;
  int min_v = 0x7fffffff;
  for (int i0 = 0; i0 < n; i0 += 4) {
    double x1[4], y1[4], x2[4], y2[4], sl[4], sc[4];
    int ln[4], sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const smx_seg_rec* r = m.sg_rec + cand[(i0 + u < n ? i0 + u : i0) * stride];
      x1[u] = r->x1;
      y1[u] = r->y1;
      x2[u] = r->x2;
      y2[u] = r->y2;
      sl[u] = r->len;
      sc[u] = r->cum;
      ln[u] = r->lane;
      sv[u] = r->v0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u < n && ln[u] == lane) {
        // a vertex that equals the point wins (the first such vertex): its arclength is the segment's, or the
        // segment's end (the same sum the vertex table holds: cum + len, added vertex by vertex on the host)
        if (x1[u] == px && y1[u] == py && sv[u] < vertex_hit) {
          vertex_hit = sv[u];
          hit_cum = sc[u];
        }
        if (x2[u] == px && y2[u] == py && sv[u] + 1 < vertex_hit) {
          vertex_hit = sv[u] + 1;
          hit_cum = sc[u] + sl[u];
        }
        const double gx = fmax(fmax(fmin(x1[u], x2[u]) - px, px - fmax(x1[u], x2[u])), 0.0);
        const double gy2 = fmax(fmax(fmin(y1[u], y2[u]) - py, py - fmax(y1[u], y2[u])), 0.0);
        const double keep = fmin(min_dist, out.dist) + 1e-6;
        if (!(gx * gx + gy2 * gy2 > keep * keep)) {
          const double d = sl[u];
          const double uq = ((px - x1[u]) * (x2[u] - x1[u])) + ((py - y1[u]) * (y2[u] - y1[u]));
          const double poff = (d == 0.0 || uq < 0.0 || uq > d * d) ? ((uq < 0.0) ? 0.0 : d) : uq / d;
          double fx, fy;
          position_at_offset(x1[u], y1[u], x2[u], y2[u], d, poff, fx, fy);
          const double dist = euclid(px, py, fx, fy);
          if (dist < min_dist || (dist == min_dist && sv[u] < min_v)) {
            min_dist = dist;
            offset = poff + sc[u];
            min_v = sv[u];
            near_cum = sc[u];
          }
        }
      }
    }
  }
  int v_near = v0;
  double v_near_cum = 0.0;  // cum of the lane's first vertex
  if (vertex_hit != 0x7fffffff) {
    offset = hit_cum;
    v_near = vertex_hit;
    v_near_cum = hit_cum;
  } else if (min_v != 0x7fffffff) {
    v_near = min_v;
    v_near_cum = near_cum;
  }
  double s_off, e_off;
  if (offset >= lane_len) {
    s_off = lane_len - 1.0;
    e_off = lane_len;
  } else {
    s_off = offset;
    e_off = offset + 1.0;
  }
  s_off = fmax(s_off, 0.0);
  double p1x, p1y, p2x, p2y;
  const int vfrom = (v_near_cum <= s_off) ? v_near : v0;
  SMX_TSTAMP(tf3);
  SMX_TACC(42, tf2, tf3);
  lane_positions_at_offsets(m, v0, v1, vfrom, s_off, e_off, p1x, p1y, p2x, p2y);
  SMX_TSTAMP(tf4);
  SMX_TACC(43, tf3, tf4);
  const double ang = vec_to_radians(p2x - p1x, p2y - p1y);
  const double half = ang * 0.5;
  const double qz = sin(half), qw = cos(half);
  lane_heading = wrap_heading(atan2(2.0 * (0.0 * 0.0 + qw * qz), qw * qw + 0.0 * 0.0 - 0.0 * 0.0 - qz * qz));
  SMX_TSTAMP(tf5);
  SMX_TACC(44, tf4, tf5);
  return true;
}

// ---------------------------------------------------------------------------------
// Path seeds, one lane per vehicle (large batches), without the ten-nearest list.
// closest_lanepoints (lanepoints.py:526-590) keeps, of the 10 nearest lanepoints, the one with the smallest
// d2 + |heading difference| (the nearest one always eligible, the others if within the radius).  The heading term
// is at most pi, so the winner's d2 is at most T = d2(nearest) + pi: it lies in the set C of lanepoints with
// d2 <= T.  C is a prefix of the (d2, index) order; when it has at most ten members it is a prefix of the ten
// nearest, every one of the ten outside it scores more than the nearest lanepoint does, and the winner over C —
// smallest (score, d2, index), the order the reference's first-minimum scan induces — is the reference's winner.
// With more than ten members (stacked lanes) the vehicle goes to the ten-nearest form.
// Seeded like team_nearest10_seeded: d(nearest) now <= d(nearest) then + the way driven, which bounds T before the
// visit; pass 1 visits the cells of that reach once (nearest lanepoint, nearest per lane of last tick's road,
// survivors d2 <= the a-priori T into a per-lane LDS list), pass 2 scores the survivors.
// ---------------------------------------------------------------------------------
#define SMX_SEEDS_CAND 20
struct ClosePick {  // running minimum of (score, d2, index); the winner's lane rides along
  double score, d2;
  int idx, lane;
  __device__ __forceinline__ void none() {
    score = d2 = SMX_INF;
    idx = -1;
    lane = 0;
  }
  __device__ __forceinline__ void offer(double sc, double q2, int i, int ln) {
    if (sc < score || (sc == score && (q2 < d2 || (q2 == d2 && i < idx)))) {
      score = sc;
      d2 = q2;
      idx = i;
      lane = ln;
    }
  }
};

// Returns false when the vehicle has to take the ten-nearest form (no usable carry, a list overflow, more than ten
// members in C, a reach beyond the seeded span, the closest lanepoint on a junction road, a seed road other than last
// tick's).  `d1sq`: d2 of the nearest lanepoint (for the next tick's carry).
// Loads come in batches (the kernel runs two wavefronts per SIMD, every dependent load is a microsecond under load):
// the four start records together, the rows' offsets together, the grid members eight at a time, the survivors'
// lanepoint records four at a time.
#define SMX_SEEDS_BATCH 8
__device__ inline bool seeds_one_lane(const MapDev& m, double px, double py, double heading, double within_radius,
                                      const SeedsCarry& c, int* cand, int stride, PathSeeds& s, double& d1sq) {
  if (!(c.valid && c.d1 >= 0.0 && c.d1 < 1.0e290 && c.qx == c.qx && c.qy == c.qy)) return false;
  const double b = (sqrt(c.d1) + euclid(px, py, c.qx, c.qy)) * (1.0 + 1e-12) + 1e-9;  // d(nearest) now is at most this
  const double t_pre = (b * b + SMX_PI) * (1.0 + 1e-12) + 1e-9;                       // ... and T at most this
  double reach = sqrt(t_pre) * (1.0 + 1e-12) + 1e-9;
  // the road the paths started on last tick: its lanes' nearest lanepoints ride along (a guess, checked below).  Lane
  // q's key is the lane of last tick's start lanepoint q (the nearest lanepoint ON lane q of that road).
  int kq[4] = {-9, -9, -9, -9};
  int nk = 0;
  {
    double sx[4], sy[4];
    int sl[4];
#pragma unroll
    for (int q = 0; q < SMX_SEED_LANES; ++q) {
      const smx_lp_rec* r = m.lp_rec + (c.prev_start[q] >= 0 ? c.prev_start[q] : 0);
      sx[q] = r->x;
      sy[q] = r->y;
      sl[q] = r->lane;
    }
    if (c.prev_road >= 0 && c.prev_lanes >= 1) {
      nk = min(c.prev_lanes, SMX_SEED_LANES);
      bool all = true;
#pragma unroll
      for (int q = 0; q < SMX_SEED_LANES; ++q) {
        if (q < nk) {
          if (c.prev_start[q] >= 0) {
            kq[q] = sl[q];
            reach = fmax(reach, euclid(px, py, sx[q], sy[q]) * (1.0 + 1e-12) + 1e-9);
          } else {
            all = false;
          }
        }
      }
      if (!all) nk = 0;
    }
  }
  int cx0 = (int)floor((px - reach - m.lpg_x0) / m.lpg_cell), cx1 = (int)floor((px + reach - m.lpg_x0) / m.lpg_cell);
  int cy0 = (int)floor((py - reach - m.lpg_y0) / m.lpg_cell), cy1 = (int)floor((py + reach - m.lpg_y0) / m.lpg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.lpg_nx - 1);
  cy1 = min(cy1, m.lpg_ny - 1);
  if (cx0 > cx1 || cy0 > cy1 || cy1 - cy0 >= SMX_SEEDED_SPAN || cx1 - cx0 >= SMX_SEEDED_SPAN) return false;
  // ---- pass 1
  int ra[SMX_SEEDED_SPAN], rn[SMX_SEEDED_SPAN];
  int total = 0;
#pragma unroll
  for (int i = 0; i < SMX_SEEDED_SPAN; ++i) {
    const int y = cy0 + i;
    const bool in = y <= cy1;
    const int row = (in ? y : cy0) * m.lpg_nx;
    const int va = m.lpg_off[row + cx0], vb = m.lpg_off[row + cx1 + 1];
    ra[i] = va;
    total += in ? vb - va : 0;
    rn[i] = total;
  }
  auto member = [&](int n) {
    int k = ra[0] + n;
#pragma unroll
    for (int i = 1; i < SMX_SEEDED_SPAN; ++i)
      if (n >= rn[i - 1]) k = ra[i] + (n - rn[i - 1]);
    return k;
  };
  double g2 = SMX_INF;  // nearest lanepoint: smallest (d2, index)
  int gi = 0x7fffffff;
  double bd[4] = {SMX_INF, SMX_INF, SMX_INF, SMX_INF};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  int n = 0;
  bool overflow = false;
  auto take = [&](const smx_pt_rec& p) {
    const double dx = p.x - px, dy = p.y - py;
    const double d2 = dx * dx + dy * dy;
    if (d2 < g2 || (d2 == g2 && p.idx < gi)) {
      g2 = d2;
      gi = p.idx;
    }
    if (d2 <= t_pre) {
      if (n < SMX_SEEDS_CAND)
        cand[n * stride] = p.idx;
      else
        overflow = true;
      ++n;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < nk && p.lane == kq[q] && (d2 < bd[q] || (d2 == bd[q] && p.idx < bi[q]))) {
        bd[q] = d2;
        bi[q] = p.idx;
      }
    }
  };
  for (int n0 = 0; n0 < total; n0 += SMX_SEEDS_BATCH) {
    smx_pt_rec rec[SMX_SEEDS_BATCH];
#pragma unroll
    for (int u = 0; u < SMX_SEEDS_BATCH; ++u) rec[u] = m.lpg_pts[member(n0 + u < total ? n0 + u : 0)];
#pragma unroll
    for (int u = 0; u < SMX_SEEDS_BATCH; ++u)
      if (n0 + u < total) take(rec[u]);
  }
  if (overflow || gi == 0x7fffffff) return false;
  d1sq = g2;
  // ---- pass 2: the members of C, scored (their lanepoint records: position, heading, lane)
  const double t_c = g2 + SMX_PI;
  const double r2 = within_radius * within_radius;
  ClosePick any, near;  // closest_lanepoints(pose) without a radius, and within `within_radius`
  any.none();
  near.none();
  int n_c = 0;
  for (int i0 = 0; i0 < n; i0 += 4) {
    double qx[4], qy[4], qh[4];
    int qi[4], ql[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qi[u] = cand[(i0 + u < n ? i0 + u : i0) * stride];
      const smx_lp_rec* r = m.lp_rec + qi[u];
      qx[u] = r->x;
      qy[u] = r->y;
      qh[u] = r->heading;
      ql[u] = r->lane;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u < n) {
        const double dx = qx[u] - px, dy = qy[u] - py;
        const double d2 = dx * dx + dy * dy;
        if (d2 <= t_c) {
          ++n_c;
          const double rel = fabs(heading_relative_to(heading, qh[u]));
          const double score = d2 + rel;
          any.offer(score, d2, qi[u], ql[u]);
          if (qi[u] == gi || d2 <= r2) near.offer(score, d2, qi[u], ql[u]);
        }
      }
    }
  }
  if (n_c > 10 || any.idx < 0) return false;
  // ---- team_compute_path_seeds, an agent with a (possibly empty) route object and no fixed route
  s.f.none();
  s.road = -1;
  s.n_lanes = 0;
#pragma unroll
  for (int q = 0; q < SMX_SEED_LANES; ++q) s.start[q] = -1;
  const int road_any = m.lane_road[any.lane], road_near = m.lane_road[near.lane];
  // _resolve_in_junction (sumo_road_network.py:842-860): the closest lanepoint lies on a junction road — the
  // searches by road that follow are the team form's (few vehicles are inside a junction at any time)
  if (m.road_is_junction[road_any]) return false;
  s.road = near.idx >= 0 ? road_near : -1;
  if (s.road >= 0) {
    // the start lanepoints were found on the way if this is last tick's road; a new road is the team form's
    if (s.road != c.prev_road || nk == 0) return false;
    s.n_lanes = c.prev_lanes;
#pragma unroll
    for (int q = 0; q < SMX_SEED_LANES; ++q) {
      if (q < nk && bi[q] == 0x7fffffff) return false;
      s.start[q] = q < nk ? bi[q] : -1;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------
// One-lane (serial) forms for the rare questions of the event code: offset_along_lane
// (sumo_road_network.py:476-491), from_lane_coord (:502-506, s only), vector_at_offset (road_map.py:377-388).
// Same arithmetic as the team forms above.
// ---------------------------------------------------------------------------------
__device__ inline double lane_offset_along(const MapDev& m, int lane, double px, double py) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  for (int v = v0; v < v1; ++v) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.x == px && a.y == py) return a.cum;
  }
  double min_dist = SMX_INF, offset = -1.0;
  for (int v = v0; v + 1 < v1; ++v) {
    const smx_shape_rec a = m.shape_rec[v], b = m.shape_rec[v + 1];
    const double gx = fmax(fmax(fmin(a.x, b.x) - px, px - fmax(a.x, b.x)), 0.0);
    const double gy = fmax(fmax(fmin(a.y, b.y) - py, py - fmax(a.y, b.y)), 0.0);
    const double keep = min_dist + 1e-6;
    if (gx * gx + gy * gy > keep * keep) continue;
    const double d = a.len;
    const double u = ((px - a.x) * (b.x - a.x)) + ((py - a.y) * (b.y - a.y));
    const double poff = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
    double fx, fy;
    position_at_offset(a.x, a.y, b.x, b.y, d, poff, fx, fy);
    const double dist = euclid(px, py, fx, fy);
    if (dist < min_dist) {
      min_dist = dist;
      offset = poff + a.cum;
    }
  }
  return offset;
}

__device__ inline void lane_point_at_offset(const MapDev& m, int lane, double offset, double& ox, double& oy) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  for (int v = v0; v + 1 < v1; ++v) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.cum + a.len > offset) {
      const smx_shape_rec b = m.shape_rec[v + 1];
      position_at_offset(a.x, a.y, b.x, b.y, a.len, offset - a.cum, ox, oy);
      return;
    }
  }
  const smx_shape_rec z = m.shape_rec[v1 - 1];
  ox = z.x;
  oy = z.y;
}

__device__ inline void lane_vector_at_offset(const MapDev& m, int lane, double offset, double& vx, double& vy) {
  const double L = m.lane_length[lane];
  double s_off, e_off;
  if (offset >= L) {
    s_off = L - 1.0;
    e_off = L;
  } else {
    s_off = offset;
    e_off = offset + 1.0;
  }
  s_off = fmax(s_off, 0.0);
  double p1x, p1y, p2x, p2y;
  lane_point_at_offset(m, lane, s_off, p1x, p1y);
  lane_point_at_offset(m, lane, e_off, p2x, p2y);
  vx = p2x - p1x;
  vy = p2y - p1y;
}

// Is one of Lane.oncoming_lanes_at_offset(offset) (sumo_road_network.py:371-395) on a road of the route?
// (sensors.py:566-571: the vehicle may have crossed the centre line into an oncoming lane of its route.)
// nearest_lanes(pt, radius): every lane with a centre-line segment closer than the radius, found through the
// segment grid like road_facts_scan; a lane met through several segments is simply asked again.
__device__ inline bool oncoming_lane_on_route(const MapDev& m, const RouteFilter& f, int lane, double offset) {
  const double radius = 1.1 * m.lane_width[lane];
  double ptx, pty;
  lane_point_at_offset(m, lane, offset, ptx, pty);
  double mvx, mvy;
  lane_vector_at_offset(m, lane, offset, mvx, mvy);
  const double my_norm = sqrt(mvx * mvx + mvy * mvy + 0.0);
  if (my_norm == 0.0) return false;
  const double threshold = -0.995562;  // cos(175 deg)
  int cx0 = (int)floor((ptx - radius - m.sg_x0) / m.sg_cell), cx1 = (int)floor((ptx + radius - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((pty - radius - m.sg_y0) / m.sg_cell), cy1 = (int)floor((pty + radius - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  int asked = -1;
  for (int gy = cy0; gy <= cy1; ++gy) {
    const int row = gy * m.sg_nx;
    const int a = m.sg_off[row + cx0], b = m.sg_off[row + cx1 + 1];
    for (int k = a; k < b; ++k) {
      const smx_seg_rec s = m.sg_rec[k];
      if (s.lane == lane || s.lane == asked) continue;
      if (!f.has(m, m.lane_road[s.lane])) continue;  // only a lane of the route can change the answer
      // distance_point_to_line (math.py:393-411), as in road_facts_scan
      const double ex = s.x1 - s.x2, ey = s.y1 - s.y2;
      const double d = sqrt(ex * ex + ey * ey);
      const double sx = s.x2 - s.x1, sy = s.y2 - s.y1;
      const double u = ((ptx - s.x1) * sx) + ((pty - s.y1) * sy);
      const double off = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
      double dist;
      if (off == 0.0) {
        dist = sqrt((ptx - s.x1) * (ptx - s.x1) + (pty - s.y1) * (pty - s.y1));
      } else {
        const double uu = off / d;
        const double ix = s.x1 + uu * sx, iy = s.y1 + uu * sy;
        dist = sqrt((ptx - ix) * (ptx - ix) + (pty - iy) * (pty - iy));
      }
      if (!(dist < radius)) continue;
      asked = s.lane;
      const double ls = lane_offset_along(m, s.lane, ptx, pty);
      double lvx, lvy;
      lane_vector_at_offset(m, s.lane, ls, lvx, lvy);
      const double lv_norm = sqrt(lvx * lvx + lvy * lvy + 0.0);
      if (lv_norm == 0.0) continue;
      const double lane_angle = (mvx * lvx + mvy * lvy + 0.0) / (my_norm * lv_norm);
      if (lane_angle < threshold) return true;
    }
  }
  return false;
}

// Lane.oncoming_lanes_at_offset(offset) (sumo_road_network.py:371-395) in the reference's order: the lanes
// nearest_lanes(pt, 1.1 x width) finds — sorted by centre-line distance, equal distances in lane-table order
// (the stable sort over sumolib's _allLanes order) — that run against this lane.  visit(lane) per result.
#define SMX_ONCOMING_CAP 16
template <class Visit>
__device__ inline void oncoming_lanes_at_offset(const MapDev& m, int lane, double offset, Visit&& visit) {
  const double radius = 1.1 * m.lane_width[lane];
  double ptx, pty;
  lane_point_at_offset(m, lane, offset, ptx, pty);
  double mvx, mvy;
  lane_vector_at_offset(m, lane, offset, mvx, mvy);
  const double my_norm = sqrt(mvx * mvx + mvy * mvy + 0.0);
  int cand[SMX_ONCOMING_CAP];
  double cdist[SMX_ONCOMING_CAP];
  int nc = 0;
  int cx0 = (int)floor((ptx - radius - m.sg_x0) / m.sg_cell), cx1 = (int)floor((ptx + radius - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((pty - radius - m.sg_y0) / m.sg_cell), cy1 = (int)floor((pty + radius - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  for (int gy = cy0; gy <= cy1; ++gy) {
    const int row = gy * m.sg_nx;
    const int a = m.sg_off[row + cx0], b = m.sg_off[row + cx1 + 1];
    for (int k = a; k < b; ++k) {
      const smx_seg_rec s = m.sg_rec[k];
      const double ex = s.x1 - s.x2, ey = s.y1 - s.y2;
      const double d = sqrt(ex * ex + ey * ey);
      const double sx = s.x2 - s.x1, sy = s.y2 - s.y1;
      const double u = ((ptx - s.x1) * sx) + ((pty - s.y1) * sy);
      const double off = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
      double dist;
      if (off == 0.0) {
        dist = sqrt((ptx - s.x1) * (ptx - s.x1) + (pty - s.y1) * (pty - s.y1));
      } else {
        const double uu = off / d;
        const double ix = s.x1 + uu * sx, iy = s.y1 + uu * sy;
        dist = sqrt((ptx - ix) * (ptx - ix) + (pty - iy) * (pty - iy));
      }
      if (!(dist < radius)) continue;
      int at = -1;
      for (int q = 0; q < nc; ++q)
        if (cand[q] == s.lane) at = q;
      if (at >= 0) {
        cdist[at] = fmin(cdist[at], dist);
      } else if (nc < SMX_ONCOMING_CAP) {
        cand[nc] = s.lane;
        cdist[nc] = dist;
        ++nc;
      }
    }
  }
  if (nc == 0 || my_norm == 0.0) return;
  // order: (distance, lane index)
  for (int i = 1; i < nc; ++i) {
    const int cl = cand[i];
    const double cd = cdist[i];
    int j = i - 1;
    while (j >= 0 && (cdist[j] > cd || (cdist[j] == cd && cand[j] > cl))) {
      cand[j + 1] = cand[j];
      cdist[j + 1] = cdist[j];
      --j;
    }
    cand[j + 1] = cl;
    cdist[j + 1] = cd;
  }
  const double threshold = -0.995562;  // cos(175 deg)
  for (int q = 0; q < nc; ++q) {
    const int other = cand[q];
    if (other == lane) continue;
    const double ls = lane_offset_along(m, other, ptx, pty);
    double lvx, lvy;
    lane_vector_at_offset(m, other, ls, lvx, lvy);
    const double lv_norm = sqrt(lvx * lvx + lvy * lvy + 0.0);
    if (lv_norm == 0.0) continue;
    const double lane_angle = (mvx * lvx + mvy * lvy + 0.0) / (my_norm * lv_norm);
    if (lane_angle < threshold) visit(other);
  }
}

// ---------------------------------------------------------------------------------
// Lane.center_at_point (road_map.py:357-360): from_lane_coord(offset_along_lane(point)) — the point of
// the lane's centre line closest to (px, py).  One-lane form (the via sensor asks it for a handful of
// lanes per agent); same arithmetic as team_lane_heading_at_point's first half.
// ---------------------------------------------------------------------------------
__device__ inline void lane_center_at_point(const MapDev& m, int lane, double px, double py, double& cx, double& cy) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  double offset = -1.0;
  bool on_vertex = false;
  for (int v = v0; v < v1 && !on_vertex; ++v) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.x == px && a.y == py) {
      on_vertex = true;
      offset = a.cum;
    }
  }
  if (!on_vertex) {
    double min_dist = SMX_INF;
    for (int v = v0; v + 1 < v1; ++v) {
      const smx_shape_rec a = m.shape_rec[v], b = m.shape_rec[v + 1];
      const double gx = fmax(fmax(fmin(a.x, b.x) - px, px - fmax(a.x, b.x)), 0.0);
      const double gy = fmax(fmax(fmin(a.y, b.y) - py, py - fmax(a.y, b.y)), 0.0);
      const double keep = min_dist + 1e-6;
      if (gx * gx + gy * gy > keep * keep) continue;
      const double d = a.len;
      const double u = ((px - a.x) * (b.x - a.x)) + ((py - a.y) * (b.y - a.y));
      const double poff = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
      double fx, fy;
      position_at_offset(a.x, a.y, b.x, b.y, d, poff, fx, fy);
      const double dist = euclid(px, py, fx, fy);
      if (dist < min_dist) {
        min_dist = dist;
        offset = poff + a.cum;
      }
    }
  }
  // position_at_shape_offset (utils/math.py:319-331)
  for (int v = v0; v + 1 < v1; ++v) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.cum + a.len > offset) {
      const smx_shape_rec b = m.shape_rec[v + 1];
      position_at_offset(a.x, a.y, b.x, b.y, a.len, offset - a.cum, cx, cy);
      return;
    }
  }
  const smx_shape_rec z = m.shape_rec[v1 - 1];
  cx = z.x;
  cy = z.y;
}
