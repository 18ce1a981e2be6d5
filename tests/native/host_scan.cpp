// The scan's map searches (smarts_amd/csrc/smx_scan.h) compiled for the HOST with one-lane teams, so that the
// SEEDED forms (searches that start from last tick's answers) can be held to the unseeded ones pose by pose
// without a GPU, under AddressSanitizer + UBSan (tests/test_host_scan.py).  Test infrastructure only.
#include "smx_scan.h"

extern "C" {

// facts half at (x, y, heading).  seeded != 0: start from (qx, qy, prev_dist) as scan_role does.
// out[0] = distance, out[1] = lane heading (grid form), out[2] = lane heading (vertex walk); flags: bit 0 on road,
// bits 1..4 corners.  Returns the nearest lane.
int host_scan_facts(const smx_map_tables* t, double x, double y, double heading, int seeded, double qx, double qy,
                    double prev_dist, double thr_max, double* out, int* out_flags) {
  MapDev m(*t);
  const double cxs[4] = {-0.5, 0.5, 0.5, -0.5};
  const double cys[4] = {0.5, 0.5, -0.5, -0.5};
  double cx[4], cy[4];
  const double ch = cos(heading), sh = sin(heading);
  for (int q = 0; q < 4; ++q) {
    double px = x + cxs[q] * 1.47, py = y + cys[q] * 3.68;
    cx[q] = x + ch * (px - x) + sh * (py - y);
    cy[q] = y + -sh * (px - x) + ch * (py - y);
  }
  FactsCarry fc;
  fc.valid = seeded != 0;
  fc.qx = qx;
  fc.qy = qy;
  fc.prev_dist = prev_dist;
  RoadFacts h;
  int cand[SMX_FACTS_CAND];
  double one_heading = 0.0;
  // seeded == 2: the one-lane two-pass form (out_flags bit 8 tells whether it served the vehicle)
  const bool one = seeded == 2 && facts_one_lane(m, x, y, 10.0, 4, cx, cy, fc, thr_max, cand, 1, true, h, one_heading);
  if (!one) h = team_road_facts_seeded<1>(m, x, y, 10.0, 4, cx, cy, fc, thr_max);
  out[0] = h.dist;
  out[1] = out[2] = 0.0;
  if (h.lane >= 0 && !m.lane_in_junction[h.lane]) {
    out[1] = one ? one_heading : team_lane_heading_at_point<1>(m, h.lane, x, y, h.dist);
    out[2] = team_lane_heading_at_point_linear<1>(m, h.lane, x, y, h.dist);
  }
  *out_flags = (h.on_road ? 1 : 0) | ((h.corner_mask & 15) << 1) | (one ? 256 : 0);
  return h.lane;
}

// seeds half at (x, y, heading).  carry_valid != 0: start from (qx, qy, d10, prev seeds).
// out_i: [0..9] ten nearest, [10] road, [11] filter n, [12..13] filter roads, [14] n_lanes, [15..18] starts,
// [19] the guess's road (-1: the guess was not available); out_d: [0..9] their d2.
void host_scan_seeds(const smx_map_tables* t, double x, double y, double heading, int carry_valid, double qx, double qy,
                     double d10, double d1, int prev_road, int prev_lanes, const int* prev_start, int* out_i, double* out_d) {
  MapDev m(*t);
  SeedsCarry c;
  c.valid = carry_valid != 0;
  c.qx = qx;
  c.qy = qy;
  c.d10 = d10;
  c.d1 = d1;
  c.prev_road = prev_road;
  c.prev_lanes = prev_lanes;
  for (int q = 0; q < 4; ++q) c.prev_start[q] = prev_start[q];
  Top10 top;
  LaneGuess guess;
  for (int k = 0; k < 20; ++k) out_i[k] = -7;
  for (int k = 0; k < 10; ++k) out_d[k] = 0.0;
  if (carry_valid == 2) {  // the one-lane form without the ten-nearest list; out_i[19] = 1 when it served the vehicle
    int cand[SMX_SEEDS_CAND];
    PathSeeds one;
    double d1sq = -1.0;
    if (seeds_one_lane(m, x, y, heading, 5.0, c, cand, 1, one, d1sq)) {
      out_i[10] = one.road;
      out_i[11] = one.f.n;
      out_i[12] = one.f.road[0];
      out_i[13] = one.f.road[1];
      out_i[14] = one.n_lanes;
      for (int q = 0; q < 4; ++q) out_i[15 + q] = one.start[q];
      out_i[19] = 1;
      out_d[0] = d1sq;
      return;
    }
    out_i[19] = 0;
    return;
  }
  team_nearest10_carried<1>(m, x, y, c, top, guess);
  const Top10Scores sc = team_top10_heading_terms<1>(m, top, heading);
  MissionsDev ms{nullptr, nullptr};
  const PathSeeds s = team_compute_path_seeds<1, false>(m, x, y, heading, 5.0, true, top, sc, ms, 0, &guess);
  for (int k = 0; k < 10; ++k) {
    out_i[k] = top.idx[k];
    out_d[k] = top.d2[k];
  }
  out_i[10] = s.road;
  out_i[11] = s.f.n;
  out_i[12] = s.f.road[0];
  out_i[13] = s.f.road[1];
  out_i[14] = s.n_lanes;
  for (int q = 0; q < 4; ++q) out_i[15 + q] = s.start[q];
  out_i[19] = guess.road;
}

}  // extern "C"
