// smx_kernels.hip — the per-tick kernels of the SMARTS hot path on gfx950, and the C-ABI.
//
// One thread per vehicle slot, one workgroup per group of whole environment instances, so
// that everything an instance's vehicles exchange in a tick (poses for the neighbourhood
// sensor and the collision check) goes through LDS behind one workgroup barrier.  Envs are
// independent (reference: one process per env, parallel_env.py:96-122), so there is no
// inter-workgroup communication at all.
//
// Tick order = SMARTS._step (smarts.py:236-327):
//   A controllers   (_perform_agent_actions, smarts.py:1233-1263)
//   B physics       (_step_pybullet, smarts.py:923-931)
//   C collisions    (_process_collisions, smarts.py:1270-1291)
//   D sensors       (Sensors.observe, sensors.py:238-396; events :443-594)
//   E teardown / auto-reset (smarts.py:314; parallel_env.py:303-309)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smx_vehicle.h"

#define SMX_BLOCK 64
#define SMX_COLLISION_LEEWAY 0.05  // chassis.py:75-78

struct KernelArgs {
  smx_config cfg;
  MapDev map;
  smx_state st;
  smx_spawns sp;
  smx_outputs out;
  const int8_t* actions;
  const uint8_t* env_mask;  // reset kernel only
  const double* lidar_rays;
  int envs_per_block;
  double heading_gain_pos, lateral_gain_pos;  // lateral gains for target_speed > 0
  int debug_skip;  // developer ablation mask (SMX_DEBUG_SKIP env var), 0 in production
};

// Pose block shared by the vehicles of the envs of one workgroup (LDS).
struct __align__(16) SharedPose {
  double x, y, heading, speed;
  double lane_dist;  // centre-line distance of `lane`
  int lane;          // nearest lane within SMX_POSE_SCAN_RADIUS, -1 none
  int alive;
  int on_road;       // road_with_point(centre) is not None
  int corner_mask;   // bit q: road_with_point(bounding-box corner q) is not None
};

// ---------------------------------------------------------------------------------
// oriented-box proximity (substitution for pybullet getClosestPoints)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void box_corners(double x, double y, double h, double len, double wid, double* cx,
                                            double* cy) {
  double fx = -sin(h), fy = cos(h), rx = cos(h), ry = sin(h);
  double hl = 0.5 * len, hw = 0.5 * wid;
  cx[0] = x + fx * hl + rx * hw;
  cy[0] = y + fy * hl + ry * hw;
  cx[1] = x + fx * hl - rx * hw;
  cy[1] = y + fy * hl - ry * hw;
  cx[2] = x - fx * hl - rx * hw;
  cy[2] = y - fy * hl - ry * hw;
  cx[3] = x - fx * hl + rx * hw;
  cy[3] = y - fy * hl + ry * hw;
}

__device__ __forceinline__ bool point_in_box(double px, double py, double x, double y, double h, double len,
                                             double wid) {
  double fx = -sin(h), fy = cos(h), rx = cos(h), ry = sin(h);
  double dx = px - x, dy = py - y;
  return fabs(dx * fx + dy * fy) <= 0.5 * len && fabs(dx * rx + dy * ry) <= 0.5 * wid;
}

__device__ __forceinline__ double seg_point_dist2(double px, double py, double ax, double ay, double bx, double by) {
  double dx = bx - ax, dy = by - ay;
  double ll = dx * dx + dy * dy;
  double t = (ll == 0.0) ? 0.0 : ((px - ax) * dx + (py - ay) * dy) / ll;
  t = fmin(1.0, fmax(0.0, t));
  double ex = ax + t * dx - px, ey = ay + t * dy - py;
  return ex * ex + ey * ey;
}

__device__ __noinline__ bool boxes_within(double ax, double ay, double ah, double bx, double by, double bh, double len,
                                    double wid, double leeway) {
  // broad phase: circumscribed circles
  double dx = ax - bx, dy = ay - by;
  double reach = sqrt(len * len + wid * wid) + leeway;
  if (dx * dx + dy * dy > reach * reach) return false;
  double cax[4], cay[4], cbx[4], cby[4];
  box_corners(ax, ay, ah, len, wid, cax, cay);
  box_corners(bx, by, bh, len, wid, cbx, cby);
  double best = SMX_INF;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (point_in_box(cax[i], cay[i], bx, by, bh, len, wid)) return true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      best = fmin(best, seg_point_dist2(cax[i], cay[i], cbx[k], cby[k], cbx[(k + 1) & 3], cby[(k + 1) & 3]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (point_in_box(cbx[i], cby[i], ax, ay, ah, len, wid)) return true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      best = fmin(best, seg_point_dist2(cbx[i], cby[i], cax[k], cay[k], cax[(k + 1) & 3], cay[(k + 1) & 3]));
  }
  if (best <= leeway * leeway) return true;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double a0x = cax[i], a0y = cay[i], a1x = cax[(i + 1) & 3], a1y = cay[(i + 1) & 3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double b0x = cbx[k], b0y = cby[k], b1x = cbx[(k + 1) & 3], b1y = cby[(k + 1) & 3];
      double d1 = (a1x - a0x) * (b0y - a0y) - (a1y - a0y) * (b0x - a0x);
      double d2 = (a1x - a0x) * (b1y - a0y) - (a1y - a0y) * (b1x - a0x);
      double d3 = (b1x - b0x) * (a0y - b0y) - (b1y - b0y) * (a0x - b0x);
      double d4 = (b1x - b0x) * (a1y - b0y) - (b1y - b0y) * (a1x - b0x);
      if (((d1 > 0) != (d2 > 0)) && ((d3 > 0) != (d4 > 0))) return true;
    }
  }
  return false;
}

// ---------------------------------------------------------------------------------
// state load / store (struct-of-arrays, coalesced across the wave)
// ---------------------------------------------------------------------------------
#define SF(field) a.st.f64[(size_t)(field) * total + gid]

__device__ __forceinline__ void zero_outputs(const KernelArgs& a, size_t gid) {
  const smx_config& c = a.cfg;
  const smx_outputs& o = a.out;
  for (int k = 0; k < 3; ++k) o.ego_pos[gid * 3 + k] = 0.0;
  for (int k = 0; k < SMX_EGO_F32_COUNT; ++k) o.ego_f32[gid * SMX_EGO_F32_COUNT + k] = 0.0f;
  o.ego_lane[gid * 2] = -1;
  o.ego_lane[gid * 2 + 1] = -1;
  for (int k = 0; k < SMX_EV_COUNT; ++k) o.events[gid * SMX_EV_COUNT + k] = 0;
  o.reward[gid] = 0.0;
  if (c.sensors & SMX_SENSOR_WAYPOINTS) {
    size_t per = (size_t)c.wp_paths * c.wp_len;
    for (size_t k = 0; k < per; ++k) {
      size_t q = gid * per + k;
      o.wp_pos[q * 3] = 0.0;
      o.wp_pos[q * 3 + 1] = 0.0;
      o.wp_pos[q * 3 + 2] = 0.0;
      o.wp_heading[q] = 0.0f;
      o.wp_lane_width[q] = 0.0f;
      o.wp_speed_limit[q] = 0.0f;
      o.wp_lane_index[q] = 0;
      o.wp_lane_id[q] = -1;
    }
    for (int k = 0; k <= c.wp_paths; ++k) o.wp_count[gid * (c.wp_paths + 1) + k] = 0;
  }
  if (c.sensors & SMX_SENSOR_NEIGHBORS) {
    for (int k = 0; k < c.nb_max; ++k) {
      size_t q = gid * c.nb_max + k;
      o.nb_pos[q * 3] = o.nb_pos[q * 3 + 1] = o.nb_pos[q * 3 + 2] = 0.0;
      o.nb_box[q * 3] = o.nb_box[q * 3 + 1] = o.nb_box[q * 3 + 2] = 0.0f;
      o.nb_heading[q] = 0.0f;
      o.nb_speed[q] = 0.0f;
      o.nb_lane_index[q] = 0;
      o.nb_lane_id[q] = -1;
      o.nb_slot[q] = -1;
    }
    o.nb_count[gid] = 0;
  }
}


__device__ __forceinline__ void store_seeds(const KernelArgs& a, size_t gid, size_t total, const PathSeeds& s) {
  int32_t* c = a.st.seed_cache;
  c[0 * total + gid] = s.road;
  c[1 * total + gid] = s.f.n;
  c[2 * total + gid] = s.f.n > 0 ? s.f.road[0] : -1;
  c[3 * total + gid] = s.f.n > 1 ? s.f.road[1] : -1;
  c[4 * total + gid] = s.n_lanes;
  c[5 * total + gid] = s.start[0];
  c[6 * total + gid] = s.start[1];
  c[7 * total + gid] = s.start[2];
  c[8 * total + gid] = s.start[3];
}

__device__ __forceinline__ PathSeeds load_seeds(const KernelArgs& a, size_t gid, size_t total) {
  const int32_t* c = a.st.seed_cache;
  PathSeeds s;
  s.road = c[0 * total + gid];
  s.f.n = c[1 * total + gid];
  s.f.road[0] = c[2 * total + gid];
  s.f.road[1] = c[3 * total + gid];
  s.n_lanes = c[4 * total + gid];
  s.start[0] = c[5 * total + gid];
  s.start[1] = c[6 * total + gid];
  s.start[2] = c[7 * total + gid];
  s.start[3] = c[8 * total + gid];
  return s;
}

// ---------------------------------------------------------------------------------
// lane heading at the centre-line point closest to (px, py):
//   Lane.center_pose_at_point(point).heading  (road_map.py:390-396)
//   = offset_along_lane (sumo_road_network.py:476-491, math.py:370-390)
//   + vector_at_offset (road_map.py:377-388) over from_lane_coord (math.py:333-345)
//   + Pose(fast_quaternion_from_angle(vec_to_radians(v))).heading (coordinates.py:394-403)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ bool is_close_ref(double a, double b) {
  return fabs(a - b) <= fmax(1e-09 * fmax(fabs(a), fabs(b)), 0.0);
}

__device__ __forceinline__ void position_at_offset(double x1, double y1, double x2, double y2, double offset,
                                                   double& ox, double& oy) {
  if (is_close_ref(offset, 0.0)) {
    ox = x1;
    oy = y1;
    return;
  }
  double dist = euclid(x1, y1, x2, y2);
  if (is_close_ref(dist, offset)) {
    ox = x2;
    oy = y2;
    return;
  }
  ox = x1 + (x2 - x1) * (offset / dist);
  oy = y1 + (y2 - y1) * (offset / dist);
}

__device__ inline void position_at_shape_offset(const MapDev& m, int v0, int v1, double offset, double& ox,
                                                double& oy) {
  double seen_length = 0.0;
  double cx = m.shape_x[v0], cy = m.shape_y[v0];
  for (int v = v0 + 1; v < v1; ++v) {
    double nx = m.shape_x[v], ny = m.shape_y[v];
    double next_length = euclid(cx, cy, nx, ny);
    if (seen_length + next_length > offset) {
      position_at_offset(cx, cy, nx, ny, offset - seen_length, ox, oy);
      return;
    }
    seen_length += next_length;
    cx = nx;
    cy = ny;
  }
  ox = m.shape_x[v1 - 1];
  oy = m.shape_y[v1 - 1];
}

__device__ __noinline__ double lane_heading_at_point(const MapDev& m, int lane, double px, double py) {
  const int v0 = m.lane_shape_off[SMX_BCHK(31, lane, m.n_lanes)], v1 = m.lane_shape_off[lane + 1];
  // offset_along_lane
  double offset;
  {
    bool on_vertex = false;
    double acc = 0.0, vertex_offset = 0.0;
    for (int v = v0; v < v1; ++v) {
      if (m.shape_x[v] == px && m.shape_y[v] == py) {
        on_vertex = true;
        vertex_offset = acc;
        break;
      }
      if (v + 1 < v1) acc += euclid(m.shape_x[v], m.shape_y[v], m.shape_x[v + 1], m.shape_y[v + 1]);
    }
    if (on_vertex) {
      offset = vertex_offset;
    } else {
      double seen = 0.0, min_dist = SMX_INF, min_offset = -1.0;
      for (int v = v0; v + 1 < v1; ++v) {
        double x1 = m.shape_x[v], y1 = m.shape_y[v], x2 = m.shape_x[v + 1], y2 = m.shape_y[v + 1];
        double d = euclid(x1, y1, x2, y2);
        double u = ((px - x1) * (x2 - x1)) + ((py - y1) * (y2 - y1));
        double poff = (d == 0.0 || u < 0.0 || u > d * d) ? ((u < 0.0) ? 0.0 : d) : u / d;
        double fx, fy;
        position_at_offset(x1, y1, x2, y2, poff, fx, fy);
        double dist = euclid(px, py, fx, fy);
        if (dist < min_dist) {
          min_dist = dist;
          min_offset = poff + seen;
        }
        seen += d;
      }
      offset = min_offset;
    }
  }
  // vector_at_offset
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
  position_at_shape_offset(m, v0, v1, s_off, p1x, p1y);
  position_at_shape_offset(m, v0, v1, e_off, p2x, p2y);
  double ang = vec_to_radians(p2x - p1x, p2y - p1y);
  double half = ang * 0.5;
  double qz = sin(half), qw = cos(half);
  return wrap_heading(atan2(2.0 * (0.0 * 0.0 + qw * qz), qw * qw + 0.0 * 0.0 - 0.0 * 0.0 - qz * qz));
}

// ---------------------------------------------------------------------------------
// phase D: Sensors.observe for one vehicle (sensors.py:238-396, 443-594)
// ---------------------------------------------------------------------------------
struct ObserveCtx {
  size_t gid, total;
  int slot, n_veh;
  const SharedPose* env_pose;  // the env's vehicles, [n_veh]
  bool collided;
  int steps;      // SensorState._step after this observation's increment
  int env_ticks;  // ticks since reset (elapsed_sim_time / dt)
  double prev_x, prev_y;  // position at the previous observation
  bool first;     // observation produced by a reset
  bool write_reward;
  int* knots;     // per-thread knot scratch (LDS), stride SMX_BLOCK
};

__device__ inline bool observe_vehicle(const KernelArgs& a, const ObserveCtx& k, const VehState& s, int& flags) {
  const smx_config& c = a.cfg;
  const smx_outputs& o = a.out;
  const MapDev& m = a.map;
  const size_t gid = k.gid, total = k.total;
  const double px = s.x, py = s.y;
  const double speed = vehicle_speed(s);
  double lng, lat;
  long_lat_speed(s, lng, lat);
  const SharedPose& me = k.env_pose[k.slot];

  // ---- ego lane (sensors.py:277-285): nearest lane within max(10, 2 * default lane width)
  const int ego_lane = (me.lane >= 0 && me.lane_dist < fmax(10.0, 2.0 * m.default_lane_width)) ? me.lane : -1;

  // ---- ego vehicle state (sensors.py:314-329; read-back of chassis.py:493-566)
  o.ego_pos[gid * 3 + 0] = px;
  o.ego_pos[gid * 3 + 1] = py;
  o.ego_pos[gid * 3 + 2] = SMX_BASE_HEIGHT;
  float* ef = o.ego_f32 + gid * SMX_EGO_F32_COUNT;
  ef[SMX_EGO_HEADING] = (float)wrap_heading(s.heading);
  ef[SMX_EGO_SPEED] = (float)speed;
  ef[SMX_EGO_STEERING] = (float)(-s.delta);
  ef[SMX_EGO_YAW_RATE] = (float)vec_to_radians(0.0, 0.0);  // chassis.py:552-556 on a planar body
  ef[SMX_EGO_LIN_VEL + 0] = (float)lng;
  ef[SMX_EGO_LIN_VEL + 1] = (float)lat;
  ef[SMX_EGO_LIN_VEL + 2] = 0.0f;
  ef[SMX_EGO_ANG_VEL + 0] = 0.0f;
  ef[SMX_EGO_ANG_VEL + 1] = 0.0f;
  ef[SMX_EGO_ANG_VEL + 2] = (float)s.r;
  ef[SMX_EGO_BOX + 0] = (float)SMX_CHASSIS_LENGTH;
  ef[SMX_EGO_BOX + 1] = (float)SMX_CHASSIS_WIDTH;
  ef[SMX_EGO_BOX + 2] = (float)SMX_CHASSIS_HEIGHT;
  o.ego_lane[gid * 2 + 0] = (int16_t)ego_lane;
  o.ego_lane[gid * 2 + 1] = (int16_t)(ego_lane >= 0 ? m.lane_index[ego_lane] : -1);

  // ---- accelerometer (sensors.py:1053-1084): finite differences over a 3-deep history
  {
    double la[3] = {0, 0, 0}, aa[3] = {0, 0, 0}, lj[3] = {0, 0, 0}, aj[3] = {0, 0, 0};
    if (c.sensors & SMX_SENSOR_ACCELEROMETER) {
      int hist = k.first ? 0 : ((flags >> SMX_F_HIST_SHIFT) & 3);  // samples held before this one
      double l0x = SF(SMX_S_LV0_LONG), l0y = SF(SMX_S_LV0_LAT), a0z = SF(SMX_S_AV0_Z);
      double l1x = SF(SMX_S_LV1_LONG), l1y = SF(SMX_S_LV1_LAT), a1z = SF(SMX_S_AV1_Z);
      if (hist >= 1) {
        la[0] = (lng - l0x) / c.dt;
        la[1] = (lat - l0y) / c.dt;
        aa[2] = (s.r - a0z) / c.dt;
        if (hist >= 2) {
          lj[0] = la[0] - (l0x - l1x) / c.dt;
          lj[1] = la[1] - (l0y - l1y) / c.dt;
          aj[2] = aa[2] - (a0z - a1z) / c.dt;
        }
      }
      SF(SMX_S_LV1_LONG) = l0x;
      SF(SMX_S_LV1_LAT) = l0y;
      SF(SMX_S_AV1_Z) = a0z;
      SF(SMX_S_LV0_LONG) = lng;
      SF(SMX_S_LV0_LAT) = lat;
      SF(SMX_S_AV0_Z) = s.r;
      hist = hist < 2 ? hist + 1 : 2;
      flags = (flags & ~(3 << SMX_F_HIST_SHIFT)) | (hist << SMX_F_HIST_SHIFT);
    }
    for (int q = 0; q < 3; ++q) {
      ef[SMX_EGO_LIN_ACC + q] = (float)la[q];
      ef[SMX_EGO_ANG_ACC + q] = (float)aa[q];
      ef[SMX_EGO_LIN_JERK + q] = (float)lj[q];
      ef[SMX_EGO_ANG_JERK + q] = (float)aj[q];
    }
  }

  // ---- neighbourhood (sensors.py:241-266, smarts.py:1191-1208): every other vehicle of the
  //      instance within `radius` (3-D distance), in slot order, first nb_max kept
  if ((c.sensors & SMX_SENSOR_NEIGHBORS) && !(a.debug_skip & 8)) {
    int cnt = 0;
    for (int j = 0; j < k.n_veh; ++j) {
      if (j == k.slot) continue;
      const SharedPose& p = k.env_pose[j];
      if (!p.alive) continue;
      if (c.nb_radius >= 0.0) {
        double dx = p.x - px, dy = p.y - py, dz = SMX_BASE_HEIGHT - SMX_BASE_HEIGHT;
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (!(d <= c.nb_radius)) continue;
      }
      if (cnt < c.nb_max) {
        size_t q = gid * c.nb_max + cnt;
        o.nb_pos[q * 3 + 0] = p.x;
        o.nb_pos[q * 3 + 1] = p.y;
        o.nb_pos[q * 3 + 2] = SMX_BASE_HEIGHT;
        o.nb_box[q * 3 + 0] = (float)SMX_CHASSIS_LENGTH;
        o.nb_box[q * 3 + 1] = (float)SMX_CHASSIS_WIDTH;
        o.nb_box[q * 3 + 2] = (float)SMX_CHASSIS_HEIGHT;
        o.nb_heading[q] = (float)p.heading;
        o.nb_speed[q] = (float)p.speed;
        // nearest_lane(nv.pose.point, radius=vehicle.length) (sensors.py:244-246)
        int nl = (p.lane >= 0 && p.lane_dist < SMX_CHASSIS_LENGTH) ? p.lane : -1;
        o.nb_lane_id[q] = (int16_t)nl;
        o.nb_lane_index[q] = (int8_t)(nl >= 0 ? m.lane_index[nl] : -1);
        o.nb_slot[q] = (int8_t)j;
      }
      ++cnt;
    }
    for (int q0 = cnt; q0 < c.nb_max; ++q0) {
      size_t q = gid * c.nb_max + q0;
      o.nb_pos[q * 3] = o.nb_pos[q * 3 + 1] = o.nb_pos[q * 3 + 2] = 0.0;
      o.nb_box[q * 3] = o.nb_box[q * 3 + 1] = o.nb_box[q * 3 + 2] = 0.0f;
      o.nb_heading[q] = 0.0f;
      o.nb_speed[q] = 0.0f;
      o.nb_lane_index[q] = 0;
      o.nb_lane_id[q] = -1;
      o.nb_slot[q] = -1;
    }
    o.nb_count[gid] = (uint8_t)(cnt > 255 ? 255 : cnt);
  }

  // ---- trip meter construction on a fresh vehicle (TripMeterSensor.__init__, sensors.py:885-898)
  double dist = SF(SMX_S_DIST);
  if (k.first) {
    PathSeeds seed = compute_path_seeds(m, px, py, s.heading, SMX_CHASSIS_LENGTH, false);
    flags &= ~SMX_F_TRIP_HAS_WP;
    if (seed.road >= 0 && seed.start[0] >= 0) {
      BranchState bs;
      bs.reset();
      equally_spaced_path(m, seed.f, bs, seed.start[0], 1, px, py, k.knots, SMX_BLOCK, 1,
                          [&](int, const WaypointOut& w) {
                            SF(SMX_S_TRIP_X) = w.x;
                            SF(SMX_S_TRIP_Y) = w.y;
                            SF(SMX_S_TRIP_H) = w.heading;
                            flags |= SMX_F_TRIP_HAS_WP;
                          });
    }
    dist = 0.0;
  }
  const double last_dist = dist;

  // ---- waypoint paths (sensors.py:268-275, 972-985)
  bool have_first_wp = false;
  double fwx = 0, fwy = 0, fwh = 0;
  {
    const bool wp_on = (c.sensors & SMX_SENSOR_WAYPOINTS) != 0;
    const int lookahead = wp_on ? c.wp_lookahead : 1;
    PathSeeds seed = wp_on ? compute_path_seeds(m, px, py, s.heading, 5.0, true)
                           : compute_path_seeds(m, px, py, s.heading, SMX_CHASSIS_LENGTH, false);
    // the next tick's controller asks for paths at this same pose with the agent's route
    if (wp_on) {
      store_seeds(a, gid, total, seed);
    } else {
      PathSeeds none;
      none.road = -2;  // "not cached": the controller computes its own
      none.f.n = 0;
      none.n_lanes = 0;
      none.start[0] = none.start[1] = none.start[2] = none.start[3] = -1;
      store_seeds(a, gid, total, none);
    }
    int n_paths = 0;
    const size_t per = (size_t)c.wp_paths * c.wp_len;
    if (seed.road >= 0 && !(a.debug_skip & 16)) {
      for (int li = 0; li < seed.n_lanes; ++li) {
        int start = seed_start(m, seed, li, px, py);
        if (start < 0) continue;
        BranchState bs;
        bs.reset();
        do {
          const bool keep = wp_on && n_paths < c.wp_paths;
          const int max_emit = keep ? c.wp_len : (n_paths == 0 ? 1 : 0);
          const size_t base = gid * per + (size_t)n_paths * c.wp_len;
          const bool is_first_path = (n_paths == 0);
          int n = equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, k.knots, SMX_BLOCK, max_emit,
                                      [&](int i, const WaypointOut& w) {
                                        if (is_first_path && i == 0) {
                                          have_first_wp = true;
                                          fwx = w.x;
                                          fwy = w.y;
                                          fwh = w.heading;
                                        }
                                        if (keep) {
                                          size_t q = base + i;
                                          o.wp_pos[q * 3 + 0] = w.x;
                                          o.wp_pos[q * 3 + 1] = w.y;
                                          o.wp_pos[q * 3 + 2] = 0.0;
                                          o.wp_heading[q] = (float)w.heading;
                                          o.wp_lane_width[q] = (float)w.width;
                                          o.wp_speed_limit[q] = (float)w.speed;
                                          o.wp_lane_index[q] = (int8_t)m.lane_index[SMX_BCHK(30, w.lane, m.n_lanes)];
                                          o.wp_lane_id[q] = (int16_t)w.lane;
                                        }
                                      });
          if (keep) {
            int kept = n < c.wp_len ? n : c.wp_len;
            for (int i = kept; i < c.wp_len; ++i) {
              size_t q = base + i;
              o.wp_pos[q * 3] = o.wp_pos[q * 3 + 1] = o.wp_pos[q * 3 + 2] = 0.0;
              o.wp_heading[q] = 0.0f;
              o.wp_lane_width[q] = 0.0f;
              o.wp_speed_limit[q] = 0.0f;
              o.wp_lane_index[q] = 0;
              o.wp_lane_id[q] = -1;
            }
            o.wp_count[gid * (c.wp_paths + 1) + 1 + n_paths] = (uint8_t)kept;
          }
          ++n_paths;
        } while (bs.advance());
      }
    }
    if (wp_on) {
      for (int p = n_paths; p < c.wp_paths; ++p) {
        for (int i = 0; i < c.wp_len; ++i) {
          size_t q = gid * per + (size_t)p * c.wp_len + i;
          o.wp_pos[q * 3] = o.wp_pos[q * 3 + 1] = o.wp_pos[q * 3 + 2] = 0.0;
          o.wp_heading[q] = 0.0f;
          o.wp_lane_width[q] = 0.0f;
          o.wp_speed_limit[q] = 0.0f;
          o.wp_lane_index[q] = 0;
          o.wp_lane_id[q] = -1;
        }
        o.wp_count[gid * (c.wp_paths + 1) + 1 + p] = 0;
      }
      o.wp_count[gid * (c.wp_paths + 1)] = (uint8_t)(n_paths > 255 ? 255 : n_paths);
    }
  }

  // ---- trip meter (sensors.py:900-944); reward = increment (agent_manager.py:233-234)
  if (have_first_wp) {
    if (!(flags & SMX_F_TRIP_HAS_WP)) {
      SF(SMX_S_TRIP_X) = fwx;
      SF(SMX_S_TRIP_Y) = fwy;
      SF(SMX_S_TRIP_H) = fwh;
      flags |= SMX_F_TRIP_HAS_WP;
    } else {
      double tx = SF(SMX_S_TRIP_X), ty = SF(SMX_S_TRIP_Y), th = SF(SMX_S_TRIP_H);
      double dx = fwx - tx, dy = fwy - ty;
      double nrm = sqrt(dx * dx + dy * dy);
      if (nrm > 0.5) {
        double hvx, hvy;
        radians_to_vec(th, hvx, hvy);
        double dot = hvx * dx + hvy * dy;
        double sgn = dot > 0.0 ? 1.0 : (dot < 0.0 ? -1.0 : 0.0);
        dist += sgn * nrm;
        SF(SMX_S_TRIP_X) = fwx;
        SF(SMX_S_TRIP_Y) = fwy;
        SF(SMX_S_TRIP_H) = fwh;
      }
    }
  }
  SF(SMX_S_DIST) = dist;
  o.dist[gid] = dist;
  if (k.write_reward) o.reward[gid] = dist - last_dist;

  // ---- driven path (sensors.py:842-877): running length of the last window
  bool is_not_moving = false;
  if (a.st.driven_path != nullptr) {
    double* ring = a.st.driven_path + gid * (size_t)SMX_DRIVEN_PATH_LEN;
    double sum = SF(SMX_S_PATH_SUM);
    int window_pts = (int)floor(c.not_moving_time / c.dt + 1e-9) + 1;
    if (window_pts > SMX_DRIVEN_PATH_LEN) window_pts = SMX_DRIVEN_PATH_LEN;
    const int K = window_pts - 1;  // segments in a full window
    if (k.first) {
      sum = 0.0;  // a reset records a point but no segment
    } else {
      double dx = k.prev_x - px, dy = k.prev_y - py;
      double seg = sqrt(dx * dx + dy * dy);
      int nseg = k.steps - 1;  // segments recorded so far, this one included
      ring[(nseg - 1) % SMX_DRIVEN_PATH_LEN] = seg;
      sum += seg;
      if (nseg > K) sum -= ring[(nseg - 1 - K) % SMX_DRIVEN_PATH_LEN];
    }
    SF(SMX_S_PATH_SUM) = sum;
    double elapsed = (double)k.env_ticks * c.dt;
    if (!(elapsed < c.not_moving_time)) is_not_moving = sum < c.not_moving_distance;
  }

  // ---- events + done (sensors.py:443-489)
  const bool reached_goal = false;  // EndlessGoal (plan.py:76-84)
  const bool is_off_road = !me.on_road;  // sensors.py:498-500
  const bool is_on_shoulder = (me.corner_mask & 15) != 15;  // sensors.py:502-509 (any corner off road)
  const bool reached_max = c.max_episode_steps > 0 && k.steps >= c.max_episode_steps;
  bool is_off_route, is_wrong_way;
  {
    // sensors.py:527-594 with no route roads (endless mission)
    double radius = sqrt(SMX_CHASSIS_LENGTH * SMX_CHASSIS_LENGTH + SMX_CHASSIS_WIDTH * SMX_CHASSIS_WIDTH) * 0.5 + 5.0;
    int nl = (me.lane >= 0 && me.lane_dist < radius) ? me.lane : -1;
    if (nl < 0) {
      is_off_route = true;
      is_wrong_way = false;
    } else {
      is_off_route = false;
      is_wrong_way = false;
      if (!m.lane_in_junction[nl] && !(a.debug_skip & 64)) {
        double target = lane_heading_at_point(m, nl, px, py);
        is_wrong_way = fabs(heading_relative_to(s.heading, target)) > 0.5 * SMX_PI;
      }
    }
  }
  uint8_t* ev = o.events + gid * SMX_EV_COUNT;
  ev[SMX_EV_COLLISIONS] = k.collided ? 1 : 0;
  ev[SMX_EV_OFF_ROAD] = is_off_road ? 1 : 0;
  ev[SMX_EV_OFF_ROUTE] = is_off_route ? 1 : 0;
  ev[SMX_EV_ON_SHOULDER] = is_on_shoulder ? 1 : 0;
  ev[SMX_EV_WRONG_WAY] = is_wrong_way ? 1 : 0;
  ev[SMX_EV_NOT_MOVING] = is_not_moving ? 1 : 0;
  ev[SMX_EV_REACHED_GOAL] = reached_goal ? 1 : 0;
  ev[SMX_EV_REACHED_MAX_EPISODE_STEPS] = reached_max ? 1 : 0;
  ev[SMX_EV_AGENTS_ALIVE_DONE] = 0;
  const uint32_t dc = c.done_criteria;
  return (is_off_road && (dc & SMX_DONE_OFF_ROAD)) || reached_goal || reached_max ||
         (is_on_shoulder && (dc & SMX_DONE_ON_SHOULDER)) || (k.collided && (dc & SMX_DONE_COLLISION)) ||
         (is_not_moving && (dc & SMX_DONE_NOT_MOVING)) || (is_off_route && (dc & SMX_DONE_OFF_ROUTE)) ||
         (is_wrong_way && (dc & SMX_DONE_WRONG_WAY));
}

// Pose pass: publish this vehicle's pose and its road facts to the env-mates.  One sweep of the
// segment grid answers the tick's nearest_lane queries at the vehicle centre (ego lane, neighbour
// lanes, off-route check: all "nearest lane if closer than r") and road_with_point at the centre
// and at the four bounding-box corners (Vehicle.bounding_box, vehicle.py:315-332, through
// rotate_around_point, math.py:436-444).
#define SMX_POSE_SCAN_RADIUS 10.0
__device__ __forceinline__ void publish_pose(const MapDev& m, SharedPose& p, const VehState& s, bool alive, int dbg) {
  p.x = s.x;
  p.y = s.y;
  p.heading = wrap_heading(s.heading);
  p.speed = vehicle_speed(s);
  p.alive = alive ? 1 : 0;
  p.lane = -1;
  p.lane_dist = SMX_INF;
  p.on_road = 0;
  p.corner_mask = 0;
  if (alive && !(dbg & 2)) {
    const double cxs[4] = {-0.5, 0.5, 0.5, -0.5};
    const double cys[4] = {0.5, 0.5, -0.5, -0.5};
    double cx[4], cy[4];
    const double ch = cos(s.heading), sh = sin(s.heading);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double qx = s.x + cxs[q] * SMX_CHASSIS_WIDTH;
      double qy = s.y + cys[q] * SMX_CHASSIS_LENGTH;
      cx[q] = s.x + ch * (qx - s.x) + sh * (qy - s.y);
      cy[q] = s.y + -sh * (qx - s.x) + ch * (qy - s.y);
    }
    RoadFacts h = road_facts_scan(m, s.x, s.y, fmax(SMX_POSE_SCAN_RADIUS, 2.0 * m.default_lane_width),
                                  (dbg & 32) ? 0 : 4, cx, cy);
    p.lane = h.lane;
    p.lane_dist = h.dist;
    p.on_road = h.on_road ? 1 : 0;
    p.corner_mask = (dbg & 32) ? 15 : h.corner_mask;
  }
}

// ---------------------------------------------------------------------------------
// the tick kernel.  mode 0: one SMARTS step for every env.  mode 1: reset the masked envs.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(SMX_BLOCK) smx_tick_kernel(const KernelArgs a, const int mode) {
  __shared__ SharedPose pose[SMX_BLOCK];
  __shared__ int env_new_done[SMX_BLOCK];
  __shared__ int env_need_reset[SMX_BLOCK];
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  int* knots = knot_scratch + threadIdx.x;

  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const int n_veh = c.num_vehicles;
  const int epb = a.envs_per_block;
  const int local = threadIdx.x;
  const int env_local = local / n_veh;
  const int slot = local - env_local * n_veh;
  const int env = blockIdx.x * epb + env_local;
  const bool valid = (env_local < epb) && (env < c.num_envs);
  const size_t total = (size_t)c.num_envs * n_veh;
  const size_t gid = valid ? ((size_t)env * n_veh + slot) : 0;
  const SharedPose* env_pose = pose + env_local * n_veh;

  env_new_done[local] = 0;
  env_need_reset[local] = 0;

  VehState s = {0, 0, 0, 0, 0, 0, 0};
  int flags = 0, steps = 0, env_ticks = 0;
  bool alive = false;
  if (valid) {
    flags = a.st.flags[gid];
    steps = a.st.steps[gid];
    env_ticks = a.st.env_ticks[env];
    alive = (flags & SMX_F_ALIVE) != 0;
    s.x = SF(SMX_S_X);
    s.y = SF(SMX_S_Y);
    s.heading = SF(SMX_S_HEADING);
    s.u = SF(SMX_S_U);
    s.v = SF(SMX_S_V);
    s.r = SF(SMX_S_R);
    s.delta = SF(SMX_S_DELTA);
  }
  __syncthreads();

  if (mode == 0) {
    // ================= A + B: controllers, physics =================
    const double prev_x = s.x, prev_y = s.y;
    if (valid && alive) {
      CtrlState cs;
      cs.lat_int = SF(SMX_S_LAT_INT);
      cs.spd_int = SF(SMX_S_SPD_INT);
      cs.steer = SF(SMX_S_STEER);
      cs.throttle = SF(SMX_S_THROTTLE);
      cs.spd_err = SF(SMX_S_SPD_ERR);
      cs.mcl_x = SF(SMX_S_MCL_X);
      cs.mcl_y = SF(SMX_S_MCL_Y);
      cs.mcl_set = (flags & SMX_F_MCL_SET) != 0;
      const int action = a.actions[gid];
      ControlOut co;
      if (action >= 0 && !(a.debug_skip & 1)) {
        // Controllers.perform_action, Lane space (controllers/__init__.py:125-144)
        double target_speed = action == SMX_ACTION_KEEP_LANE ? 15.0 : (action == SMX_ACTION_SLOW_DOWN ? 0.0 : 12.5);
        int lane_change = action == SMX_ACTION_CHANGE_LANE_LEFT ? 1 : (action == SMX_ACTION_CHANGE_LANE_RIGHT ? -1 : 0);
        double hg = target_speed > 0.0 ? a.heading_gain_pos : 0.01;
        double lg = target_speed > 0.0 ? a.lateral_gain_pos : 0.36;
        PathSeeds seed = load_seeds(a, gid, total);
        if (seed.road == -2) seed = compute_path_seeds(m, s.x, s.y, s.heading, 5.0, true);
        co = lane_following_control(m, s, cs, c.dt, target_speed, lane_change, hg, lg, seed, knots, SMX_BLOCK);
      } else {
        // no action this tick: wheel torques do not persist, the steer motor target does
        co.throttle = 0.0;
        co.brake = 0.0;
        co.steering = cs.steer;
      }
      vehicle_step(s, co, c.dt);
      SF(SMX_S_LAT_INT) = cs.lat_int;
      SF(SMX_S_SPD_INT) = cs.spd_int;
      SF(SMX_S_STEER) = cs.steer;
      SF(SMX_S_THROTTLE) = cs.throttle;
      SF(SMX_S_SPD_ERR) = cs.spd_err;
      SF(SMX_S_MCL_X) = cs.mcl_x;
      SF(SMX_S_MCL_Y) = cs.mcl_y;
      flags = cs.mcl_set ? (flags | SMX_F_MCL_SET) : (flags & ~SMX_F_MCL_SET);
    }
    if (valid) ++env_ticks;  // smarts.py:261-262 (every thread of the env keeps the same copy)
    publish_pose(m, pose[local], s, valid && alive, a.debug_skip);
    __syncthreads();

    // ================= C: collisions =================
    bool collided = false;
    if (valid && alive && !(a.debug_skip & 4)) {
      for (int j = 0; j < n_veh; ++j) {
        if (j == slot) continue;
        const SharedPose& p = env_pose[j];
        if (!p.alive) continue;
        if (boxes_within(s.x, s.y, wrap_heading(s.heading), p.x, p.y, p.heading, SMX_CHASSIS_LENGTH,
                         SMX_CHASSIS_WIDTH, SMX_COLLISION_LEEWAY))
          collided = true;
      }
    }

    // ================= D: sensors =================
    bool done = false;
    if (valid) {
      if (alive) {
        ++steps;
        ObserveCtx k;
        k.gid = gid;
        k.total = total;
        k.slot = slot;
        k.n_veh = n_veh;
        k.env_pose = env_pose;
        k.collided = collided;
        k.steps = steps;
        k.env_ticks = env_ticks;
        k.prev_x = prev_x;
        k.prev_y = prev_y;
        k.first = false;
        k.write_reward = true;
        k.knots = knots;
        done = observe_vehicle(a, k, s, flags);
        // ================= E: teardown (smarts.py:314, 329-363) =================
        if (done) {
          flags &= ~SMX_F_ALIVE;
          atomicAdd(&env_new_done[env_local], 1);
        }
        SF(SMX_S_X) = s.x;
        SF(SMX_S_Y) = s.y;
        SF(SMX_S_HEADING) = s.heading;
        SF(SMX_S_U) = s.u;
        SF(SMX_S_V) = s.v;
        SF(SMX_S_R) = s.r;
        SF(SMX_S_DELTA) = s.delta;
        a.st.steps[gid] = steps;
        a.st.flags[gid] = flags;
        a.out.done[gid] = done ? 1 : 0;
        a.out.active[gid] = done ? 0 : 1;
      } else {
        zero_outputs(a, gid);
        a.out.dist[gid] = 0.0;
        a.out.done[gid] = 0;
        a.out.active[gid] = 0;
      }
    }
    __syncthreads();
    if (valid && slot == 0) {
      int dc = a.st.env_done_count[env] + env_new_done[env_local];
      a.st.env_done_count[env] = dc;
      a.st.env_ticks[env] = env_ticks;
      bool all_done = dc >= n_veh;  // hiway_env.py:258-261
      a.out.env_done[env] = all_done ? 1 : 0;
      env_need_reset[env_local] = (all_done && c.auto_reset) ? 1 : 0;
    }
  } else {
    if (valid && slot == 0) env_need_reset[env_local] = (a.env_mask == nullptr || a.env_mask[env]) ? 1 : 0;
  }
  __syncthreads();

  // ================= reset (SMARTS.reset, smarts.py:365-460; ParallelEnv auto-reset) =================
  const bool do_reset = valid && env_need_reset[env_local] != 0;
  const int episode = do_reset ? a.st.env_episode[env] + 1 : 0;  // every reset starts the next spawn row
  if (do_reset) {
    const int row = a.sp.episodes > 0 ? (((episode % a.sp.episodes) + a.sp.episodes) % a.sp.episodes) : 0;
    const double* sp = a.sp.pose + ((size_t)row * total + gid) * 4;
    s.x = sp[0];
    s.y = sp[1];
    s.heading = wrap_heading(sp[2]);
    s.u = sp[3];  // AckermannChassis._initialize_speed (chassis.py:668-671)
    s.v = 0.0;
    s.r = 0.0;
    s.delta = 0.0;
    flags = SMX_F_ALIVE;
    steps = 1;  // SensorState.step runs in the tick that creates the vehicle (agent_manager.py:250-258)
    env_ticks = c.reset_elapsed_steps;
    for (int f = SMX_S_LAT_INT; f < SMX_S_COUNT; ++f) SF(f) = 0.0;
  }
  publish_pose(m, pose[local], s, do_reset, a.debug_skip);
  __syncthreads();
  if (do_reset) {
    ObserveCtx k;
    k.gid = gid;
    k.total = total;
    k.slot = slot;
    k.n_veh = n_veh;
    k.env_pose = env_pose;
    k.collided = false;
    k.steps = steps;
    k.env_ticks = env_ticks;
    k.prev_x = s.x;
    k.prev_y = s.y;
    k.first = true;
    k.write_reward = (mode != 0);
    k.knots = knots;
    observe_vehicle(a, k, s, flags);
    SF(SMX_S_X) = s.x;
    SF(SMX_S_Y) = s.y;
    SF(SMX_S_HEADING) = s.heading;
    SF(SMX_S_U) = s.u;
    SF(SMX_S_V) = s.v;
    SF(SMX_S_R) = s.r;
    SF(SMX_S_DELTA) = s.delta;
    a.st.steps[gid] = steps;
    a.st.flags[gid] = flags;
    a.out.active[gid] = 1;
    if (mode != 0) {
      a.out.done[gid] = 0;
      a.out.reward[gid] = 0.0;
    }
    if (slot == 0) {
      a.st.env_episode[env] = episode;
      a.st.env_done_count[env] = 0;
      a.st.env_ticks[env] = env_ticks;
      if (mode != 0) a.out.env_done[env] = 0;
    }
  }
}

// =================================================================================
// C-ABI (include/smx.h)
// =================================================================================
struct smx_handle_s {
  smx_config cfg;
  int device;
  bool map_loaded;
  MapDev map;
  void* map_blob;  // one device allocation holding every table
  size_t map_bytes;
  const double* lidar_rays;
  double heading_gain_pos, lateral_gain_pos;
  int debug_skip;
  bool timing;
  std::vector<hipEvent_t> ev_pool;  // pairs: [2*i] start, [2*i+1] stop
  size_t ev_used;                   // pairs recorded since the last read
  std::string err;
};

static int fail(smx_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define SMX_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) return fail(h, SMX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

extern "C" const char* smx_version(void) { return "smarts-mi355x 0.1 (gfx950)"; }

#ifdef SMX_DEBUG_BOUNDS
extern "C" int smx_debug_read(int* site, long long* value) {
  if (hipMemcpyFromSymbol(site, HIP_SYMBOL(smx_dbg_site), sizeof(int)) != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(value, HIP_SYMBOL(smx_dbg_value), sizeof(long long)) != hipSuccess) return -2;
  int aux[8];
  if (hipMemcpyFromSymbol(aux, HIP_SYMBOL(smx_dbg_aux), sizeof(aux)) != hipSuccess) return -2;
  printf("dbg aux: first=%d remaining=%d hops=%d n_next=%d lane=%d next0=%d cur_idx=%d mem_next0=%d\n", aux[0], aux[1], aux[2], aux[3], aux[4], aux[5], aux[6], aux[7]);
  return 0;
}
#endif

extern "C" uint64_t smx_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(smx_config);
    case 1: return sizeof(smx_map_tables);
    case 2: return sizeof(smx_state);
    case 3: return sizeof(smx_spawns);
    case 4: return sizeof(smx_outputs);
    default: return 0;
  }
}

extern "C" int smx_create(const smx_config* cfg, int device, smx_handle* out) {
  if (!cfg || !out) return SMX_ERR_INVALID;
  *out = nullptr;
  smx_handle h = new (std::nothrow) smx_handle_s();
  if (!h) return SMX_ERR_NOMEM;
  h->cfg = *cfg;
  h->device = device;
  h->map_loaded = false;
  h->map_blob = nullptr;
  h->map_bytes = 0;
  h->lidar_rays = nullptr;
  // lane_following_controller.py:426-430: place_poles gains clipped to [0.02, 0.04] / [3.4, 4.1];
  // for the sedan they saturate at (0.04, 3.4) for both Lane-space target speeds.
  h->heading_gain_pos = 0.04;
  h->lateral_gain_pos = 3.4;
  h->timing = false;
  h->ev_used = 0;
  {
    const char* dbg = getenv("SMX_DEBUG_SKIP");
    h->debug_skip = dbg ? atoi(dbg) : 0;
  }
  *out = h;
  const smx_config& c = h->cfg;
  if (c.num_envs <= 0 || c.num_vehicles <= 0 || c.num_vehicles > SMX_BLOCK)
    return fail(h, SMX_ERR_INVALID, "num_envs must be > 0 and 0 < num_vehicles <= 64");
  if (!(c.dt > 0.0)) return fail(h, SMX_ERR_INVALID, "dt must be > 0");
  if ((c.sensors & SMX_SENSOR_WAYPOINTS) &&
      (c.wp_lookahead < 1 || c.wp_lookahead > SMX_MAX_KNOTS - 2 || c.wp_paths < 1 || c.wp_paths > 64 || c.wp_len < 1 || c.wp_len > c.wp_lookahead + 1))
    return fail(h, SMX_ERR_INVALID, "waypoints: need lookahead >= 1, 1 <= wp_paths <= 64, 1 <= wp_len <= lookahead + 1");
  if ((c.sensors & SMX_SENSOR_NEIGHBORS) && (c.nb_max < 1 || c.nb_max > 127))
    return fail(h, SMX_ERR_INVALID, "neighbours: need 1 <= nb_max <= 127");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(h, SMX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return SMX_OK;
}

extern "C" int smx_set_controller_gains(smx_handle h, double heading_gain, double lateral_gain) {
  if (!h) return SMX_ERR_INVALID;
  h->heading_gain_pos = heading_gain;
  h->lateral_gain_pos = lateral_gain;
  return SMX_OK;
}

namespace {
struct BlobWriter {
  std::string host;
  size_t add(const void* p, size_t bytes) {
    size_t off = (host.size() + 255) & ~size_t(255);
    host.resize(off + bytes);
    if (bytes) memcpy(&host[off], p, bytes);
    return off;
  }
};
}  // namespace

extern "C" int smx_load_map(smx_handle h, const smx_map_tables* t) {
  if (!h || !t) return SMX_ERR_INVALID;
  if (t->n_lanes <= 0 || t->n_roads <= 0 || t->n_lanepoints <= 0) return fail(h, SMX_ERR_INVALID, "empty map tables");
  if (t->n_lanes > 32767) return fail(h, SMX_ERR_INVALID, "lane ids are reported as int16: at most 32767 lanes");
  SMX_HIP(hipSetDevice(h->device));
  BlobWriter w;
  const size_t nl = t->n_lanes, nr = t->n_roads, np_ = t->n_lanepoints, nv = t->n_shape_pts;
  const size_t lpg_cells = (size_t)t->lpg_nx * t->lpg_ny, sg_cells = (size_t)t->sg_nx * t->sg_ny;
  // every record index stored in the tables is range-checked here, once, so that the kernels can
  // follow them without bounds tests
  for (size_t i = 0; i < np_; ++i) {
    const smx_lp_rec& r = t->lp_rec[i];
    if (r.lane < 0 || r.lane >= t->n_lanes || r.next0 >= t->n_lanepoints || r.knot_next >= t->n_lanepoints ||
        (r.n_next > 0 && (r.next_off < 0 || r.next_off + r.n_next > t->n_succ || r.next0 < 0 || r.knot_next < 0)))
      return fail(h, SMX_ERR_INVALID, "lanepoint record out of range");
  }
  for (int i = 0; i < t->n_succ; ++i) {
    const smx_succ_rec& r = t->succ_rec[i];
    if (r.idx < 0 || r.idx >= t->n_lanepoints || r.knot < 0 || r.knot >= t->n_lanepoints || r.lane < 0 ||
        r.lane >= t->n_lanes || r.hops < 1)
      return fail(h, SMX_ERR_INVALID, "successor record out of range");
  }
#define ADD(field, count, type) size_t off_##field = w.add(t->field, (size_t)(count) * sizeof(type))
  ADD(lane_road, nl, int32_t);
  ADD(lane_index, nl, int32_t);
  ADD(lane_width, nl, double);
  ADD(lane_speed, nl, double);
  ADD(lane_length, nl, double);
  ADD(lane_in_junction, nl, uint8_t);
  ADD(lane_shape_off, nl + 1, int32_t);
  ADD(shape_x, nv, double);
  ADD(shape_y, nv, double);
  ADD(lane_out_off, nl + 1, int32_t);
  ADD(lane_out_idx, t->lane_out_off[nl], int32_t);
  ADD(road_lane_off, nr + 1, int32_t);
  ADD(road_lanes, t->road_lane_off[nr], int32_t);
  ADD(road_is_junction, nr, uint8_t);
  ADD(road_out_road, nr, int32_t);
  ADD(lp_rec, np_, smx_lp_rec);
  ADD(succ_rec, t->n_succ, smx_succ_rec);
  ADD(lpg_off, lpg_cells + 1, int32_t);
  ADD(lpg_pts, t->lpg_off[lpg_cells], smx_pt_rec);
  ADD(sg_off, sg_cells + 1, int32_t);
  ADD(sg_rec, t->sg_off[sg_cells], smx_seg_rec);
#undef ADD
  if (h->map_blob) {
    (void)hipFree(h->map_blob);
    h->map_blob = nullptr;
  }
  SMX_HIP(hipMalloc(&h->map_blob, w.host.size()));
  SMX_HIP(hipMemcpy(h->map_blob, w.host.data(), w.host.size(), hipMemcpyHostToDevice));
  h->map_bytes = w.host.size();
  char* base = (char*)h->map_blob;
  MapDev& m = h->map;
  m = *t;  // scalars; every pointer is re-pointed into the device blob below
#define PTR(field, type) m.field = (const type*)(base + off_##field)
  PTR(lane_road, int32_t);
  PTR(lane_index, int32_t);
  PTR(lane_width, double);
  PTR(lane_speed, double);
  PTR(lane_length, double);
  PTR(lane_in_junction, uint8_t);
  PTR(lane_shape_off, int32_t);
  PTR(shape_x, double);
  PTR(shape_y, double);
  PTR(lane_out_off, int32_t);
  PTR(lane_out_idx, int32_t);
  PTR(road_lane_off, int32_t);
  PTR(road_lanes, int32_t);
  PTR(road_is_junction, uint8_t);
  PTR(road_out_road, int32_t);
  PTR(lp_rec, smx_lp_rec);
  PTR(succ_rec, smx_succ_rec);
  PTR(lpg_off, int32_t);
  PTR(lpg_pts, smx_pt_rec);
  PTR(sg_off, int32_t);
  PTR(sg_rec, smx_seg_rec);
#undef PTR
  h->map_loaded = true;
  return SMX_OK;
}

extern "C" int smx_set_lidar_rays(smx_handle h, const double* rays_dev, int32_t n_rays) {
  if (!h) return SMX_ERR_INVALID;
  if (n_rays != h->cfg.lidar_rays) return fail(h, SMX_ERR_INVALID, "n_rays != cfg.lidar_rays");
  h->lidar_rays = rays_dev;
  return SMX_OK;
}

static int check_buffers(smx_handle h, const smx_state* st, const smx_spawns* sp, const smx_outputs* o) {
  if (!st || !sp || !o) return fail(h, SMX_ERR_INVALID, "null state / spawns / outputs");
  if (!st->f64 || !st->flags || !st->steps || !st->env_ticks || !st->env_done_count || !st->env_episode ||
      !st->seed_cache)
    return fail(h, SMX_ERR_INVALID, "null state buffer");
  if (!sp->pose || sp->episodes < 1) return fail(h, SMX_ERR_INVALID, "spawn table is empty");
  if (!o->ego_pos || !o->ego_f32 || !o->ego_lane || !o->events || !o->reward || !o->dist || !o->done || !o->active ||
      !o->env_done)
    return fail(h, SMX_ERR_INVALID, "null output buffer");
  const smx_config& c = h->cfg;
  if ((c.sensors & SMX_SENSOR_WAYPOINTS) && (!o->wp_pos || !o->wp_heading || !o->wp_lane_width || !o->wp_speed_limit ||
                                             !o->wp_lane_index || !o->wp_lane_id || !o->wp_count))
    return fail(h, SMX_ERR_INVALID, "waypoints sensor enabled but an output buffer is null");
  if ((c.sensors & SMX_SENSOR_NEIGHBORS) && (!o->nb_pos || !o->nb_box || !o->nb_heading || !o->nb_speed ||
                                             !o->nb_lane_index || !o->nb_lane_id || !o->nb_slot || !o->nb_count))
    return fail(h, SMX_ERR_INVALID, "neighbourhood sensor enabled but an output buffer is null");
  if ((c.done_criteria & SMX_DONE_NOT_MOVING) && !st->driven_path)
    return fail(h, SMX_ERR_INVALID, "not_moving done criterion needs the driven_path ring");
  return SMX_OK;
}

static int launch(smx_handle h, int mode, const int8_t* actions, const uint8_t* mask, const smx_state* st,
                  const smx_spawns* sp, const smx_outputs* out, void* stream_) {
  if (!h) return SMX_ERR_INVALID;
  if (!h->map_loaded) return fail(h, SMX_ERR_STATE, "smx_load_map has not been called");
  int rc = check_buffers(h, st, sp, out);
  if (rc != SMX_OK) return rc;
  if (mode == 0 && !actions) return fail(h, SMX_ERR_INVALID, "null actions");
  hipStream_t stream = (hipStream_t)stream_;
  KernelArgs a;
  a.cfg = h->cfg;
  a.map = h->map;
  a.st = *st;
  a.sp = *sp;
  a.out = *out;
  a.actions = actions;
  a.env_mask = mask;
  a.lidar_rays = h->lidar_rays;
  a.envs_per_block = SMX_BLOCK / h->cfg.num_vehicles;
  a.heading_gain_pos = h->heading_gain_pos;
  a.lateral_gain_pos = h->lateral_gain_pos;
  a.debug_skip = h->debug_skip;
  const int blocks = (h->cfg.num_envs + a.envs_per_block - 1) / a.envs_per_block;
  const bool timed = h->timing && mode == 0 && h->ev_used < 65536;
  if (timed) {
    if (h->ev_pool.size() < 2 * (h->ev_used + 1)) {
      hipEvent_t e0, e1;
      SMX_HIP(hipEventCreate(&e0));
      SMX_HIP(hipEventCreate(&e1));
      h->ev_pool.push_back(e0);
      h->ev_pool.push_back(e1);
    }
    SMX_HIP(hipEventRecord(h->ev_pool[2 * h->ev_used], stream));
  }
  hipLaunchKernelGGL(smx_tick_kernel, dim3(blocks), dim3(SMX_BLOCK), 0, stream, a, mode);
  SMX_HIP(hipGetLastError());
  if (timed) {
    SMX_HIP(hipEventRecord(h->ev_pool[2 * h->ev_used + 1], stream));
    h->ev_used += 1;
  }
  return SMX_OK;
}

extern "C" int smx_reset(smx_handle h, const uint8_t* env_mask_dev, const smx_state* st, const smx_spawns* sp,
                         const smx_outputs* out, void* hip_stream) {
  return launch(h, 1, nullptr, env_mask_dev, st, sp, out, hip_stream);
}

extern "C" int smx_step(smx_handle h, const int8_t* actions_dev, const smx_state* st, const smx_spawns* sp,
                        const smx_outputs* out, void* hip_stream) {
  return launch(h, 0, actions_dev, nullptr, st, sp, out, hip_stream);
}

extern "C" int smx_sync(smx_handle h, void* hip_stream) {
  if (!h) return SMX_ERR_INVALID;
  SMX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return SMX_OK;
}

extern "C" int smx_set_timing(smx_handle h, int enabled) {
  if (!h) return SMX_ERR_INVALID;
  h->timing = enabled != 0;
  return SMX_OK;
}

extern "C" int smx_read_step_ms(smx_handle h, float* ms, int32_t max, int32_t* n) {
  if (!h || !ms || !n || max < 0) return SMX_ERR_INVALID;
  int32_t count = 0;
  for (size_t i = 0; i < h->ev_used; ++i) {
    SMX_HIP(hipEventSynchronize(h->ev_pool[2 * i + 1]));
    float t = 0.f;
    SMX_HIP(hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
    if (count < max) ms[count++] = t;
  }
  h->ev_used = 0;
  *n = count;
  return SMX_OK;
}

extern "C" int smx_last_step_ms(smx_handle h, float* ms) {
  if (!h || !ms) return SMX_ERR_INVALID;
  if (h->ev_used == 0) return fail(h, SMX_ERR_STATE, "no timed step recorded (smx_set_timing(1) then smx_step)");
  const size_t i = h->ev_used - 1;
  SMX_HIP(hipEventSynchronize(h->ev_pool[2 * i + 1]));
  SMX_HIP(hipEventElapsedTime(ms, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
  return SMX_OK;
}

extern "C" const char* smx_last_error(smx_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void smx_destroy(smx_handle h) {
  if (!h) return;
  if (h->map_blob) (void)hipFree(h->map_blob);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  delete h;
}
