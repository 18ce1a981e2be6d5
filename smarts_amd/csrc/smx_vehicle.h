// smx_vehicle.h — per-vehicle control + dynamics on the device.
//   Controllers.perform_action (Lane space)          controllers/__init__.py:125-144
//   LaneFollowingController.perform_lane_following   lane_following_controller.py:63-365
//   TrajectoryTrackingController.curvature_calculation trajectory_tracking_controller.py:444-473
//   AckermannChassis.control / _apply_*              chassis.py:678-794
//   SMARTS._step_pybullet substep schedule           smarts.py:602-613, 923-931
// The rigid-body solve itself (pybullet, not in the reference tree) is the planar single-track
// model documented in DESIGN.md "Substitutions"; constants come from models/vehicle.urdf,
// models/controller_parameters.yaml and models/plane.urdf.
#pragma once
#include "smx_roadmap.h"

// ---- models/controller_parameters.yaml:17-25 ("sedan") ----
#define SMX_WHEEL_RADIUS 0.31265
#define SMX_MAX_TORQUE 1600.0
#define SMX_MAX_BTORQUE 1400.0
#define SMX_MAX_STEERING 12.56
#define SMX_STEERING_GEAR_RATIO 17.4
// ---- models/vehicle.urdf ----
#define SMX_TOTAL_MASS (2356.0 + 4 * 15.0 + 2 * 1.0 + 1.0)
#define SMX_AXLE_DIST 1.5
#define SMX_TRACK_HALF 0.5
#define SMX_TOTAL_INERTIA_Z (2681.95008628 + (4 * 15.0 + 2 * 1.0) * (1.5 * 1.5 + 0.5 * 0.5))
#define SMX_WHEELBASE 3.0
#define SMX_STEER_LIMIT 0.8727
#define SMX_CHASSIS_LENGTH 3.68
#define SMX_CHASSIS_WIDTH 1.47
#define SMX_CHASSIS_HEIGHT 1.0
#define SMX_BASE_HEIGHT (0.31265 - 0.3)
// ---- models/plane.urdf, smarts.py:67,615 ----
#define SMX_CORNERING_STIFFNESS 100000.0
#define SMX_GROUND_FRICTION 3.0
#define SMX_GRAVITY 9.8
#define SMX_MAX_PYBULLET_FREQ 240
#define SMX_STEER_POSITION_GAIN 0.1
#define SMX_KINEMATIC_BELOW_SPEED 2.0

struct VehState {
  double x, y, heading, u, v, r, delta;
};

struct CtrlState {
  double lat_int, spd_int, steer, throttle, spd_err, mcl_x, mcl_y;
  bool mcl_set;
};

// sin / cos of the heading, evaluated once per use site
struct HeadingTrig {
  double sh, ch;
};
__device__ __forceinline__ HeadingTrig heading_trig(double heading) {
  HeadingTrig t;
  sincos(heading, &t.sh, &t.ch);
  return t;
}

// chassis.py:558-566 (body-frame speeds recovered from the world velocity, as the reference does)
__device__ __forceinline__ void world_velocity(const VehState& s, const HeadingTrig& t, double& vx, double& vy) {
  // forward = (-sin h, cos h); left = (-cos h, -sin h)
  vx = s.u * (-t.sh) + s.v * (-t.ch);
  vy = s.u * t.ch + s.v * (-t.sh);
}

__device__ __forceinline__ void long_lat_speed(const VehState& s, const HeadingTrig& t, double& lng, double& lat) {
  double vx, vy;
  world_velocity(s, t, vx, vy);
  lng = vy * t.ch - vx * t.sh;
  lat = vy * t.sh + vx * t.ch;
}

__device__ __forceinline__ double vehicle_speed(const VehState& s, const HeadingTrig& t) {
  double vx, vy;
  world_velocity(s, t, vx, vy);
  return sqrt(vx * vx + vy * vy + 0.0 * 0.0);
}

// curvature_calculation on the first waypoints of the chosen path, kept in registers.
struct Traj10 {
  double x[10], y[10], h[10];
  int n;
};

__device__ __forceinline__ double curvature_calculation(const Traj10& t, int offset) {
  const int num_points = 5;
  if (t.n <= num_points + offset) return 1e20;
  double hs = 0.0, ds = 0.0;
#pragma unroll
  for (int i = 0; i < num_points; ++i) {
    // static indexing after unroll: offset is 0 or 4
    double h1, h0, x0, x1, y0, y1;
    if (offset == 0) {
      h1 = t.h[i + 1];
      h0 = t.h[i];
      x0 = t.x[i];
      x1 = t.x[i + 1];
      y0 = t.y[i];
      y1 = t.y[i + 1];
    } else {
      h1 = t.h[i + 5];
      h0 = t.h[i + 4];
      x0 = t.x[i + 4];
      x1 = t.x[i + 5];
      y0 = t.y[i + 4];
      y1 = t.y[i + 5];
    }
    hs += min_angles_difference_signed(h1, h0);
    double ex = x0 - x1, ey = y0 - y1;
    ds += fabs(sqrt(ex * ex + ey * ey));
  }
  if (hs == 0.0) return 1e20;
  return ds / hs;
}


// LaneFollowingController.calculate_lateral_gains (lane_following_controller.py:376-437) for an
// arbitrary target speed (ActionSpaceType.LaneWithContinuousSpeed).  The reference places the poles
// (-35, -15, -2, -3) of the linearised lateral dynamics with scipy.signal.place_poles; for this
// single-input system the gain row is unique and equals Ackermann's formula
// K = e4^T C^-1 phi(A) (agreement with scipy 2e-12 relative, tests/test_host_logic.py), after which
// both gains are clipped.  The lateral gain is 0.587 at every speed (clipped to 3.4); the heading
// gain grows linearly and crosses the clip window [0.02, 0.04] between 2.02 and 2.06 m/s.
__device__ inline void lateral_gains_for_speed(double v, double& heading_gain, double& lateral_gain) {
  if (!(v > 0.0)) {
    heading_gain = 0.01;
    lateral_gain = 0.36;
    return;
  }
  // vehicle.chassis.mass_and_inertia is the chassis link's (chassis.py:575-582, models/vehicle.urdf)
  const double L = 0.5 * SMX_CHASSIS_LENGTH, M = 2356.0, IZ = 2681.95008628, C = SMX_CORNERING_STIFFNESS;
  double A[4][4] = {{0.0, v, 0.0, v},
                    {0.0, 0.0, 1.0, 0.0},
                    {0.0, 0.0, -(2.0 * C * (L * L)) / (v * IZ), 0.0},
                    {0.0, 0.0, -1.0, -2.0 * C / (M * v)}};
  const double B[4] = {0.0, 0.0, L * C / IZ, C / (M * v)};
  const double poles[4] = {-35.0, -15.0, -2.0, -3.0};
  // controllability matrix columns B, AB, A^2 B, A^3 B
  double Cm[4][4];
  double col[4] = {B[0], B[1], B[2], B[3]};
  for (int k = 0; k < 4; ++k) {
    for (int i = 0; i < 4; ++i) Cm[i][k] = col[i];
    double nxt[4];
    for (int i = 0; i < 4; ++i) {
      double acc = 0.0;
      for (int j = 0; j < 4; ++j) acc += A[i][j] * col[j];
      nxt[i] = acc;
    }
    for (int i = 0; i < 4; ++i) col[i] = nxt[i];
  }
  // phi(A) = prod (A - p_i I)
  double P[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int q = 0; q < 4; ++q) {
    double T[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
        for (int k = 0; k < 4; ++k) acc += P[i][k] * (A[k][j] - (k == j ? poles[q] : 0.0));
        T[i][j] = acc;
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) P[i][j] = T[i][j];
  }
  // solve Cm^T x = e4 by Gaussian elimination with partial pivoting
  double G[4][5];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 4; ++j) G[i][j] = Cm[j][i];
    G[i][4] = (i == 3) ? 1.0 : 0.0;
  }
  for (int k = 0; k < 4; ++k) {
    int piv = k;
    for (int i = k + 1; i < 4; ++i)
      if (fabs(G[i][k]) > fabs(G[piv][k])) piv = i;
    for (int j = 0; j < 5; ++j) {
      const double t = G[k][j];
      G[k][j] = G[piv][j];
      G[piv][j] = t;
    }
    for (int i = k + 1; i < 4; ++i) {
      const double f = G[i][k] / G[k][k];
      for (int j = k; j < 5; ++j) G[i][j] -= f * G[k][j];
    }
  }
  double x[4];
  for (int i = 3; i >= 0; --i) {
    double acc = G[i][4];
    for (int j = i + 1; j < 4; ++j) acc -= G[i][j] * x[j];
    x[i] = acc / G[i][i];
  }
  double K0 = 0.0, K1 = 0.0;
  for (int i = 0; i < 4; ++i) {
    K0 += x[i] * P[i][0];
    K1 += x[i] * P[i][1];
  }
  heading_gain = clip_ref(K1, 0.02, 0.04);
  lateral_gain = clip_ref(K0, 3.4, 4.1);
}

// The chosen waypoint path of the controller (lookahead 16 => at most 17 waypoints), in registers.
#define SMX_CTRL_WPS 17
struct CtrlPath {
  double x[SMX_CTRL_WPS], y[SMX_CTRL_WPS], h[SMX_CTRL_WPS];
  int n;
};

// statically indexed stores / loads (dynamic indexing would put the arrays in scratch)
__device__ __forceinline__ void ctrl_path_put(CtrlPath& p, int i, double x, double y, double h) {
#pragma unroll
  for (int k = 0; k < SMX_CTRL_WPS; ++k)
    if (k == i) {
      p.x[k] = x;
      p.y[k] = y;
      p.h[k] = h;
    }
}

struct ControlOut {
  double throttle, brake, steering;
};

// lane_following_controller.py:63-365 after the waypoint query: `path` is wp_paths[want] where
// want = clip(current_lane + lane_change) (:100-103, find_current_lane :367-374).  Lateral gains:
// place_poles (:376-437) lands outside the clip window for every target speed of the Lane action
// space, so they are the clip bounds (0.04, 3.4) for target_speed > 0 and the literals
// (0.01, 0.36) at 0 — verified against scipy in tests/test_host_logic.py.
__device__ inline ControlOut lane_following_from_path(const VehState& s, CtrlState& cs, double dt, double target_speed,
                                                      int lane_change, double heading_error_gain,
                                                      double lateral_error_gain, const CtrlPath& path) {
  const double px = s.x, py = s.y;
  const HeadingTrig trig = heading_trig(s.heading);
  const double speed = vehicle_speed(s, trig);
  ControlOut out;
  out.throttle = cs.throttle;
  out.brake = 0.0;
  out.steering = cs.steer;
  const int wp_n = path.n;
  if (wp_n <= 0) return out;  // reference asserts "no waypoints found"; keep the last command
  Traj10 tr;
  tr.n = wp_n < 10 ? wp_n : 10;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    tr.x[k] = path.x[k];
    tr.y[k] = path.y[k];
    tr.h[k] = path.h[k];
  }
  const double wp0h = path.h[0];
  const double wp4x = path.x[4], wp4y = path.y[4];

  // road curviness (:105-118): EWMA over reversed consecutive pairs
  double ewma = 0.0;
#pragma unroll
  for (int k = 15; k >= 0; --k) {
    if (k + 1 < wp_n) {
      double rel = heading_relative_to(path.h[k], path.h[k + 1]);
      ewma = lerp_ref(ewma, fabs(rel) * (180.0 / SMX_PI), 0.03);
    }
  }
  double road_curviness = clip_ref(ewma / 2.5, 0.0, 1.0);

  double look_ahead_curvature = fabs(curvature_calculation(tr, 4));
  if (look_ahead_curvature <= 2.0) {
    // wp_path[4] exists whenever the curvature is finite (needs 10 points)
    cs.mcl_x = wp4x;
    cs.mcl_y = wp4y;
    cs.mcl_set = true;
  }
  int look_ahead_wp_num = road_curviness > 0.5 ? 3 : 4;
  look_ahead_wp_num = look_ahead_wp_num < wp_n - 1 ? look_ahead_wp_num : wp_n - 1;
  double lax = path.x[0], lay = path.y[0], lah = path.h[0];
#pragma unroll
  for (int k = 1; k <= 4; ++k)
    if (k == look_ahead_wp_num) {
      lax = path.x[k];
      lay = path.y[k];
      lah = path.h[k];
    }
  double reference_heading = wp0h;
  double ldx = lax - px, ldy = lay - py;
  double look_ahead_dist = sqrt(ldx * ldx + ldy * ldy);
  double vlx = px - look_ahead_dist * trig.sh;
  double vly = py + look_ahead_dist * trig.ch;

  double raw_throttle;
  if (road_curviness < 0.3) {
    raw_throttle = -3.6 * 1.8 * (speed - target_speed);
  } else if (road_curviness > 0.3 && road_curviness < 0.8) {
    raw_throttle = -0.6 * 3.6 * (speed - clip_ref(target_speed, 0.0, 6.94));
  } else {
    raw_throttle = -0.6 * 3.6 * (speed - clip_ref(target_speed, 0.0, 5.56));
  }
  double speed_error = speed - target_speed;
  cs.spd_int += speed_error * dt;
  double velocity_error_damping_term = (speed_error - cs.spd_err) / dt;
  double lateral_force_coefficient = 1.5;
  if (speed < 8.0 || target_speed < 6.0) lateral_force_coefficient = 0.0;
  raw_throttle += (-0.2 * velocity_error_damping_term - 0.1 * cs.spd_int +
                   fabs(lateral_force_coefficient * sin(cs.steer * (SMX_MAX_STEERING / SMX_STEERING_GEAR_RATIO))));
  cs.spd_err = speed_error;

  if (cs.mcl_set) {
    double mx = px - cs.mcl_x, my = py - cs.mcl_y;
    if (sqrt(mx * mx + my * my) < 2.0) reference_heading = lah;
  }

  double dvx, dvy;
  radians_to_vec(lah, dvx, dvy);
  double controller_lat_error = signed_dist_to_line(vlx, vly, lax, lay, dvx, dvy);

  double curvature_radius = curvature_calculation(tr, 0);
  double brake_norm = 0.0, throttle_norm;
  double lng, lat;
  long_lat_speed(s, trig, lng, lat);
  if (raw_throttle < 0.0) {
    brake_norm = clip_ref(-raw_throttle, 0.0, 1.0);
    throttle_norm = 0.0;
  } else {
    double traction_gain;
    if (speed > 70.0 / 3.6 && fabs(curvature_radius) <= 1e3)
      traction_gain = 4.5;
    else if (40.0 / 3.6 <= speed && speed <= 70.0 / 3.6 && fabs(curvature_radius) <= 3.0)
      traction_gain = 2.5;
    else
      traction_gain = 0.5;
    throttle_norm = clip_ref(raw_throttle - traction_gain * 3.6 * fabs(lat), 0.0, 1.0);
  }
  cs.lat_int += dt * controller_lat_error;
  double steering_feed_forward_gain = 0.15;
  if (fabs(curvature_radius) < 7.0) steering_feed_forward_gain = 0.45;
  double steering_controller_feed_forward = 1.0 * steering_feed_forward_gain * (1.0 / curvature_radius) * (speed * speed);
  double normalized_speed = clip_ref(speed * 3.6 / 100.0, 0.0, 1.0);
  double heading_speed_gain = -lerp_ref(0.5, 14.0, normalized_speed);
  double yaw_rate_speed_gain = lerp_ref(5.75, 11.75, normalized_speed);
  double lateral_speed_gain = clip_ref(lerp_ref(-1.0, 14.0, normalized_speed), 1.0, 2.0);
  double max_steering_normalized = 1.0;
  if (fabs(curvature_radius) > 1e7 && lane_change != 0) {
    heading_speed_gain = -4.95;
    yaw_rate_speed_gain = 1.0;
    lateral_speed_gain = 0.22;
    max_steering_normalized = 0.12;
  }
  double z_yaw = s.r;
  double heading_error = min_angles_difference_signed(py_mod(s.heading, SMX_TWO_PI), reference_heading);
  double steering_norm =
      clip_ref(-heading_speed_gain * (heading_error_gain * (180.0 / SMX_PI)) * heading_error +
                   lateral_speed_gain * lateral_error_gain * controller_lat_error + yaw_rate_speed_gain * z_yaw +
                   0.3 * cs.lat_int - steering_controller_feed_forward,
               -max_steering_normalized, max_steering_normalized);
  // low_pass_filter (math.py:219-244)
  cs.steer += dt * 5.5 * (steering_norm - cs.steer);
  cs.steer = clip_ref(cs.steer + 0.0, -1.0, 1.0);
  cs.throttle += dt * 2.0 * (throttle_norm - cs.throttle);
  cs.throttle = clip_ref(cs.throttle + 0.0, 0.0, 1.0);
  out.throttle = cs.throttle;
  out.brake = brake_norm;
  out.steering = cs.steer;
  return out;
}


// Serial search (one lane does everything): number the paths in the reference's order (lanes by
// index, branches depth-first), pick the one nearest by its first waypoint, synthesise
// wp_paths[clip(nearest + lane_change)].  k_control uses it only when the wanted path is not one of
// the four its team synthesises in parallel.
__device__ inline void ctrl_path_serial(const MapDev& m, const PathSeeds& seed, double px, double py, int want,
                                        int* knots, int kstride, CtrlPath& path) {
  path.n = 0;
  int idx = 0;
  for (int li = 0; li < seed.n_lanes; ++li) {
    int start = seed_start(m, seed, li, px, py);
    if (start < 0) continue;
    BranchState bs;
    bs.reset();
    do {
      if (idx == want) {
        path.n = equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, kstride, SMX_CTRL_WPS,
                                     [&](int i, const WaypointOut& w) { ctrl_path_put(path, i, w.x, w.y, w.heading); });
        return;
      }
      // not the wanted path: walk it only to discover its branchings
      equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, kstride, 0,
                          [&](int, const WaypointOut&) {});
      ++idx;
    } while (bs.advance());
  }
}

// AckermannChassis.control (chassis.py:678-718) + one SMARTS tick of the body model.
// The heading's sine / cosine are carried from substep to substep by rotating them through the
// substep's yaw increment d = h * r (|d| <= a few mrad; sin d / cos d from their Taylor series to
// d^7 / d^8, truncation < 1e-20): 24 libm sincos calls become one, with < 1e-14 drift per tick.
__device__ inline void vehicle_step(VehState& s, ControlOut c, double dt) {
  double sh, ch;
  sincos(s.heading, &sh, &ch);
  HeadingTrig t0;
  t0.sh = sh;
  t0.ch = ch;
  double lng, lat;
  long_lat_speed(s, t0, lng, lat);
  double brake = c.brake;
  if (brake > 0.0 && lng < 1.0 / 36.0) brake = 0.0;
  int substeps = (int)(dt * SMX_MAX_PYBULLET_FREQ);
  if (substeps < 1) substeps = 1;
  const double h = dt / (double)substeps;
  const double delta_target = -c.steering * SMX_MAX_STEERING * (1.0 / SMX_STEERING_GEAR_RATIO);
  const double drive_accel = 4.0 * (c.throttle * SMX_MAX_TORQUE) / SMX_WHEEL_RADIUS / SMX_TOTAL_MASS;
  const double brake_decel = 4.0 * (brake * SMX_MAX_BTORQUE) / SMX_WHEEL_RADIUS / SMX_TOTAL_MASS;
  const double f_max = 0.5 * SMX_GROUND_FRICTION * SMX_TOTAL_MASS * SMX_GRAVITY;
  const double inv_mass = 1.0 / SMX_TOTAL_MASS, inv_inertia = 1.0 / SMX_TOTAL_INERTIA_Z;
  for (int k = 0; k < substeps; ++k) {
    s.delta += SMX_STEER_POSITION_GAIN * (delta_target - s.delta);
    s.delta = fmin(fmax(s.delta, -SMX_STEER_LIMIT), SMX_STEER_LIMIT);
    double u = s.u, v = s.v, r = s.r;
    double u_new = u + h * (drive_accel + v * r);
    if (brake_decel > 0.0 && u_new > 0.0) u_new = fmax(0.0, u_new - h * brake_decel);
    double v_new, r_new;
    if (u_new >= SMX_KINEMATIC_BELOW_SPEED) {
      // one reciprocal per substep, reciprocal constants (the form oracle/dynamics.py states)
      const double inv_u = 1.0 / u_new;
      double alpha_f = s.delta - (v + SMX_AXLE_DIST * r) * inv_u;
      double alpha_r = -(v - SMX_AXLE_DIST * r) * inv_u;
      double f_f = fmin(fmax(SMX_CORNERING_STIFFNESS * alpha_f, -f_max), f_max);
      double f_r = fmin(fmax(SMX_CORNERING_STIFFNESS * alpha_r, -f_max), f_max);
      v_new = v + h * ((f_f + f_r) * inv_mass - u_new * r);
      r_new = r + h * (SMX_AXLE_DIST * (f_f - f_r) * inv_inertia);
    } else {
      r_new = u_new * tan(s.delta) / SMX_WHEELBASE;
      v_new = r_new * SMX_AXLE_DIST;
    }
    const double hd = s.heading;
    s.x += h * (-u_new * sh - v_new * ch);
    s.y += h * (u_new * ch - v_new * sh);
    const double d = h * r_new;
    s.heading = hd + d;
    if (fabs(d) < 0.05) {
      const double d2 = d * d;
      const double sd = d * (1.0 + d2 * (-1.0 / 6.0 + d2 * (1.0 / 120.0 + d2 * (-1.0 / 5040.0))));
      const double cd = 1.0 + d2 * (-0.5 + d2 * (1.0 / 24.0 + d2 * (-1.0 / 720.0 + d2 * (1.0 / 40320.0))));
      const double nsh = sh * cd + ch * sd;
      ch = ch * cd - sh * sd;
      sh = nsh;
    } else {
      sincos(s.heading, &sh, &ch);
    }
    s.u = u_new;
    s.v = v_new;
    s.r = r_new;
  }
  s.heading = wrap_heading(s.heading);
}


// ---------------------------------------------------------------------------------
// ActionSpaceType.Trajectory: TrajectoryTrackingController.perform_trajectory_tracking_PD
// (trajectory_tracking_controller.py:176-331) with calculate_raw_throttle_feedback (:333-395),
// calculate_heading_lateral_error (:398-441), curvature_calculation (:444-473) and the sedan's
// parameters (models/controller_parameters.yaml:1-16).  The controller reads trajectory points
// 0..9, the last point and the length only, which is the form it travels in (smx.h).
// Controller state reuse: lat_int = lateral_error, spd_int = integral_velocity_error, spd_err =
// velocity_error, mcl_x = integral_windup_error, mcl_y = heading_error, steer / throttle = filters.
// ---------------------------------------------------------------------------------
struct PackedTraj {
  const double* p;  // [4][SMX_TRAJ_COLS]: x, y, heading, speed rows; column 10 = the last point
  int n;
  __device__ __forceinline__ double at(int row, int i) const {
    const double* r = p + row * SMX_TRAJ_COLS;
    if (i == n - 1) return n > 10 ? r[10] : r[i];
    return r[i];
  }
  __device__ __forceinline__ double last(int row) const { return at(row, n - 1); }
};

__device__ inline double traj_curvature(const PackedTraj& t, int offset, int num_points) {
  if (t.n <= num_points + offset) return 1e20;
  double hs = 0.0, ds = 0.0;
  for (int i = 0; i < num_points; ++i) {
    hs += min_angles_difference_signed(t.at(2, i + 1 + offset), t.at(2, i + offset));
    const double ex = t.at(0, i + offset) - t.at(0, i + offset + 1), ey = t.at(1, i + offset) - t.at(1, i + offset + 1);
    ds += fabs(sqrt(ex * ex + ey * ey));
  }
  if (hs == 0.0) return 1e20;
  return ds / hs;
}

__device__ __forceinline__ double low_pass_filter_ref(double input, double prev, double filter_constant, double dt,
                                                      double raw_value) {
  // utils/math.py:219-244 (lower bound -1)
  prev += dt * filter_constant * (input - prev);
  return clip_ref(prev + raw_value, -1.0, 1.0);
}

__device__ inline ControlOut trajectory_tracking_pd(const VehState& s, CtrlState& cs, double dt, const PackedTraj& t) {
  const HeadingTrig trig = heading_trig(s.heading);
  const double speed = vehicle_speed(s, trig);
  double lng, lat;
  long_lat_speed(s, trig, lng, lat);
  // models/controller_parameters.yaml, sedan.control
  const double final_steering_filter_constant = 23.5, velocity_gain = 5.1, velocity_integral_gain = 0.0,
               traction_gain = 6.0, derivative_activation = 1.0, velocity_damping_gain = 0.001, windup_gain = 0.01;
  double throttle_filter_constant = 22.5;
  const int initial_look_ahead = 6;
  double lateral_gain = 0.61, heading_gain = 0.01, lateral_error_derivative_gain = 0.15,
         heading_error_derivative_gain = 0.5;
  const double normalized_speed = clip_ref((3.6 * t.at(3, 0) - 20.0) / (80.0 - 20.0), 0.0, 1.0);
  if (speed > 70.0 / 3.6) {
    lateral_gain = 1.51;
    heading_error_derivative_gain = 0.1;
  }
  const double steering_filter_constant = lerp_ref(12.0, final_steering_filter_constant, normalized_speed);
  if (fabs(min_angles_difference_signed(t.last(2), t.at(2, 0))) > 2.0) throttle_filter_constant = 2.5;
  if (fabs(traj_curvature(t, 0, 3)) < 150.0) {
    heading_gain = 0.05;
    lateral_error_derivative_gain = 0.015;
    heading_error_derivative_gain = 0.05;
  }
  const double ahead_curvature = fabs(traj_curvature(t, 4, 5));
  // ---- calculate_heading_lateral_error
  const double heading_error = min_angles_difference_signed(py_mod(s.heading, SMX_TWO_PI), t.at(2, 0));
  int look = initial_look_ahead;
  double look_dist = (double)initial_look_ahead;
  if (ahead_curvature < 30.0) {  // speed_reduction_activation = 1
    look = 1;
    look_dist = 1.0;
  }
  const int k = look < t.n - 1 ? look : t.n - 1;
  double pvx, pvy;
  radians_to_vec(t.at(2, k), pvx, pvy);
  const double lx = s.x - look_dist * trig.sh, ly = s.y + look_dist * trig.ch;
  const double lateral_error = signed_dist_to_line(lx, ly, t.at(0, k), t.at(1, k), pvx, pvy);
  // ---- steering
  const double curvature_radius = traj_curvature(t, 0, 5);
  const double derivative_term =
      +heading_error_derivative_gain * s.r + lateral_error_derivative_gain * (lateral_error - cs.lat_int) / dt;
  const double feed_forward = 0.1 * (1.0 / curvature_radius) * (speed * speed);
  const double steering_raw = clip_ref(derivative_activation * derivative_term + (heading_gain * heading_error) * (180.0 / SMX_PI) +
                                           1.0 * lateral_gain * lateral_error - feed_forward,
                                       -1.0, 1.0);
  cs.steer = low_pass_filter_ref(steering_raw, cs.steer, steering_filter_constant, dt, 0.0);
  // ---- calculate_raw_throttle_feedback
  double desired_speed = t.last(3);
  if (ahead_curvature < 30.0)
    desired_speed = clip_ref(0.8 * desired_speed, 0.0, 8.3);
  else if (ahead_curvature < 100.0)
    desired_speed *= 0.8;
  const double velocity_error = speed - desired_speed;
  const double damping = (velocity_error - cs.spd_err) / dt;
  const double raw = 3.6 * (-0.5 * velocity_gain * velocity_error -
                            velocity_integral_gain * (cs.spd_int + windup_gain * cs.mcl_x) - velocity_damping_gain * damping);
  cs.spd_err = velocity_error;
  cs.mcl_x = clip_ref(raw, -1.0, 1.0) - raw;
  cs.throttle = low_pass_filter_ref(raw, cs.throttle, throttle_filter_constant, dt, -traction_gain * fabs(lat));
  ControlOut out;
  if (cs.throttle > 0.0) {
    out.brake = 0.0;
    out.throttle = clip_ref(cs.throttle, 0.0, 1.0);
  } else {
    out.brake = clip_ref(-cs.throttle, 0.0, 1.0);
    out.throttle = 0.0;
  }
  out.steering = cs.steer;
  cs.mcl_y = heading_error;
  cs.lat_int = lateral_error;
  cs.spd_int += (speed - desired_speed) * dt;
  return out;
}

// ---------------------------------------------------------------------------------
// Scripted social vehicle (include/smx.h smx_config.num_social): one tick of a kinematic lane
// follower.  `lane` / `offset` locate it on a centre line (smx_shape_rec.cum is the arclength);
// at a lane's end it continues on outgoing lane (slot + crossed) mod #outgoing, or stops there if
// the lane has none.  Pose: the point at `offset`, heading of the segment that holds it.
// ---------------------------------------------------------------------------------
__device__ inline void social_pose(const MapDev& m, int lane, double offset, double& x, double& y, double& heading) {
  const int v0 = m.lane_shape_off[lane], v1 = m.lane_shape_off[lane + 1];
  int seg = v1 - 2;
  for (int v = v0; v + 1 < v1; ++v) {
    const smx_shape_rec a = m.shape_rec[v];
    if (a.cum + a.len > offset) {
      seg = v;
      break;
    }
  }
  if (seg < v0) seg = v0;  // a one-vertex lane cannot exist (the map compiler drops them)
  const smx_shape_rec a = m.shape_rec[seg], b = m.shape_rec[seg + 1];
  const double along = fmin(fmax(offset - a.cum, 0.0), a.len);
  const double f = a.len > 0.0 ? along / a.len : 0.0;
  x = a.x + (b.x - a.x) * f;
  y = a.y + (b.y - a.y) * f;
  heading = wrap_heading(atan2(b.y - a.y, b.x - a.x) - 0.5 * SMX_PI);
}

// `cmd` >= 0: the speed decided for this tick (car following); < 0: the constant fraction of the limit
__device__ inline void social_step(const MapDev& m, int slot, double factor, double dt, int& lane, double& offset,
                                   int& crossed, double& speed, double cmd = -1.0) {
  speed = cmd >= 0.0 ? cmd : m.lane_speed[lane] * factor;
  offset += speed * dt;
  for (int guard = 0; guard < 64; ++guard) {
    const int v1 = m.lane_shape_off[lane + 1];
    const double L = m.shape_rec[v1 - 1].cum;
    if (offset < L) break;
    const int a = m.lane_out_off[lane], n_out = m.lane_out_off[lane + 1] - a;
    if (n_out <= 0) {
      offset = L;
      speed = 0.0;
      break;
    }
    offset -= L;
    lane = m.lane_out_idx[a + (slot + crossed) % n_out];
    ++crossed;
  }
}
