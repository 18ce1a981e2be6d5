// smx_kernels.hip — the per-tick kernels of the SMARTS hot path on gfx950, and the C-ABI.
//
// A tick = SMARTS._step (smarts.py:236-327) for every environment instance of the shard, as a short
// sequence of kernels on one stream (each stage has its own natural thread mapping; together
// they stay far below the register pressure of one fused kernel):
//
//   k_control   4 lanes / vehicle       A controllers (_perform_agent_actions, smarts.py:1233-1263):
//                                         the team finds the controller's waypoint path together
//                                       B physics     (_step_pybullet, smarts.py:923-931)
//   k_scan      8 lanes / vehicle       map sweeps at the new pose: nearest lane / road_with_point
//                                       at centre + 4 corners, lane heading (wrong way); 10 nearest
//                                       lanepoints, path seeds
//   k_sensors   workgroup roles         waypoints role  4 lanes / vehicle: waypoint paths streamed
//                                         into the dense rows, trip meter / reward
//                                       observe role    1 lane / vehicle, whole envs / workgroup:
//                                         collisions, neighbours, ego block, accelerometer, driven
//                                         path, events, done
//                                       lidar / OGM roles  1 wavefront / vehicle
//   k_commit    1 lane / vehicle        new flags (teardown), dones["__all__"], auto-reset respawn
//
// followed, for envs whose episode ended under auto_reset (parallel_env.py:303-309), by k_scan /
// k_sensors / k_commit restricted to the re-created vehicles.  Above 16384 vehicles every role is
// launched on its own (k_waypoints, k_observe, k_lidar, k_ogm; see enqueue()).  Envs are independent
// (reference: one process per env, parallel_env.py:96-122): no inter-workgroup communication.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "smx_scan.h"
#include "smx_vehicle.h"

#define SMX_BLOCK 64
#define SMX_COLLISION_LEEWAY 0.05  // chassis.py:75-78
#define SMX_WP_LANES 4             // lanes of a wavefront that share one vehicle (k_control, waypoints role)
#define SMX_POSE_SCAN_RADIUS 10.0
// SMX_LAUNCH_AUTO: the LARGE launch form above this many vehicles.  Measured crossover (round 2, C4's shape:
// 8 192 vehicles 0.165 / 0.216 ms small / large, 32 768: 0.461 / 0.303, 65 536: 0.886 / 0.505; C3 at 32 768:
// 0.514 / 0.504; C2 at 8 192: 0.133 / 0.187).  At 16 384 the two cross: every agent alive 0.258 / 0.226, over
// ticks 50-550 of a run (fewer alive) 0.223 / 0.250 — the longer run decides, 16 384 stays small.
#define SMX_LARGE_BATCH_VEHICLES 16384
#define SMX_WPT_PRELOAD 8           // knots of a path held in registers while it is interpolated
#ifndef SMX_WP_STAGED
#define SMX_WP_STAGED 0             // developer variant (-DSMX_WP_STAGED=1): k_waypoints_tables instead of k_waypoints_emit
#endif
#define SMX_SLOW_BLOCKS 512          // workgroups of the slow lists' kernels on a map without junctions (smx_load_map: slow_blocks)
#ifndef SMX_SCAN_UNSEEDED
#define SMX_SCAN_UNSEEDED 0         // developer variant (-DSMX_SCAN_UNSEEDED=1): the scan never starts from last tick's answers
#endif

struct KernelArgs {
  smx_config cfg;
  MapDev map;
  smx_state st;
  smx_spawns sp;
  smx_outputs out;
  const int8_t* actions;
  const float* actions_f32;  // float action spaces: [E*N][3]
  const double* traj;        // Trajectory space: [E*N][4][SMX_TRAJ_COLS]
  const int32_t* traj_n;     // ... true lengths, 0 = no action
  const uint8_t* env_mask;  // k_reset: explicit mask (NULL = use env_reset_pending / all)
  const double* lidar_rays;
  const smx_via* vias;          // device copy of smx_set_vias
  const int32_t* via_slot_off;  // [num_vehicles + 1]
  int first_only;           // restrict to vehicles carrying SMX_F_FIRST (reset observations)
  int keep_reward_done;     // auto-reset: the terminal step's reward / done / env_done stay
  int reset_all;            // k_reset: every env (explicit reset with NULL mask)
  double heading_gain_pos, lateral_gain_pos;  // lateral gains for target_speed > 0
  int wp_pool_limit;        // k_waypoints_emit: records of its LDS pool in use (SMX_WPE_POOL; less: developer / tests)
  char* wp_spill;           // k_waypoints_emit: knot records of the paths its LDS pool has no room for, [workgroup][column][11] + lanes
  int walk_new;             // k_first: also walk the new vehicles' knot lists (large batches: for the next tick's k_control_fast)
  double nb_d2_max;         // the largest squared distance whose rounded square root is <= cfg.nb_radius (radius_threshold)
  int debug_skip;
  int wp_blocks, obs_blocks, lidar_blocks;  // k_sensors: workgroups per role (OGM takes the rest)
  double dagm_reach;        // widest lane's half width (which segments can touch a DAGM view)
  KnotLists knots;          // library-owned hand-off: k_wp_walk -> k_waypoints_tables
  MissionsDev missions;     // device copy of smx_set_missions (null pointers: every mission endless)
  // large batches: the tick's alive vehicles, compacted by k_alive_list at the start of the tick (null: launch
  // index = vehicle).  The per-vehicle team kernels then run over full wavefronts however many agents are gone.
  const int32_t* alive_list;
  const int32_t* alive_count;
  int32_t* status;          // library-owned device word of SMX_DEVICE_* bits, read and cleared by smx_sync
  // library-owned, kept from tick to tick: what the scan's seeded searches start from (smx_scan.h).  Null: unseeded.
  double* seeds_carry;      // [4][E*N]: pose (x, y) the seeds half last ran at, d2 of its 10th nearest and of its nearest lanepoint (< 0: none)
  double* facts_carry;      // [2][E*N]: pose (x, y) the facts half last ran at (its answers are facts_i32 / facts_f64)
  // large batches: vehicles the one-lane scan kernels could not serve (k_scan_fast -> k_scan_half over this list)
  int32_t* slow_list;
  int32_t* slow_count;
  // [E*N], 1: the vehicle's path seeds, walks and rows are the slow chain's this tick (k_scan_fast<1> decides; null: none)
  uint8_t* seed_pending;
};
enum { SMX_DEVICE_BAD_LANE_ACTION = 1 };  // a Lane action code outside -1..3 was met (and treated as "no action")

#define SF(field) a.st.f64[(size_t)(field) * total + gid]

// The vehicle that team (or lane) i of a per-vehicle launch works on; `total` = none (i is past the last one).
__device__ __forceinline__ size_t launch_vehicle(const KernelArgs& a, size_t i, size_t total) {
  if (a.alive_list == nullptr) return i < total ? i : total;
  // (the entry is loaded beside the count, not behind it: one round trip; entries past the count are old vehicle
  // numbers or zeros, in range either way)
  const int32_t entry = a.alive_list[i < total ? i : total - 1];
  return i < (size_t)*a.alive_count ? (size_t)entry : total;
}

// Developer timing switches ("switch a piece off and see what the tick costs without it"): they exist
// only in the -DSMX_DEBUG_TIMING variant of the library (smarts_amd/build.py --prof); in the shipped
// library the test is the constant false and nothing, environment included, can drop work from a tick.
#if defined(SMX_ABLATE)  // developer variant without the stamps: the pieces named by a compile-time mask are off
#define SMX_SKIP(args, bit) (((SMX_ABLATE) & (bit)) != 0)
#elif defined(SMX_DEBUG_TIMING)
#define SMX_SKIP(args, bit) (((args).debug_skip & (bit)) != 0)
#else
#define SMX_SKIP(args, bit) false
#endif

// ---------------------------------------------------------------------------------
// oriented-box proximity (substitution for pybullet getClosestPoints)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void box_corners(double x, double y, double sh, double ch, double len, double wid,
                                            double* cx, double* cy) {
  double fx = -sh, fy = ch, rx = ch, ry = sh;
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

__device__ __forceinline__ bool point_in_box(double px, double py, double x, double y, double sh, double ch,
                                             double len, double wid) {
  double fx = -sh, fy = ch, rx = ch, ry = sh;
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

__device__ inline bool boxes_within(double ax, double ay, double ah, double bx, double by, double bh, double len,
                                    double wid, double leeway) {
  // broad phase: circumscribed circles
  double dx = ax - bx, dy = ay - by;
  double reach = sqrt(len * len + wid * wid) + leeway;
  if (dx * dx + dy * dy > reach * reach) return false;
  double cax[4], cay[4], cbx[4], cby[4];
  double sa, ca, sb, cb;
  sincos(ah, &sa, &ca);
  sincos(bh, &sb, &cb);
  box_corners(ax, ay, sa, ca, len, wid, cax, cay);
  box_corners(bx, by, sb, cb, len, wid, cbx, cby);
  double best = SMX_INF;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (point_in_box(cax[i], cay[i], bx, by, sb, cb, len, wid)) return true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      best = fmin(best, seg_point_dist2(cax[i], cay[i], cbx[k], cby[k], cbx[(k + 1) & 3], cby[(k + 1) & 3]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (point_in_box(cbx[i], cby[i], ax, ay, sa, ca, len, wid)) return true;
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

// TripMeterSensor.append_waypoint_if_new's should_count_wp (sensors.py:908-913): with a fixed route only waypoints
// on the route's roads count.  The first waypoint of the first path lies on the seed road: with the waypoints
// sensor that is one of the route's roads by construction (_waypoint_paths_along_route); without it the path
// comes from the unrouted lookahead-1 query (sensors.py:271-275), whose start lanepoint is SMX_FI_OBS_START.
__device__ __forceinline__ bool trip_counts_waypoint(const KernelArgs& a, const MapDev& m, size_t gid, size_t total) {
  RouteFilter f;
  if (!f.fixed_route(a.missions, (int)(gid % (size_t)a.cfg.num_vehicles), m.n_roads)) return true;
  if (a.cfg.sensors & SMX_SENSOR_WAYPOINTS) return true;
  const int os = a.st.facts_i32[(size_t)SMX_FI_OBS_START * total + gid];
  if (os < 0) return true;  // no waypoint this tick anyway
  return f.has(m, m.lane_road[m.lp_rec[os].lane]);
}

__device__ __forceinline__ VehState load_vehicle(const KernelArgs& a, size_t gid, size_t total) {
  VehState s;
  s.x = SF(SMX_S_X);
  s.y = SF(SMX_S_Y);
  s.heading = SF(SMX_S_HEADING);
  s.u = SF(SMX_S_U);
  s.v = SF(SMX_S_V);
  s.r = SF(SMX_S_R);
  s.delta = SF(SMX_S_DELTA);
  return s;
}
// =================================================================================
// k_control: controllers (a1-a3) + vehicle dynamics (a4-a6), SMX_WP_LANES lanes per vehicle.
// The controller's waypoint query (lane_following_controller.py:96-98) is the long part: the team
// walks the candidate paths together — lane p synthesises path p, every lane measures the first
// waypoint of the paths it owns (find_current_lane :367-374) — then the wanted path moves to
// lane 0 through shuffles and lane 0 runs the control law and the 24 physics substeps.
// =================================================================================

// One instantiation per action space: the Lane kernel does not carry the registers of the others.
// waves_per_eu(2): at most 256 registers, so that two wavefronts share a SIMD on large batches.
// LDS_PATH (small batches, lane-following spaces): the candidate path is written to LDS as it is
// synthesised and read back with fixed indices by the lane that runs the control law.  In registers a
// put at a run-time index is a 17-way compare / select chain over every live element (~90
// instructions per waypoint); the LDS copy costs 26 KB per workgroup, which would halve the
// wavefronts per CU on large batches, so those keep the register form.
template <int SPACE, bool LDS_PATH = false>
__global__ void __attribute__((amdgpu_waves_per_eu(2, 8))) __launch_bounds__(SMX_BLOCK) k_control(const KernelArgs a) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  __shared__ double path_lds[LDS_PATH ? 3 * SMX_CTRL_WPS * SMX_BLOCK : 1];
  int* knots = knot_scratch + threadIdx.x;
  // element (waypoint i, component q) of this lane's path: consecutive lanes, consecutive words
  auto path_put = [&](CtrlPath& p, int i, double x, double y, double h) {
    if (LDS_PATH) {
      double* q = path_lds + (size_t)(i * 3) * SMX_BLOCK + threadIdx.x;
      q[0] = x;
      q[SMX_BLOCK] = y;
      q[2 * SMX_BLOCK] = h;
    } else {
      ctrl_path_put(p, i, x, y, h);
    }
  };
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int p0 = threadIdx.x % SMX_WP_LANES;
  const size_t gid = ((size_t)blockIdx.x * SMX_BLOCK + threadIdx.x) / SMX_WP_LANES;
  if (gid >= total) return;  // whole teams leave together
  int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE)) return;
  if (flags & SMX_F_SOCIAL) {  // scripted lane follower: no controller, no dynamics
    if (p0 != 0) return;
    int lane = (int)SF(SMX_S_MCL_X), crossed = (int)SF(SMX_S_SPD_INT);
    double offset = SF(SMX_S_MCL_Y), speed, x, y, heading;
    SF(SMX_S_PREV_X) = SF(SMX_S_X);
    SF(SMX_S_PREV_Y) = SF(SMX_S_Y);
    // SMX_SOCIAL_IDM: k_social decided this tick's speed from the state at the start of the tick
    const double cmd = c.social_model == SMX_SOCIAL_IDM ? SF(SMX_S_THROTTLE) : -1.0;
    social_step(m, (int)(gid % c.num_vehicles), c.social_speed_factor, c.dt, lane, offset, crossed, speed, cmd);
    social_pose(m, lane, offset, x, y, heading);
    SF(SMX_S_X) = x;
    SF(SMX_S_Y) = y;
    SF(SMX_S_HEADING) = heading;
    SF(SMX_S_U) = speed;
    SF(SMX_S_MCL_X) = (double)lane;
    SF(SMX_S_MCL_Y) = offset;
    SF(SMX_S_SPD_INT) = (double)crossed;
    return;
  }
  SMX_TSTAMP(tc0);
  VehState s = load_vehicle(a, gid, total);
  CtrlState cs;
  cs.lat_int = SF(SMX_S_LAT_INT);
  cs.spd_int = SF(SMX_S_SPD_INT);
  cs.steer = SF(SMX_S_STEER);
  cs.throttle = SF(SMX_S_THROTTLE);
  cs.spd_err = SF(SMX_S_SPD_ERR);
  cs.mcl_x = SF(SMX_S_MCL_X);
  cs.mcl_y = SF(SMX_S_MCL_Y);
  cs.mcl_set = (flags & SMX_F_MCL_SET) != 0;
  // ---- Controllers.perform_action (controllers/__init__.py:61-152)
  constexpr int space = SPACE;
  int action = SMX_ACTION_NONE;
  float act0 = 0.f, act1 = 0.f, act2 = 0.f;
  bool has_action;
  if (space == SMX_ACTION_SPACE_LANE) {
    action = a.actions[gid];
    if (action < SMX_ACTION_NONE || action > SMX_ACTION_CHANGE_LANE_RIGHT) {
      // the reference looks the action string up in a dict and raises (controllers/__init__.py:137-144); a code
      // that names no action is reported at the next smx_sync and moves nothing
      if (p0 == 0) atomicOr(a.status, SMX_DEVICE_BAD_LANE_ACTION);
      action = SMX_ACTION_NONE;
    }
    has_action = action >= 0;
  } else if (space == SMX_ACTION_SPACE_TRAJECTORY) {
    has_action = a.traj_n[gid] > 0;
  } else {
    act0 = a.actions_f32[gid * 3 + 0];
    act1 = a.actions_f32[gid * 3 + 1];
    act2 = a.actions_f32[gid * 3 + 2];
    has_action = !(act0 != act0);  // NaN = no action
  }
  int act_lane = 0;  // the team lane that runs the control law and the physics (uniform in the team)
  ControlOut co;
  // no action this tick: wheel torques do not persist, the steer motor target does
  co.throttle = 0.0;
  co.brake = 0.0;
  co.steering = cs.steer;
  const bool lane_following =
      space == SMX_ACTION_SPACE_LANE || space == SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED;
  if (has_action && space == SMX_ACTION_SPACE_TRAJECTORY) {
    if (p0 == 0) {
      PackedTraj t;
      t.p = a.traj + gid * (size_t)(4 * SMX_TRAJ_COLS);
      t.n = a.traj_n[gid];
      co = trajectory_tracking_pd(s, cs, c.dt, t);
    }
  } else if (has_action && !lane_following) {
    if (space == SMX_ACTION_SPACE_CONTINUOUS) {
      // :94-99
      co.throttle = clip_ref((double)act0, 0.0, 1.0);
      co.brake = clip_ref((double)act1, 0.0, 1.0);
      co.steering = clip_ref((double)act2, -1.0, 1.0);
    } else {
      // ActuatorDynamicController.perform_action (actuator_dynamic_controller.py:47-80): the third
      // component is a steering *rate*; the held angle is the controller state
      const double change = clip_ref((double)act2, -1.0, 1.0);
      co.throttle = clip_ref((double)act0, 0.0, 1.0);
      co.brake = clip_ref((double)act1, 0.0, 1.0);
      co.steering = clip_ref((1.0 - 0.001) * cs.steer + change * c.dt, -1.0, 1.0);
    }
    cs.steer = co.steering;  // last_steering_angle / the persisting steer target
  }
  if (has_action && lane_following && !SMX_SKIP(a, 1)) {  // uniform within a team
    double target_speed;
    int lane_change;
    double hg, lg;
    if (space == SMX_ACTION_SPACE_LANE) {
      // :125-144
      target_speed = action == SMX_ACTION_KEEP_LANE ? 15.0 : (action == SMX_ACTION_SLOW_DOWN ? 0.0 : 12.5);
      lane_change = action == SMX_ACTION_CHANGE_LANE_LEFT ? 1 : (action == SMX_ACTION_CHANGE_LANE_RIGHT ? -1 : 0);
      hg = target_speed > 0.0 ? a.heading_gain_pos : 0.01;
      lg = target_speed > 0.0 ? a.lateral_gain_pos : 0.36;
    } else {
      // :113-124: (target_speed, lane_change)
      target_speed = (double)act0;
      lane_change = (int)act1;
      lateral_gains_for_speed(target_speed, hg, lg);
    }
    const PathSeeds seed = load_seeds(a, gid, total);  // found by k_scan at this very pose
    const double px = s.x, py = s.y;
    CtrlPath path;
    path.n = 0;
#pragma unroll
    for (int k = 0; k < SMX_CTRL_WPS; ++k) path.x[k] = path.y[k] = path.h[k] = 0.0;
    // Paths are numbered in the reference's order: seed lanes by index, branches depth-first.
    // Team lane p walks seed lanes p, p + 4, ... on its own (no lane re-walks another lane's
    // paths); counts are exchanged by shuffles to turn (lane, branch) into the global number.
    // The first branch of the lane's first seed lane is synthesised in full while it is walked.
    int n_paths = 0;
    double my_d = SMX_INF;
    int my_idx = 0x7fffffff;
    int goff0 = 0, cnt0 = 0;
    SMX_TSTAMP(tc1);
    SMX_TACC(15, tc0, tc1);
    if (seed.road >= 0) {
      for (int r4 = 0; r4 < seed.n_lanes; r4 += SMX_WP_LANES) {  // uniform within a team
        const int li = r4 + p0;
        int cnt = 0, bj = 0x7fffffff;
        double bd = SMX_INF;
        if (li < seed.n_lanes) {
          const int start = seed_start(m, seed, li, px, py);
          if (start >= 0) {
            BranchState bs;
            bs.reset();
            do {
              double fx = 0.0, fy = 0.0;
              if (r4 == 0 && cnt == 0) {
                path.n = equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK,
                                             SMX_CTRL_WPS, [&](int i, const WaypointOut& w) {
                                               path_put(path, i, w.x, w.y, w.heading);
                                               if (i == 0) {
                                                 fx = w.x;
                                                 fy = w.y;
                                               }
                                             });
              } else {
                equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, 1,
                                    [&](int, const WaypointOut& w) {
                                      fx = w.x;
                                      fy = w.y;
                                    });
              }
              const double ex = fx - px, ey = fy - py;
              const double d = sqrt(ex * ex + ey * ey);
              if (d < bd) {  // strict: the lowest-numbered path wins ties (np.argmin)
                bd = d;
                bj = cnt;
              }
              ++cnt;
            } while (bs.advance());
          }
        }
        // exclusive prefix of the counts over the team
        int incl = cnt;
        {
          int t = __shfl_up(incl, 1, SMX_WP_LANES);
          if (p0 >= 1) incl += t;
          t = __shfl_up(incl, 2, SMX_WP_LANES);
          if (p0 >= 2) incl += t;
        }
        const int round_total = __shfl(incl, SMX_WP_LANES - 1, SMX_WP_LANES);
        const int g = n_paths + incl - cnt;
        if (r4 == 0) {
          goff0 = g;
          cnt0 = cnt;
        }
        if (bj != 0x7fffffff && (bd < my_d || (bd == my_d && g + bj < my_idx))) {
          my_d = bd;
          my_idx = g + bj;
        }
        n_paths += round_total;
      }
    }
    SMX_TSTAMP(tc2);
    SMX_TACC(16, tc1, tc2);
    // nearest path over the team: smallest distance, then smallest number
#pragma unroll
    for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) {
      const double od = __shfl_xor(my_d, msk, SMX_WP_LANES);
      const int oi = __shfl_xor(my_idx, msk, SMX_WP_LANES);
      if (od < my_d || (od == my_d && oi < my_idx)) {
        my_d = od;
        my_idx = oi;
      }
    }
    if (n_paths > 0) {  // uniform within a team
      int want = my_idx + lane_change;
      want = want < 0 ? 0 : (want > n_paths - 1 ? n_paths - 1 : want);
      // the lane whose first seed lane holds path `want`
      const bool own = cnt0 > 0 && want >= goff0 && want < goff0 + cnt0;
      if (own && want > goff0) {
        // a later branch of this lane's seed lane: walk to it again (rare: branching inside 16 hops)
        const int start = seed_start(m, seed, p0, px, py);
        BranchState bs;
        bs.reset();
        int j = 0;
        do {
          if (j == want - goff0) {
            path.n = equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK,
                                         SMX_CTRL_WPS,
                                         [&](int i, const WaypointOut& w) { path_put(path, i, w.x, w.y, w.heading); });
            break;
          }
          equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, 0,
                              [&](int, const WaypointOut&) {});
          ++j;
        } while (bs.advance());
      }
      int owners = own ? (1 << p0) : 0;
#pragma unroll
      for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) owners |= __shfl_xor(owners, msk, SMX_WP_LANES);
      // the lane that holds the wanted path carries on alone (control law, physics, state write):
      // nothing moves between lanes and no second copy of the path is kept in registers
      act_lane = owners ? (__ffs(owners) - 1) : 0;
      SMX_TSTAMP(tc3);
      SMX_TACC(17, tc2, tc3);
      if (p0 == act_lane) {
        if (LDS_PATH) {
#pragma unroll
          for (int k = 0; k < SMX_CTRL_WPS; ++k) {
            const double* q = path_lds + (size_t)(k * 3) * SMX_BLOCK + threadIdx.x;
            const bool held = k < path.n;
            path.x[k] = held ? q[0] : 0.0;
            path.y[k] = held ? q[SMX_BLOCK] : 0.0;
            path.h[k] = held ? q[2 * SMX_BLOCK] : 0.0;
          }
        }
        // beyond the team's first seed lanes (roads with more than 4 lanes): serial search
        if (!owners) ctrl_path_serial(m, seed, px, py, want, knots, SMX_BLOCK, path);
        if (!SMX_SKIP(a, 1048576)) co = lane_following_from_path(s, cs, c.dt, target_speed, lane_change, hg, lg, path);
      }
      SMX_TSTAMP(tc4);
      SMX_TACC(18, tc3, tc4);
    } else {
      // reference asserts "no waypoints found"; keep the last command
      co.throttle = cs.throttle;
      co.brake = 0.0;
      co.steering = cs.steer;
    }
  }
  if (p0 != act_lane) return;
  SMX_TSTAMP(tc5);
  SF(SMX_S_PREV_X) = s.x;  // the position recorded by the previous observation
  SF(SMX_S_PREV_Y) = s.y;
  if (!SMX_SKIP(a, 2097152)) vehicle_step(s, co, c.dt);
  SMX_TSTAMP(tc6);
  SMX_TACC(19, tc5, tc6);
  SF(SMX_S_X) = s.x;
  SF(SMX_S_Y) = s.y;
  SF(SMX_S_HEADING) = s.heading;
  SF(SMX_S_U) = s.u;
  SF(SMX_S_V) = s.v;
  SF(SMX_S_R) = s.r;
  SF(SMX_S_DELTA) = s.delta;
  SF(SMX_S_LAT_INT) = cs.lat_int;
  SF(SMX_S_SPD_INT) = cs.spd_int;
  SF(SMX_S_STEER) = cs.steer;
  SF(SMX_S_THROTTLE) = cs.throttle;
  SF(SMX_S_SPD_ERR) = cs.spd_err;
  SF(SMX_S_MCL_X) = cs.mcl_x;
  SF(SMX_S_MCL_Y) = cs.mcl_y;
  a.st.flags[gid] = cs.mcl_set ? (flags | SMX_F_MCL_SET) : (flags & ~SMX_F_MCL_SET);
  SMX_TSTAMP(tc7);
  SMX_TACC(20, tc0, tc7);
}

// =================================================================================
// Large batches: the controller as two launches.  In k_control the control law and the 24 physics substeps
// run on ONE lane of each vehicle's team of four (154 of its 223 us at 131 k vehicles, three quarters of the
// lanes masked off).  Here k_control_paths keeps the team work — candidate paths, nearest path, the wanted
// path written as 17 waypoints to a hand-off in device memory, laid out [waypoint][component][vehicle] —
// and k_control_law runs law + physics with one lane per vehicle: a quarter of the wavefronts, every lane
// busy.  Small batches keep the single launch (one wavefront's latency is what they wait for).
// =================================================================================
struct CtrlHandoff {
  double* path;  // [SMX_CTRL_WPS][3][E*N]: x, y, heading of the wanted path's waypoints
  int32_t* n;    // [E*N] waypoints held; 0 = no path found (the reference asserts; the last command is kept)
};

// Controllers.perform_action's decoding of the Lane / LaneWithContinuousSpeed action (controllers/__init__.py:113-144)
template <int SPACE>
__device__ __forceinline__ bool decode_lane_action(const KernelArgs& a, size_t gid, double& target_speed, int& lane_change,
                                                   double& hg, double& lg) {
  if (SPACE == SMX_ACTION_SPACE_LANE) {
    const int action = a.actions[gid];
    if (action < SMX_ACTION_NONE || action > SMX_ACTION_CHANGE_LANE_RIGHT) {
      atomicOr(a.status, SMX_DEVICE_BAD_LANE_ACTION);  // reported at the next smx_sync; the code moves nothing
      return false;
    }
    if (action < 0) return false;
    target_speed = action == SMX_ACTION_KEEP_LANE ? 15.0 : (action == SMX_ACTION_SLOW_DOWN ? 0.0 : 12.5);
    lane_change = action == SMX_ACTION_CHANGE_LANE_LEFT ? 1 : (action == SMX_ACTION_CHANGE_LANE_RIGHT ? -1 : 0);
    hg = target_speed > 0.0 ? a.heading_gain_pos : 0.01;
    lg = target_speed > 0.0 ? a.lateral_gain_pos : 0.36;
    return true;
  }
  const float act0 = a.actions_f32[gid * 3 + 0], act1 = a.actions_f32[gid * 3 + 1];
  if (act0 != act0) return false;  // NaN = no action
  target_speed = (double)act0;
  lane_change = (int)act1;
  lateral_gains_for_speed(target_speed, hg, lg);
  return true;
}

template <int SPACE>
__device__ __forceinline__ void control_paths_for(const KernelArgs& a, const CtrlHandoff& ho, const size_t gid, int* knots) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int p0 = threadIdx.x % SMX_WP_LANES;
  if (gid >= total) return;  // whole teams leave together
  const int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE) || (flags & SMX_F_SOCIAL)) return;
  double target_speed, hg, lg;
  int lane_change;
  if (!decode_lane_action<SPACE>(a, gid, target_speed, lane_change, hg, lg)) return;  // uniform within a team
  const PathSeeds seed = load_seeds(a, gid, total);  // found by k_scan at this very pose
  const double px = SF(SMX_S_X), py = SF(SMX_S_Y);
  auto put = [&](int i, const WaypointOut& w) {
    double* q = ho.path + (size_t)(i * 3) * total + gid;
    q[0] = w.x;
    q[total] = w.y;
    q[2 * total] = w.heading;
  };
  // ---- no walk at all when the waypoints sensor's chain walks of the previous tick can be reused: they started
  // from these very seeds (the controller asks its paths at the pose of the last observation), and a
  // lookahead-16 path is the first 17 lanepoints of the lookahead-32 one: its knots are the longer path's
  // knots less than 16 hops down plus the lanepoint 16 hops down (KnotLists.end16).  The first waypoint of
  // every path of a seed lane is the projection of the vehicle on the start lanepoint's heading line
  // (interpolate_knots at t = 0), so find_current_lane needs the start records only.  Taken when every seed
  // lane of the team has a list for its start and filter and none branches inside the lookahead.
  if (a.knots.key != nullptr && c.wp_lookahead >= SMX_CTRL_WPS - 1 && seed.road >= 0 && seed.n_lanes <= SMX_WP_LANES) {
    const size_t paths = total * SMX_WP_LANES, path = gid * SMX_WP_LANES + p0;
    const int start = p0 < seed.n_lanes ? seed_start(m, seed, p0, px, py) : -1;
    bool reusable = true;
    int n32 = 0;
    if (start >= 0) {
      reusable = a.knots.key[path] == start && a.knots.key[paths + path] == (seed.f.n > 0 ? seed.f.road[0] : -1) &&
                 a.knots.key[2 * paths + path] == (seed.f.n > 1 ? seed.f.road[1] : -1) && a.knots.cnt[path] == 1;
      n32 = a.knots.n[path];
      reusable = reusable && n32 > 0;
    }
    int bad = reusable ? 0 : 1;
#pragma unroll
    for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) bad |= __shfl_xor(bad, msk, SMX_WP_LANES);
    if (!bad) {  // uniform within a team
      int started = start >= 0 ? (1 << p0) : 0;
#pragma unroll
      for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) started |= __shfl_xor(started, msk, SMX_WP_LANES);
      const int n_paths = __popc(started);
      if (n_paths == 0) {
        if (p0 == 0) ho.n[gid] = 0;
        return;
      }
      const int mine = __popc(started & ((1 << p0) - 1));  // this lane's path number
      double my_d = SMX_INF;
      int my_idx = 0x7fffffff;
      smx_lp_rec r0 = smx_lp_rec{};
      if (start >= 0) {
        r0 = load_lp(m, start, 46);
        const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
        const double fx = n32 == 1 ? r0.x : r0.x + proj * r0.dirx, fy = n32 == 1 ? r0.y : r0.y + proj * r0.diry;
        const double ex = fx - px, ey = fy - py;
        my_d = sqrt(ex * ex + ey * ey);
        my_idx = mine;
      }
#pragma unroll
      for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) {
        const double od = __shfl_xor(my_d, msk, SMX_WP_LANES);
        const int oi = __shfl_xor(my_idx, msk, SMX_WP_LANES);
        if (od < my_d || (od == my_d && oi < my_idx)) {
          my_d = od;
          my_idx = oi;
        }
      }
      int want = my_idx + lane_change;
      want = want < 0 ? 0 : (want > n_paths - 1 ? n_paths - 1 : want);
      if (start < 0 || mine != want) return;
      // ---- the wanted path's 17 waypoints from the list: knots into registers (every load in flight
      // together), their arclength in path order (pass 1's additions), then the interpolation
      const int n16 = n32 < SMX_CTRL_WPS ? n32 : SMX_CTRL_WPS;
      const int nk16 = a.knots.nk16[path];
      const int last = a.knots.end16[path];  // the last knot when it is not one of the list's
      auto fetch = [&](int k) { return (k == nk16 - 1 && last >= 0) ? last : a.knots.idx[(size_t)(k + 1) * paths + path]; };
      constexpr int KP = SMX_WPT_PRELOAD;
      double kx[KP], ky[KP], kh[KP], kw[KP], ks_[KP];
      int kl[KP];
      {
        int kid[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) kid[k] = k < nk16 ? fetch(k) : 0;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const smx_lp_rec* r = m.lp_rec + kid[k];
          const bool have = k < nk16;
          kx[k] = have ? r->x : 0.0;
          ky[k] = have ? r->y : 0.0;
          kh[k] = have ? r->heading : 0.0;
          kl[k] = have ? r->lane : 0;
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const bool ask = k < nk16 && kl[k] != (k == 0 ? (int)r0.lane : kl[k > 0 ? k - 1 : 0]);
          kw[k] = ask ? m.lane_width[kl[k]] : 0.0;
          ks_[k] = ask ? m.lane_speed[kl[k]] : 0.0;
        }
      }
      double D = 0.0;
      {
        const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
        double lastx = r0.x + proj * r0.dirx, lasty = r0.y + proj * r0.diry;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          if (k < nk16) {
            const double ex = kx[k] - lastx, ey = ky[k] - lasty;
            D += sqrt(ex * ex + ey * ey);
            lastx = kx[k];
            lasty = ky[k];
          }
        }
        for (int k = KP; k < nk16; ++k) {
          const smx_lp_rec* r = m.lp_rec + fetch(k);
          const double qx = r->x, qy = r->y;
          const double ex = qx - lastx, ey = qy - lasty;
          D += sqrt(ex * ex + ey * ey);
          lastx = qx;
          lasty = qy;
        }
      }
      interpolate_knots_preloaded<KP>(m, r0, m.lane_width[r0.lane], m.lane_speed[r0.lane], nk16, n16, D, px, py, SMX_CTRL_WPS,
                                      kx, ky, kh, kl, kw, ks_, fetch, put);
      ho.n[gid] = n16;
      return;
    }
  }
  // Paths are numbered in the reference's order: seed lanes by index, branches depth-first.  Team lane p
  // walks seed lanes p, p + 4, ... and measures the first waypoint of every path it meets
  // (find_current_lane, lane_following_controller.py:367-374); counts exchanged by shuffles turn
  // (lane, branch) into the global number.
  int n_paths = 0;
  double my_d = SMX_INF;
  int my_idx = 0x7fffffff;
  int goff0 = 0, cnt0 = 0;
  if (seed.road >= 0) {
    for (int r4 = 0; r4 < seed.n_lanes; r4 += SMX_WP_LANES) {  // uniform within a team
      const int li = r4 + p0;
      int cnt = 0, bj = 0x7fffffff;
      double bd = SMX_INF;
      if (li < seed.n_lanes) {
        const int start = seed_start(m, seed, li, px, py);
        if (start >= 0) {
          BranchState bs;
          bs.reset();
          do {
            double fx = 0.0, fy = 0.0;
            equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, 1,
                                [&](int, const WaypointOut& w) {
                                  fx = w.x;
                                  fy = w.y;
                                });
            const double ex = fx - px, ey = fy - py;
            const double d = sqrt(ex * ex + ey * ey);
            if (d < bd) {  // strict: the lowest-numbered path wins ties (np.argmin)
              bd = d;
              bj = cnt;
            }
            ++cnt;
          } while (bs.advance());
        }
      }
      int incl = cnt;
      {
        int t = __shfl_up(incl, 1, SMX_WP_LANES);
        if (p0 >= 1) incl += t;
        t = __shfl_up(incl, 2, SMX_WP_LANES);
        if (p0 >= 2) incl += t;
      }
      const int round_total = __shfl(incl, SMX_WP_LANES - 1, SMX_WP_LANES);
      const int g = n_paths + incl - cnt;
      if (r4 == 0) {
        goff0 = g;
        cnt0 = cnt;
      }
      if (bj != 0x7fffffff && (bd < my_d || (bd == my_d && g + bj < my_idx))) {
        my_d = bd;
        my_idx = g + bj;
      }
      n_paths += round_total;
    }
  }
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) {
    const double od = __shfl_xor(my_d, msk, SMX_WP_LANES);
    const int oi = __shfl_xor(my_idx, msk, SMX_WP_LANES);
    if (od < my_d || (od == my_d && oi < my_idx)) {
      my_d = od;
      my_idx = oi;
    }
  }
  if (n_paths <= 0) {  // uniform within a team
    if (p0 == 0) ho.n[gid] = 0;
    return;
  }
  int want = my_idx + lane_change;
  want = want < 0 ? 0 : (want > n_paths - 1 ? n_paths - 1 : want);
  // the lane whose first seed lane holds path `want` walks to it again and writes its waypoints; a path of a
  // later seed lane (roads with more than four lanes) is found by lane 0 the long way
  const bool own = cnt0 > 0 && want >= goff0 && want < goff0 + cnt0;
  int owners = own ? (1 << p0) : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) owners |= __shfl_xor(owners, msk, SMX_WP_LANES);
  const int act_lane = owners ? (__ffs(owners) - 1) : 0;
  if (p0 != act_lane) return;
  int n = 0;
  if (owners) {
    const int start = seed_start(m, seed, p0, px, py);
    BranchState bs;
    bs.reset();
    int j = 0;
    do {
      if (j == want - goff0) {
        n = equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, SMX_CTRL_WPS, put);
        break;
      }
      equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, 0, [&](int, const WaypointOut&) {});
      ++j;
    } while (bs.advance());
  } else {
    int idx = 0;
    for (int li = 0; li < seed.n_lanes && n == 0; ++li) {
      const int start = seed_start(m, seed, li, px, py);
      if (start < 0) continue;
      BranchState bs;
      bs.reset();
      do {
        if (idx == want) {
          n = equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, SMX_CTRL_WPS, put);
          break;
        }
        equally_spaced_path(m, seed.f, bs, start, SMX_CTRL_WPS - 1, px, py, knots, SMX_BLOCK, 0, [&](int, const WaypointOut&) {});
        ++idx;
      } while (bs.advance());
    }
  }
  ho.n[gid] = n < SMX_CTRL_WPS ? n : SMX_CTRL_WPS;
}

// One team of four lanes per vehicle.  With a slow list in the arguments (large batches: the vehicles k_control_fast
// could not serve) a fixed grid strides that list, whose length is only known on the device.
// (the list form is a kernel of its own: with both forms in one kernel the role was inlined twice and the team cut's
// launch paid for it — 111 -> 137 registers here, 168 -> 256 in k_control_law, one wavefront per SIMD: C5's control
// phase 0.169 -> 0.209 ms)
template <int SPACE>
__global__ void __launch_bounds__(SMX_BLOCK) k_control_paths(const KernelArgs a, const CtrlHandoff ho) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  const size_t total = (size_t)a.cfg.num_envs * a.cfg.num_vehicles;
  control_paths_for<SPACE>(a, ho, launch_vehicle(a, ((size_t)blockIdx.x * SMX_BLOCK + threadIdx.x) / SMX_WP_LANES, total), knot_scratch + threadIdx.x);
}
template <int SPACE>
__global__ void __launch_bounds__(SMX_BLOCK) k_control_paths_listed(const KernelArgs a, const CtrlHandoff ho) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  constexpr int VPB = SMX_BLOCK / SMX_WP_LANES;
  const int count = *a.slow_count;
  for (int i = (int)blockIdx.x * VPB + (int)threadIdx.x / SMX_WP_LANES; i < count; i += (int)gridDim.x * VPB)
    control_paths_for<SPACE>(a, ho, (size_t)a.slow_list[i], knot_scratch + threadIdx.x);
}

// Control law + vehicle dynamics, one lane per vehicle (see k_control_paths).  Every action space; the
// lane-following ones read the wanted path from the hand-off.
template <int SPACE>
__device__ __forceinline__ void control_law_for(const KernelArgs& a, const CtrlHandoff& ho, const size_t gid) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  if (gid >= total) return;
  int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE)) return;
  if (flags & SMX_F_SOCIAL) {  // scripted lane follower: no controller, no dynamics
    int lane = (int)SF(SMX_S_MCL_X), crossed = (int)SF(SMX_S_SPD_INT);
    double offset = SF(SMX_S_MCL_Y), speed, x, y, heading;
    SF(SMX_S_PREV_X) = SF(SMX_S_X);
    SF(SMX_S_PREV_Y) = SF(SMX_S_Y);
    const double cmd = c.social_model == SMX_SOCIAL_IDM ? SF(SMX_S_THROTTLE) : -1.0;
    social_step(m, (int)(gid % c.num_vehicles), c.social_speed_factor, c.dt, lane, offset, crossed, speed, cmd);
    social_pose(m, lane, offset, x, y, heading);
    SF(SMX_S_X) = x;
    SF(SMX_S_Y) = y;
    SF(SMX_S_HEADING) = heading;
    SF(SMX_S_U) = speed;
    SF(SMX_S_MCL_X) = (double)lane;
    SF(SMX_S_MCL_Y) = offset;
    SF(SMX_S_SPD_INT) = (double)crossed;
    return;
  }
  VehState s = load_vehicle(a, gid, total);
  CtrlState cs;
  cs.lat_int = SF(SMX_S_LAT_INT);
  cs.spd_int = SF(SMX_S_SPD_INT);
  cs.steer = SF(SMX_S_STEER);
  cs.throttle = SF(SMX_S_THROTTLE);
  cs.spd_err = SF(SMX_S_SPD_ERR);
  cs.mcl_x = SF(SMX_S_MCL_X);
  cs.mcl_y = SF(SMX_S_MCL_Y);
  cs.mcl_set = (flags & SMX_F_MCL_SET) != 0;
  ControlOut co;
  // no action this tick: wheel torques do not persist, the steer motor target does
  co.throttle = 0.0;
  co.brake = 0.0;
  co.steering = cs.steer;
  constexpr bool lane_following = SPACE == SMX_ACTION_SPACE_LANE || SPACE == SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED;
  if (lane_following) {
    double target_speed, hg, lg;
    int lane_change;
    if (decode_lane_action<SPACE>(a, gid, target_speed, lane_change, hg, lg)) {
      const int n = ho.n[gid];
      if (n > 0) {
        CtrlPath path;
        path.n = n;
#pragma unroll
        for (int k = 0; k < SMX_CTRL_WPS; ++k) {
          const double* q = ho.path + (size_t)(k * 3) * total + gid;
          const bool held = k < n;
          path.x[k] = held ? q[0] : 0.0;
          path.y[k] = held ? q[total] : 0.0;
          path.h[k] = held ? q[2 * total] : 0.0;
        }
        co = lane_following_from_path(s, cs, c.dt, target_speed, lane_change, hg, lg, path);
      } else {
        // reference asserts "no waypoints found"; keep the last command
        co.throttle = cs.throttle;
        co.brake = 0.0;
        co.steering = cs.steer;
      }
    }
  } else if (SPACE == SMX_ACTION_SPACE_TRAJECTORY) {
    if (a.traj_n[gid] > 0) {
      PackedTraj t;
      t.p = a.traj + gid * (size_t)(4 * SMX_TRAJ_COLS);
      t.n = a.traj_n[gid];
      co = trajectory_tracking_pd(s, cs, c.dt, t);
    }
  } else {
    const float act0 = a.actions_f32[gid * 3 + 0], act1 = a.actions_f32[gid * 3 + 1], act2 = a.actions_f32[gid * 3 + 2];
    if (!(act0 != act0)) {  // NaN = no action
      if (SPACE == SMX_ACTION_SPACE_CONTINUOUS) {
        co.throttle = clip_ref((double)act0, 0.0, 1.0);
        co.brake = clip_ref((double)act1, 0.0, 1.0);
        co.steering = clip_ref((double)act2, -1.0, 1.0);
      } else {
        // ActuatorDynamicController.perform_action (actuator_dynamic_controller.py:47-80)
        const double change = clip_ref((double)act2, -1.0, 1.0);
        co.throttle = clip_ref((double)act0, 0.0, 1.0);
        co.brake = clip_ref((double)act1, 0.0, 1.0);
        co.steering = clip_ref((1.0 - 0.001) * cs.steer + change * c.dt, -1.0, 1.0);
      }
      cs.steer = co.steering;  // last_steering_angle / the persisting steer target
    }
  }
  SF(SMX_S_PREV_X) = s.x;  // the position recorded by the previous observation
  SF(SMX_S_PREV_Y) = s.y;
  vehicle_step(s, co, c.dt);
  SF(SMX_S_X) = s.x;
  SF(SMX_S_Y) = s.y;
  SF(SMX_S_HEADING) = s.heading;
  SF(SMX_S_U) = s.u;
  SF(SMX_S_V) = s.v;
  SF(SMX_S_R) = s.r;
  SF(SMX_S_DELTA) = s.delta;
  SF(SMX_S_LAT_INT) = cs.lat_int;
  SF(SMX_S_SPD_INT) = cs.spd_int;
  SF(SMX_S_STEER) = cs.steer;
  SF(SMX_S_THROTTLE) = cs.throttle;
  SF(SMX_S_SPD_ERR) = cs.spd_err;
  SF(SMX_S_MCL_X) = cs.mcl_x;
  SF(SMX_S_MCL_Y) = cs.mcl_y;
  a.st.flags[gid] = cs.mcl_set ? (flags | SMX_F_MCL_SET) : (flags & ~SMX_F_MCL_SET);
}

template <int SPACE>
__global__ void __launch_bounds__(SMX_BLOCK) k_control_law(const KernelArgs a, const CtrlHandoff ho) {
  control_law_for<SPACE>(a, ho, (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x);
}
template <int SPACE>
__global__ void __launch_bounds__(SMX_BLOCK) k_control_law_listed(const KernelArgs a, const CtrlHandoff ho) {
  const int count = *a.slow_count;  // the vehicles k_control_fast left to k_control_paths_listed (see there)
  for (int i = (int)blockIdx.x * SMX_BLOCK + (int)threadIdx.x; i < count; i += (int)gridDim.x * SMX_BLOCK)
    control_law_for<SPACE>(a, ho, (size_t)a.slow_list[i]);
}

// =================================================================================
// k_control_fast (large batches, lane-following action spaces): controller + dynamics with ONE lane per vehicle and
// no hand-off.  The controller's candidate paths start on the seeds the waypoints sensor walked last tick
// (k_control_paths explains the reuse): find_current_lane needs the start lanepoints only, and the wanted path's
// 17 waypoints are interpolated from its knot list straight into this lane's LDS column, read back with fixed
// indices by the control law.  k_control_paths wrote them to device memory for k_control_law to read back (115 MB
// each way at 131 k vehicles) with three of its four lanes idle during the interpolation.  A vehicle whose lists
// cannot be reused — a new vehicle, a branching inside the lookahead, a road of more than four lanes — goes to the
// slow list and through k_control_paths / k_control_law as before.
// =================================================================================
template <int SPACE>
__global__ void __launch_bounds__(SMX_BLOCK) k_control_fast(const KernelArgs a) {
  SMX_TSTAMP(span0);
  // the wanted path's waypoints, [element][lane]: 17 headings, then x and y of the first ten (lane_following_from_path
  // reads no position beyond waypoint 9).  19 KB: eight workgroups per CU, every vehicle of 131 k resident at once — at
  // 26 KB (all 51 elements) six fit, and the kernel ran a second round for a quarter of its workgroups.
  constexpr int CTRL_XY = 10;
  __shared__ double path_lds[(SMX_CTRL_WPS + 2 * CTRL_XY) * SMX_BLOCK];
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = launch_vehicle(a, (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x, total);
  bool slow = false;
  // ---- every word whose address only needs the vehicle, loaded together and whatever the flags say (two wavefronts
  // per SIMD hide nothing: flags -> action -> seeds -> list keys -> ... one behind the other was a microsecond each)
  const bool in_range = gid < total;
  const size_t g = in_range ? gid : 0;
  int flags = a.st.flags[g];
  int action = SMX_ACTION_NONE;
  float act0 = 0.f, act1 = 0.f;
  if (SPACE == SMX_ACTION_SPACE_LANE) {
    action = a.actions[g];
  } else {
    act0 = a.actions_f32[g * 3 + 0];
    act1 = a.actions_f32[g * 3 + 1];
  }
  VehState s;
  CtrlState cs;
  {
    const double* f = a.st.f64 + g;
    s.x = f[(size_t)SMX_S_X * total];
    s.y = f[(size_t)SMX_S_Y * total];
    s.heading = f[(size_t)SMX_S_HEADING * total];
    s.u = f[(size_t)SMX_S_U * total];
    s.v = f[(size_t)SMX_S_V * total];
    s.r = f[(size_t)SMX_S_R * total];
    s.delta = f[(size_t)SMX_S_DELTA * total];
    cs.lat_int = f[(size_t)SMX_S_LAT_INT * total];
    cs.spd_int = f[(size_t)SMX_S_SPD_INT * total];
    cs.steer = f[(size_t)SMX_S_STEER * total];
    cs.throttle = f[(size_t)SMX_S_THROTTLE * total];
    cs.spd_err = f[(size_t)SMX_S_SPD_ERR * total];
    cs.mcl_x = f[(size_t)SMX_S_MCL_X * total];
    cs.mcl_y = f[(size_t)SMX_S_MCL_Y * total];
  }
  const PathSeeds seed = load_seeds(a, g, total);
  const bool lists = a.knots.key != nullptr;
  const size_t paths = total * SMX_WP_LANES;
  int kn_n[SMX_WP_LANES], kn_key0[SMX_WP_LANES], kn_key1[SMX_WP_LANES], kn_key2[SMX_WP_LANES], kn_cnt[SMX_WP_LANES],
      kn_nk16[SMX_WP_LANES], kn_end16[SMX_WP_LANES];
#pragma unroll
  for (int q = 0; q < SMX_WP_LANES; ++q) {
    const size_t pth = g * SMX_WP_LANES + q;
    kn_n[q] = lists ? (int)a.knots.n[pth] : 0;
    kn_key0[q] = lists ? a.knots.key[pth] : -1;
    kn_key1[q] = lists ? a.knots.key[paths + pth] : -1;
    kn_key2[q] = lists ? a.knots.key[2 * paths + pth] : -1;
    kn_cnt[q] = lists ? (int)a.knots.cnt[pth] : 0;
    kn_nk16[q] = lists ? (int)a.knots.nk16[pth] : 0;
    kn_end16[q] = lists ? a.knots.end16[pth] : -1;
  }
  if (in_range && (flags & SMX_F_ALIVE)) {
    if (flags & SMX_F_SOCIAL) {  // scripted lane follower: no controller, no dynamics
      int lane = (int)cs.mcl_x, crossed = (int)cs.spd_int;
      double offset = cs.mcl_y, speed, x, y, heading;
      SF(SMX_S_PREV_X) = s.x;
      SF(SMX_S_PREV_Y) = s.y;
      const double cmd = c.social_model == SMX_SOCIAL_IDM ? cs.throttle : -1.0;
      social_step(m, (int)(gid % c.num_vehicles), c.social_speed_factor, c.dt, lane, offset, crossed, speed, cmd);
      social_pose(m, lane, offset, x, y, heading);
      SF(SMX_S_X) = x;
      SF(SMX_S_Y) = y;
      SF(SMX_S_HEADING) = heading;
      SF(SMX_S_U) = speed;
      SF(SMX_S_MCL_X) = (double)lane;
      SF(SMX_S_MCL_Y) = offset;
      SF(SMX_S_SPD_INT) = (double)crossed;
    } else {
      // Controllers.perform_action's decoding (decode_lane_action, on the words loaded above)
      double target_speed = 0.0, hg = 0.0, lg = 0.0;
      int lane_change = 0;
      bool has_action = false;
      if (SPACE == SMX_ACTION_SPACE_LANE) {
        if (action < SMX_ACTION_NONE || action > SMX_ACTION_CHANGE_LANE_RIGHT) {
          atomicOr(a.status, SMX_DEVICE_BAD_LANE_ACTION);  // reported at the next smx_sync; the code moves nothing
        } else if (action >= 0) {
          has_action = true;
          target_speed = action == SMX_ACTION_KEEP_LANE ? 15.0 : (action == SMX_ACTION_SLOW_DOWN ? 0.0 : 12.5);
          lane_change = action == SMX_ACTION_CHANGE_LANE_LEFT ? 1 : (action == SMX_ACTION_CHANGE_LANE_RIGHT ? -1 : 0);
          hg = target_speed > 0.0 ? a.heading_gain_pos : 0.01;
          lg = target_speed > 0.0 ? a.lateral_gain_pos : 0.36;
        }
      } else if (!(act0 != act0)) {  // NaN = no action
        has_action = true;
        target_speed = (double)act0;
        lane_change = (int)act1;
        lateral_gains_for_speed(target_speed, hg, lg);
      }
      CtrlPath path;
      path.n = 0;
      if (has_action && !SMX_SKIP(a, 1 << 26)) {
        // ---- the wanted path from the sensor's knot lists (k_control_paths' reuse, one lane doing the team's part)
        const double px = s.x, py = s.y;
        if (!lists || c.wp_lookahead < SMX_CTRL_WPS - 1 || (seed.road >= 0 && seed.n_lanes > SMX_WP_LANES)) {
          slow = true;
        } else if (seed.road >= 0) {
          const int f0 = seed.f.n > 0 ? seed.f.road[0] : -1, f1 = seed.f.n > 1 ? seed.f.road[1] : -1;
          int started = 0, qw = 0;
          double my_d = SMX_INF;
          int my_idx = 0;
          // the start lanepoints' records, together
          double rx[SMX_WP_LANES], ry[SMX_WP_LANES], rdx[SMX_WP_LANES], rdy[SMX_WP_LANES], rh[SMX_WP_LANES];
          int st[SMX_WP_LANES];
#pragma unroll
          for (int q = 0; q < SMX_WP_LANES; ++q) {
            st[q] = q < seed.n_lanes ? seed_start(m, seed, q, px, py) : -1;
            const smx_lp_rec* r = m.lp_rec + (st[q] >= 0 ? st[q] : 0);
            rx[q] = r->x;
            ry[q] = r->y;
            rdx[q] = r->dirx;
            rdy[q] = r->diry;
            rh[q] = r->heading;
          }
          // every seed lane's list must be this tick's; the nearest first waypoint (find_current_lane,
          // lane_following_controller.py:367-374: np.argmin, the lowest path number wins ties)
#pragma unroll
          for (int q = 0; q < SMX_WP_LANES; ++q) {
            if (st[q] >= 0) {
              const int n32 = kn_n[q];
              if (!(kn_key0[q] == st[q] && kn_key1[q] == f0 && kn_key2[q] == f1 && kn_cnt[q] == 1 && n32 > 0)) slow = true;
              const double proj = (px - rx[q]) * rdx[q] + (py - ry[q]) * rdy[q];
              const double fx = n32 == 1 ? rx[q] : rx[q] + proj * rdx[q], fy = n32 == 1 ? ry[q] : ry[q] + proj * rdy[q];
              const double ex = fx - px, ey = fy - py;
              const double d = sqrt(ex * ex + ey * ey);
              if (d < my_d) {
                my_d = d;
                my_idx = __popc(started);
              }
              started |= 1 << q;
            }
          }
          const int n_paths = __popc(started);
          if (!slow && n_paths > 0) {
            int want = my_idx + lane_change;
            want = want < 0 ? 0 : (want > n_paths - 1 ? n_paths - 1 : want);
            // the want-th started seed lane
#pragma unroll
            for (int q = 0; q < SMX_WP_LANES; ++q)
              if ((started >> q) & 1)
                if (__popc(started & ((1 << q) - 1)) == want) qw = q;
            int n32w = 0, nk16 = 0, last = -1;
            smx_lp_rec r0 = smx_lp_rec{};
#pragma unroll
            for (int q = 0; q < SMX_WP_LANES; ++q)
              if (q == qw) {
                n32w = kn_n[q];
                nk16 = kn_nk16[q];
                last = kn_end16[q];  // the last knot when it is not one of the list's
                r0.x = rx[q];
                r0.y = ry[q];
                r0.dirx = rdx[q];
                r0.diry = rdy[q];
                r0.heading = rh[q];
              }
            r0.lane = 0;  // (lane, width and speed limit of the waypoints are not the controller's business)
            const size_t pth = gid * SMX_WP_LANES + qw;
            const int n16 = n32w < SMX_CTRL_WPS ? n32w : SMX_CTRL_WPS;
            auto fetch = [&](int k) { return (k == nk16 - 1 && last >= 0) ? last : a.knots.idx[(size_t)(k + 1) * paths + pth]; };
            constexpr int KP = SMX_WPT_PRELOAD;
            double kx[KP], ky[KP], kh[KP], kw[KP], ks_[KP];
            int kl[KP];
            {
              int kid[KP];
#pragma unroll
              for (int k = 0; k < KP; ++k) kid[k] = a.knots.idx[(size_t)(k + 1) * paths + pth];  // (in range whatever nk16 is)
#pragma unroll
              for (int k = 0; k < KP; ++k) {
                const bool have = k < nk16;
                const int id = (k == nk16 - 1 && last >= 0) ? last : kid[k];
                const smx_lp_rec* r = m.lp_rec + (have ? id : 0);
                kx[k] = have ? r->x : 0.0;
                ky[k] = have ? r->y : 0.0;
                kh[k] = have ? r->heading : 0.0;
                kl[k] = 0;
                kw[k] = 0.0;
                ks_[k] = 0.0;
              }
            }
            double D = 0.0;
            {
              const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
              double lastx = r0.x + proj * r0.dirx, lasty = r0.y + proj * r0.diry;
#pragma unroll
              for (int k = 0; k < KP; ++k) {
                if (k < nk16) {
                  const double ex = kx[k] - lastx, ey = ky[k] - lasty;
                  D += sqrt(ex * ex + ey * ey);
                  lastx = kx[k];
                  lasty = ky[k];
                }
              }
              for (int k = KP; k < nk16; ++k) {
                const smx_lp_rec* r = m.lp_rec + fetch(k);
                const double qx = r->x, qy = r->y;
                const double ex = qx - lastx, ey = qy - lasty;
                D += sqrt(ex * ex + ey * ey);
                lastx = qx;
                lasty = qy;
              }
            }
            double* col = path_lds + threadIdx.x;
            interpolate_knots_preloaded<KP>(m, r0, 0.0, 0.0, nk16, n16, D, px, py, SMX_CTRL_WPS, kx, ky, kh, kl, kw, ks_, fetch,
                                            [&](int i, const WaypointOut& w) {
                                              col[(size_t)i * SMX_BLOCK] = w.heading;
                                              if (i < CTRL_XY) {
                                                col[(size_t)(SMX_CTRL_WPS + i) * SMX_BLOCK] = w.x;
                                                col[(size_t)(SMX_CTRL_WPS + CTRL_XY + i) * SMX_BLOCK] = w.y;
                                              }
                                            });
            path.n = n16;
#pragma unroll
            for (int k = 0; k < SMX_CTRL_WPS; ++k) {
              const bool held = k < n16;
              path.h[k] = held ? col[(size_t)k * SMX_BLOCK] : 0.0;
              path.x[k] = (held && k < CTRL_XY) ? col[(size_t)(SMX_CTRL_WPS + (k < CTRL_XY ? k : 0)) * SMX_BLOCK] : 0.0;
              path.y[k] = (held && k < CTRL_XY) ? col[(size_t)(SMX_CTRL_WPS + CTRL_XY + (k < CTRL_XY ? k : 0)) * SMX_BLOCK] : 0.0;
            }
          }
        }
      }
      if (!slow) {
        cs.mcl_set = (flags & SMX_F_MCL_SET) != 0;
        ControlOut co;
        // no action this tick: wheel torques do not persist, the steer motor target does
        co.throttle = 0.0;
        co.brake = 0.0;
        co.steering = cs.steer;
        if (has_action) {
          if (path.n > 0 && !SMX_SKIP(a, 1 << 28)) {
            co = lane_following_from_path(s, cs, c.dt, target_speed, lane_change, hg, lg, path);
          } else {
            // reference asserts "no waypoints found"; keep the last command
            co.throttle = cs.throttle;
            co.brake = 0.0;
            co.steering = cs.steer;
          }
        }
        SF(SMX_S_PREV_X) = s.x;  // the position recorded by the previous observation
        SF(SMX_S_PREV_Y) = s.y;
        if (!SMX_SKIP(a, 1 << 27)) vehicle_step(s, co, c.dt);
        SF(SMX_S_X) = s.x;
        SF(SMX_S_Y) = s.y;
        SF(SMX_S_HEADING) = s.heading;
        SF(SMX_S_U) = s.u;
        SF(SMX_S_V) = s.v;
        SF(SMX_S_R) = s.r;
        SF(SMX_S_DELTA) = s.delta;
        SF(SMX_S_LAT_INT) = cs.lat_int;
        SF(SMX_S_SPD_INT) = cs.spd_int;
        SF(SMX_S_STEER) = cs.steer;
        SF(SMX_S_THROTTLE) = cs.throttle;
        SF(SMX_S_SPD_ERR) = cs.spd_err;
        SF(SMX_S_MCL_X) = cs.mcl_x;
        SF(SMX_S_MCL_Y) = cs.mcl_y;
        a.st.flags[gid] = cs.mcl_set ? (flags | SMX_F_MCL_SET) : (flags & ~SMX_F_MCL_SET);
      }
    }
  }
  // the wavefront's slow vehicles, appended with one atomic
  const unsigned long long mask = __ballot(slow);
  if (mask != 0ull) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(a.slow_count, __popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1);
    if (slow) a.slow_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)gid;
  }
  SMX_TSTAMP(span1);
  SMX_TSPAN(0, span0, span1);
}

// =================================================================================
// k_social (SMX_SOCIAL_IDM only): car following of the scripted social vehicles, one thread per
// vehicle, before k_control moves anything: every follower reads its env-mates' poses and speeds as
// they stand at the start of the tick and leaves its speed for the tick in SMX_S_THROTTLE (unused by a
// kinematic vehicle).  The arithmetic is oracle/sim.py::SocialBody.idm_speed.
// =================================================================================
__global__ void __launch_bounds__(SMX_BLOCK) k_social(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x;
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE) || !(flags & SMX_F_SOCIAL)) return;
  const int n_veh = c.num_vehicles;
  const size_t env0 = (gid / n_veh) * n_veh;
  const int slot = (int)(gid - env0);
  const double x = SF(SMX_S_X), y = SF(SMX_S_Y), h = SF(SMX_S_HEADING), v = SF(SMX_S_U);
  const double v0 = m.lane_speed[(int)SF(SMX_S_MCL_X)] * c.social_speed_factor;
  const double fx = -sin(h), fy = cos(h), rx = cos(h), ry = sin(h);
  double best = 60.0, lead_u = 0.0;
  bool found = false;
  for (int j = 0; j < n_veh; ++j) {
    if (j == slot) continue;
    const size_t og = env0 + j;
    if (!(a.st.flags[og] & SMX_F_ALIVE)) continue;
    const double dx = a.st.f64[(size_t)SMX_S_X * total + og] - x, dy = a.st.f64[(size_t)SMX_S_Y * total + og] - y;
    const double lon = dx * fx + dy * fy, lat = dx * rx + dy * ry;
    if (lon > 0.0 && lon < best && fabs(lat) < 1.6) {
      best = lon;
      lead_u = a.st.f64[(size_t)SMX_S_U * total + og];
      found = true;
    }
  }
  double v_new;
  if (v0 <= 0.0) {
    v_new = fmax(0.0, v - 4.5 * c.dt);
  } else {
    const double ratio = v / v0;
    const double free_term = 1.0 - (ratio * ratio) * (ratio * ratio);
    double inter = 0.0;
    if (found) {
      const double gap = fmax(best - SMX_CHASSIS_LENGTH, 0.1);
      const double dv = v - lead_u;
      const double sstar = 2.5 + fmax(0.0, v * 1.0 + v * dv / (2.0 * sqrt(2.6 * 4.5)));
      const double q = sstar / gap;
      inter = q * q;
    }
    const double acc = 2.6 * (free_term - inter);
    v_new = fmin(fmax(v + acc * c.dt, 0.0), v0);
  }
  SF(SMX_S_THROTTLE) = v_new;
}

// =================================================================================
// k_scan: map sweeps at the vehicle's pose, SMX_TEAM lanes per vehicle.
//   facts: nearest lane + distance (nearest_lane with any radius up to 10 m is "that lane if
//          closer than r": ego lane sensors.py:277, neighbour lanes :244, off-route :552),
//          road_with_point at the centre (:498-500) and at the four bounding-box corners
//          (:502-509; Vehicle.bounding_box vehicle.py:315-332, rotate_around_point math.py:436-444)
//   seeds: start road / route filter / start lanepoints of waypoint_paths(pose, route)
//          (sumo_road_network.py:815-882), used by the waypoints role now and by k_control next tick
// =================================================================================
// one half of k_scan for one vehicle team (see k_scan)
template <int TEAM, bool ROUTED>
__device__ __forceinline__ void scan_role(const KernelArgs& a, const MapDev& m, const smx_config& c, size_t gid,
                                          size_t total, int rank, int flags, int role) {
  SMX_TSTAMP(ts0);
  const VehState s = load_vehicle(a, gid, total);
  int32_t* fi = a.st.facts_i32;
  if (SMX_SKIP(a, 512) && role == 1) return;
  if (SMX_SKIP(a, 1024) && role == 0) return;
  if (role == 0) {
    // ---- road facts
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
    const bool social = (flags & SMX_F_SOCIAL) != 0;  // only its nearest lane is ever asked for (neighbour rows)
    FactsCarry fc;
    fc.valid = false;
    if (a.facts_carry != nullptr && !(flags & SMX_F_FIRST) && !SMX_SCAN_UNSEEDED) {
      fc.qx = a.facts_carry[gid];
      fc.qy = a.facts_carry[total + gid];
      fc.prev_dist = a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + gid];
      fc.valid = fi[(size_t)SMX_FI_LANE * total + gid] >= 0;
    }
    RoadFacts h = team_road_facts_seeded<TEAM>(m, s.x, s.y, SMX_POSE_SCAN_RADIUS, social ? 0 : 4, cx, cy, fc, a.dagm_reach + 0.1);
    if (a.facts_carry != nullptr && rank == 0) {
      a.facts_carry[gid] = s.x;
      a.facts_carry[total + gid] = s.y;
    }
    SMX_TSTAMP(ts1);
    SMX_TACC(10, ts0, ts1);
    // wrong-way test input (sensors.py:556-562, 581-586): the lane heading at the point of the
    // nearest lane closest to the vehicle; junction lanes are exempt (:548-551)
    double lane_heading = 0.0;
    if (SMX_SKIP(a, 2048)) return;
    const bool want_heading = !social && h.lane >= 0 && !m.lane_in_junction[h.lane] && !SMX_SKIP(a, 64);  // uniform in the team
    if (want_heading) lane_heading = team_lane_heading_at_point<TEAM>(m, h.lane, s.x, s.y, h.dist);
    SMX_TSTAMP(ts2);
    SMX_TACC(11, ts1, ts2);
    if (rank == 0) {
      fi[(size_t)SMX_FI_LANE * total + gid] = h.lane;
      fi[(size_t)SMX_FI_FLAGS * total + gid] =
          (h.on_road ? SMX_FACT_ON_ROAD : 0) | ((h.corner_mask & 15) << SMX_FACT_CORNER_SHIFT);
      a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + gid] = h.dist;
      a.st.facts_f64[(size_t)SMX_FF_LANE_HEADING * total + gid] = lane_heading;
    }
    return;
  }
  // ---- path seeds
  if (flags & SMX_F_SOCIAL) return;
  SMX_TSTAMP(ts3);
  Top10 t;
  LaneGuess guess;
  SeedsCarry scy;
  scy.valid = false;
  if (a.seeds_carry != nullptr && !(flags & SMX_F_FIRST) && !SMX_SCAN_UNSEEDED) {
    const int32_t* sc_ = a.st.seed_cache;
    scy.qx = a.seeds_carry[gid];
    scy.qy = a.seeds_carry[total + gid];
    scy.d10 = a.seeds_carry[2 * total + gid];
    scy.d1 = a.seeds_carry[3 * total + gid];
    scy.prev_road = sc_[0 * total + gid];
    scy.prev_lanes = sc_[4 * total + gid];
#pragma unroll
    for (int q = 0; q < SMX_SEED_LANES; ++q) scy.prev_start[q] = sc_[(size_t)(5 + q) * total + gid];
    scy.valid = true;
  }
  const bool wp_on = (c.sensors & SMX_SENSOR_WAYPOINTS) != 0;
  team_nearest10_carried<TEAM>(m, s.x, s.y, scy, t, guess);
  if (a.seeds_carry != nullptr && rank == 0) {
    a.seeds_carry[gid] = s.x;
    a.seeds_carry[total + gid] = s.y;
    a.seeds_carry[2 * total + gid] = t.idx[9] >= 0 ? t.d2[9] : -1.0;
    a.seeds_carry[3 * total + gid] = t.idx[0] >= 0 ? t.d2[0] : -1.0;
  }
  SMX_TSTAMP(ts4);
  SMX_TACC(12, ts3, ts4);
  // what the controller (and the waypoints sensor) ask: paths at this pose with the agent's route
  if (SMX_SKIP(a, 4096)) return;
  Top10Scores sc;
  if (SMX_SKIP(a, 256)) {
#pragma unroll
    for (int k = 0; k < 10; ++k) sc.rel[k] = 0.0;
  } else {
    sc = team_top10_heading_terms<TEAM>(m, t, s.heading);
  }
  SMX_TSTAMP(ts4b);
  SMX_TACC(9, ts4, ts4b);
  const PathSeeds seed = team_compute_path_seeds<TEAM, ROUTED>(m, s.x, s.y, s.heading, 5.0, true, t, sc, a.missions, (int)(gid % (size_t)c.num_vehicles), &guess);
  // without the waypoints sensor the observation still takes the first waypoint of
  // waypoint_paths(pose, lookahead=1, within_radius=length) for the trip meter (sensors.py:270-275,
  // 349-351); TripMeterSensor.__init__ (sensors.py:885-898) asks the same on a new vehicle
  int obs_start = -1, trip_start = -1;
  if (!wp_on || (flags & SMX_F_FIRST)) {
    const PathSeeds ts = team_compute_path_seeds<TEAM, false>(m, s.x, s.y, s.heading, SMX_CHASSIS_LENGTH, false, t, sc, a.missions, 0, &guess);
    trip_start = (ts.road >= 0) ? ts.start[0] : -1;
    obs_start = trip_start;
  }
  SMX_TSTAMP(ts5);
  SMX_TACC(13, ts4, ts5);
  if (rank == 0) {
    store_seeds(a, gid, total, seed);
    fi[(size_t)SMX_FI_TRIP_START * total + gid] = (flags & SMX_F_FIRST) ? trip_start : -1;
    fi[(size_t)SMX_FI_OBS_START * total + gid] = obs_start;
  }
  SMX_TSTAMP(ts6);
  SMX_TACC(14, ts0, ts6);
}

// Register budgets (amdgpu_waves_per_eu): the split form runs on small batches, where at most two or
// three wavefronts per SIMD exist anyway, and takes the ~160 registers it wants — capped at 128 it
// spilled 136 B per lane and cost 10 us of 48 at 8 k vehicles, plus 9 MB of scratch write-back per
// tick.  The back-to-back form (large batches) is capped at 128 registers: four wavefronts per
// SIMD; five (96 registers, 276 B of spills) was 1.5x slower at 131 k vehicles.
#ifndef SMX_SCAN_WAVES
#define SMX_SCAN_WAVES 3
#endif
// ROUTED: the instance that knows fixed routes (smx_set_missions); batches without missions run the other one
template <bool SPLIT, bool ROUTED = false>
__global__ void __attribute__((amdgpu_waves_per_eu(SPLIT ? 2 : SMX_SCAN_WAVES, 8))) __launch_bounds__(SMX_BLOCK) k_scan(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  // the two halves of the scan are independent: on small batches they run as different workgroups
  // of one launch (even: road facts + lane heading, odd: lanepoint search + path seeds) and
  // overlap in time; on large ones every workgroup does both, one after the other
  const size_t gid = ((size_t)(SPLIT ? (blockIdx.x >> 1) : blockIdx.x) * SMX_BLOCK + threadIdx.x) / SMX_TEAM;
  const int rank = team_rank<SMX_TEAM>();
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE)) return;
  if (a.first_only && !(flags & SMX_F_FIRST)) return;
  if (SPLIT) {
    if (blockIdx.x & 1)
      scan_role<SMX_TEAM, ROUTED>(a, m, c, gid, total, rank, flags, 1);
    else
      scan_role<SMX_TEAM, ROUTED>(a, m, c, gid, total, rank, flags, 0);
  } else {
    scan_role<SMX_TEAM, ROUTED>(a, m, c, gid, total, rank, flags, 0);
    scan_role<SMX_TEAM, ROUTED>(a, m, c, gid, total, rank, flags, 1);
  }
}

// One half of the scan as a launch of its own (large batches): the road facts feed the observe role only and
// the path seeds the waypoint kernels only, so the two go to different streams and each keeps the registers
// it needs (the facts half alone fits more wavefronts per SIMD than the pair).
// (TEAM: four lanes a vehicle when the batch fills the chip, eight up to SMX_SCAN_WIDE_MAX_VEHICLES on maps whose lanes
// split: a quarter-full chip is bound by one team's latency, and the wider team's is the shorter)
template <int ROLE, bool ROUTED = false, int TEAM = SMX_TEAM_LARGE>
__global__ void __attribute__((amdgpu_waves_per_eu(ROLE == 0 ? 4 : 3, 8))) __launch_bounds__(SMX_BLOCK) k_scan_half(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = launch_vehicle(a, ((size_t)blockIdx.x * SMX_BLOCK + threadIdx.x) / TEAM, total);
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE)) return;
  if (a.first_only && !(flags & SMX_F_FIRST)) return;
  scan_role<TEAM, ROUTED>(a, a.map, c, gid, total, team_rank<TEAM>(), flags, ROLE);
}

// k_scan_fast (large batches): one half of the scan with ONE lane per vehicle (smx_scan.h facts_one_lane /
// seeds_one_lane: searches seeded from last tick's answers, two passes over per-lane candidate lists).  A vehicle it
// cannot serve — no usable carry, a list overflow, the in-junction rule, stacked lanes — is appended to the slow list
// and served by k_scan_half's teams afterwards: one such vehicle would otherwise hold its whole wavefront for the
// length of the searches from scratch.
template <int ROLE>
__global__ void __launch_bounds__(SMX_BLOCK) k_scan_fast(const KernelArgs a) {
  SMX_TSTAMP(span0);
  __shared__ int cand_lds[(ROLE == 0 ? SMX_FACTS_CAND : SMX_SEEDS_CAND) * SMX_BLOCK];
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = launch_vehicle(a, (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x, total);
  bool slow = false;
  // every word whose address only needs the vehicle is loaded here, together and whatever the flags say (one round
  // trip instead of flags -> pose -> carry one behind the other: two wavefronts per SIMD hide nothing)
  const bool in_range = gid < total;
  const size_t g = in_range ? gid : 0;
  const int flags = a.st.flags[g];
  const double sx_ = a.st.f64[(size_t)SMX_S_X * total + g], sy_ = a.st.f64[(size_t)SMX_S_Y * total + g],
               sh_ = a.st.f64[(size_t)SMX_S_HEADING * total + g];
  int32_t* fi = a.st.facts_i32;
  FactsCarry fc;
  SeedsCarry scy;
  fc.valid = false;
  scy.valid = false;
  int prev_lane = -1;
  if (ROLE == 0) {
    fc.qx = a.facts_carry[g];
    fc.qy = a.facts_carry[total + g];
    fc.prev_dist = a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + g];
    prev_lane = fi[(size_t)SMX_FI_LANE * total + g];
  } else {
    const int32_t* sc_ = a.st.seed_cache;
    scy.qx = a.seeds_carry[g];
    scy.qy = a.seeds_carry[total + g];
    scy.d10 = a.seeds_carry[2 * total + g];
    scy.d1 = a.seeds_carry[3 * total + g];
    scy.prev_road = sc_[0 * total + g];
    scy.prev_lanes = sc_[4 * total + g];
#pragma unroll
    for (int q = 0; q < SMX_SEED_LANES; ++q) scy.prev_start[q] = sc_[(size_t)(5 + q) * total + g];
  }
  if (in_range && (flags & SMX_F_ALIVE) && (!a.first_only || (flags & SMX_F_FIRST))) {
    int* cand = cand_lds + threadIdx.x;
    const bool seeded = !(flags & SMX_F_FIRST) && !SMX_SCAN_UNSEEDED;
    if (ROLE == 0) {
      const double cxs[4] = {-0.5, 0.5, 0.5, -0.5};
      const double cys[4] = {0.5, 0.5, -0.5, -0.5};
      double cx[4], cy[4];
      const double ch = cos(sh_), sh = sin(sh_);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double qx = sx_ + cxs[q] * SMX_CHASSIS_WIDTH;
        double qy = sy_ + cys[q] * SMX_CHASSIS_LENGTH;
        cx[q] = sx_ + ch * (qx - sx_) + sh * (qy - sy_);
        cy[q] = sy_ + -sh * (qx - sx_) + ch * (qy - sy_);
      }
      const bool social = (flags & SMX_F_SOCIAL) != 0;
      fc.valid = seeded && prev_lane >= 0;
      RoadFacts h;
      double lane_heading = 0.0;
      bool served;
      if (SMX_SKIP(a, 1 << 23)) {  // (developer ablation: the prologue and the stores only)
        h.lane = prev_lane;
        h.dist = fc.prev_dist + cx[0] * 1e-30;
        h.on_road = true;
        h.corner_mask = 15;
        served = true;
      } else {
        served = facts_one_lane(m, sx_, sy_, SMX_POSE_SCAN_RADIUS, social ? 0 : 4, cx, cy, fc, a.dagm_reach + 0.1, cand, SMX_BLOCK,
                                !social, h, lane_heading);
      }
      if (served) {
        a.facts_carry[gid] = sx_;
        a.facts_carry[total + gid] = sy_;
        fi[(size_t)SMX_FI_LANE * total + gid] = h.lane;
        fi[(size_t)SMX_FI_FLAGS * total + gid] =
            (h.on_road ? SMX_FACT_ON_ROAD : 0) | ((h.corner_mask & 15) << SMX_FACT_CORNER_SHIFT);
        a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + gid] = h.dist;
        a.st.facts_f64[(size_t)SMX_FF_LANE_HEADING * total + gid] = lane_heading;
      } else {
        slow = true;
      }
    } else if (!(flags & SMX_F_SOCIAL)) {
      scy.valid = seeded;
      PathSeeds one;
      double d1sq = -1.0;
      if (seeds_one_lane(m, sx_, sy_, sh_, 5.0, scy, cand, SMX_BLOCK, one, d1sq)) {
        a.seeds_carry[gid] = sx_;
        a.seeds_carry[total + gid] = sy_;
        a.seeds_carry[2 * total + gid] = -1.0;  // (the tenth nearest was not looked for)
        a.seeds_carry[3 * total + gid] = d1sq;
        store_seeds(a, gid, total, one);
        fi[(size_t)SMX_FI_TRIP_START * total + gid] = -1;  // (only a new vehicle or a batch without the waypoints
        fi[(size_t)SMX_FI_OBS_START * total + gid] = -1;   //  sensor asks these: neither comes here)
      } else {
        // its seeds, walks and rows are the slow chain's (k_scan_listed -> k_waypoints_listed, beside the tick's main
        // chain): k_wp_walk and k_waypoints_emit pass over the vehicle
        slow = true;
      }
      if (a.seed_pending != nullptr) a.seed_pending[gid] = slow ? 1 : 0;
    }
  }
  // the wavefront's slow vehicles, appended with one atomic
  const unsigned long long mask = __ballot(slow);
  if (mask != 0ull) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(a.slow_count, __popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1);
    if (slow) a.slow_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)gid;
  }
  SMX_TSTAMP(span1);
  SMX_TSPAN(ROLE == 1 ? 1 : 2, span0, span1);
}

// k_scan_half over a list whose length is only known on the device (the slow list): a fixed grid, teams striding it
// (TEAM: four lanes a vehicle where the lists are long — maps whose lanes split —, eight where they hold a few
// hundred vehicles and the kernel is one team's latency at the end of the slow chain: 70 against 55 us)
template <int ROLE, bool ROUTED = false, int TEAM = SMX_TEAM_LARGE>
__global__ void __attribute__((amdgpu_waves_per_eu(ROLE == 0 ? 4 : 3, 8))) __launch_bounds__(SMX_BLOCK) k_scan_listed(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int count = *a.slow_count;
  constexpr int VPB = SMX_BLOCK / TEAM;
  for (int i = (int)blockIdx.x * VPB + (int)threadIdx.x / TEAM; i < count; i += (int)gridDim.x * VPB) {
    const size_t gid = (size_t)a.slow_list[i];
    const int flags = a.st.flags[gid];
    scan_role<TEAM, ROUTED>(a, a.map, c, gid, total, team_rank<TEAM>(), flags, ROLE);
  }
}

// =================================================================================
// waypoints role: waypoint paths (sensors.py:268-275, 972-985) + trip meter (sensors.py:880-947).
// SMX_WP_LANES lanes per vehicle.  Team lane p takes seed lane p and writes its first path straight
// into the dense rows at the slot it would have if no lower seed lane branches (its provisional
// number); the team then exchanges the real counts.  Almost always that guess was right and every
// lane has walked exactly one path.  Otherwise (a branching inside the lookahead, or a road with
// more than four lanes) the team numbers the paths the long way and rewrites the rows.
// Rows are written whole every tick (unused waypoints and paths as zeros, format_obs.py:589-596),
// each lane streaming its own row in order, so L2 assembles full lines before they leave.
// =================================================================================
struct WpRows {
  double* pos;
  float *heading, *width, *speed;
  int16_t* lid;
  int8_t* lidx;
  int cached_lane;  // lane whose index is held in cached_index (-1: none): consecutive waypoints
  int cached_index; // mostly share their lane, and a look-up per waypoint stalls its own store
};

__device__ __forceinline__ WpRows wp_rows(const smx_outputs& o, size_t gid, int P, int W, int slot) {
  const size_t q = (gid * P + slot) * (size_t)W;
  WpRows r;
  r.pos = o.wp_pos + q * 3;
  r.heading = o.wp_heading + q;
  r.width = o.wp_lane_width + q;
  r.speed = o.wp_speed_limit + q;
  r.lid = o.wp_lane_id + q;
  r.lidx = o.wp_lane_index + q;
  r.cached_lane = -1;
  r.cached_index = 0;
  return r;
}

__device__ __forceinline__ void wp_put(const MapDev& m, WpRows& r, int i, const WaypointOut& w) {
  if (r.pos == nullptr) {  // developer switch (SMX_DEBUG_SKIP & 32768): compute, do not store
    if (i == 0x7fffffff) r.cached_index = (int)(w.x + w.y + w.heading + w.width + w.speed) + w.lane;
    return;
  }
  if (w.lane != r.cached_lane) {
    r.cached_lane = w.lane;
    r.cached_index = m.lane_index[w.lane];
  }
  r.pos[i * 3 + 0] = w.x;
  r.pos[i * 3 + 1] = w.y;
  r.pos[i * 3 + 2] = 0.0;
  r.heading[i] = (float)w.heading;
  r.width[i] = (float)w.width;
  r.speed[i] = (float)w.speed;
  r.lid[i] = (int16_t)w.lane;
  r.lidx[i] = (int8_t)r.cached_index;
}

__device__ __forceinline__ void wp_zero(const WpRows& r, int from, int W) {
  if (r.pos == nullptr) return;
  for (int i = from; i < W; ++i) {
    r.pos[i * 3 + 0] = 0.0;
    r.pos[i * 3 + 1] = 0.0;
    r.pos[i * 3 + 2] = 0.0;
    r.heading[i] = 0.0f;
    r.width[i] = 0.0f;
    r.speed[i] = 0.0f;
    r.lid[i] = -1;
    r.lidx[i] = 0;
  }
}

// `knots`: this thread's column of a [SMX_MAX_KNOTS][KSTRIDE] LDS scratch owned by the kernel
template <int KSTRIDE>
__device__ __forceinline__ void waypoints_for(const KernelArgs& a, const size_t gid, int* knots) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const smx_outputs& o = a.out;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int p0 = threadIdx.x % SMX_WP_LANES;
  if (gid >= total) return;  // whole teams leave together
  if (SMX_SKIP(a, 16384)) return;
  SMX_TSTAMP(tw0);
  int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE) || (flags & SMX_F_SOCIAL) || (a.first_only && !(flags & SMX_F_FIRST))) return;
  const bool wp_on = (c.sensors & SMX_SENSOR_WAYPOINTS) != 0;
  const int P = c.wp_paths, W = c.wp_len;
  const VehState s = load_vehicle(a, gid, total);
  const double px = s.x, py = s.y;

  bool have_first_wp = false;  // first waypoint of path 0 (trip meter), valid on team lane 0
  double fwx = 0, fwy = 0, fwh = 0;
  if (!wp_on) {
    // only the first waypoint of the first path is needed (trip meter)
    const int os = a.st.facts_i32[(size_t)SMX_FI_OBS_START * total + gid];
    if (p0 == 0 && os >= 0) {
      BranchState bs;
      bs.reset();
      RouteFilter nof;
      nof.none();
      equally_spaced_path(m, nof, bs, os, 1, px, py, knots, KSTRIDE, 1, [&](int, const WaypointOut& w) {
        have_first_wp = true;
        fwx = w.x;
        fwy = w.y;
        fwh = w.heading;
      });
    }
  } else {
    const PathSeeds seed = load_seeds(a, gid, total);
    const int lookahead = c.wp_lookahead;
    int n_paths_total = 0;
    SMX_TSTAMP(tw1);
    SMX_TACC(0, tw0, tw1);
    if (seed.road >= 0 && !SMX_SKIP(a, 16)) {
      // ---- the guess: seed lane p holds exactly one path
      const int start = (p0 < seed.n_lanes) ? seed_start(m, seed, p0, px, py) : -1;
      int started = start >= 0 ? (1 << p0) : 0;
#pragma unroll
      for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) started |= __shfl_xor(started, msk, SMX_WP_LANES);
      const int prov = __popc(started & ((1 << p0) - 1));
      int cnt = 0;
      double gx = 0, gy = 0, gh = 0;  // first waypoint of this lane's first path
      if (start >= 0) {
        BranchState bs;
        bs.reset();
        do {
          if (cnt == 0 && prov < P) {
            WpRows rows = wp_rows(o, gid, P, W, prov);
            if (SMX_SKIP(a, 32768)) rows.pos = nullptr;
            const int n = equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, knots, KSTRIDE, W,
                                              [&](int i, const WaypointOut& w) {
                                                if (i == 0) {
                                                  gx = w.x;
                                                  gy = w.y;
                                                  gh = w.heading;
                                                }
                                                wp_put(m, rows, i, w);
                                              });
            wp_zero(rows, n < W ? n : W, W);
            o.wp_count[gid * (P + 1) + 1 + prov] = (uint8_t)(n < W ? n : W);
          } else {
            equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, knots, KSTRIDE, 0,
                                [&](int, const WaypointOut&) {});
          }
          ++cnt;
        } while (bs.advance());
      }
      int branching = (cnt > 1) ? 1 : 0;
#pragma unroll
      for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) branching |= __shfl_xor(branching, msk, SMX_WP_LANES);
      if (!branching && seed.n_lanes <= SMX_WP_LANES) {
        // ---- the guess held: path numbers are the provisional ones
        n_paths_total = __popc(started);
        const int src = started ? (__ffs(started) - 1) : 0;
        fwx = __shfl(gx, src, SMX_WP_LANES);
        fwy = __shfl(gy, src, SMX_WP_LANES);
        fwh = __shfl(gh, src, SMX_WP_LANES);
        have_first_wp = n_paths_total > 0;
      } else if (seed.n_lanes <= SMX_WP_LANES) {
        // ---- a lane branches inside the lookahead: paths are numbered lanes by index, branches
        // depth-first, i.e. lane p's paths follow those of the lower lanes.  The counts are known from
        // the first pass, so an exclusive prefix over the team gives every lane the numbers of its
        // own paths, and each lane writes its own branches again, now into the right rows.  Only a
        // first path whose provisional row was already the right one stays as written: provisional
        // rows are distinct (a lane's rank among the started lanes), so nobody else wrote there in the
        // first pass, and whoever owns another lane's stale provisional row rewrites it here, later.
        int incl = cnt;
        {
          int t = __shfl_up(incl, 1, SMX_WP_LANES);
          if (p0 >= 1) incl += t;
          t = __shfl_up(incl, 2, SMX_WP_LANES);
          if (p0 >= 2) incl += t;
        }
        n_paths_total = __shfl(incl, SMX_WP_LANES - 1, SMX_WP_LANES);
        const int base = incl - cnt;
        const bool first_in_place = base == prov;
        if (start >= 0 && base < P && !(cnt == 1 && first_in_place)) {
          BranchState bs;
          bs.reset();
          int idx = base;
          do {
            if (idx >= P) break;
            if (idx == base && first_in_place) {
              // only walked, to learn the branchings the enumeration continues from
              equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, knots, KSTRIDE, 0,
                                  [&](int, const WaypointOut&) {});
            } else {
              WpRows rows = wp_rows(o, gid, P, W, idx);
              const int n = equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, knots, KSTRIDE, W,
                                                [&](int i, const WaypointOut& w) { wp_put(m, rows, i, w); });
              wp_zero(rows, n < W ? n : W, W);
              o.wp_count[gid * (P + 1) + 1 + idx] = (uint8_t)(n < W ? n : W);
            }
            ++idx;
          } while (bs.advance());
        }
        const int src = started ? (__ffs(started) - 1) : 0;  // path 0 is the lowest started lane's first path
        fwx = __shfl(gx, src, SMX_WP_LANES);
        fwy = __shfl(gy, src, SMX_WP_LANES);
        fwh = __shfl(gh, src, SMX_WP_LANES);
        have_first_wp = n_paths_total > 0;
      } else {
        // ---- roads with more than four lanes: number the paths the long way (lanes by index, branches
        // depth-first): every lane walks every path to discover the branchings, lane (idx % 4) writes kept path idx
        int idx = 0;
        for (int li = 0; li < seed.n_lanes; ++li) {
          const int st = seed_start(m, seed, li, px, py);
          if (st < 0) continue;
          BranchState bs;
          bs.reset();
          do {
            const bool kept = idx < P && (idx % SMX_WP_LANES) == p0;
            const bool first_path = (idx == 0 && p0 == 0);
            if (kept || first_path) {
              WpRows rows = wp_rows(o, gid, P, W, kept ? idx : 0);
              const int n = equally_spaced_path(m, seed.f, bs, st, lookahead, px, py, knots, KSTRIDE, kept ? W : 1,
                                                [&](int i, const WaypointOut& w) {
                                                  if (first_path && i == 0) {
                                                    have_first_wp = true;
                                                    fwx = w.x;
                                                    fwy = w.y;
                                                    fwh = w.heading;
                                                  }
                                                  if (kept) wp_put(m, rows, i, w);
                                                });
              if (kept) {
                wp_zero(rows, n < W ? n : W, W);
                o.wp_count[gid * (P + 1) + 1 + idx] = (uint8_t)(n < W ? n : W);
              }
            } else if (p0 == 0 || idx < P) {
              equally_spaced_path(m, seed.f, bs, st, lookahead, px, py, knots, KSTRIDE, 0,
                                  [&](int, const WaypointOut&) {});
            }
            ++idx;
          } while (bs.advance() && (p0 == 0 || idx < P));
        }
        n_paths_total = __shfl(idx, 0, SMX_WP_LANES);  // lane 0 counts them all
      }
    }
    SMX_TSTAMP(tw2);
    SMX_TACC(1, tw1, tw2);
    // rows of the paths that do not exist
    for (int slot = n_paths_total + ((p0 - n_paths_total) & (SMX_WP_LANES - 1)); slot < P; slot += SMX_WP_LANES) {
      wp_zero(wp_rows(o, gid, P, W, slot), 0, W);
      o.wp_count[gid * (P + 1) + 1 + slot] = 0;
    }
    if (p0 == 0) o.wp_count[gid * (P + 1)] = (uint8_t)(n_paths_total > 255 ? 255 : n_paths_total);
  }

  if (p0 != 0) return;
  // ---- trip meter (sensors.py:880-947); reward = increment (agent_manager.py:233-234)
  double dist = SF(SMX_S_DIST);
  int32_t* trip_has_wp_p = a.st.facts_i32 + (size_t)SMX_FI_TRIP_HAS_WP * total + gid;
  bool trip_has_wp = *trip_has_wp_p != 0;
  if (flags & SMX_F_FIRST) {
    // TripMeterSensor.__init__: first waypoint of the lowest lane, lookahead-1 path, no route
    trip_has_wp = false;
    const int ts = a.st.facts_i32[(size_t)SMX_FI_TRIP_START * total + gid];
    if (ts >= 0) {
      BranchState bs;
      bs.reset();
      RouteFilter nof;
      nof.none();
      equally_spaced_path(m, nof, bs, ts, 1, px, py, knots, KSTRIDE, 1, [&](int, const WaypointOut& w) {
        SF(SMX_S_TRIP_X) = w.x;
        SF(SMX_S_TRIP_Y) = w.y;
        SF(SMX_S_TRIP_H) = w.heading;
        trip_has_wp = true;
      });
    }
    dist = 0.0;
  }
  const double last_dist = dist;
  if (have_first_wp && trip_counts_waypoint(a, m, gid, total)) {
    if (!trip_has_wp) {
      SF(SMX_S_TRIP_X) = fwx;
      SF(SMX_S_TRIP_Y) = fwy;
      SF(SMX_S_TRIP_H) = fwh;
      trip_has_wp = true;
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
  if (!a.keep_reward_done) {
    o.reward[gid] = dist - last_dist;
    if (o.learner) o.learner[gid] = (float)(dist - last_dist);
  }
  *trip_has_wp_p = trip_has_wp ? 1 : 0;  // the flags word itself is not written here (the observe role owns it)
  SMX_TSTAMP(tw3);
  SMX_TACC(3, tw0, tw3);
}

__device__ __forceinline__ void waypoints_role(const KernelArgs& a, const int block) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  waypoints_for<SMX_BLOCK>(a, ((size_t)block * SMX_BLOCK + threadIdx.x) / SMX_WP_LANES, knot_scratch + threadIdx.x);
}

// =================================================================================
// waypoints role, staged form (large batches): the same rows as waypoints_for, written as whole
// contiguous pieces.  The chain walks are k_wp_walk's (one lane per path, nothing else, many wavefronts per
// SIMD); this kernel re-reads the knots with independent loads.  A workgroup = one wavefront = 16 vehicles x
// 4 team lanes:
//   1. number — the team exchanges what k_wp_walk found; when no lane branches inside the lookahead (almost
//               always) path numbers are the lanes' ranks among the started lanes, and every output row of
//               the vehicle is bound to a team lane, or to "zeros";
//   2. stage  — every lane interpolates its own path (interpolate_knots: the serial emitter's arithmetic,
//               one path per lane, all lanes busy) into an LDS stage [waypoint][lane] of 16-byte cells;
//   3. copy   — the wavefront's lanes sweep the 16 x P x W waypoint slots of its vehicles in memory order
//               (rows of consecutive vehicles are adjacent in every output array): lane l takes element
//               e = 64 k + l and copies its cell, or zeros.  A store instruction writes 64 consecutive
//               elements.
// Steps 2-3 run twice over the same stage: positions (x, y: 16 bytes), then heading / lane width / speed limit
// / lane id / lane index (packed into 16 bytes); the interpolation is cheap next to a second stage's LDS.
// Teams that need more — a branching inside the lookahead, a road with more than four lanes, a knot list
// cut at SMX_WPK_CAP — write their rows afterwards with the serial emitter, exactly as waypoints_for does;
// their rows are skipped in step 3.
// =================================================================================
#define SMX_WPT_MAX_PATHS 8  // dense rows per vehicle (wp_paths) the staged form handles
// floor(e / d) for 0 <= e < 4096, 1 <= d <= 64, rcp = 1.0f / d: (e + 0.5) / d is at least 0.5 / d away from an integer,
// the float32 product is off by less than 4096 / d * 2^-22
__device__ __forceinline__ int small_quotient(int e, float rcp) { return (int)(((float)e + 0.5f) * rcp); }
#define SMX_WPT_VEHICLES (SMX_BLOCK / SMX_WP_LANES)
enum { SMX_ROW_SKIP = -2, SMX_ROW_ZERO = -1 };

struct WpRowBook {  // which table column feeds which output row of the workgroup's vehicles
  short src[SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS];          // column, SMX_ROW_ZERO or SMX_ROW_SKIP
  unsigned char count[SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS];  // wp_count of the row
  unsigned char paths[SMX_WPT_VEHICLES];                     // total number of paths of the vehicle
  unsigned char tabled[SMX_WPT_VEHICLES];                    // the vehicle's counts come from this book
  unsigned int veh[SMX_WPT_VEHICLES];                        // the vehicle of team v (launch_vehicle: the alive list)
};

// Every path of the vehicle the long way (lanes by index, branches depth-first): every team lane walks
// every path to discover the branchings, lane (idx % 4) writes kept path idx; rows of the paths that do
// not exist are zeroed.  Returns the number of paths; lane 0 of the team learns the first waypoint.
template <int KSTRIDE>
__device__ inline int waypoints_long_way(const KernelArgs& a, const MapDev& m, const PathSeeds& seed, size_t gid, int p0,
                                         double px, double py, int* knots, bool& have_first_wp, double& fwx, double& fwy,
                                         double& fwh) {
  const smx_outputs& o = a.out;
  const int P = a.cfg.wp_paths, W = a.cfg.wp_len, lookahead = a.cfg.wp_lookahead;
  int idx = 0;
  for (int li = 0; li < seed.n_lanes; ++li) {
    const int st = seed_start(m, seed, li, px, py);
    if (st < 0) continue;
    BranchState bs;
    bs.reset();
    do {
      const bool kept = idx < P && (idx % SMX_WP_LANES) == p0;
      const bool first_path = (idx == 0 && p0 == 0);
      if (kept || first_path) {
        WpRows rows = wp_rows(o, gid, P, W, kept ? idx : 0);
        const int n = equally_spaced_path(m, seed.f, bs, st, lookahead, px, py, knots, KSTRIDE, kept ? W : 1,
                                          [&](int i, const WaypointOut& w) {
                                            if (first_path && i == 0) {
                                              have_first_wp = true;
                                              fwx = w.x;
                                              fwy = w.y;
                                              fwh = w.heading;
                                            }
                                            if (kept) wp_put(m, rows, i, w);
                                          });
        if (kept) {
          wp_zero(rows, n < W ? n : W, W);
          o.wp_count[gid * (P + 1) + 1 + idx] = (uint8_t)(n < W ? n : W);
        }
      } else if (p0 == 0 || idx < P) {
        equally_spaced_path(m, seed.f, bs, st, lookahead, px, py, knots, KSTRIDE, 0, [&](int, const WaypointOut&) {});
      }
      ++idx;
    } while (bs.advance() && (p0 == 0 || idx < P));
  }
  const int n_paths_total = __shfl(idx, 0, SMX_WP_LANES);  // lane 0 counts them all
  for (int slot = n_paths_total + ((p0 - n_paths_total) & (SMX_WP_LANES - 1)); slot < P; slot += SMX_WP_LANES) {
    wp_zero(wp_rows(o, gid, P, W, slot), 0, W);
    o.wp_count[gid * (P + 1) + 1 + slot] = 0;
  }
  if (p0 == 0) o.wp_count[gid * (P + 1)] = (uint8_t)(n_paths_total > 255 ? 255 : n_paths_total);
  return n_paths_total;
}

// Trip meter (sensors.py:880-947) and reward = its increment (agent_manager.py:233-234): lane 0 of the team.
template <int KSTRIDE>
__device__ __forceinline__ void trip_meter_update(const KernelArgs& a, const MapDev& m, size_t gid, size_t total, int flags,
                                                  double px, double py, int* knots, bool have_first_wp, double fwx,
                                                  double fwy, double fwh) {
  const smx_outputs& o = a.out;
  double dist = SF(SMX_S_DIST);
  int32_t* trip_has_wp_p = a.st.facts_i32 + (size_t)SMX_FI_TRIP_HAS_WP * total + gid;
  bool trip_has_wp = *trip_has_wp_p != 0;
  if (flags & SMX_F_FIRST) {
    // TripMeterSensor.__init__: first waypoint of the lowest lane, lookahead-1 path, no route
    trip_has_wp = false;
    const int ts = a.st.facts_i32[(size_t)SMX_FI_TRIP_START * total + gid];
    if (ts >= 0) {
      BranchState bs;
      bs.reset();
      RouteFilter nof;
      nof.none();
      equally_spaced_path(m, nof, bs, ts, 1, px, py, knots, KSTRIDE, 1, [&](int, const WaypointOut& w) {
        SF(SMX_S_TRIP_X) = w.x;
        SF(SMX_S_TRIP_Y) = w.y;
        SF(SMX_S_TRIP_H) = w.heading;
        trip_has_wp = true;
      });
    }
    dist = 0.0;
  }
  const double last_dist = dist;
  if (have_first_wp && trip_counts_waypoint(a, m, gid, total)) {
    if (!trip_has_wp) {
      SF(SMX_S_TRIP_X) = fwx;
      SF(SMX_S_TRIP_Y) = fwy;
      SF(SMX_S_TRIP_H) = fwh;
      trip_has_wp = true;
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
  if (!a.keep_reward_done) {
    o.reward[gid] = dist - last_dist;
    if (o.learner) o.learner[gid] = (float)(dist - last_dist);
  }
  *trip_has_wp_p = trip_has_wp ? 1 : 0;  // the flags word itself is not written here (the observe role owns it)
}

// k_alive_list: the alive vehicles of the tick, compacted (large batches).  Late in an episode most agents of an env
// are gone while the env waits for its last one (hiway_env.py:258-261): a per-vehicle team kernel then runs its
// wavefronts a quarter to a half full.  Order within the list is that of the atomics (it varies from run to run;
// every result is indexed by vehicle, never by list position).  The counter of the next tick is zeroed here.
#define SMX_ALIVE_BLOCK 1024
__global__ void __launch_bounds__(SMX_ALIVE_BLOCK) k_alive_list(const KernelArgs a, int32_t* list, int32_t* count, int32_t* count_next,
                                                                  int32_t* slow_next) {
  // one atomic per workgroup of 1024 (a wavefront each was 2 048 atomics on one counter: 28 us at 131 k vehicles)
  __shared__ int wave_base[SMX_ALIVE_BLOCK / 64];
  __shared__ int group_base;
  const size_t total = (size_t)a.cfg.num_envs * a.cfg.num_vehicles;
  const size_t gid = (size_t)blockIdx.x * SMX_ALIVE_BLOCK + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool alive = gid < total && (a.st.flags[gid] & SMX_F_ALIVE);
  const unsigned long long mask = __ballot(alive);
  if (gid == 0) {
    *count_next = 0;
    slow_next[0] = slow_next[1] = slow_next[2] = slow_next[3] = 0;  // the slow lists of the next tick's fast kernels
  }
  if (lane == 0) wave_base[wave] = __popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    int sum = 0;
    for (int w = 0; w < SMX_ALIVE_BLOCK / 64; ++w) {
      const int n = wave_base[w];
      wave_base[w] = sum;
      sum += n;
    }
    group_base = sum > 0 ? atomicAdd(count, sum) : 0;
  }
  __syncthreads();
  if (alive) list[group_base + wave_base[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)gid;
}

// k_wp_walk: the chain walks of the waypoints sensor, one lane per (vehicle, seed lane) and nothing else —
// no LDS, few registers, so that many wavefronts per SIMD hide the dependent loads.  Leaves the knot list of
// the seed lane's first path and the number of paths that start there (KnotLists).
// (`honour_pending`: pass over a vehicle whose seeds the slow chain is still looking for — k_scan_fast; the slow chain's
// own walk and the walk of the reset pass's new vehicles take every vehicle they are given)
__device__ __forceinline__ void wp_walk_for(const KernelArgs& a, const size_t gid, const int p0, const bool honour_pending,
                                            const bool whatever_the_pass = false) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t paths = total * SMX_WP_LANES;
  if (gid >= total) return;
  const size_t path = gid * SMX_WP_LANES + p0;
  const int flags = a.st.flags[gid];
  int n = 0, nk = 0, cnt = 0;
  double D = 0.0;
  // (nothing of a pending vehicle's lists is touched here: the slow chain writes them meanwhile, on another stream)
  if (honour_pending && a.seed_pending != nullptr && a.seed_pending[gid]) return;
  if ((flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || whatever_the_pass || (flags & SMX_F_FIRST))) {
    const PathSeeds seed = load_seeds(a, gid, total);
    if (seed.road >= 0 && seed.n_lanes <= SMX_WP_LANES && p0 < seed.n_lanes) {
      const double px = SF(SMX_S_X), py = SF(SMX_S_Y);
      const int start = seed_start(m, seed, p0, px, py);
      if (start >= 0) {
        a.knots.idx[path] = start;
        BranchState bs;
        bs.reset();
        int below16 = 0;        // knots less than 16 hops down
        bool knot_at_16 = false;
        int base_idx = start, base_hop = 0;  // the last lanepoint of the list less than 16 hops down (the start if no knot is)
        const PathWalk w = walk_knots(m, seed.f, bs, start, c.wp_lookahead, px, py, [&](int k, int idx, int hop) {
          if (k <= SMX_WPK_CAP) a.knots.idx[(size_t)k * paths + path] = idx;
          if (hop < SMX_CTRL_WPS - 1) {
            ++below16;
            base_idx = idx;
            base_hop = hop;
          }
          if (hop == SMX_CTRL_WPS - 1) knot_at_16 = true;
        });
        n = w.n;
        nk = w.nk;
        D = w.D;
        // the lookahead-16 path: the whole path when it is no longer than that, else the knots less than 16
        // hops down and the lanepoint 16 hops down (a knot of the list, or the probed interpolated lanepoint)
        a.knots.nk16[path] = (uint8_t)(w.n <= SMX_CTRL_WPS ? w.nk : below16 + 1);
        // (an interpolated lanepoint of the run that leaves the last knot before it: followed here, after the walk,
        // from that knot's record — the successor the route filter allows where the knot branches; the list is only
        // reused when exactly one path starts on this seed lane, so there is exactly one)
        int end16 = -1;
        if (w.n > SMX_CTRL_WPS && !knot_at_16) {
          const smx_lp_rec br = load_lp(m, base_idx, 47);
          int first = br.next0;
          if (br.n_next > 1) {
            first = -1;
            for (int k = br.next_off; k < br.next_off + br.n_next && first < 0; ++k) {
              const smx_succ_rec sr = m.succ_rec[k];
              if (lane_allowed(m, seed.f, sr.lane)) first = sr.idx;
            }
          }
          if (first >= 0) end16 = chain_at(m, first, SMX_CTRL_WPS - 1 - base_hop - 1, (br.flags & 1) != 0);
        }
        a.knots.end16[path] = end16;
        a.knots.key[path] = start;
        a.knots.key[paths + path] = seed.f.n > 0 ? seed.f.road[0] : -1;
        a.knots.key[2 * paths + path] = seed.f.n > 1 ? seed.f.road[1] : -1;
        cnt = 1;
        while (bs.advance()) {
          walk_knots(m, seed.f, bs, start, c.wp_lookahead, px, py, [](int, int, int) {});
          if (cnt < 255) ++cnt;
        }
      }
    }
  }
  a.knots.n[path] = (int16_t)n;
  a.knots.nk[path] = (int16_t)nk;
  a.knots.cnt[path] = (uint8_t)cnt;
  a.knots.D[path] = D;
  if (n == 0) a.knots.key[path] = -1;  // nothing here for the next tick's controller to reuse
}

__global__ void __launch_bounds__(SMX_BLOCK) k_wp_walk(const KernelArgs a) {
  SMX_TSTAMP(span0);
  const size_t total = (size_t)a.cfg.num_envs * a.cfg.num_vehicles;
  const size_t lane_no = (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x;
  wp_walk_for(a, launch_vehicle(a, lane_no / SMX_WP_LANES, total), (int)(lane_no % SMX_WP_LANES), true);
  SMX_TSTAMP(span1);
  SMX_TSPAN(3, span0, span1);
}

// the slow chain's vehicles (their seeds come from k_scan_listed): a fixed grid striding the slow list.  Their rows
// leave through k_waypoints_listed; the lists are for the next tick's k_control_fast.
__global__ void __launch_bounds__(SMX_BLOCK) k_wp_walk_listed(const KernelArgs a) {
  const int count = *a.slow_count;
  constexpr int VPB = SMX_BLOCK / SMX_WP_LANES;
  for (int i = (int)blockIdx.x * VPB + (int)threadIdx.x / SMX_WP_LANES; i < count; i += (int)gridDim.x * VPB)
    wp_walk_for(a, (size_t)a.slow_list[i], (int)threadIdx.x % SMX_WP_LANES, false);
}

struct __align__(16) WpStageCell {  // second pass: everything of a waypoint but its position
  float heading, width, speed;
  short lane;
  signed char lane_index;
  signed char pad;
};

__device__ __forceinline__ void waypoints_tables_role(const KernelArgs& a, const int block) {
  extern __shared__ __align__(16) unsigned char stage_raw[];  // [wp_len][SMX_BLOCK] cells of 16 bytes
  __shared__ WpRowBook book;
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const smx_outputs& o = a.out;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int P = c.wp_paths, W = c.wp_len, lookahead = c.wp_lookahead;
  const int p0 = threadIdx.x % SMX_WP_LANES, v = threadIdx.x / SMX_WP_LANES;
  // team v works on the v-th vehicle of the workgroup's sixteen launch slots (alive vehicles only when the tick
  // has its list): the rows of a vehicle are contiguous in the outputs, the vehicles of a workgroup need not be
  const size_t gid = launch_vehicle(a, (size_t)block * SMX_WPT_VEHICLES + v, total);
  const size_t path = gid * SMX_WP_LANES + p0, paths = total * SMX_WP_LANES;
  const int col = threadIdx.x;
  SMX_TSTAMP(tw0);
  int flags = gid < total ? a.st.flags[gid] : 0;
  const bool live = gid < total && (flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST));
  double px = 0.0, py = 0.0;
  PathSeeds seed;
  seed.road = -1;
  seed.n_lanes = 0;
  seed.f.none();
  int n_first = 0, nk = 0, cnt = 0;
  double D = 0.0;
  if (live) {
    px = SF(SMX_S_X);
    py = SF(SMX_S_Y);
    seed = load_seeds(a, gid, total);
    n_first = a.knots.n[path];  // what k_wp_walk found on seed lane p0 (0: no path starts there)
    nk = a.knots.nk[path];
    cnt = a.knots.cnt[path];
    D = a.knots.D[path];
  }
  SMX_TSTAMP(tw1);
  SMX_TACC(0, tw0, tw1);
  // ---- 1. number the paths
  const bool seeded = live && seed.road >= 0;
  const bool long_way = seeded && seed.n_lanes > SMX_WP_LANES;  // uniform in the team
  int started = n_first > 0 ? (1 << p0) : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) started |= __shfl_xor(started, msk, SMX_WP_LANES);
  const int prov = __popc(started & ((1 << p0) - 1));  // this lane's path number if nobody branches
  int branching = (cnt > 1) ? 1 : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) branching |= __shfl_xor(branching, msk, SMX_WP_LANES);
  // the team's rows come from the stage unless it has to number its paths the long way
  const bool serial_team = seeded && (long_way || branching != 0);  // uniform in the team
  const int n_paths_staged = __popc(started);
  const bool staged = n_first > 0 && !serial_team;            // this lane interpolates a path
  const bool listed = nk <= SMX_WPK_CAP;                      // ... from a complete knot list
  const bool my_row = staged && prov < P;                     // ... that one of the kept rows holds
  if (p0 == 0) {
    book.veh[v] = (unsigned int)(gid < total ? gid : 0);
    book.tabled[v] = (live && !serial_team) ? 1 : 0;
    book.paths[v] = (unsigned char)(seeded ? n_paths_staged : 0);
  }
  for (int slot = p0; slot < P; slot += SMX_WP_LANES) {
    // rows without a path read zeros; a dead / absent vehicle's rows are not this kernel's to write
    book.src[v * P + slot] = (short)((live && !serial_team) ? SMX_ROW_ZERO : SMX_ROW_SKIP);
    book.count[v * P + slot] = 0;
  }
  __syncthreads();  // (one wavefront: orders the LDS writes of the team's other lanes)
  if (my_row) {
    book.src[v * P + prov] = (short)(listed ? col : SMX_ROW_SKIP);
    book.count[v * P + prov] = (unsigned char)min(n_first, W);
  }
  // ---- 2 + 3, positions
  const smx_lp_rec r0 = (staged && listed) ? load_lp(m, a.knots.idx[path], 46) : smx_lp_rec{};
  auto fetch = [&](int k) { return a.knots.idx[(size_t)(k + 1) * paths + path]; };
  // the knots of the path into registers, every load issued before anything waits for one (paths of more
  // knots than SMX_WPT_PRELOAD interpolate straight from the list, a dependent load per knot)
  constexpr int KP = SMX_WPT_PRELOAD;
  const bool pre = staged && listed;
  double kx[KP], ky[KP], kh[KP], kw[KP], ks_[KP];
  int kl[KP];
  double w0 = 0.0, s0 = 0.0;
  {
    int kid[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) kid[k] = (pre && k < nk) ? fetch(k) : 0;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const smx_lp_rec* r = m.lp_rec + kid[k];
      const bool have = pre && k < nk;
      kx[k] = have ? r->x : 0.0;
      ky[k] = have ? r->y : 0.0;
      kh[k] = have ? r->heading : 0.0;
      kl[k] = have ? r->lane : 0;
    }
    if (pre) {
      w0 = m.lane_width[r0.lane];
      s0 = m.lane_speed[r0.lane];
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      // the tables are read only where the lane changes (few lanes of the wavefront ask at all)
      const bool ask = pre && k < nk && kl[k] != (k == 0 ? (int)r0.lane : kl[k > 0 ? k - 1 : 0]);
      kw[k] = ask ? m.lane_width[kl[k]] : 0.0;
      ks_[k] = ask ? m.lane_speed[kl[k]] : 0.0;
    }
  }
  const int elems = SMX_WPT_VEHICLES * P * W;
  double gx = 0.0, gy = 0.0, gh = 0.0;  // first waypoint of this lane's path
  {
    double2* stage = reinterpret_cast<double2*>(stage_raw);
    auto put_xy = [&](int i, const WaypointOut& w) {
      if (i == 0) {
        gx = w.x;
        gy = w.y;
      }
      stage[i * SMX_BLOCK + col] = make_double2(w.x, w.y);
    };
    if (pre)
      interpolate_knots_preloaded<KP>(m, r0, w0, s0, nk, n_first, D, px, py, my_row ? W : 1, kx, ky, kh, kl, kw, ks_, fetch, put_xy);
    __syncthreads();
    SMX_TSTAMP(tw2);
    SMX_TACC(1, tw1, tw2);
    int row = threadIdx.x / W, i = threadIdx.x - row * W;  // element e = row * W + i, advanced by 64 per round
    int vv = row / P, slot = row - vv * P;                 // row = vv * P + slot: team vv's path row `slot`
    const int drow = SMX_BLOCK / W, di = SMX_BLOCK - drow * W;
    for (int e = threadIdx.x; e < elems; e += SMX_BLOCK) {
      const int src = book.src[row];
      if (src != SMX_ROW_SKIP) {
        double2 xy = make_double2(0.0, 0.0);
        if (src >= 0 && i < (int)book.count[row]) xy = stage[i * SMX_BLOCK + src];
        double* dst = o.wp_pos + (((size_t)book.veh[vv] * P + slot) * W + i) * 3;
        dst[0] = xy.x;
        dst[1] = xy.y;
        dst[2] = 0.0;
      }
      row += drow;
      slot += drow;
      i += di;
      if (i >= W) {
        i -= W;
        ++row;
        ++slot;
      }
      while (slot >= P) {
        slot -= P;
        ++vv;
      }
    }
    __syncthreads();  // the stage is reused
    SMX_TSTAMP(tw2b);
    SMX_TACC(24, tw2, tw2b);
  }
  SMX_TSTAMP(tw3);
  // ---- 2 + 3, the other fields
  {
    WpStageCell* stage = reinterpret_cast<WpStageCell*>(stage_raw);
    if (staged && listed) {
      int cached_lane = -1, cached_index = 0;  // consecutive waypoints mostly share their lane
      auto put_rest = [&](int i, const WaypointOut& w) {
        if (i == 0) gh = w.heading;
        if (w.lane != cached_lane) {
          cached_lane = w.lane;
          cached_index = m.lane_index[w.lane];
        }
        WpStageCell cell;
        cell.heading = (float)w.heading;
        cell.width = (float)w.width;
        cell.speed = (float)w.speed;
        cell.lane = (short)w.lane;
        cell.lane_index = (signed char)cached_index;
        cell.pad = 0;
        stage[i * SMX_BLOCK + col] = cell;
      };
      interpolate_knots_preloaded<KP>(m, r0, w0, s0, nk, n_first, D, px, py, my_row ? W : 1, kx, ky, kh, kl, kw, ks_, fetch, put_rest);
    }
    __syncthreads();
    SMX_TSTAMP(tw3b);
    SMX_TACC(25, tw3, tw3b);
    int row = threadIdx.x / W, i = threadIdx.x - row * W;
    int vv = row / P, slot = row - vv * P;
    const int drow = SMX_BLOCK / W, di = SMX_BLOCK - drow * W;
    for (int e = threadIdx.x; e < elems; e += SMX_BLOCK) {
      const int src = book.src[row];
      if (src != SMX_ROW_SKIP) {
        WpStageCell cell;
        cell.heading = 0.0f;
        cell.width = 0.0f;
        cell.speed = 0.0f;
        cell.lane = -1;
        cell.lane_index = 0;
        if (src >= 0 && i < (int)book.count[row]) cell = stage[i * SMX_BLOCK + src];
        const size_t q = ((size_t)book.veh[vv] * P + slot) * W + i;
        o.wp_heading[q] = cell.heading;
        o.wp_lane_width[q] = cell.width;
        o.wp_speed_limit[q] = cell.speed;
        o.wp_lane_id[q] = cell.lane;
        o.wp_lane_index[q] = cell.lane_index;
      }
      row += drow;
      slot += drow;
      i += di;
      if (i >= W) {
        i -= W;
        ++row;
        ++slot;
      }
      while (slot >= P) {
        slot -= P;
        ++vv;
      }
    }
    // wp_count: [vehicle][0] = number of paths, [1 + slot] = waypoints kept of the path in that row
    const int cells = SMX_WPT_VEHICLES * (P + 1);
    for (int e = threadIdx.x; e < cells; e += SMX_BLOCK) {
      const int vv = e / (P + 1), qq = e - vv * (P + 1);
      if (!book.tabled[vv]) continue;
      o.wp_count[(size_t)book.veh[vv] * (P + 1) + qq] = qq == 0 ? book.paths[vv] : book.count[vv * P + qq - 1];
    }
    SMX_TSTAMP(tw3c);
    SMX_TACC(26, tw3b, tw3c);
  }
  __syncthreads();  // the stage is done with: its memory becomes the serial emitter's knot scratch
  SMX_TSTAMP(tw4);
  SMX_TACC(2, tw3, tw4);
  int* knots = reinterpret_cast<int*>(stage_raw) + threadIdx.x;
  bool have_first_wp = false;
  double fwx = 0.0, fwy = 0.0, fwh = 0.0;
  if (live) {
    if (long_way) {
      waypoints_long_way<SMX_BLOCK>(a, m, seed, gid, p0, px, py, knots, have_first_wp, fwx, fwy, fwh);
    } else if (serial_team) {
      // ---- a lane branches inside the lookahead: paths are numbered lanes by index, branches depth-first,
      // i.e. lane p's paths follow those of the lower lanes.  k_wp_walk counted them, so an exclusive prefix
      // over the team gives every lane the numbers of its own paths, and each lane writes its own branches
      // with the serial emitter (its own walks: the knot list only holds the first one).
      int incl = cnt;
      {
        int t = __shfl_up(incl, 1, SMX_WP_LANES);
        if (p0 >= 1) incl += t;
        t = __shfl_up(incl, 2, SMX_WP_LANES);
        if (p0 >= 2) incl += t;
      }
      const int n_paths_total = __shfl(incl, SMX_WP_LANES - 1, SMX_WP_LANES);
      const int base = incl - cnt;
      if (cnt > 0 && base < P) {
        const int start = a.knots.idx[path];
        BranchState bs;
        bs.reset();
        int idx = base;
        do {
          if (idx >= P) break;
          WpRows rows = wp_rows(o, gid, P, W, idx);
          const int n = equally_spaced_path(m, seed.f, bs, start, lookahead, px, py, knots, SMX_BLOCK, W,
                                            [&](int i, const WaypointOut& w) {
                                              if (i == 0 && idx == 0) {
                                                gx = w.x;
                                                gy = w.y;
                                                gh = w.heading;
                                              }
                                              wp_put(m, rows, i, w);
                                            });
          wp_zero(rows, n < W ? n : W, W);
          o.wp_count[gid * (P + 1) + 1 + idx] = (uint8_t)(n < W ? n : W);
          ++idx;
        } while (bs.advance());
      }
      for (int slot = n_paths_total + ((p0 - n_paths_total) & (SMX_WP_LANES - 1)); slot < P; slot += SMX_WP_LANES) {
        wp_zero(wp_rows(o, gid, P, W, slot), 0, W);
        o.wp_count[gid * (P + 1) + 1 + slot] = 0;
      }
      if (p0 == 0) o.wp_count[gid * (P + 1)] = (uint8_t)(n_paths_total > 255 ? 255 : n_paths_total);
      const int src = started ? (__ffs(started) - 1) : 0;  // path 0 is the lowest started lane's first path
      fwx = __shfl(gx, src, SMX_WP_LANES);
      fwy = __shfl(gy, src, SMX_WP_LANES);
      fwh = __shfl(gh, src, SMX_WP_LANES);
      have_first_wp = n_paths_total > 0;
    } else {
      if (staged && !listed) {
        // more knots than the list holds: this path leaves through the serial emitter (its own walk)
        BranchState bs;
        bs.reset();
        WpRows rows = wp_rows(o, gid, P, W, my_row ? prov : 0);
        const int n = equally_spaced_path(m, seed.f, bs, a.knots.idx[path], lookahead, px, py, knots, SMX_BLOCK, my_row ? W : 1,
                                          [&](int i, const WaypointOut& w) {
                                            if (i == 0) {
                                              gx = w.x;
                                              gy = w.y;
                                              gh = w.heading;
                                            }
                                            if (my_row) wp_put(m, rows, i, w);
                                          });
        if (my_row) wp_zero(rows, n < W ? n : W, W);
      }
      const int src = started ? (__ffs(started) - 1) : 0;  // path 0 is the lowest started lane's
      fwx = __shfl(gx, src, SMX_WP_LANES);
      fwy = __shfl(gy, src, SMX_WP_LANES);
      fwh = __shfl(gh, src, SMX_WP_LANES);
      have_first_wp = n_paths_staged > 0;
    }
  }
  SMX_TSTAMP(tw5);
  SMX_TACC(5, tw4, tw5);
  if (live && p0 == 0) trip_meter_update<SMX_BLOCK>(a, m, gid, total, flags, px, py, knots, have_first_wp, fwx, fwy, fwh);
  SMX_TSTAMP(tw6);
  SMX_TACC(3, tw0, tw6);
}

// =================================================================================
// waypoints role, emit-parallel form (large batches, round 3): the same rows as waypoints_tables_role, one lane per
// WAYPOINT instead of one lane per path.
//   The staged form interpolates one path per lane: 64 paths of a wavefront meet their knots at different
// waypoints, so the wavefront runs every knot's arithmetic and the longest run of waypoints of every knot interval
// (~50 emit rounds for 20 waypoints), twice (two 16-byte stage passes), and then copies the stage out element by
// element: 7.8 k vector instructions per wavefront with 20 KB of LDS (two wavefronts per SIMD).
//   Here a path lane only walks its knots ONCE for what is sequential by nature — the running arclength and the
// running heading unwrap, in the reference's order of additions — and leaves the knots it needs for its W waypoints
// in an LDS pool shared by the workgroup's paths (a path on a straight needs two records, one in a bend ten:
// records are handed out by a prefix sum over the wavefront).  Then the wavefront sweeps its 16 x P x W waypoint
// slots in memory order: a lane finds its waypoint's knot interval in the path's records, takes np.interp's slope
// from the two knots and stores the waypoint straight from registers — every store instruction writes consecutive
// elements, no stage, no divergence between lanes beyond the interval search.
//   Same expressions, same bits: t_i, (q - j) / (cum_q - cum_j), slope * (t - cum_j) + j, the lane rules.
// Vehicles with a path the pool cannot hold (more knots than SMX_WPE_KNOTS inside the kept waypoints, a pool
// overflow) or whose team numbers its paths the long way go to a slow list and through the serial emitter of
// k_waypoints_listed (waypoints_for: its own walks, every row, the trip meter): a serial tail inside this kernel would
// hold its whole wavefront, and its registers would set this kernel's occupancy.
// =================================================================================
#define SMX_WPE_KNOTS 10  // knots after the start a path lane holds in registers
// knot records per workgroup (64 paths; loop: 48 paths of 5.5 records on average, sigma 14).  A workgroup whose paths need
// more sends the teams that do not fit to the slow list, whose kernel is a launch of pure latency at the end of the tick's
// longest chain: at 320 a dozen workgroups of 8 192 overflowed per tick (76 us each tick for 200 vehicles); two sweeps of
// the pool inside the kernel kept the knots in registers across the sweep (256 registers, one wavefront per SIMD).
#ifndef SMX_WPE_POOL  // (a developer build with a small pool drives most paths through the overflow area: tests)
#define SMX_WPE_POOL 352
#endif
struct __align__(8) WpKnot {
  double x, y, h, cum;  // position, unwrapped heading, arclength from the projected start
};
struct WpKnotLanes {
  short lane, strict;   // the knot's lane; lane of the last knot with an arclength strictly below this one's
};
// the overflow area of one workgroup: per column SMX_WPE_KNOTS + 1 records, then as many lane pairs
#define SMX_WPE_SPILL_GROUP_BYTES ((size_t)SMX_BLOCK * (SMX_WPE_KNOTS + 1) * (sizeof(WpKnot) + sizeof(WpKnotLanes)))

__device__ __forceinline__ void waypoints_emit_role(const KernelArgs& a, const int block) {
  // 15.6 KB of LDS in all: ten workgroups per CU (its registers allow twelve: three wavefronts per SIMD)
  __shared__ WpKnot pool[SMX_WPE_POOL];
  __shared__ WpKnotLanes pool_lanes[SMX_WPE_POOL];
  __shared__ WpRowBook book;
  __shared__ double hdr_step[SMX_BLOCK], hdr_D[SMX_BLOCK];
  __shared__ unsigned short hdr_off[SMX_BLOCK];
  __shared__ unsigned char hdr_nrec[SMX_BLOCK], hdr_n[SMX_BLOCK];
  // lane width / speed limit / lane index of the path's start lane, and whether every knot held lies on that lane (almost
  // always): the waypoint lanes then need no table look-up behind their interval search
  __shared__ float hdr_w0[SMX_BLOCK], hdr_s0[SMX_BLOCK];  // (as they leave: the rows hold them as float32)
  __shared__ signed char hdr_li0[SMX_BLOCK];
  __shared__ unsigned char hdr_one_lane[SMX_BLOCK];
  __shared__ double first_wp[SMX_WPT_VEHICLES][3];
  __shared__ unsigned char rows_live[SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS], rows_zero[SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS];
  __shared__ unsigned char rows_spill[SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS];  // rows whose knots lie in the overflow area
  __shared__ unsigned char hdr_spill[SMX_BLOCK];
  static_assert(SMX_WPT_VEHICLES * SMX_WPT_MAX_PATHS <= 256, "row numbers are bytes");
  static_assert(sizeof(WpKnot) == 32 && sizeof(WpKnotLanes) == 4, "WpKnot layout");
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const smx_outputs& o = a.out;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const int P = c.wp_paths, W = c.wp_len;
  const int p0 = threadIdx.x % SMX_WP_LANES, v = threadIdx.x / SMX_WP_LANES;
  const size_t gid = launch_vehicle(a, (size_t)block * SMX_WPT_VEHICLES + v, total);
  const size_t path = gid * SMX_WP_LANES + p0, paths = total * SMX_WP_LANES;
  const int col = threadIdx.x;
  // Every load whose address only needs the vehicle is issued here, together and whatever the flags say (a dead slot's
  // words are loaded and dropped): the kernel runs two wavefronts per SIMD, and flags -> seeds -> knot list -> records
  // -> trip meter taken one after the other was five round trips of 2-4 us each under load (round 3: 85 of its 217 us).
  constexpr int KP = SMX_WPE_KNOTS;
  SMX_TSTAMP(te0);
  const bool in_range = gid < total;
  const size_t g = in_range ? gid : 0, pth = g * SMX_WP_LANES + p0;
  int flags = a.st.flags[g];
  const int pend = a.seed_pending != nullptr ? (int)a.seed_pending[g] : 0;
  double px = a.st.f64[(size_t)SMX_S_X * total + g], py = a.st.f64[(size_t)SMX_S_Y * total + g];
  PathSeeds seed = load_seeds(a, g, total);
  int n_first = a.knots.n[pth];  // what k_wp_walk found on seed lane p0 (0: no path starts there)
  int nk = a.knots.nk[pth], cnt = a.knots.cnt[pth];
  double D = a.knots.D[pth];
  const int kid0 = a.knots.idx[pth];
  int kid[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) kid[k] = a.knots.idx[(size_t)(k + 1) * paths + pth];
  // the trip meter's words (lane 0 of the team uses them)
  const double trip_dist = a.st.f64[(size_t)SMX_S_DIST * total + g], trip_x = a.st.f64[(size_t)SMX_S_TRIP_X * total + g],
               trip_y = a.st.f64[(size_t)SMX_S_TRIP_Y * total + g], trip_h = a.st.f64[(size_t)SMX_S_TRIP_H * total + g];
  const int trip_has = a.st.facts_i32[(size_t)SMX_FI_TRIP_HAS_WP * total + g];
  // (the slow chain writes the rows of a vehicle whose seeds it is still looking for: k_scan_fast; uniform in the team)
  const bool live = in_range && (flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST)) && !pend;
  if (!live) {
    seed.road = -1;
    seed.n_lanes = 0;
    seed.f.none();
    n_first = nk = cnt = 0;
    D = 0.0;
  }
  // ---- 1. number the paths (as the staged form does)
  const bool seeded = live && seed.road >= 0;
  const bool long_way = seeded && seed.n_lanes > SMX_WP_LANES;  // uniform in the team
  int started = n_first > 0 ? (1 << p0) : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) started |= __shfl_xor(started, msk, SMX_WP_LANES);
  const int prov = __popc(started & ((1 << p0) - 1));  // this lane's path number if nobody branches
  int branching = (cnt > 1) ? 1 : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) branching |= __shfl_xor(branching, msk, SMX_WP_LANES);
  const bool serial_team = seeded && (long_way || branching != 0);  // uniform in the team
  const int n_paths_staged = __popc(started);
  const bool staged = n_first > 0 && !serial_team;  // this lane's path leaves through the table (or the serial emitter alone)
  const bool listed = nk <= SMX_WPK_CAP;            // ... k_wp_walk's knot list holds all its knots
  const bool my_row = staged && prov < P;           // ... and one of the kept rows holds it
  // ---- 2. the path lane's walk over its knots: arclength, unwrapped headings, how many it needs
  SMX_TSTAMP(te1);
  SMX_TACC(32, te0, te1);
  bool tabled_path = false;
  int nrec = 0;
  double kx[KP], ky[KP], kh[KP], cum[KP];
  int kl[KP], kstrict[KP];
  double k0x = 0.0, k0y = 0.0, k0h = 0.0;
  int lane0 = 0;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    kx[k] = ky[k] = kh[k] = cum[k] = 0.0;
    kl[k] = kstrict[k] = 0;
  }
  const bool walk = my_row && listed && !SMX_SKIP(a, 1 << 22);
  if (walk) {
    const smx_lp_rec r0 = load_lp(m, kid0, 46);
    lane0 = r0.lane;
    const int nkp = nk < KP ? nk : KP;
    {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const bool have = k < nkp;
        const smx_lp_rec* r = m.lp_rec + (have ? kid[k] : 0);
        kx[k] = have ? r->x : 0.0;
        ky[k] = have ? r->y : 0.0;
        kh[k] = have ? r->heading : 0.0;
        kl[k] = have ? r->lane : 0;
      }
    }
    const int n = n_first;
    if (n == 1) {
      // :1379-1390 (a one-point path): the lanepoint itself, not the projection; its heading as it is
      k0x = r0.x;
      k0y = r0.y;
      k0h = r0.heading;
      nrec = 1;
      tabled_path = true;
    } else {
      const double proj = (px - r0.x) * r0.dirx + (py - r0.y) * r0.diry;
      k0x = r0.x + proj * r0.dirx;
      k0y = r0.y + proj * r0.diry;
      k0h = r0.heading;
      const int n_emit = n < W ? n : W;
      const double step = D / (double)(n - 1);  // np.linspace(0, D, n)
      const double t_last = (n_emit - 1 == n - 1) ? D : (double)(n_emit - 1) * step;
      double jx = k0x, jy = k0y, jcum = 0.0;
      int jlane = lane0, strict_lane = lane0;
      Unwrap uw;
      uw.start(k0h);
      int need = -1;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (k < nkp && need < 0) {
          const double ex = kx[k] - jx, ey = ky[k] - jy;
          const double qcum = jcum + sqrt(ex * ex + ey * ey);
          kh[k] = uw.push(kh[k]);
          cum[k] = qcum;
          if (qcum > jcum) strict_lane = jlane;
          kstrict[k] = strict_lane;
          if (t_last < qcum) need = k + 2;  // the last kept waypoint lies inside this interval
          jx = kx[k];
          jy = ky[k];
          jcum = qcum;
          jlane = kl[k];
        }
      }
      if (need < 0 && nkp == nk) need = nk + 1;  // kept waypoints at or beyond the last knot: every knot, the last one as the tail
      if (need > 0) {
        nrec = need;
        tabled_path = true;
      }
    }
  }
  // ---- records handed out by an exclusive prefix sum over the wavefront's paths
  SMX_TSTAMP(te2);
  SMX_TACC(33, te1, te2);
  int incl = tabled_path ? nrec : 0;
  {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
  }
  const int off = incl - (tabled_path ? nrec : 0);
  // The pool is full (a workgroup whose sixteen vehicles all sit in a bend: one or two of 8 192 per tick): the path's
  // records go to the workgroup's overflow area in device memory instead, and its rows are swept after the others by
  // a plain loop that reads them back (L2).  The rows list's serial emitter — 50 us of one team's latency behind this
  // kernel, at the end of the tick's longest chain, on two ticks of three — is left to the teams that need it.
  const bool spilled = tabled_path && off + nrec > a.wp_pool_limit && a.wp_spill != nullptr;
  if (tabled_path && off + nrec > a.wp_pool_limit && !spilled) tabled_path = false;
  WpKnot* const spill_knots = reinterpret_cast<WpKnot*>(a.wp_spill + (size_t)block * SMX_WPE_SPILL_GROUP_BYTES);
  WpKnotLanes* const spill_lanes = reinterpret_cast<WpKnotLanes*>(spill_knots + (size_t)SMX_BLOCK * (SMX_WPE_KNOTS + 1));
  // A team with a row the pool does not hold (more knots than the table form takes, a cut knot list, a full pool), or
  // that has to number its paths the long way (a branching inside the lookahead, a road of more than four lanes),
  // leaves all its rows and its trip meter to k_waypoints_listed: its vehicle goes to the slow list.
  // (so does a new vehicle, whose trip meter starts with a walk of its own: none comes here on a tick, whose new
  // vehicles are the reset pass's)
  // (developer counts of the reasons, per path lane)
  SMX_COUNT(50, live && p0 == 0 && long_way);
  SMX_COUNT(51, live && p0 == 0 && branching != 0);
  SMX_COUNT(52, my_row && !listed);
  SMX_COUNT(53, my_row && listed && nrec == 0);
  SMX_COUNT(54, my_row && listed && nrec > 0 && !tabled_path);
  SMX_COUNT(55, live && p0 == 0 && (flags & SMX_F_FIRST));
  SMX_COUNT(56, live && p0 == 0);
  int slow_i = (live && (serial_team || (my_row && !tabled_path) || (flags & SMX_F_FIRST))) ? 1 : 0;
#pragma unroll
  for (int msk = SMX_WP_LANES / 2; msk >= 1; msk >>= 1) slow_i |= __shfl_xor(slow_i, msk, SMX_WP_LANES);
  const bool slow_team = slow_i != 0;  // uniform in the team
  if (slow_team) tabled_path = false;
  if (p0 == 0) {
    book.veh[v] = (unsigned int)(gid < total ? gid : 0);
    book.tabled[v] = (live && !slow_team) ? 1 : 0;
    book.paths[v] = (unsigned char)(seeded ? n_paths_staged : 0);
  }
  for (int slot = p0; slot < P; slot += SMX_WP_LANES) {
    // rows without a path read zeros; a dead / absent vehicle's rows are not this kernel's to write
    book.src[v * P + slot] = (short)((live && !slow_team) ? SMX_ROW_ZERO : SMX_ROW_SKIP);
    book.count[v * P + slot] = 0;
  }
  {
    // the wavefront's slow vehicles, appended with one atomic
    const bool app = slow_team && p0 == 0;
    const unsigned long long mask = __ballot(app);
    if (mask != 0ull) {
      const int lane = threadIdx.x & 63;
      int base = 0;
      if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(a.slow_count, __popcll(mask));
      base = __shfl(base, __ffsll((long long)mask) - 1);
      if (app) a.slow_list[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)gid;
    }
  }
  __syncthreads();  // (one wavefront: orders the LDS writes of the team's other lanes)
  if (my_row && !slow_team) {
    book.src[v * P + prov] = (short)col;
    book.count[v * P + prov] = (unsigned char)min(n_first, W);
  }
  if (tabled_path) {
    hdr_step[col] = n_first > 1 ? D / (double)(n_first - 1) : 0.0;
    hdr_D[col] = D;
    hdr_off[col] = (unsigned short)off;
    hdr_nrec[col] = (unsigned char)nrec;
    hdr_n[col] = (unsigned char)n_first;
    hdr_w0[col] = (float)m.lane_width[lane0];
    hdr_s0[col] = (float)m.lane_speed[lane0];
    hdr_li0[col] = (signed char)m.lane_index[lane0];
    {
      bool one = true;
#pragma unroll
      for (int k = 0; k < KP; ++k) one = one && (k + 1 >= nrec || kl[k] == lane0);
      hdr_one_lane[col] = one ? 1 : 0;
    }
    WpKnot r;
    r.x = k0x;
    r.y = k0y;
    r.h = k0h;
    r.cum = 0.0;
    WpKnotLanes rl;
    rl.lane = (short)lane0;
    rl.strict = (short)lane0;
    if (!spilled) {
      pool[off] = r;
      pool_lanes[off] = rl;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (k + 1 < nrec) {
          WpKnot q;
          q.x = kx[k];
          q.y = ky[k];
          q.h = kh[k];
          q.cum = cum[k];
          pool[off + k + 1] = q;
          WpKnotLanes ql;
          ql.lane = (short)kl[k];
          ql.strict = (short)kstrict[k];
          pool_lanes[off + k + 1] = ql;
        }
      }
    } else {
      WpKnot* gk = spill_knots + col * (SMX_WPE_KNOTS + 1);
      WpKnotLanes* gl = spill_lanes + col * (SMX_WPE_KNOTS + 1);
      gk[0] = r;
      gl[0] = rl;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (k + 1 < nrec) {
          WpKnot q;
          q.x = kx[k];
          q.y = ky[k];
          q.h = kh[k];
          q.cum = cum[k];
          gk[k + 1] = q;
          WpKnotLanes ql;
          ql.lane = (short)kl[k];
          ql.strict = (short)kstrict[k];
          gl[k + 1] = ql;
        }
      }
    }
  }
  hdr_spill[col] = spilled ? 1 : 0;
  __threadfence_block();  // (the overflow area is read back by other lanes of the workgroup)
  __syncthreads();
  SMX_TSTAMP(te3);
  SMX_TACC(34, te2, te3);
  // ---- 3. the waypoint slots of the workgroup's rows, a lane per waypoint.  The rows that hold a path first, in memory
  // order (a road of three lanes leaves every fourth row empty, and the slot arithmetic is straight-line code that a lane
  // without a path went through all the same: a fifth round of four for nothing); then the empty rows, zeros only.
  // Four slots per lane
  // and round, their arithmetic written without branches (every read at a clamped index, results selected at the end):
  // the four dependency chains — header, interval search, two knot records, three divisions, the heading wrap — are
  // independent, and straight-line code is what lets the compiler interleave them (a wavefront alone on its SIMD half
  // the time issues one chain's instruction every ten cycles or so).  Only the rare lane look-ups keep their branch.
  int n_rows_live = 0, n_rows_zero = 0, n_rows_spill = 0;  // (uniform: the workgroup is one wavefront)
  {
    static_assert(SMX_BLOCK == 64, "one ballot per 64 rows");
    const int n_rows = SMX_WPT_VEHICLES * P;
    const unsigned long long below = (1ull << threadIdx.x) - 1ull;
    for (int r0 = 0; r0 < n_rows; r0 += SMX_BLOCK) {
      const int r = r0 + (int)threadIdx.x;
      const int src = r < n_rows ? (int)book.src[r] : (int)SMX_ROW_SKIP;
      const bool in_pool = src >= 0 && hdr_spill[src] == 0, in_spill = src >= 0 && hdr_spill[src] != 0;
      const unsigned long long ml = __ballot(in_pool), mz = __ballot(src == SMX_ROW_ZERO), ms = __ballot(in_spill);
      if (in_pool) rows_live[n_rows_live + __popcll(ml & below)] = (unsigned char)r;
      if (src == SMX_ROW_ZERO) rows_zero[n_rows_zero + __popcll(mz & below)] = (unsigned char)r;
      if (in_spill) rows_spill[n_rows_spill + __popcll(ms & below)] = (unsigned char)r;
      n_rows_live += __popcll(ml);
      n_rows_zero += __popcll(mz);
      n_rows_spill += __popcll(ms);
    }
  }
  __syncthreads();
  {
    const float rcp_p = 1.0f / (float)P;  // (row -> vehicle: small_quotient, rows below 128)
    const int drow = SMX_BLOCK / W, di = SMX_BLOCK - drow * W;
    {  // the empty rows of the workgroup's tabled vehicles: zeros, lane ids -1
      int ridx = threadIdx.x / W, i = threadIdx.x - ridx * W;
      for (int e = threadIdx.x; e < n_rows_zero * W; e += SMX_BLOCK) {
        const int row = rows_zero[ridx];
        const int vq = small_quotient(row, rcp_p);
        const size_t q = ((size_t)book.veh[vq] * P + (row - vq * P)) * W + i;
        double* dst = o.wp_pos + q * 3;
        dst[0] = 0.0;
        dst[1] = 0.0;
        dst[2] = 0.0;
        o.wp_heading[q] = 0.0f;
        o.wp_lane_width[q] = 0.0f;
        o.wp_speed_limit[q] = 0.0f;
        o.wp_lane_id[q] = (int16_t)-1;
        o.wp_lane_index[q] = (int8_t)0;
        ridx += drow;
        i += di;
        if (i >= W) {
          i -= W;
          ++ridx;
        }
      }
    }
    const int elems = n_rows_live * W;
    int ridx = threadIdx.x / W, i = threadIdx.x - ridx * W;  // element e = ridx * W + i, advanced by 64 per slot
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < (SMX_SKIP(a, 1 << 21) ? 0 : elems); e0 += U * SMX_BLOCK) {
      int s_row[U], s_i[U], s_vv[U], s_slot[U];
      bool s_in[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s_in[u] = e0 + u * SMX_BLOCK < elems;
        s_row[u] = rows_live[min(ridx, n_rows_live - 1)];
        s_i[u] = i;
        s_vv[u] = small_quotient(s_row[u], rcp_p);
        s_slot[u] = s_row[u] - s_vv[u] * P;
        ridx += drow;
        i += di;
        if (i >= W) {
          i -= W;
          ++ridx;
        }
      }
      SMX_TSTAMP(tr0);
      double ox[U], oy[U], oh[U];
      float ow[U], os[U];
      int oln[U], oli[U], osrc[U], okl[U], oql[U];
      bool owrite[U], ohave[U], oone[U], ointerior[U];
      double oden[U], odt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = s_in[u] ? (int)book.src[s_row[u]] : (int)SMX_ROW_SKIP;
        const bool have = src >= 0 && s_i[u] < (int)book.count[s_row[u]];
        const int sc = have ? src : 0;
        const int n = hdr_n[sc], nr = max((int)hdr_nrec[sc], 1), koff = hdr_off[sc];
        const double t = (s_i[u] == n - 1) ? hdr_D[sc] : (double)s_i[u] * hdr_step[sc];
        // np.interp's interval: the last knot with cum <= t.  Cums do not decrease along the path, so that is the
        // number of knots 1 .. nr-1 with cum <= t (the reads are issued together)
        int j = 0;
#pragma unroll
        for (int k = 1; k <= SMX_WPE_KNOTS; ++k) {
          const double ck = pool[min(koff + k, SMX_WPE_POOL - 1)].cum;
          j += (k < nr && ck <= t) ? 1 : 0;
        }
        const bool interior = j + 1 < nr;
        const int kj = min(koff + j, SMX_WPE_POOL - 1), qj = min(koff + (interior ? j + 1 : j), SMX_WPE_POOL - 1);
        const WpKnot K = pool[kj];
        const WpKnot Q = pool[qj];
        const WpKnotLanes KL = pool_lanes[kj], QL = pool_lanes[qj];
        const double den = interior ? Q.cum - K.cum : 1.0;
        const double dt_ = t - K.cum;
        const double sx = (Q.x - K.x) / den, sy = (Q.y - K.y) / den, sh = (Q.h - K.h) / den;
        // (at or beyond the last knot: the knot itself; t == cum: the lane of the last knot strictly passed)
        double h = interior ? sh * dt_ + K.h : K.h;
        h = (n == 1) ? h : wrap_heading(h);
        ox[u] = have ? (interior ? sx * dt_ + K.x : K.x) : 0.0;
        oy[u] = have ? (interior ? sy * dt_ + K.y : K.y) : 0.0;
        oh[u] = have ? h : 0.0;
        ow[u] = have ? hdr_w0[sc] : 0.0f;
        os[u] = have ? hdr_s0[sc] : 0.0f;
        oln[u] = have ? ((t == K.cum) ? (int)KL.strict : (int)KL.lane) : -1;
        oli[u] = have ? (int)hdr_li0[sc] : 0;
        osrc[u] = src;
        owrite[u] = src != SMX_ROW_SKIP;
        ohave[u] = have;
        oone[u] = hdr_one_lane[sc] != 0;
        ointerior[u] = interior;
        okl[u] = KL.lane;
        oql[u] = QL.lane;
        oden[u] = den;
        odt[u] = dt_;
      }
      SMX_TSTAMP(tr1);
      // a path whose knots do not all lie on its start lane (seldom): lane width / speed limit / index from the tables,
      // interpolated where the interval joins two lanes
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ohave[u] && !oone[u]) {
          double wj = m.lane_width[okl[u]], sj = m.lane_speed[okl[u]];
          if (ointerior[u] && oql[u] != okl[u]) {
            const double sw = (m.lane_width[oql[u]] - wj) / oden[u], ss = (m.lane_speed[oql[u]] - sj) / oden[u];
            wj = sw * odt[u] + wj;
            sj = ss * odt[u] + sj;
          }
          ow[u] = (float)wj;
          os[u] = (float)sj;
          oli[u] = m.lane_index[oln[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ohave[u] && s_i[u] == 0 && s_slot[u] == 0) {  // first waypoint of the vehicle's first path: the trip meter's
          first_wp[s_vv[u]][0] = ox[u];
          first_wp[s_vv[u]][1] = oy[u];
          first_wp[s_vv[u]][2] = oh[u];
        }
        if (owrite[u] && (!SMX_SKIP(a, 1 << 20) || ox[u] == 1.2345e300)) {  // (developer ablation: compute, do not store)
          const size_t q = ((size_t)book.veh[s_vv[u]] * P + s_slot[u]) * W + s_i[u];
          double* dst = o.wp_pos + q * 3;
          dst[0] = ox[u];
          dst[1] = oy[u];
          dst[2] = 0.0;
          o.wp_heading[q] = (float)oh[u];
          o.wp_lane_width[q] = ow[u];
          o.wp_speed_limit[q] = os[u];
          o.wp_lane_id[q] = (int16_t)oln[u];
          o.wp_lane_index[q] = (int8_t)oli[u];
        }
      }
      SMX_TSTAMP(tr2);
      SMX_TACC(38, tr0, tr1);
      SMX_TACC(39, tr1, tr2);
    }
    // the rows whose knots lie in the overflow area (seldom any): the same expressions, one slot at a time, the knot
    // records read back from device memory
    if (n_rows_spill > 0) {
      int ridx = threadIdx.x / W, i = threadIdx.x - ridx * W;
      for (int e = threadIdx.x; e < n_rows_spill * W; e += SMX_BLOCK) {
        const int row = rows_spill[ridx];
        const int vq = small_quotient(row, rcp_p), slot_q = row - vq * P;
        const int sc = (int)book.src[row];
        const bool have = i < (int)book.count[row];
        const WpKnot* gk = spill_knots + sc * (SMX_WPE_KNOTS + 1);
        const WpKnotLanes* gl = spill_lanes + sc * (SMX_WPE_KNOTS + 1);
        const int n = hdr_n[sc], nr = max((int)hdr_nrec[sc], 1);
        const double t = (i == n - 1) ? hdr_D[sc] : (double)i * hdr_step[sc];
        int j = 0;
        for (int k = 1; k < nr; ++k) j += (gk[k].cum <= t) ? 1 : 0;
        const bool interior = j + 1 < nr;
        const WpKnot K = gk[j];
        const WpKnot Q = gk[interior ? j + 1 : j];
        const WpKnotLanes KL = gl[j], QL = gl[interior ? j + 1 : j];
        const double den = interior ? Q.cum - K.cum : 1.0;
        const double dt_ = t - K.cum;
        const double sx = (Q.x - K.x) / den, sy = (Q.y - K.y) / den, sh = (Q.h - K.h) / den;
        double h = interior ? sh * dt_ + K.h : K.h;
        h = (n == 1) ? h : wrap_heading(h);
        const double ox = have ? (interior ? sx * dt_ + K.x : K.x) : 0.0;
        const double oy = have ? (interior ? sy * dt_ + K.y : K.y) : 0.0;
        const double oh = have ? h : 0.0;
        float ow = have ? hdr_w0[sc] : 0.0f, os = have ? hdr_s0[sc] : 0.0f;
        const int oln = have ? ((t == K.cum) ? (int)KL.strict : (int)KL.lane) : -1;
        int oli = have ? (int)hdr_li0[sc] : 0;
        if (have && hdr_one_lane[sc] == 0) {
          double wj = m.lane_width[KL.lane], sj = m.lane_speed[KL.lane];
          if (interior && QL.lane != KL.lane) {
            const double sw = (m.lane_width[QL.lane] - wj) / den, ss = (m.lane_speed[QL.lane] - sj) / den;
            wj = sw * dt_ + wj;
            sj = ss * dt_ + sj;
          }
          ow = (float)wj;
          os = (float)sj;
          oli = m.lane_index[oln];
        }
        if (have && i == 0 && slot_q == 0) {
          first_wp[vq][0] = ox;
          first_wp[vq][1] = oy;
          first_wp[vq][2] = oh;
        }
        const size_t q = ((size_t)book.veh[vq] * P + slot_q) * W + i;
        double* dst = o.wp_pos + q * 3;
        dst[0] = ox;
        dst[1] = oy;
        dst[2] = 0.0;
        o.wp_heading[q] = (float)oh;
        o.wp_lane_width[q] = ow;
        o.wp_speed_limit[q] = os;
        o.wp_lane_id[q] = (int16_t)oln;
        o.wp_lane_index[q] = (int8_t)oli;
        ridx += drow;
        i += di;
        if (i >= W) {
          i -= W;
          ++ridx;
        }
      }
    }
    SMX_TSTAMP(te4);
    SMX_TACC(35, te3, te4);
    // wp_count: [vehicle][0] = number of paths, [1 + slot] = waypoints kept of the path in that row
    const int cells = SMX_WPT_VEHICLES * (P + 1);
    for (int e = threadIdx.x; e < cells; e += SMX_BLOCK) {
      const int vq = e / (P + 1), qq = e - vq * (P + 1);
      if (!book.tabled[vq]) continue;
      o.wp_count[(size_t)book.veh[vq] * (P + 1) + qq] = qq == 0 ? book.paths[vq] : book.count[vq * P + qq - 1];
    }
  }
  __syncthreads();  // first_wp is complete
  SMX_TSTAMP(te5);
  // ---- trip meter + reward (lane 0 of the team): path 0 is the lowest started lane's, row 0 of the vehicle
  if (live && !slow_team && p0 == 0) {
    // trip_meter_update on the words loaded at the top (no new vehicle comes here)
    double dist = trip_dist;
    bool trip_has_wp = trip_has != 0;
    const double last_dist = dist;
    if (n_paths_staged > 0 && trip_counts_waypoint(a, m, gid, total)) {
      const double fwx = first_wp[v][0], fwy = first_wp[v][1], fwh = first_wp[v][2];
      if (!trip_has_wp) {
        SF(SMX_S_TRIP_X) = fwx;
        SF(SMX_S_TRIP_Y) = fwy;
        SF(SMX_S_TRIP_H) = fwh;
        trip_has_wp = true;
      } else {
        const double dx = fwx - trip_x, dy = fwy - trip_y;
        const double nrm = sqrt(dx * dx + dy * dy);
        if (nrm > 0.5) {
          double hvx, hvy;
          radians_to_vec(trip_h, hvx, hvy);
          const double dot = hvx * dx + hvy * dy;
          const double sgn = dot > 0.0 ? 1.0 : (dot < 0.0 ? -1.0 : 0.0);
          dist += sgn * nrm;
          SF(SMX_S_TRIP_X) = fwx;
          SF(SMX_S_TRIP_Y) = fwy;
          SF(SMX_S_TRIP_H) = fwh;
        }
      }
    }
    SF(SMX_S_DIST) = dist;
    o.dist[gid] = dist;
    if (!a.keep_reward_done) {
      o.reward[gid] = dist - last_dist;
      if (o.learner) o.learner[gid] = (float)(dist - last_dist);
    }
    a.st.facts_i32[(size_t)SMX_FI_TRIP_HAS_WP * total + gid] = trip_has_wp ? 1 : 0;
  }
  SMX_TSTAMP(te6);
  SMX_TACC(36, te5, te6);
  SMX_TACC(37, te0, te6);
}

// =================================================================================
// observe role: the rest of Sensors.observe (sensors.py:238-396) and the events / done logic
// (sensors.py:443-594), one thread per vehicle, whole envs per workgroup (env-mates' poses in LDS)
// =================================================================================
// A vehicle of a freshly reset env: state from the spawn table row of `episode`
// (SMARTS.reset / TrapManager, smarts.py:365-460, trap_manager.py:212-230).
__device__ __forceinline__ void respawn_vehicle(const KernelArgs& a, size_t gid, size_t total, int episode) {
  const int row = a.sp.episodes > 0 ? (((episode % a.sp.episodes) + a.sp.episodes) % a.sp.episodes) : 0;
  const double* sp = a.sp.pose + ((size_t)row * total + gid) * 4;
  for (int f = 0; f < SMX_S_COUNT; ++f) SF(f) = 0.0;
  SF(SMX_S_X) = sp[0];
  SF(SMX_S_Y) = sp[1];
  SF(SMX_S_HEADING) = wrap_heading(sp[2]);
  SF(SMX_S_U) = sp[3];
  SF(SMX_S_PREV_X) = sp[0];
  SF(SMX_S_PREV_Y) = sp[1];
  int fl = SMX_F_ALIVE | SMX_F_FIRST;
  const int n_veh = a.cfg.num_vehicles;
  if ((int)(gid % n_veh) >= n_veh - a.cfg.num_social) {
    const double* so = a.sp.social + ((size_t)row * total + gid) * 2;
    SF(SMX_S_MCL_X) = so[0];  // lane
    SF(SMX_S_MCL_Y) = so[1];  // arclength offset
    fl |= SMX_F_SOCIAL;
  }
  a.st.flags[gid] = fl;
  // the slot's knot lists belong to the vehicle that is gone (k_wp_walk only visits alive vehicles)
  if (a.knots.key != nullptr)
    for (int p = 0; p < SMX_WP_LANES; ++p) a.knots.key[gid * SMX_WP_LANES + p] = -1;
  a.st.facts_i32[(size_t)SMX_FI_TRIP_HAS_WP * total + gid] = 0;
  a.st.steps[gid] = 1;  // SensorState.step runs in the tick that creates the vehicle (agent_manager.py:250-258)
}

struct __align__(16) SharedPose {
  double x, y, heading, speed;
  double lane_dist;
  int lane;
  int alive;
  short nl, nlidx;  // lane and lane index an observer reports for this vehicle (-1: none within its length)
  int pad;
};

// The rows of an agent that is gone read as an absent agent's: zeros (lane ids and slots -1).  Written by the whole
// workgroup, thread t of nth striding each array — one lane on its own took 4 096 byte stores for an OGM tile and
// 640 for the waypoint rows, one after the other, and the wavefront that held such a lane ended the kernel (C4: 380
// agents go per tick, one in six wavefronts of k_observe held one: 98 us for a kernel whose wavefronts take 43).
__device__ __forceinline__ void zero_dense_rows(const KernelArgs& a, size_t gid, int t, int nth) {
  const smx_config& c = a.cfg;
  const smx_outputs& o = a.out;
  for (int k = t; k < 3; k += nth) o.ego_pos[gid * 3 + k] = 0.0;
  for (int k = t; k < SMX_EGO_F32_COUNT; k += nth) o.ego_f32[gid * SMX_EGO_F32_COUNT + k] = 0.0f;
  for (int k = t; k < 2; k += nth) o.ego_lane[gid * 2 + k] = -1;
  for (int k = t; k < SMX_EV_COUNT; k += nth) o.events[gid * SMX_EV_COUNT + k] = 0;
  if (t == 0) {
    if (o.collidees) o.collidees[gid] = 0ull;
    o.reward[gid] = 0.0;
    o.dist[gid] = 0.0;
  }
  if (c.sensors & SMX_SENSOR_WAYPOINTS) {
    const int per = c.wp_paths * c.wp_len;
    for (int k = t; k < per * 3; k += nth) o.wp_pos[gid * per * 3 + k] = 0.0;
    for (int k = t; k < per; k += nth) {
      const size_t q = gid * per + k;
      o.wp_heading[q] = 0.0f;
      o.wp_lane_width[q] = 0.0f;
      o.wp_speed_limit[q] = 0.0f;
      o.wp_lane_index[q] = 0;
      o.wp_lane_id[q] = -1;
    }
    for (int k = t; k <= c.wp_paths; k += nth) o.wp_count[gid * (c.wp_paths + 1) + k] = 0;
  }
  if (c.sensors & SMX_SENSOR_NEIGHBORS) {
    for (int k = t; k < c.nb_max * 3; k += nth) {
      o.nb_pos[gid * c.nb_max * 3 + k] = 0.0;
      o.nb_box[gid * c.nb_max * 3 + k] = 0.0f;
    }
    for (int k = t; k < c.nb_max; k += nth) {
      const size_t q = gid * c.nb_max + k;
      o.nb_heading[q] = 0.0f;
      o.nb_speed[q] = 0.0f;
      o.nb_lane_index[q] = 0;
      o.nb_lane_id[q] = -1;
      o.nb_slot[q] = -1;
    }
    if (t == 0) o.nb_count[gid] = 0;
  }
  if (c.via_max > 0 && o.via_near) {
    for (int k = t; k < c.via_max; k += nth) o.via_near[gid * (size_t)c.via_max + k] = -1;
    if (t == 0) {
      o.via_near_count[gid] = 0;
      o.via_hit[gid] = 0;
    }
  }
  auto zero_bytes = [&](uint8_t* base, size_t n) {
    uint8_t* p = base + gid * n;
    if (n % 16 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      for (size_t k = t; k < n / 16; k += nth) reinterpret_cast<int4*>(p)[k] = make_int4(0, 0, 0, 0);
    } else {
      for (size_t k = t; k < n; k += nth) p[k] = 0;
    }
  };
  if ((c.sensors & SMX_SENSOR_OGM) && o.ogm) zero_bytes(o.ogm, (size_t)c.ogm_width * c.ogm_height);
  if ((c.sensors & SMX_SENSOR_DAGM) && o.dagm) zero_bytes(o.dagm, (size_t)c.dagm_width * c.dagm_height);
  if ((c.sensors & SMX_SENSOR_LIDAR) && o.lidar_hit) {
    for (int k = t; k < c.lidar_rays; k += nth) o.lidar_hit[gid * (size_t)c.lidar_rays + k] = 0;
    for (int k = t; k < c.lidar_rays * 3; k += nth) o.lidar_point[gid * (size_t)c.lidar_rays * 3 + k] = 0.0;
  }
}

// The rows of several elements per agent (ego block, events, neighbour rows) go through LDS: every lane fills
// its own vehicle's cells, then the workgroup sweeps each output array in memory order — the rows of its
// vehicles are adjacent — so a store instruction writes consecutive elements instead of one element of 64 rows.
// (Scalars per agent — done, active, reward, counts — are consecutive across lanes as they are.)
#define SMX_NB_STAGE 16  // neighbour rows per agent the staged form handles (nb_max; StdObs keeps 10)
static_assert(SMX_BLOCK * SMX_NB_STAGE * 3 < 4096, "small_quotient's range");
struct ObsStage {
  float ego_f32[SMX_BLOCK][SMX_EGO_F32_COUNT];
  short ego_lane[SMX_BLOCK][2];
  unsigned char events[SMX_BLOCK][SMX_EV_COUNT + 1];
  signed char nb_list[SMX_BLOCK][SMX_NB_STAGE];  // env-mate slot of every neighbour row kept
  unsigned char nb_kept[SMX_BLOCK];
  unsigned char mode[SMX_BLOCK];                 // 1: this vehicle's rows are written this pass
};

__device__ __forceinline__ void observe_role(const KernelArgs& a, const int block) {
  __shared__ SharedPose pose[SMX_BLOCK];
  __shared__ ObsStage stage;
  __shared__ unsigned long long zero_mask;  // vehicles of the workgroup whose rows are to read as an absent agent's
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const smx_outputs& o = a.out;
  const int n_veh = c.num_vehicles;
  const int epb = SMX_BLOCK / n_veh;
  const int local = threadIdx.x;
  if (local == 0) zero_mask = 0ull;
  const int env_local = local / n_veh;
  const int slot = local - env_local * n_veh;
  const int env = block * epb + env_local;
  const bool valid = (env_local < epb) && (env < c.num_envs);
  const size_t total = (size_t)c.num_envs * n_veh;
  const size_t gid = valid ? ((size_t)env * n_veh + slot) : 0;
  const SharedPose* env_pose = pose + env_local * n_veh;

  if (SMX_SKIP(a, 8192)) return;
  SMX_TSTAMP(to0);
  VehState s = {0, 0, 0, 0, 0, 0, 0};
  int flags = 0;
  bool alive = false;
  int my_lane = -1, my_facts = 0;
  double my_lane_dist = SMX_INF;
  if (valid) {
    flags = a.st.flags[gid];
    alive = (flags & SMX_F_ALIVE) != 0;
    s = load_vehicle(a, gid, total);
    if (alive) {
      my_lane = a.st.facts_i32[(size_t)SMX_FI_LANE * total + gid];
      my_facts = a.st.facts_i32[(size_t)SMX_FI_FLAGS * total + gid];
      my_lane_dist = a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + gid];
    }
  }
  const HeadingTrig trig = heading_trig(s.heading);
  const double speed = vehicle_speed(s, trig);
  if (local < SMX_BLOCK) {  // k_first runs this role in a wider workgroup: the extra threads hold nothing
    SharedPose& p = pose[local];
    p.x = s.x;
    p.y = s.y;
    p.heading = wrap_heading(s.heading);
    p.speed = speed;
    p.lane = my_lane;
    p.lane_dist = my_lane_dist;
    p.alive = (valid && alive) ? 1 : 0;
    // what an observer reports as this vehicle's lane: nearest_lane(nv.pose.point, radius=vehicle.length)
    // (sensors.py:244-246; every vehicle on this path has the sedan's length)
    const int nl = (my_lane >= 0 && my_lane_dist < SMX_CHASSIS_LENGTH) ? my_lane : -1;
    p.nl = (short)nl;
    p.nlidx = (short)(nl >= 0 ? m.lane_index[nl] : -1);
    stage.mode[local] = 0;
  }
  __syncthreads();
  SMX_TSTAMP(to1);
  SMX_TACC(7, to0, to1);

  const bool first = (flags & SMX_F_FIRST) != 0;
  const bool social = (flags & SMX_F_SOCIAL) != 0;
  const bool mine = valid && alive && !social && (!a.first_only || first);
  bool done = false;
  int new_flags = flags;  // what the commit kernel makes the vehicle's flags word after this pass
  if (valid && alive && social && first) {
    // nothing to observe: its rows read as an absent agent's
    atomicOr(&zero_mask, 1ull << local);
    o.active[gid] = 0;
    o.done[gid] = 0;
    new_flags = flags & ~SMX_F_FIRST;
  }
  const bool nb_staged = c.nb_max <= SMX_NB_STAGE;
  if (mine) {
    stage.mode[local] = 1;
    const double px = s.x, py = s.y;
    int steps = a.st.steps[gid];
    const int env_ticks = first ? a.st.env_ticks[env] : a.st.env_ticks[env] + 1;  // smarts.py:261-262
    if (!first) ++steps;  // SensorState.step (agent_manager.py:250-258); a new vehicle starts at 1

    // ---- collisions (smarts.py:1270-1291): a new vehicle has not been through a physics step
    bool collided = false;
    unsigned long long collidee_mask = 0ull;
    // One pass over the env-mates serves the collisions' broad phase (circumscribed circles; the narrow phase then
    // runs over the survivors only: a wavefront pays for max-over-lanes(candidates) box tests, not for n_veh) and
    // the neighbourhood (sensors.py:241-266, smarts.py:1191-1208: every other vehicle of the instance within
    // `radius`, in slot order, first nb_max kept) — both look at the same squared distance.
    const bool want_col = !first && !SMX_SKIP(a, 4);
    const bool want_nb = (c.sensors & SMX_SENSOR_NEIGHBORS) && !SMX_SKIP(a, 8);
    const bool nb_in_pass = want_nb && nb_staged;  // (more rows than the staged form holds: the loop further down)
    int nb_cnt = 0;
    {
      unsigned long long cand = 0ull;
      const double reach = sqrt(SMX_CHASSIS_LENGTH * SMX_CHASSIS_LENGTH + SMX_CHASSIS_WIDTH * SMX_CHASSIS_WIDTH) +
                           SMX_COLLISION_LEEWAY;
      const double reach2 = reach * reach;
      const bool nb_all = !(c.nb_radius >= 0.0);
      for (int j = 0; j < n_veh; ++j) {
        if (j == slot) continue;
        const SharedPose& q = env_pose[j];
        if (!q.alive) continue;
        const double dx = px - q.x, dy = py - q.y;
        const double d2 = dx * dx + dy * dy;
        if (want_col && !(d2 > reach2)) cand |= 1ull << j;
        // sqrt(dx^2 + dy^2 + 0^2) <= radius, as a threshold on the squared distance (radius_threshold)
        if (nb_in_pass && (nb_all || d2 <= a.nb_d2_max)) {
          if (nb_cnt < c.nb_max) stage.nb_list[local][nb_cnt] = (signed char)j;
          ++nb_cnt;
        }
      }
      const double my_h = wrap_heading(s.heading);
      // one Collision per collidee (smarts.py:1270-1291): every survivor is tested, not just the first hit
      while (cand != 0ull) {
        const int j = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        const SharedPose& q = env_pose[j];
        if (boxes_within(px, py, my_h, q.x, q.y, q.heading, SMX_CHASSIS_LENGTH, SMX_CHASSIS_WIDTH,
                         SMX_COLLISION_LEEWAY))
          collidee_mask |= 1ull << j;
      }
      collided = collidee_mask != 0ull;
    }
    if (o.collidees) o.collidees[gid] = collidee_mask;

    double lng, lat;
    long_lat_speed(s, trig, lng, lat);
    // ---- ego lane (sensors.py:277-285): nearest lane within max(10, 2 * default lane width)
    const int ego_lane = (my_lane >= 0 && my_lane_dist < fmax(10.0, 2.0 * m.default_lane_width)) ? my_lane : -1;
    // ---- ego vehicle state (sensors.py:314-329; read-back of chassis.py:493-566)
    float* ef = stage.ego_f32[local];  // (ego_pos comes from the pose block)
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
    stage.ego_lane[local][0] = (short)ego_lane;
    stage.ego_lane[local][1] = (short)(ego_lane >= 0 ? m.lane_index[ego_lane] : -1);

    // ---- accelerometer (sensors.py:1053-1084): finite differences over a 3-deep history
    {
      double la[3] = {0, 0, 0}, aa[3] = {0, 0, 0}, lj[3] = {0, 0, 0}, aj[3] = {0, 0, 0};
      if (c.sensors & SMX_SENSOR_ACCELEROMETER) {
        int hist = first ? 0 : ((flags >> SMX_F_HIST_SHIFT) & 3);  // samples held before this one
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

    // ---- neighbourhood: the staged form's rows were listed above; more rows than it holds leave from here
    if (want_nb) {
      int cnt = nb_cnt;
      for (int j = 0; j < n_veh && !nb_staged; ++j) {
        if (j == slot) continue;
        const SharedPose& q = env_pose[j];
        if (!q.alive) continue;
        if (c.nb_radius >= 0.0) {
          double dx = q.x - px, dy = q.y - py, dz = SMX_BASE_HEIGHT - SMX_BASE_HEIGHT;
          double d = sqrt(dx * dx + dy * dy + dz * dz);
          if (!(d <= c.nb_radius)) continue;
        }
        if (cnt < c.nb_max) {
          size_t w = gid * c.nb_max + cnt;
          o.nb_pos[w * 3 + 0] = q.x;
          o.nb_pos[w * 3 + 1] = q.y;
          o.nb_pos[w * 3 + 2] = SMX_BASE_HEIGHT;
          o.nb_box[w * 3 + 0] = (float)SMX_CHASSIS_LENGTH;
          o.nb_box[w * 3 + 1] = (float)SMX_CHASSIS_WIDTH;
          o.nb_box[w * 3 + 2] = (float)SMX_CHASSIS_HEIGHT;
          o.nb_heading[w] = (float)q.heading;
          o.nb_speed[w] = (float)q.speed;
          // nearest_lane(nv.pose.point, radius=vehicle.length) (sensors.py:244-246)
          int nl = (q.lane >= 0 && q.lane_dist < SMX_CHASSIS_LENGTH) ? q.lane : -1;
          o.nb_lane_id[w] = (int16_t)nl;
          o.nb_lane_index[w] = (int8_t)(nl >= 0 ? m.lane_index[nl] : -1);
          o.nb_slot[w] = (int8_t)j;
        }
        ++cnt;
      }
      stage.nb_kept[local] = (unsigned char)(cnt < c.nb_max ? cnt : c.nb_max);
      for (int q0 = cnt; q0 < c.nb_max && !nb_staged; ++q0) {
        size_t w = gid * c.nb_max + q0;
        o.nb_pos[w * 3] = o.nb_pos[w * 3 + 1] = o.nb_pos[w * 3 + 2] = 0.0;
        o.nb_box[w * 3] = o.nb_box[w * 3 + 1] = o.nb_box[w * 3 + 2] = 0.0f;
        o.nb_heading[w] = 0.0f;
        o.nb_speed[w] = 0.0f;
        o.nb_lane_index[w] = 0;
        o.nb_lane_id[w] = -1;
        o.nb_slot[w] = -1;
      }
      o.nb_count[gid] = (uint8_t)(cnt > 255 ? 255 : cnt);
    }

    SMX_TSTAMP(to2);
    SMX_TACC(8, to1, to2);
    // ---- driven path (sensors.py:842-877): running length of the last window
    bool is_not_moving = false;
    if (a.st.driven_path != nullptr) {
      double* ring = a.st.driven_path + gid * (size_t)SMX_DRIVEN_PATH_LEN;
      double sum = SF(SMX_S_PATH_SUM);
      int window_pts = (int)floor(c.not_moving_time / c.dt + 1e-9) + 1;
      if (window_pts > SMX_DRIVEN_PATH_LEN) window_pts = SMX_DRIVEN_PATH_LEN;
      const int K = window_pts - 1;  // segments in a full window
      if (first) {
        sum = 0.0;  // a reset records a point but no segment
      } else {
        double dx = SF(SMX_S_PREV_X) - px, dy = SF(SMX_S_PREV_Y) - py;
        double seg = sqrt(dx * dx + dy * dy);
        int nseg = steps - 1;  // segments recorded so far, this one included
        ring[(nseg - 1) % SMX_DRIVEN_PATH_LEN] = seg;
        sum += seg;
        if (nseg > K) sum -= ring[(nseg - 1 - K) % SMX_DRIVEN_PATH_LEN];
      }
      SF(SMX_S_PATH_SUM) = sum;
      double elapsed = (double)env_ticks * c.dt;
      if (!(elapsed < c.not_moving_time)) is_not_moving = sum < c.not_moving_distance;
    }

    // ---- via sensor (ViaSensor.__call__, sensors.py:1103-1146; acquisition range 40 m and speed
    //      tolerance 1.5 m/s from vehicle.py:553-557)
    if (c.via_max > 0 && a.vias != nullptr) {
      const int v_a = a.via_slot_off[slot], v_b = a.via_slot_off[slot + 1];
      int32_t* consumed_p = a.st.facts_i32 + (size_t)SMX_FI_VIA_CONSUMED * total + gid;
      unsigned consumed = first ? 0u : (unsigned)*consumed_p;
      int hit = 0, cnt = 0;
      int8_t* near = o.via_near + gid * (size_t)c.via_max;
      for (int v = v_a; v < v_b; ++v) {
        const smx_via via = a.vias[v];
        double qx, qy;
        lane_center_at_point(m, via.lane, px, py, qx, qy);
        const double lx = qx - px, ly = qy - py;
        if (lx * lx + ly * ly > 40.0 * 40.0) continue;
        const double dx = via.x - px, dy = via.y - py;
        const double d2 = dx * dx + dy * dy;
        // sorted(near_points, key=squared distance): stable insertion keeps list order among equals
        // (the kept rows live in the output itself; a row's distance is recomputed from the table)
        int pos = cnt < c.via_max ? cnt : c.via_max;
        while (pos > 0) {
          const smx_via prev = a.vias[v_a + near[pos - 1]];
          const double ex = prev.x - px, ey = prev.y - py;
          if (ex * ex + ey * ey > d2)
            --pos;
          else
            break;
        }
        if (pos < c.via_max) {
          const int last = (cnt < c.via_max ? cnt : c.via_max - 1);
          for (int k = last; k > pos; --k) near[k] = near[k - 1];
          near[pos] = (int8_t)(v - v_a);
        }
        ++cnt;
        const int bit = 1 << (v - v_a);
        // np.isclose(speed, required_speed, atol=1.5) with the default rtol = 1e-5
        const bool speed_ok = fabs(speed - via.required_speed) <= 1.5 + 1e-5 * fabs(via.required_speed);
        if (d2 <= via.hit_distance * via.hit_distance && !(consumed & bit) && speed_ok) {
          consumed |= bit;
          hit |= bit;
        }
      }
      for (int k = (cnt < c.via_max ? cnt : c.via_max); k < c.via_max; ++k) near[k] = -1;
      o.via_near_count[gid] = (uint8_t)(cnt > 255 ? 255 : cnt);
      o.via_hit[gid] = hit;
      *consumed_p = (int32_t)consumed;
    }

    // ---- events + done (sensors.py:443-489)
    // Mission.is_complete -> PositionalGoal.is_reached (plan.py:116-120, 220-222); EndlessGoal never (:76-84)
    RouteFilter route;
    const bool fixed_route = route.fixed_route(a.missions, slot, m.n_roads);
    bool reached_goal = false;
    if (fixed_route) {
      const double gx = a.missions.goal[3 * slot], gy = a.missions.goal[3 * slot + 1], gr = a.missions.goal[3 * slot + 2];
      const double sqr_dist = (s.x - gx) * (s.x - gx) + (s.y - gy) * (s.y - gy);
      reached_goal = sqr_dist <= gr * gr;
    }
    const bool is_off_road = !(my_facts & SMX_FACT_ON_ROAD);           // sensors.py:498-500
    const bool is_on_shoulder = ((my_facts >> SMX_FACT_CORNER_SHIFT) & 15) != 15;  // sensors.py:502-509
    const bool reached_max = c.max_episode_steps > 0 && steps >= c.max_episode_steps;
    bool is_off_route, is_wrong_way;
    {
      // sensors.py:527-594
      double radius = sqrt(SMX_CHASSIS_LENGTH * SMX_CHASSIS_LENGTH + SMX_CHASSIS_WIDTH * SMX_CHASSIS_WIDTH) * 0.5 + 5.0;
      int nl = (my_lane >= 0 && my_lane_dist < radius) ? my_lane : -1;
      if (nl < 0) {
        is_off_route = true;
        is_wrong_way = false;
      } else {
        is_off_route = false;
        is_wrong_way = false;
        if (!m.lane_in_junction[nl] && !SMX_SKIP(a, 64)) {
          const double target = a.st.facts_f64[(size_t)SMX_FF_LANE_HEADING * total + gid];  // k_scan
          is_wrong_way = fabs(heading_relative_to(s.heading, target)) > 0.5 * SMX_PI;
        }
        // an endless mission has no route roads: on route (:556-561); else the nearest lane's road must be
        // one of them, or a junction, or the lane has an oncoming neighbour that is (:563-574)
        if (fixed_route && !route.has(m, m.lane_road[nl]) && !m.lane_in_junction[nl])
          is_off_route = !oncoming_lane_on_route(m, route, nl, lane_offset_along(m, nl, s.x, s.y));
      }
    }
    unsigned char* ev = stage.events[local];
    ev[SMX_EV_COLLISIONS] = collided ? 1 : 0;
    ev[SMX_EV_OFF_ROAD] = is_off_road ? 1 : 0;
    ev[SMX_EV_OFF_ROUTE] = is_off_route ? 1 : 0;
    ev[SMX_EV_ON_SHOULDER] = is_on_shoulder ? 1 : 0;
    ev[SMX_EV_WRONG_WAY] = is_wrong_way ? 1 : 0;
    ev[SMX_EV_NOT_MOVING] = is_not_moving ? 1 : 0;
    ev[SMX_EV_REACHED_GOAL] = reached_goal ? 1 : 0;
    ev[SMX_EV_REACHED_MAX_EPISODE_STEPS] = reached_max ? 1 : 0;
    // ---- DoneCriteria.agents_alive (sensors.py:404-441): agents registered at the start of the tick
    bool agents_alive_done = false;
    if (c.alive_min_ego > 0 || c.alive_min_total > 0 || c.alive_lists > 0) {
      unsigned long long alive_mask = 0ull;
      const int n_agents = n_veh - c.num_social;
      for (int j = 0; j < n_agents; ++j)
        if (env_pose[j].alive) alive_mask |= 1ull << j;
      const int n_alive = __popcll(alive_mask);
      // no social *agents* exist on this path, so every registered agent is an ego agent
      if (c.alive_min_ego > 0 && n_alive < c.alive_min_ego) agents_alive_done = true;
      if (c.alive_min_total > 0 && n_alive < c.alive_min_total) agents_alive_done = true;
      for (int k = 0; k < c.alive_lists && k < SMX_MAX_ALIVE_LISTS; ++k)
        if (__popcll(alive_mask & c.alive_list_mask[k]) < c.alive_list_min[k]) agents_alive_done = true;
    }
    ev[SMX_EV_AGENTS_ALIVE_DONE] = agents_alive_done ? 1 : 0;
    const uint32_t dc = c.done_criteria;
    done = (is_off_road && (dc & SMX_DONE_OFF_ROAD)) || reached_goal || reached_max ||
           (is_on_shoulder && (dc & SMX_DONE_ON_SHOULDER)) || (collided && (dc & SMX_DONE_COLLISION)) ||
           (is_not_moving && (dc & SMX_DONE_NOT_MOVING)) || (is_off_route && (dc & SMX_DONE_OFF_ROUTE)) ||
           (is_wrong_way && (dc & SMX_DONE_WRONG_WAY)) || agents_alive_done;
    if (first) done = false;  // sensors.py:465: `not sim.resetting and (...)`: reset observations never end an agent

    // ---- teardown (smarts.py:314, 329-363)
    flags &= ~SMX_F_FIRST;
    if (done) flags &= ~SMX_F_ALIVE;
    a.st.steps[gid] = steps;
    new_flags = flags;  // applied by k_commit: the waypoints role of this launch still reads the old word
    o.active[gid] = done ? 0 : 1;
    if (!a.keep_reward_done) {
      o.done[gid] = done ? 1 : 0;
      if (o.learner) o.learner[total + gid] = done ? 1.0f : 0.0f;
    }
  } else if (valid && !a.first_only) {
    if (o.learner && (!alive || social)) {
      o.learner[gid] = 0.0f;  // no agent in this slot: absent from the learner block
      o.learner[total + gid] = 0.0f;
    }
    // an agent whose vehicle is gone: absent from the observations (zeros), done stays 0
    if (!alive && (o.active[gid] != 0 || o.done[gid] != 0)) {
      atomicOr(&zero_mask, 1ull << local);
      o.done[gid] = 0;
      o.active[gid] = 0;
    }
  }
  if (valid) a.st.facts_i32[(size_t)SMX_FI_FLAGS_NEXT * total + gid] = new_flags;
  // ---- copy-out of the staged rows, every array in memory order over the workgroup's vehicles
  __syncthreads();
  SMX_TSTAMP(to2c);
  {
    const int nth = (int)blockDim.x;
    for (unsigned long long zm = zero_mask; zm != 0ull; zm &= zm - 1ull)  // (uniform in the workgroup)
      zero_dense_rows(a, (size_t)block * epb * n_veh + (__ffsll((long long)zm) - 1), local, nth);
    // (element -> vehicle, row: quotients by the run-time row lengths, below 4096 / by at most 64 — exact in float32
    // with half a unit added; an integer division is some forty instructions, and these sweeps were half the kernel)
    const float rcp_veh = 1.0f / (float)n_veh;
    const int wg_veh = epb * n_veh;                           // vehicles of this workgroup (<= 64), gids g0 ...
    const size_t g0 = (size_t)block * epb * n_veh;
    for (int e = local; e < wg_veh * 3; e += nth) {           // ego_pos
      const int v = e / 3, q = e - v * 3;
      if (!stage.mode[v]) continue;
      o.ego_pos[g0 * 3 + e] = q == 0 ? pose[v].x : (q == 1 ? pose[v].y : SMX_BASE_HEIGHT);
    }
    for (int e = local; e < wg_veh * SMX_EGO_F32_COUNT; e += nth) {
      const int v = e / SMX_EGO_F32_COUNT;
      if (stage.mode[v]) o.ego_f32[g0 * SMX_EGO_F32_COUNT + e] = stage.ego_f32[v][e - v * SMX_EGO_F32_COUNT];
    }
    for (int e = local; e < wg_veh * 2; e += nth)
      if (stage.mode[e >> 1]) o.ego_lane[g0 * 2 + e] = stage.ego_lane[e >> 1][e & 1];
    for (int e = local; e < wg_veh * SMX_EV_COUNT; e += nth) {
      const int v = e / SMX_EV_COUNT;
      if (stage.mode[v]) o.events[g0 * SMX_EV_COUNT + e] = stage.events[v][e - v * SMX_EV_COUNT];
    }
    if ((c.sensors & SMX_SENSOR_NEIGHBORS) && nb_staged && !SMX_SKIP(a, 8)) {
      const int K = c.nb_max;
      const float rcp_k3 = 1.0f / (float)(K * 3), rcp_k = 1.0f / (float)K;
      for (int e = local; e < wg_veh * K * 3; e += nth) {     // nb_pos, nb_box
        const int v = small_quotient(e, rcp_k3), r = e - v * (K * 3), k = r / 3, q = r - k * 3;
        if (!stage.mode[v]) continue;
        const bool held = k < (int)stage.nb_kept[v];
        const SharedPose& P = pose[small_quotient(v, rcp_veh) * n_veh + (held ? (int)stage.nb_list[v][k] : 0)];
        o.nb_pos[g0 * K * 3 + e] = held ? (q == 0 ? P.x : (q == 1 ? P.y : SMX_BASE_HEIGHT)) : 0.0;
        o.nb_box[g0 * K * 3 + e] =
            held ? (float)(q == 0 ? SMX_CHASSIS_LENGTH : (q == 1 ? SMX_CHASSIS_WIDTH : SMX_CHASSIS_HEIGHT)) : 0.0f;
      }
      for (int e = local; e < wg_veh * K; e += nth) {         // the scalar neighbour rows
        const int v = small_quotient(e, rcp_k), k = e - v * K;
        if (!stage.mode[v]) continue;
        const bool held = k < (int)stage.nb_kept[v];
        const int j = held ? (int)stage.nb_list[v][k] : 0;
        const SharedPose& P = pose[small_quotient(v, rcp_veh) * n_veh + j];
        const size_t w = g0 * K + e;
        o.nb_heading[w] = held ? (float)P.heading : 0.0f;
        o.nb_speed[w] = held ? (float)P.speed : 0.0f;
        o.nb_lane_id[w] = (int16_t)(held ? P.nl : -1);
        o.nb_lane_index[w] = (int8_t)(held ? P.nlidx : 0);
        o.nb_slot[w] = (int8_t)(held ? j : -1);
      }
    }
  }
  SMX_TSTAMP(to3);
  SMX_TACC(6, to0, to3);
  SMX_TACC(45, to2c, to3);
}

// =================================================================================
// k_commit: the end of a pass, after every sensor role has read the old flags: apply the flags the
// observe role decided (teardown of done agents, smarts.py:314, 329-363), per-env done count and
// dones["__all__"] (hiway_env.py:258-261), and the auto-reset respawn (parallel_env.py:303-309) —
// the reset pass that follows builds the first observations of the restarted envs.
// One thread per vehicle, whole envs per workgroup.
// =================================================================================
__device__ __forceinline__ void commit_role(const KernelArgs& a, const int block) {
  __shared__ int env_new_done[SMX_BLOCK];
  __shared__ int env_respawn[SMX_BLOCK];
  __shared__ int env_first_alive[SMX_BLOCK];
  const smx_config& c = a.cfg;
  const smx_outputs& o = a.out;
  const int n_veh = c.num_vehicles;
  const int epb = SMX_BLOCK / n_veh;
  const int local = threadIdx.x;
  const int env_local = local / n_veh;
  const int slot = local - env_local * n_veh;
  const int env = block * epb + env_local;
  const bool valid = (env_local < epb) && (env < c.num_envs);
  const size_t total = (size_t)c.num_envs * n_veh;
  const size_t gid = valid ? ((size_t)env * n_veh + slot) : 0;
  if (local < SMX_BLOCK) {
    env_new_done[local] = 0;
    env_respawn[local] = 0;
  }
  __syncthreads();
  if (valid) {
    const int old_flags = a.st.flags[gid];
    const int new_flags = a.st.facts_i32[(size_t)SMX_FI_FLAGS_NEXT * total + gid];
    a.st.flags[gid] = new_flags;
    if ((old_flags & SMX_F_ALIVE) && !(new_flags & SMX_F_ALIVE)) atomicAdd(&env_new_done[env_local], 1);
    if (slot == 0) env_first_alive[env_local] = (new_flags & SMX_F_ALIVE) ? 1 : 0;
  }
  __syncthreads();
  if (valid && slot == 0) {
    if (!a.first_only) {
      int dcnt = a.st.env_done_count[env] + env_new_done[env_local];
      a.st.env_done_count[env] = dcnt;
      a.st.env_ticks[env] = a.st.env_ticks[env] + 1;
      bool all_done = dcnt >= n_veh - c.num_social;  // every agent (hiway_env.py:258-261)
      o.env_done[env] = all_done ? 1 : 0;
      a.st.env_reset_pending[env] = 0;
      env_respawn[env_local] = (all_done && c.auto_reset) ? 1 : 0;
    } else if (!a.keep_reward_done) {
      if (a.st.env_done_count[env] == 0 && env_first_alive[env_local]) o.env_done[env] = 0;
    }
  }
  __syncthreads();
  const bool respawn = valid && env_respawn[valid ? env_local : 0] != 0;
  int next_episode = 0;
  if (respawn && o.final_ego_pos != nullptr) {
    // the finishing tick's rows, before the reset pass writes the next episode's first observation over them
    // (parallel_env.py:303-309: what info[agent]["env_obs"] holds for the agents that ended with their env)
    for (int k = 0; k < 3; ++k) o.final_ego_pos[gid * 3 + k] = o.ego_pos[gid * 3 + k];
    for (int k = 0; k < SMX_EGO_F32_COUNT; ++k) o.final_ego_f32[gid * SMX_EGO_F32_COUNT + k] = o.ego_f32[gid * SMX_EGO_F32_COUNT + k];
    o.final_ego_lane[gid * 2] = o.ego_lane[gid * 2];
    o.final_ego_lane[gid * 2 + 1] = o.ego_lane[gid * 2 + 1];
    for (int k = 0; k < SMX_EV_COUNT; ++k) o.final_events[gid * SMX_EV_COUNT + k] = o.events[gid * SMX_EV_COUNT + k];
    o.final_dist[gid] = o.dist[gid];
  }
  if (respawn) {
    next_episode = a.st.env_episode[env] + 1;
    respawn_vehicle(a, gid, total, next_episode);
  }
  __syncthreads();  // every thread of the env has read env_episode
  if (respawn && slot == 0) {
    a.st.env_episode[env] = next_episode;
    a.st.env_done_count[env] = 0;
    a.st.env_ticks[env] = c.reset_elapsed_steps;
  }
}

__global__ void __launch_bounds__(SMX_BLOCK) k_commit(const KernelArgs a) { commit_role(a, (int)blockIdx.x); }

// =================================================================================
// OGM role: occupancy grid map sensor (OGMSensor, sensors.py:719-758): one wavefront per observing
// vehicle.  The H x W byte tile is built in LDS (lane j rasterises env-mate j's footprint over the
// few pixels its bounding rectangle touches) and leaves as full 16-byte pieces — the kernel is
// bound by its own 4 KiB-per-agent output stream.  Pixel rule (substitution for the Panda3D
// orthographic render, renderer.py:325-395): a pixel is 255 iff its centre lies inside a vehicle's
// oriented chassis rectangle; view centred on the vehicle, +row = behind, row 0 = ahead
// (np.flipud, sensors.py:748), extent width*res x height*res (renderer.py:384-385).
// =================================================================================
struct OgmMate {
  double cx, cy, vfx, vfy, vrx, vry;  // centre and axes of the footprint in the ego frame
  int c0, r0, bw, n_px;               // pixel rectangle: first column / row, width, pixel count
};

__device__ __forceinline__ void ogm_role(const KernelArgs& a, const int block) {
  extern __shared__ __align__(16) unsigned char tile[];
  const smx_config& c = a.cfg;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = (size_t)block;
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  const bool live = (flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST));
  if (!live) return;  // uniform for the whole workgroup
  const int W = c.ogm_width, H = c.ogm_height;
  const int n_veh = c.num_vehicles;
  const int env = (int)(gid / n_veh);
  const int bytes = W * H;
  for (int k = threadIdx.x; k < bytes / 4; k += SMX_BLOCK) reinterpret_cast<int*>(tile)[k] = 0;
  __syncthreads();
  const double res = c.ogm_resolution;
  const double ex0 = SF(SMX_S_X), ey0 = SF(SMX_S_Y), eh = wrap_heading(SF(SMX_S_HEADING));
  const double rx = cos(eh), ry = sin(eh);    // ego right axis
  const double fx = -sin(eh), fy = cos(eh);   // ego forward axis
  // Pass 1, lane j = env-mate j: its footprint in the ego frame and the pixel rectangle that can hold
  // it.  Pass 2: the mates whose rectangle meets the view (a ballot; typically a handful of the env)
  // are drawn one after the other with the wavefront's lanes over the rectangle's pixels — a lane per
  // mate would make every lane wait for the mate with the most pixels while most lanes draw nothing.
  __shared__ OgmMate mates[SMX_BLOCK];
  const double hl = 0.5 * SMX_CHASSIS_LENGTH, hw = 0.5 * SMX_CHASSIS_WIDTH;
  for (int base = 0; base < n_veh; base += SMX_BLOCK) {
    const int j = base + (int)threadIdx.x;
    bool in_view = false;
    if (j < n_veh) {
      const size_t og = (size_t)env * n_veh + j;
      if (a.st.flags[og] & SMX_F_ALIVE) {
        const double vx = a.st.f64[(size_t)SMX_S_X * total + og], vy = a.st.f64[(size_t)SMX_S_Y * total + og];
        const double vh = wrap_heading(a.st.f64[(size_t)SMX_S_HEADING * total + og]);
        const double dx = vx - ex0, dy = vy - ey0;
        const double cx = dx * rx + dy * ry, cy = dx * fx + dy * fy;  // centre in the ego frame
        // the mate's axes in the ego frame from each vehicle's own cos / sin (the rule k_ogm_env shares)
        const double cm = cos(vh), sm = sin(vh);
        const double vfx = cm * ry - sm * rx, vfy = sm * ry + cm * rx, vrx = cm * rx + sm * ry, vry = sm * rx - cm * ry;
        const double ext_x = fabs(vfx) * hl + fabs(vrx) * hw, ext_y = fabs(vfy) * hl + fabs(vry) * hw;
        // pixel centre (r, col): x = (col + 0.5 - W/2) res, y = (H/2 - (r + 0.5)) res
        int c0 = (int)floor((cx - ext_x) / res + 0.5 * W - 0.5) - 1, c1 = (int)ceil((cx + ext_x) / res + 0.5 * W - 0.5) + 1;
        int r0 = (int)floor(0.5 * H - 0.5 - (cy + ext_y) / res) - 1, r1 = (int)ceil(0.5 * H - 0.5 - (cy - ext_y) / res) + 1;
        c0 = max(c0, 0);
        r0 = max(r0, 0);
        c1 = min(c1, W - 1);
        r1 = min(r1, H - 1);
        if (c0 <= c1 && r0 <= r1) {
          in_view = true;
          OgmMate& q = mates[threadIdx.x];
          q.cx = cx;
          q.cy = cy;
          q.vfx = vfx;
          q.vfy = vfy;
          q.vrx = vrx;
          q.vry = vry;
          q.c0 = c0;
          q.r0 = r0;
          q.bw = c1 - c0 + 1;
          q.n_px = (c1 - c0 + 1) * (r1 - r0 + 1);
        }
      }
    }
    unsigned long long todo = __ballot(in_view);
    __syncthreads();
    while (todo != 0ull) {  // uniform
      const OgmMate q = mates[__ffsll((long long)todo) - 1];
      todo &= todo - 1ull;
      for (int p = (int)threadIdx.x; p < q.n_px; p += SMX_BLOCK) {
        const int r = q.r0 + p / q.bw, col = q.c0 + p % q.bw;
        const double py = (0.5 * H - (r + 0.5)) * res - q.cy;
        const double px = (col + 0.5 - 0.5 * W) * res - q.cx;
        if (fabs(px * q.vfx + py * q.vfy) <= hl && fabs(px * q.vrx + py * q.vry) <= hw) tile[r * W + col] = 255;
      }
    }
    __syncthreads();  // before the stage is reused (envs of more than 64 vehicles do not exist, but the loop is general)
  }
  __syncthreads();
  int4* dst = reinterpret_cast<int4*>(a.out.ogm + gid * (size_t)bytes);
  for (int k = threadIdx.x; k < bytes / 16; k += SMX_BLOCK) dst[k] = reinterpret_cast<const int4*>(tile)[k];
}

// k_ogm_env (large batches): the same tiles, one workgroup of four wavefronts per ENV.  The env's poses are
// loaded once, with one cos / sin pair per vehicle (a workgroup per observer reloads all its mates and takes
// a sine and a cosine per mate: n x n of each per env); every wavefront then builds the tiles of a quarter of
// the observers, one after the other, in its own LDS tile.
struct OgmPose {
  double x, y, ch, sh;  // centre, cos / sin of the wrapped heading
  int alive, observes;
};
#define SMX_OGM_WAVES 4
#ifndef SMX_SCAN_WIDE_MAX_VEHICLES  // the team scan halves take eight lanes a vehicle up to this many vehicles, four above
#define SMX_SCAN_WIDE_MAX_VEHICLES 65536
#endif
#ifndef SMX_ONE_LANE_ON_SPLIT_MAPS  // developer: the one-lane cut on maps whose lanes split too
#define SMX_ONE_LANE_ON_SPLIT_MAPS 0
#endif
#ifndef SMX_ONE_LANE_MIN_VEHICLES  // the one-lane cut's seeds half is the one-lane kernel + slow chain from this many vehicles on
#define SMX_ONE_LANE_MIN_VEHICLES 114688
#endif
#ifndef SMX_OGM_ENV_MIN_VEHICLES  // small form: OGM tiles by k_ogm_env from this many vehicles on (smarts_amd/engine.py mirrors it)
#define SMX_OGM_ENV_MIN_VEHICLES 8192
#endif
#ifndef SMX_SIDE_PRIO  // developer: side streams that get the default priority instead of the lowest (bit i = side i)
#define SMX_SIDE_PRIO 0
#endif
#ifndef SMX_FACTS_EARLY_MAX  // the facts half leaves with the grid kernels up to this many vehicles, else after the seeds half
#define SMX_FACTS_EARLY_MAX 32768
#endif
// orders a wavefront's own LDS traffic for the compiler (the hardware keeps a wavefront's LDS operations in order)
#define SMX_WAVE_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// a double of another lane of the wavefront (lane index uniform): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// OBS observers per wavefront at a time (two while the env has at most 32 vehicles and eight tiles fit the LDS a
// workgroup may have): lane = (observer, env-mate) for the footprints' rectangles, so that a wavefront of 32-vehicle
// envs is full; the footprints in view are then drawn two at a time, each by a half wavefront of 8 rows x 4 columns
// (a car ahead is 3 x 6 pixels at 0.78 m per pixel: one step), its record fetched from the lane that holds it with
// ds_bpermute.  390 -> 1xx vector instructions per tile (a third of the headline tick's were this kernel's).
template <int OBS>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 8))) __launch_bounds__(SMX_OGM_WAVES * 64) k_ogm_env(const KernelArgs a) {
  SMX_TSTAMP(span0);
  extern __shared__ __align__(16) unsigned char tiles[];  // [SMX_OGM_WAVES][OBS][H * W]
  __shared__ OgmPose pose[SMX_BLOCK];
  __shared__ unsigned char observers[SMX_BLOCK];  // the env's observing slots, compacted: the wavefronts share them evenly
  __shared__ int n_observers;                     // however many agents of the env are gone
  const smx_config& c = a.cfg;
  const int n_veh = c.num_vehicles;
  const size_t total = (size_t)c.num_envs * n_veh;
  const int env = (int)blockIdx.x;
  const int W = c.ogm_width, H = c.ogm_height, bytes = W * H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((int)threadIdx.x < n_veh) {
    const size_t gid = (size_t)env * n_veh + threadIdx.x;
    const int flags = a.st.flags[gid];
    OgmPose p;
    p.alive = (flags & SMX_F_ALIVE) ? 1 : 0;
    p.observes = ((flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST))) ? 1 : 0;
    p.x = SF(SMX_S_X);
    p.y = SF(SMX_S_Y);
    const double h = wrap_heading(SF(SMX_S_HEADING));
    p.ch = cos(h);
    p.sh = sin(h);
    pose[threadIdx.x] = p;
  }
  if (wave == 0) {  // (n_veh <= 64: the first wavefront sees every slot)
    const bool obs_here = lane < n_veh && pose[lane].observes;
    const unsigned long long om = __ballot(obs_here);
    if (obs_here) observers[__popcll(om & ((1ull << lane) - 1ull))] = (unsigned char)lane;
    if (lane == 0) n_observers = __popcll(om);
  }
  __syncthreads();
  unsigned char* tile = tiles + (size_t)wave * OBS * bytes;
  const double res = c.ogm_resolution;
  const double inv_res = 1.0 / res;  // for the pixel RECTANGLES only (an enumeration bound with a margin on every
                                     // side); the pixel test itself keeps the oracle's arithmetic
  const double hl = 0.5 * SMX_CHASSIS_LENGTH, hw = 0.5 * SMX_CHASSIS_WIDTH;
  const int half = lane >> 5, in_half = lane & 31;
  const int mate = OBS == 2 ? in_half : lane;
  const int mate_clamped = min(mate, n_veh - 1);  // (every lane computes; lanes past the env's vehicles are masked below)
  const int lr = in_half >> 2, lc = in_half & 3;  // this lane's pixel of a drawing step: 8 rows x 4 columns per half
  // pixel centre (r, col): x = (col + 0.5 - W/2) res, y = (H/2 - (r + 0.5)) res; the sums in front of `res` are
  // exact in any order (integers and halves), so the constants are folded
  const double col_bias = 0.5 - 0.5 * W, row_bias = 0.5 * H - 0.5;
  const int n_obs = n_observers;
  const int per_round = SMX_OGM_WAVES * OBS;
  const int rounds = (n_obs + per_round - 1) / per_round;
  // Each wavefront owns its tiles: inside the loop only lanes of ONE wavefront exchange data through LDS, whose
  // operations a wavefront issues in order — a scheduling fence is all that is needed (four workgroup barriers per
  // round made the four wavefronts wait for the slowest one's rectangles)
#pragma nounroll
  for (int it = 0; it < rounds; ++it) {
    const int turn0 = (it * SMX_OGM_WAVES + wave) * OBS;  // uniform in the wavefront
    const int n_live = min(OBS, n_obs - turn0);
    if (n_live <= 0) break;  // (turns grow with `it`)
    for (int k = lane; k < n_live * (bytes / 16); k += 64) reinterpret_cast<int4*>(tile)[k] = make_int4(0, 0, 0, 0);
    const int my_turn = turn0 + (OBS == 2 ? half : 0);
    const bool live = my_turn < n_obs;
    const int obs = (int)observers[live ? my_turn : turn0];
    // the footprint of vehicle `mate` in this observer's frame stays in this lane's registers (no LDS copy of it)
    const OgmPose e = pose[obs];
    const OgmPose v = pose[mate_clamped];
    const double rx = e.ch, ry = e.sh;   // ego right axis
    const double fx = -e.sh, fy = e.ch;  // ego forward axis
    const double dx = v.x - e.x, dy = v.y - e.y;
    const double cx = dx * rx + dy * ry, cy = dx * fx + dy * fy;  // centre in the ego frame
    const double cm = v.ch, sm = v.sh;
    const double vfx = cm * ry - sm * rx, vfy = sm * ry + cm * rx, vrx = cm * rx + sm * ry, vry = sm * rx - cm * ry;
    const double ext_x = fabs(vfx) * hl + fabs(vrx) * hw, ext_y = fabs(vfy) * hl + fabs(vry) * hw;
    // (a pixel centre inside the footprint lies inside its bounding box: columns ceil(lo) .. floor(hi).  The bounds
    // are rounded — a dozen operations on numbers below 1e3 pixels, errors of 1e-12 — and the pixel test accepts a
    // centre its own rounding puts on the edge: a millionth of a pixel on every side covers both.  A whole pixel of
    // margin made a car ahead, 3 x 6 pixels, a rectangle of 5 x 8.)
    const double slack = 1e-6;
    // (clamped as doubles first: a mate far away must not overflow the conversion)
    const double c_lo = fmax((cx - ext_x) * inv_res + 0.5 * W - 0.5 - slack, -1.0), c_hi = fmin((cx + ext_x) * inv_res + 0.5 * W - 0.5 + slack, (double)W);
    const double r_lo = fmax(0.5 * H - 0.5 - (cy + ext_y) * inv_res - slack, -1.0), r_hi = fmin(0.5 * H - 0.5 - (cy - ext_y) * inv_res + slack, (double)H);
    const int c0 = max((int)ceil(c_lo), 0), c1 = min((int)floor(c_hi), W - 1);
    const int r0 = max((int)ceil(r_lo), 0), r1 = min((int)floor(r_hi), H - 1);
    const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
    const bool in_view = live && mate < n_veh && v.alive != 0 && bw > 0 && bh > 0;
    unsigned long long todo = __ballot(in_view);
    SMX_WAVE_SYNC();
    while (todo != 0ull) {  // uniform in the wavefront: two footprints per turn, one per half
      const int s0 = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const int s1 = todo != 0ull ? __ffsll((long long)todo) - 1 : s0;
      const bool two = todo != 0ull;
      todo &= todo - 1ull;  // (0 & anything = 0)
      const int src = half ? s1 : s0;
      const bool drawing = half == 0 || two;
      const double qcx = __shfl(cx, src), qcy = __shfl(cy, src);
      const double qvfx = __shfl(vfx, src), qvfy = __shfl(vfy, src), qvrx = __shfl(vrx, src), qvry = __shfl(vry, src);
      const int qc0 = __shfl(c0, src), qr0 = __shfl(r0, src), qbw = __shfl(bw, src), qbh = __shfl(bh, src);
      unsigned char* dst_tile = tile + (OBS == 2 ? (src >> 5) * bytes : 0);
      const int bh_max = max(__builtin_amdgcn_readlane(bh, s0), __builtin_amdgcn_readlane(bh, s1));
      const int bw_max = max(__builtin_amdgcn_readlane(bw, s0), __builtin_amdgcn_readlane(bw, s1));
      // (one step nearly always: kept from the loop optimiser, which interleaved four column steps and paid two
      // dozen register copies per footprint for it)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
      for (int rr = 0; rr < bh_max; rr += 8) {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int cc = 0; cc < bw_max; cc += 4) {
          const int dr = rr + lr, dc = cc + lc;
          const int r = qr0 + dr, col = qc0 + dc;
          const double py = (row_bias - (double)r) * res - qcy;
          const double px = ((double)col + col_bias) * res - qcx;
          if (drawing && dr < qbh && dc < qbw && fabs(px * qvfx + py * qvfy) <= hl && fabs(px * qvrx + py * qvry) <= hw)
            dst_tile[r * W + col] = 255;
        }
      }
    }
    SMX_WAVE_SYNC();
    for (int t = 0; t < n_live; ++t) {
      const int4* src_tile = reinterpret_cast<const int4*>(tile + (size_t)t * bytes);
      int4* dst = reinterpret_cast<int4*>(a.out.ogm + ((size_t)env * n_veh + (int)observers[turn0 + t]) * (size_t)bytes);
      for (int k = lane; k < bytes / 16; k += 64) dst[k] = src_tile[k];
    }
    SMX_WAVE_SYNC();  // the tiles are reused
  }
  SMX_TSTAMP(span1);
  SMX_TSPAN(6, span0, span1);
}

// =================================================================================
// DAGM role: drivable-area grid map (DrivableAreaGridMapSensor, sensors.py:675-716): one workgroup per
// observing vehicle, the tile in LDS.  The reference renders the road mesh (lane centre lines
// buffered by half the lane width) through the OGM's camera; here a pixel is 255 when its centre
// lies within half a lane width of a segment of a lane centre line (DESIGN.md "Substitutions").
// The segment grid only prunes: a segment that can reach the view lies within the view's
// circumscribed circle grown by the widest half width, so its bounding box meets the visited cells.
// Wavefronts take segments, lanes the pixels of a segment's bounding box; a segment listed in
// several cells is drawn again (same value).
// =================================================================================
__device__ __forceinline__ void dagm_role(const KernelArgs& a, const int block) {
  extern __shared__ __align__(16) unsigned char tile[];
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = (size_t)block;
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  const bool live = (flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST));
  if (!live) return;  // uniform for the whole workgroup
  const int W = c.dagm_width, H = c.dagm_height;
  const int bytes = W * H;
  for (int k = threadIdx.x; k < bytes / 4; k += SMX_BLOCK) reinterpret_cast<int*>(tile)[k] = 0;
  __syncthreads();
  const double res = c.dagm_resolution;
  const double ex0 = SF(SMX_S_X), ey0 = SF(SMX_S_Y), eh = wrap_heading(SF(SMX_S_HEADING));
  const double rx = cos(eh), ry = sin(eh);    // ego right axis
  const double fx = -sin(eh), fy = cos(eh);   // ego forward axis
  const double vw = 0.5 * W * res, vh = 0.5 * H * res;
  const double reach = sqrt(vw * vw + vh * vh) + a.dagm_reach + 1e-6;
  int cx0 = (int)floor((ex0 - reach - m.sg_x0) / m.sg_cell), cx1 = (int)floor((ex0 + reach - m.sg_x0) / m.sg_cell);
  int cy0 = (int)floor((ey0 - reach - m.sg_y0) / m.sg_cell), cy1 = (int)floor((ey0 + reach - m.sg_y0) / m.sg_cell);
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, m.sg_nx - 1);
  cy1 = min(cy1, m.sg_ny - 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = SMX_BLOCK >> 6;
  for (int gy = cy0; gy <= cy1; ++gy) {
    if (cx0 > cx1) break;
    const int row = gy * m.sg_nx;
    // cells of one grid row are contiguous in the member array
    const int m0 = m.sg_off[row + cx0], m1 = m.sg_off[row + cx1 + 1];
    for (int k = m0 + wave; k < m1; k += n_waves) {
      const smx_seg_rec s = m.sg_rec[k];
      const double hw = 0.5 * m.lane_width[s.lane];
      // end points in the ego frame (x to the right, y ahead)
      const double d1x = s.x1 - ex0, d1y = s.y1 - ey0, d2x = s.x2 - ex0, d2y = s.y2 - ey0;
      const double ax = d1x * rx + d1y * ry, ay = d1x * fx + d1y * fy;
      const double bx = d2x * rx + d2y * ry, by = d2x * fx + d2y * fy;
      // pixel centre (r, col): x = (col + 0.5 - W/2) res, y = (H/2 - (r + 0.5)) res
      int c0 = (int)floor((fmin(ax, bx) - hw) / res + 0.5 * W - 0.5) - 1, c1 = (int)ceil((fmax(ax, bx) + hw) / res + 0.5 * W - 0.5) + 1;
      int r0 = (int)floor(0.5 * H - 0.5 - (fmax(ay, by) + hw) / res) - 1, r1 = (int)ceil(0.5 * H - 0.5 - (fmin(ay, by) - hw) / res) + 1;
      c0 = max(c0, 0);
      r0 = max(r0, 0);
      c1 = min(c1, W - 1);
      r1 = min(r1, H - 1);
      if (c0 > c1 || r0 > r1) continue;
      const int bw = c1 - c0 + 1, n_px = bw * (r1 - r0 + 1);
      for (int q = lane; q < n_px; q += 64) {
        const int r = r0 + q / bw, col = c0 + q % bw;
        const double px = (col + 0.5 - 0.5 * W) * res, py = (0.5 * H - (r + 0.5)) * res;
        if (seg_point_dist2(px, py, ax, ay, bx, by) <= hw * hw) tile[r * W + col] = 255;
      }
    }
  }
  __syncthreads();
  int4* dst = reinterpret_cast<int4*>(a.out.dagm + gid * (size_t)bytes);
  for (int k = threadIdx.x; k < bytes / 16; k += SMX_BLOCK) dst[k] = reinterpret_cast<const int4*>(tile)[k];
}

// =================================================================================
// lidar role: lidar sensor (LidarSensor sensors.py:797-827, Lidar lidar.py:58-134): one wavefront per
// observing vehicle, lanes over rays.  Ray i = [origin, origin + base_ray[i]], origin = vehicle
// position + (0, 0, 1); base rays come from the host (they do not rotate with the vehicle,
// lidar.py:109-113).  pybullet rayTestBatch is substituted by exact ray / oriented-box and
// ray / ground-plane intersection (DESIGN.md "Substitutions"); a miss reports (inf, inf, inf).
// =================================================================================
struct LidarPose {
  double x, y, fx, fy;  // centre, forward axis
  int alive;
};

__device__ __forceinline__ void lidar_role(const KernelArgs& a, const int block) {
  __shared__ LidarPose mates[SMX_BLOCK];
  const smx_config& c = a.cfg;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = (size_t)block;
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  const bool live = (flags & SMX_F_ALIVE) && !(flags & SMX_F_SOCIAL) && (!a.first_only || (flags & SMX_F_FIRST));
  if (!live) return;
  const int n_veh = c.num_vehicles;
  const int env = (int)(gid / n_veh);
  const int slot = (int)(gid - (size_t)env * n_veh);
  // env-mates that a ray can reach at all (|ray| = max_distance; a chassis box lies within 2.1 m of
  // its centre), compacted into LDS by ballot / prefix count — the order does not matter for a min
  __shared__ int n_mates;
  {
    const double ex = SF(SMX_S_X), ey = SF(SMX_S_Y);
    const double reach = c.lidar_max_distance + 2.1;
    bool keep = false;
    LidarPose p;
    p.x = p.y = p.fx = p.fy = 0.0;
    p.alive = 1;
    if ((int)threadIdx.x < n_veh && (int)threadIdx.x != slot) {
      const size_t og = (size_t)env * n_veh + threadIdx.x;
      if (a.st.flags[og] & SMX_F_ALIVE) {
        p.x = a.st.f64[(size_t)SMX_S_X * total + og];
        p.y = a.st.f64[(size_t)SMX_S_Y * total + og];
        const double dx = p.x - ex, dy = p.y - ey;
        if (dx * dx + dy * dy <= reach * reach) {
          const double h = wrap_heading(a.st.f64[(size_t)SMX_S_HEADING * total + og]);
          p.fx = -sin(h);
          p.fy = cos(h);
          keep = true;
        }
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) mates[__popcll(mask & ((1ull << threadIdx.x) - 1ull))] = p;
    if (threadIdx.x == 0) n_mates = __popcll(mask);
  }
  __syncthreads();
  const int mates_n = n_mates;
  const double ox = SF(SMX_S_X), oy = SF(SMX_S_Y), oz = SMX_BASE_HEIGHT + 1.0;
  const double hl = 0.5 * SMX_CHASSIS_LENGTH, hw = 0.5 * SMX_CHASSIS_WIDTH, hh = 0.5 * SMX_CHASSIS_HEIGHT;
  const double bz = SMX_BASE_HEIGHT + 0.6;  // chassis box centre height (models/vehicle.urdf)
  for (int i = threadIdx.x; i < c.lidar_rays; i += (int)blockDim.x) {
    const double dx = a.lidar_rays[i * 3 + 0], dy = a.lidar_rays[i * 3 + 1], dz = a.lidar_rays[i * 3 + 2];
    double best = SMX_INF;
    if (dz < 0.0) {
      const double t = -oz / dz;
      if (t >= 0.0 && t <= 1.0) best = t;
    }
    for (int j = 0; j < mates_n; ++j) {
      const LidarPose p = mates[j];
      const double relx = ox - p.x, rely = oy - p.y, relz = oz - bz;
      // slabs along the box axes: forward f, right r = (f.y, -f.x), up
      double tmin = 0.0, tmax = 1.0;
      bool miss = false;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double o, d, half;
        if (ax == 0) {
          o = relx * p.fx + rely * p.fy;
          d = dx * p.fx + dy * p.fy;
          half = hl;
        } else if (ax == 1) {
          o = relx * p.fy + rely * (-p.fx);
          d = dx * p.fy + dy * (-p.fx);
          half = hw;
        } else {
          o = relz;
          d = dz;
          half = hh;
        }
        if (d == 0.0) {
          if (fabs(o) > half) miss = true;
        } else {
          double t1 = (-half - o) / d, t2 = (half - o) / d;
          if (t1 > t2) {
            double tt = t1;
            t1 = t2;
            t2 = tt;
          }
          tmin = fmax(tmin, t1);
          tmax = fmin(tmax, t2);
          if (tmin > tmax) miss = true;
        }
      }
      if (!miss && tmin < best) best = tmin;
    }
    const size_t q = gid * (size_t)c.lidar_rays + i;
    if (best <= 1.0) {
      a.out.lidar_hit[q] = 1;
      a.out.lidar_point[q * 3 + 0] = ox + best * dx;
      a.out.lidar_point[q * 3 + 1] = oy + best * dy;
      a.out.lidar_point[q * 3 + 2] = oz + best * dz;
    } else {
      a.out.lidar_hit[q] = 0;
      const double inf = __builtin_huge_val();
      a.out.lidar_point[q * 3 + 0] = inf;
      a.out.lidar_point[q * 3 + 1] = inf;
      a.out.lidar_point[q * 3 + 2] = inf;
    }
  }
}

// =================================================================================
// k_sensors: the observation of a pass as ONE launch whose workgroups take different roles —
// waypoint paths + trip meter (4 lanes / vehicle), the rest of Sensors.observe (1 lane / vehicle,
// whole envs per workgroup) and, if enabled, lidar and OGM (1 wavefront / vehicle each).  The roles read the
// same pose / flags / facts and write disjoint outputs, so they overlap in time; the flags word
// itself only changes in k_commit.
// =================================================================================
__global__ void __launch_bounds__(SMX_BLOCK) k_sensors(const KernelArgs a) {
  const int b = (int)blockIdx.x;
  if (b < a.wp_blocks) {
    waypoints_role(a, b);
  } else if (b < a.wp_blocks + a.obs_blocks) {
    observe_role(a, b - a.wp_blocks);
  } else if (b < a.wp_blocks + a.obs_blocks + a.lidar_blocks) {
    lidar_role(a, b - a.wp_blocks - a.obs_blocks);
  } else {
    ogm_role(a, b - a.wp_blocks - a.obs_blocks - a.lidar_blocks);
  }
}

// =================================================================================
// k_first: the reset pass (first observations of re-created vehicles) as ONE launch.  A workgroup
// owns the envs of one observe-role group and runs scan -> sensors -> commit for their new vehicles
// itself, phase after phase; results pass between phases through global memory behind a fence and a
// barrier.  Almost always no env of the group has restarted and the workgroup leaves at once, so an
// auto-reset tick pays for one empty launch instead of three.  (OGM tiles need dynamic LDS and keep
// their own launch.)
// =================================================================================
// (eight wavefronts: the (vehicle, scan half) pairs of a restarted 64-vehicle env are 128 teams of eight lanes — four
// rounds of ~55 us each in a workgroup of 256, the largest piece of C5's reset pass; the waypoint teams, four lanes a
// vehicle, fit the first 256 threads, whose knot scratch is all the LDS the workgroup may have)
#define SMX_FIRST_BLOCK 512
#define SMX_FIRST_WP_THREADS 256
__global__ void __launch_bounds__(SMX_FIRST_BLOCK) k_first(const KernelArgs a) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_FIRST_WP_THREADS];
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const int n_veh = c.num_vehicles;
  const int epb = SMX_BLOCK / n_veh;  // the env groups are those of the observe / commit roles
  const int block = (int)blockIdx.x;
  const size_t total = (size_t)c.num_envs * n_veh;
  const size_t g0 = (size_t)block * epb * n_veh;
  const size_t g1 = min(total, g0 + (size_t)epb * n_veh);
  int mine_first = 0;
  if (g0 + threadIdx.x < g1) {
    const int f = a.st.flags[g0 + threadIdx.x];
    mine_first = (f & SMX_F_ALIVE) && (f & SMX_F_FIRST);
  }
  const int n_new = __syncthreads_count(mine_first);
  if (n_new == 0) return;
  SMX_TSTAMP(tk0);
  // More new vehicles than one round of (vehicle, scan half) teams holds (a 64-vehicle env of C5: 128 pairs for 64
  // teams): the seeds halves take the round — the waypoint rows wait for nothing else —, and the facts halves
  // then run BESIDE the rows' serial emitter (minicity: 220 us of a restarted env's 410) on the
  // workgroup's other four wavefronts; observe needs both.  Fewer: one round serves both halves as before.
  const bool split = 2 * n_new > SMX_FIRST_BLOCK / SMX_TEAM;
  // The workgroup is four wavefronts wide: a restarted env's chain scan -> sensors -> commit is
  // pure latency, so its independent pieces run side by side — (vehicle, scan half) pairs over the
  // teams of all four wavefronts, then the vehicles' waypoint teams — instead of one after the
  // other in a single wavefront (an env of 16 vehicles: 4 scan rounds of ~40 us became 1).
  // ---- scan: one team per (vehicle, half)
  // (one scan call site inside a two-turn loop: inlined three times the kernel spilled 2 KB per lane)
#pragma nounroll
  for (int turn = 0; turn < 2; ++turn) {
    SMX_TSTAMP(tt0);
    // turn 0: every team scans — both halves of a vehicle (pair = 2 x vehicle + half), or the seeds halves only;
    // turn 1: the first 256 threads emit the rows, and the other 32 teams take the facts halves left over
    const bool scans = turn == 0 || (split && threadIdx.x >= SMX_FIRST_WP_THREADS);
    if (scans) {
      const int teams = turn == 0 ? SMX_FIRST_BLOCK / SMX_TEAM : (SMX_FIRST_BLOCK - SMX_FIRST_WP_THREADS) / SMX_TEAM;
      const int team = ((int)threadIdx.x - (turn == 0 ? 0 : SMX_FIRST_WP_THREADS)) / SMX_TEAM;
      const bool both = turn == 0 && !split;
      const size_t units = both ? (g1 - g0) * 2 : (g1 - g0);
      for (size_t u0 = 0; u0 < units; u0 += teams) {
        const size_t u = u0 + team;
        if (u < units) {
          const size_t gid = g0 + (both ? (u >> 1) : u);
          const int half = both ? (int)(u & 1) : (turn == 0 ? 1 : 0);
          const int flags = a.st.flags[gid];
          if ((flags & SMX_F_ALIVE) && (flags & SMX_F_FIRST)) scan_role<SMX_TEAM, true>(a, m, c, gid, total, team_rank<SMX_TEAM>(), flags, half);
        }
      }
    } else if (turn == 1 && threadIdx.x < SMX_FIRST_WP_THREADS) {  // (whole wavefronts; waypoints_for holds no barrier)
      for (size_t base = g0; base < g1; base += SMX_FIRST_WP_THREADS / SMX_WP_LANES) {
        const size_t gid = base + threadIdx.x / SMX_WP_LANES;
        if (gid < g1) waypoints_for<SMX_FIRST_WP_THREADS>(a, gid, knot_scratch + threadIdx.x);
      }
    }
    __threadfence();
    __syncthreads();
    SMX_TSTAMP(tt1);
    SMX_TACC_ALL(turn == 0 ? 57 : 58, tt0, tt1);
  }
  SMX_TSTAMP(tk2);
  observe_role(a, block);
  SMX_TSTAMP(tk3);
  SMX_TACC_ALL(59, tk2, tk3);
  if ((c.sensors & SMX_SENSOR_LIDAR) && a.lidar_blocks != 0)  // (0: the reset pass launched k_lidar for the new vehicles)
    for (size_t gid = g0; gid < g1; ++gid) {
      lidar_role(a, (int)gid);
      __syncthreads();  // the role's LDS block is reused by the next vehicle
    }
  __threadfence();
  __syncthreads();
  // ---- commit
  SMX_TSTAMP(tk4);
  commit_role(a, block);
  SMX_TSTAMP(tk5);
  SMX_TACC_ALL(60, tk4, tk5);
  SMX_TACC_ALL(61, tk0, tk5);
  // ---- large batches: the new vehicles' knot lists, so that the next tick's k_control_fast serves them too (a new
  // vehicle without lists went through the slow controller, two launches of pure latency in front of everything else
  // of the tick).  New here: alive, SMX_F_FIRST just cleared by the commit, one step old.
  if (a.walk_new) {
    __threadfence();
    __syncthreads();
    for (size_t base = g0; base < g1; base += SMX_FIRST_BLOCK / SMX_WP_LANES) {
      const size_t gid = base + threadIdx.x / SMX_WP_LANES;
      if (gid >= g1) continue;
      const int f = a.st.flags[gid];
      if ((f & SMX_F_ALIVE) && !(f & SMX_F_SOCIAL) && !(f & SMX_F_FIRST) && a.st.steps[gid] == 1) {
        wp_walk_for(a, gid, (int)threadIdx.x % SMX_WP_LANES, false, true);  // (the reset pass's flag word says "new vehicles only")
      }
    }
  }
}

// single-role launches: large batches (each role then keeps its own register / LDS footprint and
// occupancy; forcing more wavefronts per SIMD onto k_waypoints / k_observe by waves_per_eu cost more
// in spills than it won: +20 % on loop 4096 x 32) and OGM tiles too large to ride along as dynamic LDS of every k_sensors workgroup
__global__ void __launch_bounds__(SMX_BLOCK) k_ogm(const KernelArgs a) { ogm_role(a, (int)blockIdx.x); }
__global__ void __launch_bounds__(SMX_BLOCK) k_dagm(const KernelArgs a) { dagm_role(a, (int)blockIdx.x); }
// =================================================================================
// k_road_waypoints: RoadWaypointsSensor (sensors.py:991-1040).  SMX_RW_LANE_CAP lanes of a wavefront share a
// vehicle: every lane of the team builds the sensor's lane list for itself (the same serial steps, so the
// team does not diverge), then team lane l follows road lane l: start `horizon` metres behind the vehicle
// along the lane (through its incoming lanes, depth first in their order, where the lane is shorter), and from
// each start every lanepoint path of lookahead 2 x horizon, interpolated like the waypoints sensor's
// (equally_spaced_path, its knots in private memory: up to 2 x horizon + 2 of them).  An optional sensor off
// the headline configurations: written for parity, not for throughput.
// Where the reference cannot answer — the nearest lane is junction-internal: Road.parallel_roads asks sumolib
// for the internal edge's from-node, which is None, and raises — the road has no parallel roads here.
// =================================================================================
#define SMX_RW_MAX_KNOTS (2 * SMX_RW_HORIZON_MAX + 4)
#define SMX_RW_STACK 24
struct RwLanes {
  int n;  // lanes the sensor reports (the list keeps the first SMX_RW_LANE_CAP)
  int lane[SMX_RW_LANE_CAP];
  // lane_paths[lane.lane_id] = ...: a lane met again keeps its first place in the dict
  __device__ __forceinline__ void add_road(const MapDev& m, int road) {
    for (int k = m.road_lane_off[road]; k < m.road_lane_off[road + 1]; ++k) {
      const int ln = m.road_lanes[k];
      bool seen = false;
      for (int q = 0; q < min(n, SMX_RW_LANE_CAP); ++q) seen = seen || lane[q] == ln;
      if (seen) continue;
      if (n < SMX_RW_LANE_CAP) lane[n] = ln;
      ++n;
    }
  }
};

__global__ void __launch_bounds__(SMX_BLOCK) k_road_waypoints(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const MapDev& m = a.map;
  const smx_outputs& o = a.out;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t tid = (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x;
  const size_t gid = tid / SMX_RW_LANE_CAP;
  const int l = (int)(tid % SMX_RW_LANE_CAP);
  if (gid >= total) return;
  const int flags = a.st.flags[gid];
  if (!(flags & SMX_F_ALIVE) || (flags & SMX_F_SOCIAL) || (a.first_only && !(flags & SMX_F_FIRST))) return;
  const int L = c.rw_lanes, P = c.rw_paths, H = c.rw_horizon, R = 2 * H + 1;
  const double px = SF(SMX_S_X), py = SF(SMX_S_Y);
  // ---- the sensor's lanes: nearest lane's road, its parallel roads, the roads oncoming at the point
  RwLanes lanes;
  lanes.n = 0;
  // road_map.nearest_lane(point) (road_map.py:91-96, the default radius); asked here rather than taken from
  // k_scan so that the kernel also serves the reset pass, whose scan runs inside k_first
  // In a tick the scan's facts half has just answered it (the nearest lane and its distance: the observe role reads
  // its ego lane the same way); only the reset pass, whose scan runs inside k_first after this kernel, asks here —
  // every lane of the team repeating a one-lane ring search was the longest piece of the kernel.
  int near_lane = -1;
  if (!a.first_only) {
    const int ln = a.st.facts_i32[(size_t)SMX_FI_LANE * total + gid];
    const double dd = a.st.facts_f64[(size_t)SMX_FF_LANE_DIST * total + gid];
    near_lane = (ln >= 0 && dd < fmax(10.0, 2.0 * m.default_lane_width)) ? ln : -1;
  } else {
    near_lane = road_facts_scan(m, px, py, fmax(10.0, 2.0 * m.default_lane_width), 0, nullptr, nullptr).lane;
  }
  if (near_lane >= 0) {
    const int road = m.lane_road[near_lane];
    lanes.add_road(m, road);
    for (int k = m.road_par_off[road]; k < m.road_par_off[road + 1]; ++k) lanes.add_road(m, m.road_par_idx[k]);
    // Road.oncoming_roads_at_point (sumo_road_network.py:596-605)
    for (int k = m.road_lane_off[road]; k < m.road_lane_off[road + 1]; ++k) {
      const int ln = m.road_lanes[k];
      const double off = lane_offset_along(m, ln, px, py);
      oncoming_lanes_at_offset(m, ln, off, [&](int other) {
        if (m.lane_road[other] != road) lanes.add_road(m, m.lane_road[other]);
      });
    }
  }
  if (l == 0) o.rw_lane_count[gid] = (uint8_t)min(lanes.n, 255);
  if (l >= L) return;
  const size_t lane_row = gid * (size_t)L + l;
  const bool mine = l < min(lanes.n, SMX_RW_LANE_CAP);
  int my_lane = -1;
#pragma unroll
  for (int q = 0; q < SMX_RW_LANE_CAP; ++q)
    if (q == l && mine) my_lane = lanes.lane[q];
  o.rw_lane[lane_row] = (int16_t)my_lane;
  int n_paths = 0;
  if (my_lane >= 0) {
    RouteFilter f;  // route = plan.route: a fixed route filters, the endless mission's empty route does not
    f.fixed_route(a.missions, (int)(gid % (size_t)c.num_vehicles), m.n_roads);
    // ---- paths_for_lane (sensors.py:1014-1040): depth first through the incoming lanes
    int st_lane[SMX_RW_STACK];
    double st_start[SMX_RW_STACK];
    int sp = 0;
    st_lane[sp] = my_lane;
    st_start[sp] = lane_offset_along(m, my_lane, px, py) - (double)H;
    ++sp;
    int knots[SMX_RW_MAX_KNOTS];
    while (sp > 0) {
      --sp;
      const int ln = st_lane[sp];
      double start_offset = st_start[sp];
      const int ia = m.lane_in_off[ln], ib = m.lane_in_off[ln + 1];
      if (start_offset < 0.0 && ib > ia) {
        // children in reverse so that the first incoming lane is taken up first (a full stack drops the rest:
        // rows of at most rw_paths paths are kept anyway, the count then reads low)
        for (int k = ib - 1; k >= ia; --k) {
          if (sp >= SMX_RW_STACK) break;
          const int child = m.lane_in_idx[k];
          st_lane[sp] = child;
          st_start[sp] = m.lane_length[child] + start_offset;
          ++sp;
        }
        continue;
      }
      start_offset = fmax(0.0, start_offset);
      double wx, wy;
      lane_point_at_offset(m, ln, start_offset, wx, wy);
      int key[4] = {ln, -9, -9, -9};
      int idx4[4];
      closest_filtered4(m, wx, wy, key, 1, false, idx4, nullptr);
      const int start = idx4[0];
      if (start < 0) continue;
      BranchState bs;
      bs.reset();
      do {
        const bool kept = n_paths < P;
        const size_t row = (lane_row * (size_t)P + (size_t)(kept ? n_paths : 0)) * (size_t)R;
        const int n = equally_spaced_path<SMX_RW_MAX_KNOTS>(m, f, bs, start, 2 * H, wx, wy, knots, 1, kept ? R : 0,
                                                            [&](int i, const WaypointOut& w) {
                                                              double* d = o.rw_pos + (row + i) * 3;
                                                              d[0] = w.x;
                                                              d[1] = w.y;
                                                              d[2] = 0.0;
                                                              o.rw_heading[row + i] = (float)w.heading;
                                                              o.rw_lane_width[row + i] = (float)w.width;
                                                              o.rw_speed_limit[row + i] = (float)w.speed;
                                                              o.rw_lane_index[row + i] = (int8_t)m.lane_index[w.lane];
                                                              o.rw_lane_id[row + i] = (int16_t)w.lane;
                                                            });
        if (kept) o.rw_count[lane_row * (size_t)P + n_paths] = (uint8_t)min(n, R);
        if (n_paths < 32767) ++n_paths;
      } while (bs.advance());
    }
  }
  o.rw_path_count[lane_row] = (int16_t)n_paths;
  for (int p = min(n_paths, P); p < P; ++p) o.rw_count[lane_row * (size_t)P + p] = 0;
}

// The reset pass of the grid sensors: almost no vehicle is new in a given tick, and one workgroup
// per vehicle that only finds that out costs ~120 us at 131 k vehicles.  Here a workgroup looks at
// the flags of 64 vehicles with one load and a ballot, and builds tiles only for the new ones.
template <bool DAGM>
__global__ void __launch_bounds__(SMX_BLOCK) k_grid_first(const KernelArgs a) {
  const size_t total = (size_t)a.cfg.num_envs * a.cfg.num_vehicles;
  const size_t g0 = (size_t)blockIdx.x * SMX_BLOCK;
  const size_t gid = g0 + threadIdx.x;
  const int f = gid < total ? a.st.flags[gid] : 0;
  unsigned long long fresh = __ballot((f & SMX_F_ALIVE) && (f & SMX_F_FIRST) && !(f & SMX_F_SOCIAL));
  while (fresh != 0ull) {  // uniform
    const int j = __ffsll((long long)fresh) - 1;
    fresh &= fresh - 1ull;
    if (DAGM)
      dagm_role(a, (int)(g0 + j));
    else
      ogm_role(a, (int)(g0 + j));
    __syncthreads();  // the tile is reused
  }
}
__global__ void __launch_bounds__(SMX_BLOCK) k_waypoints(const KernelArgs a) { waypoints_role(a, (int)blockIdx.x); }
__global__ void __launch_bounds__(SMX_BLOCK) k_waypoints_tables(const KernelArgs a) { waypoints_tables_role(a, (int)blockIdx.x); }
__global__ void __launch_bounds__(SMX_BLOCK) k_waypoints_emit(const KernelArgs a) {
  SMX_TSTAMP(span0);
  waypoints_emit_role(a, (int)blockIdx.x);
  SMX_TSTAMP(span1);
  SMX_TSPAN(4, span0, span1);
}
// the vehicles k_waypoints_emit left on the slow list: waypoints_for (a team of four lanes per vehicle), a fixed grid
// striding the list, whose length is only known on the device
__global__ void __launch_bounds__(SMX_BLOCK) k_waypoints_listed(const KernelArgs a) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  const int count = *a.slow_count;
  constexpr int VPB = SMX_BLOCK / SMX_WP_LANES;
  for (int i = (int)blockIdx.x * VPB + (int)threadIdx.x / SMX_WP_LANES; i < count; i += (int)gridDim.x * VPB)
    waypoints_for<SMX_BLOCK>(a, (size_t)a.slow_list[i], knot_scratch + threadIdx.x);
}
// the slow chain's form (short lists: its latency ends the chain): eight lanes a vehicle — four emit the rows
// (waypoints_for), four walk the knot lists for the next tick's controller (k_wp_walk_listed's work) beside them
__global__ void __launch_bounds__(SMX_BLOCK) k_waypoints_walk_listed(const KernelArgs a) {
  __shared__ int knot_scratch[SMX_MAX_KNOTS * SMX_BLOCK];
  const int count = *a.slow_count;
  constexpr int VPB = SMX_BLOCK / (2 * SMX_WP_LANES);
  const bool walker = ((threadIdx.x / SMX_WP_LANES) & 1) != 0;  // (uniform in an aligned group of four lanes)
  for (int i = (int)blockIdx.x * VPB + (int)threadIdx.x / (2 * SMX_WP_LANES); i < count; i += (int)gridDim.x * VPB) {
    const size_t gid = (size_t)a.slow_list[i];
    if (walker)
      wp_walk_for(a, gid, (int)threadIdx.x % SMX_WP_LANES, false);
    else
      waypoints_for<SMX_BLOCK>(a, gid, knot_scratch + threadIdx.x);
  }
}
// (capped at 168 registers for a third wavefront per SIMD beside the waypoint kernels it spills 52 of them: 0.796 -> 0.806 ms)
__global__ void __launch_bounds__(SMX_BLOCK) k_observe(const KernelArgs a) {
  SMX_TSTAMP(span0);
  observe_role(a, (int)blockIdx.x);
  SMX_TSTAMP(span1);
  SMX_TSPAN(5, span0, span1);
}
__global__ void __launch_bounds__(SMX_BLOCK) k_lidar(const KernelArgs a) { lidar_role(a, (int)blockIdx.x); }
// The lidar of the reset pass on large batches: almost no vehicle is new in a given tick, and when an env restarts
// all its vehicles are — neighbours in memory.  Workgroup w looks at the vehicles v = w (mod gridDim.x), 64 flags per
// load and ballot, and runs the lidar role for the new ones it finds: an env's 64 new vehicles land in 64 different
// workgroups instead of one after the other in k_first's, and a tick without restarts pays a few microseconds.
#define SMX_LIDAR_FIRST_BLOCKS 1024
__global__ void __launch_bounds__(SMX_BLOCK) k_lidar_first(const KernelArgs a) {
  const size_t total = (size_t)a.cfg.num_envs * a.cfg.num_vehicles;
  const size_t stride = (size_t)gridDim.x;
  for (size_t base = blockIdx.x; base < total; base += stride * SMX_BLOCK) {
    const size_t v = base + (size_t)threadIdx.x * stride;
    const int f = v < total ? a.st.flags[v] : 0;
    unsigned long long fresh = __ballot((f & SMX_F_ALIVE) && (f & SMX_F_FIRST) && !(f & SMX_F_SOCIAL));
    while (fresh != 0ull) {  // uniform in the (one-wavefront) workgroup
      const int l = __ffsll((long long)fresh) - 1;
      fresh &= fresh - 1ull;
      lidar_role(a, (int)(base + (size_t)l * stride));
      __syncthreads();  // the role's LDS block is reused
    }
  }
}

// =================================================================================
// k_reset: SMARTS.reset (smarts.py:365-460) for the selected envs — vehicles re-created at their
// spawn poses (AckermannChassis._initialize_speed, chassis.py:668-671); the observation kernels
// that follow produce their first observations.
// =================================================================================
__global__ void __launch_bounds__(SMX_BLOCK) k_reset(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  const size_t gid = (size_t)blockIdx.x * SMX_BLOCK + threadIdx.x;
  if (gid >= total) return;
  const int env = (int)(gid / c.num_vehicles);
  const int slot = (int)(gid - (size_t)env * c.num_vehicles);
  bool sel;
  if (a.reset_all)
    sel = true;
  else if (a.env_mask)
    sel = a.env_mask[env] != 0;
  else
    sel = a.st.env_reset_pending[env] != 0;
  if (!sel) return;
  respawn_vehicle(a, gid, total, a.st.env_episode[env] + 1);  // every reset starts the next spawn row
  // per-env words are written by every thread of the env with the same values (no ordering needed
  // inside this kernel; the env's own threads never read them here)
  (void)slot;
}

// per-env bookkeeping of a reset, after k_reset (separate launch: k_reset's threads read env_episode)
__global__ void __launch_bounds__(SMX_BLOCK) k_reset_env(const KernelArgs a) {
  const smx_config& c = a.cfg;
  const int env = blockIdx.x * SMX_BLOCK + threadIdx.x;
  if (env >= c.num_envs) return;
  bool sel;
  if (a.reset_all)
    sel = true;
  else if (a.env_mask)
    sel = a.env_mask[env] != 0;
  else
    sel = a.st.env_reset_pending[env] != 0;
  if (!sel) return;
  a.st.env_episode[env] = a.st.env_episode[env] + 1;
  a.st.env_done_count[env] = 0;
  a.st.env_ticks[env] = c.reset_elapsed_steps;
  a.st.env_reset_pending[env] = 0;
  if (!a.keep_reward_done) a.out.env_done[env] = 0;
}

// =================================================================================
// C-ABI (include/smx.h)
// =================================================================================
// sqrt(d2) <= radius without the square root: the correctly rounded root does not decrease with its argument, so
// the test holds exactly for the squared distances up to a threshold — the largest double whose rounded root is
// still <= radius (found from radius * radius by stepping a few units in the last place).  k_observe takes the
// 32 x 32 distances of an env per tick; the root and its comparison were twenty instructions each.
static double radius_threshold(double radius) {
  if (!(radius >= 0.0)) return -1.0;           // (unlimited: the kernels do not look at it)
  if (std::isinf(radius)) return radius;
  double t = radius * radius;
  if (std::isinf(t)) return t;
  while (t > 0.0 && std::sqrt(t) > radius) t = std::nextafter(t, 0.0);
  for (;;) {
    const double up = std::nextafter(t, INFINITY);
    if (std::isinf(up) || !(std::sqrt(up) <= radius)) break;
    t = up;
  }
  return t;
}

struct smx_handle_s {
  smx_config cfg;
  int device;
  bool map_loaded;
  MapDev map;
  void* map_blob;  // one device allocation holding every table
  size_t map_bytes;
  void* knots_blob;  // KnotLists of the waypoints sensor (k_wp_walk -> k_waypoints_tables)
  void* spill_blob;  // k_waypoints_emit's overflow area (KernelArgs::wp_spill)
  int wp_pool_limit; // records of k_waypoints_emit's LDS pool in use (smx_debug_set_wp_pool)
  int32_t* alive_blob;  // [total] alive list + two counters (ticks alternate), large batches
  uint8_t* pending_blob;  // [total] seed_pending
  int32_t* slow_blob;   // [4][total] slow lists of the fast kernels (scan facts, scan seeds, control, waypoint rows) + [2][4] counters (ticks alternate)
  double* scan_carry;   // [6][total]: seeds_carry (x, y, d10^2, d1^2) | facts_carry (x, y) of the seeded scan
  int alive_parity;
  KnotLists knots;
  void* ctrl_blob;   // CtrlHandoff of the two-launch controller (k_control_paths -> k_control_law)
  CtrlHandoff ctrl;
  int32_t* status_dev;  // SMX_DEVICE_* bits raised by the kernels
  // Large batches: the sensor kernels of a tick are independent of each other (they read the pose and write
  // disjoint rows) and are bound by different things — waypoint chain walks by load latency, OGM tiles by
  // their own write stream — so they are enqueued on side streams between two events and overlap.
  hipStream_t side[3];
  hipEvent_t ev_fork, ev_fork_grid, ev_join[3];
  bool side_ready;
  const double* lidar_rays;
  smx_via* vias_dev;
  int32_t* via_off_dev;
  int32_t n_vias;
  void* missions_blob;      // device copy of smx_set_missions: goals | last roads | route positions | lane table
  std::vector<int32_t> host_lane_road, host_lane_out_off, host_lane_out_idx;  // kept for smx_set_missions
  MissionsDev missions;
  double heading_gain_pos, lateral_gain_pos;
  double nb_d2_max;
  int slow_blocks;  // grid of the slow lists' kernels (smx_load_map)
  bool map_junctions;  // lanes of the map split (some lanepoint has several successors)
  double dagm_reach;  // half the widest lane width of the loaded map
  int debug_skip;
  int launch_strategy;  // SMX_LAUNCH_*
  bool timing;
  std::vector<hipEvent_t> ev_pool;  // pairs: [2*i] start, [2*i+1] stop
  size_t ev_used;                   // pairs recorded since the last read
  bool phase_timing;
  std::vector<hipEvent_t> ph_pool;  // SMX_PHASE_COUNT + 1 boundary events per step
  size_t ph_used;
  std::string err;
};

static thread_local std::string g_create_err;  // the reason of this thread's last failed smx_create

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

#ifdef SMX_DEBUG_TIMING
extern "C" int smx_span_read(unsigned int* out) {  // developer: [kernel][wavefront] spans of the last launches, 10 ns units
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(smx_span), sizeof(unsigned int) * SMX_SPAN_KERNELS * SMX_SPAN_WAVES) != hipSuccess) return -2;
  return 0;
}

extern "C" int smx_prof_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(smx_prof), 128 * sizeof(unsigned long long)) != hipSuccess) return -2;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smx_prof), z, sizeof(z)) != hipSuccess) return -2;
  }
  return 0;
}
#endif
#ifdef SMX_DEBUG_BOUNDS
extern "C" int smx_debug_read(int* site, long long* value) {
  if (hipMemcpyFromSymbol(site, HIP_SYMBOL(smx_dbg_site), sizeof(int)) != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(value, HIP_SYMBOL(smx_dbg_value), sizeof(long long)) != hipSuccess) return -2;
  int aux[8];
  if (hipMemcpyFromSymbol(aux, HIP_SYMBOL(smx_dbg_aux), sizeof(aux)) != hipSuccess) return -2;
  double f[64];
  if (hipMemcpyFromSymbol(f, HIP_SYMBOL(smx_dbg_f), sizeof(f)) != hipSuccess) return -2;
  printf("dbg ctrl: wp_n=%g la_num=%g lax=%.6f lay=%.6f lah=%.6f n_paths=%g want=%g curv=%g\n", f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
  for (int k = 0; k < 17; ++k) printf("  wp%d %.6f %.6f %.6f\n", k, f[8 + 3 * k], f[9 + 3 * k], f[10 + 3 * k]);
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

static int create_impl(const smx_config* cfg, int device, smx_handle* out) {
  smx_handle h = new (std::nothrow) smx_handle_s();
  if (!h) return SMX_ERR_NOMEM;
  h->cfg = *cfg;
  h->device = device;
  h->map_loaded = false;
  h->map_blob = nullptr;
  h->map_bytes = 0;
  h->knots_blob = nullptr;
  h->spill_blob = nullptr;
  h->wp_pool_limit = SMX_WPE_POOL;
  h->alive_blob = nullptr;
  h->scan_carry = nullptr;
  h->slow_blob = nullptr;
  h->pending_blob = nullptr;
  h->alive_parity = 0;
  h->knots = KnotLists{};
  h->ctrl_blob = nullptr;
  h->ctrl = CtrlHandoff{nullptr, nullptr};
  h->status_dev = nullptr;
  h->side_ready = false;
  h->lidar_rays = nullptr;
  // lane_following_controller.py:426-430: place_poles gains clipped to [0.02, 0.04] / [3.4, 4.1];
  // for the sedan they saturate at (0.04, 3.4) for both Lane-space target speeds.
  h->heading_gain_pos = 0.04;
  h->lateral_gain_pos = 3.4;
  h->nb_d2_max = radius_threshold(h->cfg.nb_radius);
  h->slow_blocks = SMX_SLOW_BLOCKS;
  h->map_junctions = false;
  h->timing = false;
  h->ev_used = 0;
  h->phase_timing = false;
  h->ph_used = 0;
  h->vias_dev = nullptr;
  h->via_off_dev = nullptr;
  h->n_vias = 0;
  h->missions_blob = nullptr;
  h->missions = MissionsDev{nullptr, nullptr};
  h->map.route_pos = nullptr;
  h->map.route_lane_ok = nullptr;
  h->launch_strategy = SMX_LAUNCH_AUTO;
  h->debug_skip = 0;
#ifdef SMX_DEBUG_TIMING
  if (const char* dbg = getenv("SMX_DEBUG_SKIP")) h->debug_skip = atoi(dbg);
#endif
  *out = h;
  const smx_config& c = h->cfg;
  if (c.num_envs <= 0 || c.num_vehicles <= 0 || c.num_vehicles > SMX_BLOCK)
    return fail(h, SMX_ERR_INVALID, "num_envs must be > 0 and 0 < num_vehicles <= 64");
  if (!(c.dt > 0.0)) return fail(h, SMX_ERR_INVALID, "dt must be > 0");
  if ((c.sensors & SMX_SENSOR_ROAD_WAYPOINTS) &&
      (c.rw_horizon < 1 || c.rw_horizon > SMX_RW_HORIZON_MAX || c.rw_lanes < 1 || c.rw_lanes > SMX_RW_LANE_CAP || c.rw_paths < 1 ||
       c.rw_paths > 64))
    return fail(h, SMX_ERR_INVALID, "road waypoints: need 1 <= rw_horizon <= 64, 1 <= rw_lanes <= 8, 1 <= rw_paths <= 64");
  if ((c.sensors & SMX_SENSOR_WAYPOINTS) &&
      (c.wp_lookahead < 1 || c.wp_lookahead > SMX_MAX_KNOTS - 2 || c.wp_paths < 1 || c.wp_paths > 64 || c.wp_len < 1 || c.wp_len > c.wp_lookahead + 1))
    return fail(h, SMX_ERR_INVALID, "waypoints: need lookahead >= 1, 1 <= wp_paths <= 64, 1 <= wp_len <= lookahead + 1");
  if (c.via_max < 0 || c.via_max > 32) return fail(h, SMX_ERR_INVALID, "via_max must be in 0..32");
  if (c.alive_lists < 0 || c.alive_lists > SMX_MAX_ALIVE_LISTS || c.alive_min_ego < 0 || c.alive_min_total < 0)
    return fail(h, SMX_ERR_INVALID, "agents_alive: at most 4 lists, non-negative minima");
  if (c.num_social < 0 || c.num_social >= c.num_vehicles)
    return fail(h, SMX_ERR_INVALID, "num_social must leave at least one agent slot");
  if (c.num_social > 0 && !(c.social_speed_factor >= 0.0))
    return fail(h, SMX_ERR_INVALID, "social_speed_factor must be >= 0");
  if (c.social_model != SMX_SOCIAL_CONSTANT && c.social_model != SMX_SOCIAL_IDM)
    return fail(h, SMX_ERR_INVALID, "unknown social_model");
  if (c.action_space < SMX_ACTION_SPACE_LANE || c.action_space > SMX_ACTION_SPACE_TRAJECTORY)
    return fail(h, SMX_ERR_INVALID, "unknown action_space");
  if ((c.sensors & SMX_SENSOR_OGM) &&
      (c.ogm_width < 1 || c.ogm_height < 1 || (c.ogm_width * c.ogm_height) % 16 != 0 ||
       c.ogm_width * c.ogm_height > 64 * 1024 || !(c.ogm_resolution > 0.0)))
    return fail(h, SMX_ERR_INVALID, "ogm: need width*height a multiple of 16 and at most 65536, resolution > 0");
  if ((c.sensors & SMX_SENSOR_DAGM) &&
      (c.dagm_width < 1 || c.dagm_height < 1 || (c.dagm_width * c.dagm_height) % 16 != 0 ||
       c.dagm_width * c.dagm_height > 64 * 1024 || !(c.dagm_resolution > 0.0)))
    return fail(h, SMX_ERR_INVALID, "dagm: need width*height a multiple of 16 and at most 65536, resolution > 0");
  if ((c.sensors & SMX_SENSOR_LIDAR) && (c.lidar_rays < 1 || c.lidar_rays > 65536))
    return fail(h, SMX_ERR_INVALID, "lidar: need 1 <= lidar_rays <= 65536");
  if ((c.sensors & SMX_SENSOR_NEIGHBORS) && (c.nb_max < 1 || c.nb_max > 127))
    return fail(h, SMX_ERR_INVALID, "neighbours: need 1 <= nb_max <= 127");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return fail(h, SMX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return SMX_OK;
}

extern "C" int smx_create(const smx_config* cfg, int device, smx_handle* out) {
  if (out) *out = nullptr;
  if (!cfg || !out) {
    g_create_err = "smx_create: null config or handle pointer";
    return SMX_ERR_INVALID;
  }
  smx_handle h = nullptr;
  const int rc = create_impl(cfg, device, &h);
  if (rc != SMX_OK) {  // no half-made handle for the caller to remember to destroy
    g_create_err = h ? h->err : "out of memory";
    delete h;
    return rc;
  }
  *out = h;
  return SMX_OK;
}

extern "C" int smx_set_launch_strategy(smx_handle h, int strategy) {
  if (!h) return SMX_ERR_INVALID;
  if (strategy < SMX_LAUNCH_AUTO || strategy > SMX_LAUNCH_LARGE_TEAMS) return fail(h, SMX_ERR_INVALID, "unknown launch strategy");
  h->launch_strategy = strategy;
  return SMX_OK;
}

// Which cut of the LARGE form a batch takes (smx.h, smx_launch_form): one lane per vehicle + slow lists where the lists
// stay short (a map whose lanes never split), teams of lanes for everybody elsewhere; the strategies LARGE_ONE_LANE /
// LARGE_TEAMS force a cut.
static bool one_lane_cut(const smx_handle_s* h) {
  if (h->launch_strategy == SMX_LAUNCH_LARGE_ONE_LANE) return true;
  if (h->launch_strategy == SMX_LAUNCH_LARGE_TEAMS) return false;
  return !h->map_junctions || SMX_ONE_LANE_ON_SPLIT_MAPS;
}
// ... and, inside the one-lane cut, whether the seeds half is the one-lane kernel + the slow seeds chain, or the team
// kernel for everybody: the chain — from-scratch searches and the serial emitter for the few vehicles the one-lane
// kernel cannot serve — is 110 us of latency behind the seeds kernel whatever the batch, and below
// SMX_ONE_LANE_MIN_VEHICLES it ends the tick; the team seeds kernel then costs less than it saves (C4's shards, default
// run / ticks 5-65, ms per tick, team seeds against one-lane seeds: 1024 envs 0.195 / 0.265 against 0.266 / 0.284; 2048:
// 0.251 / 0.378 against 0.288 / 0.383; 3072: 0.286 / 0.455 against 0.325 / 0.477; 4096: 0.367 / 0.603 against 0.380 / 0.579;
// the team kernels throughout: 0.227 / 0.287, 0.273 / 0.435, 0.371 / 0.593, 0.440 / 0.768).
static bool one_lane_seeds(const smx_handle_s* h) {
  if (h->launch_strategy == SMX_LAUNCH_LARGE_ONE_LANE) return true;
  const size_t total = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
  return total >= SMX_ONE_LANE_MIN_VEHICLES;
}

extern "C" int smx_launch_form(smx_handle h) {
  if (!h) return SMX_ERR_INVALID;
  if (!h->map_loaded) return fail(h, SMX_ERR_STATE, "smx_launch_form needs the map (the form depends on it)");
  const size_t total = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
  const bool small_batch = h->launch_strategy == SMX_LAUNCH_SMALL ||
                           (h->launch_strategy == SMX_LAUNCH_AUTO && total <= SMX_LARGE_BATCH_VEHICLES);
  if (small_batch) return SMX_FORM_SMALL;
  return (h->alive_blob && h->slow_blob && one_lane_cut(h)) ? SMX_FORM_LARGE_ONE_LANE : SMX_FORM_LARGE_TEAMS;
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
  for (int i = 0; i < t->sg_off[sg_cells]; ++i)
    if (t->sg_rec[i].lane < 0 || t->sg_rec[i].lane >= t->n_lanes || t->sg_rec[i].v0 < 0 || t->sg_rec[i].v0 + 1 >= t->n_shape_pts)
      return fail(h, SMX_ERR_INVALID, "segment record out of range");
  if (!t->lane_in_off || !t->lane_in_idx || !t->road_par_off || !t->road_par_idx)
    return fail(h, SMX_ERR_INVALID, "map tables: lane_in_* / road_par_* missing");
  for (int i = 0; i < t->lane_in_off[nl]; ++i)
    if (t->lane_in_idx[i] < 0 || t->lane_in_idx[i] >= t->n_lanes) return fail(h, SMX_ERR_INVALID, "incoming lane out of range");
  for (int i = 0; i < t->road_par_off[nr]; ++i)
    if (t->road_par_idx[i] < 0 || t->road_par_idx[i] >= t->n_roads) return fail(h, SMX_ERR_INVALID, "parallel road out of range");
  h->dagm_reach = 0.0;
  for (size_t i = 0; i < nl; ++i) h->dagm_reach = std::max(h->dagm_reach, 0.5 * t->lane_width[i]);
  // The slow lists' kernels run a fixed grid that strides a list whose length only the device knows.  On a map
  // whose lanes never split the lists hold a few vehicles of a hundred thousand and the grid is an empty launch's latency;
  // where lanes branch or cross, a third of the vehicles is on them (minicity, 262 144 vehicles: 77 000 rows through
  // 512 workgroups were half a wavefront per SIMD for nine passes, 1.4 ms of a 2.8 ms tick) — a team slot for every
  // second vehicle then.
  {
    // (a lanepoint with several successors: lanes that split.  Junction-internal lanes alone do not tell — the loop map's
    // two edges are joined by six of them, one successor each)
    bool junctions = false;
    for (int i = 0; i < t->n_lanepoints && !junctions; ++i) junctions = t->lp_rec[i].n_next > 1;
    const size_t tv = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
    const size_t teams_per_block = SMX_BLOCK / SMX_WP_LANES;
    h->map_junctions = junctions;
    h->slow_blocks = SMX_SLOW_BLOCKS;
    if (junctions) h->slow_blocks = (int)std::min<size_t>(8192, std::max<size_t>(SMX_SLOW_BLOCKS, tv / (2 * teams_per_block)));
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
  ADD(shape_rec, nv, smx_shape_rec);
  ADD(lane_out_off, nl + 1, int32_t);
  ADD(lane_out_idx, t->lane_out_off[nl], int32_t);
  ADD(lane_in_off, nl + 1, int32_t);
  ADD(lane_in_idx, t->lane_in_off[nl], int32_t);
  ADD(road_par_off, nr + 1, int32_t);
  ADD(road_par_idx, t->road_par_off[nr], int32_t);
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
  h->host_lane_road.assign(t->lane_road, t->lane_road + nl);
  h->host_lane_out_off.assign(t->lane_out_off, t->lane_out_off + nl + 1);
  h->host_lane_out_idx.assign(t->lane_out_idx, t->lane_out_idx + t->lane_out_off[nl]);
  // missions name roads of the map they were set for: a new map starts without any
  if (h->missions_blob) (void)hipFree(h->missions_blob);
  h->missions_blob = nullptr;
  h->missions = MissionsDev{nullptr, nullptr};
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
  PTR(shape_rec, smx_shape_rec);
  PTR(lane_out_off, int32_t);
  PTR(lane_out_idx, int32_t);
  PTR(lane_in_off, int32_t);
  PTR(lane_in_idx, int32_t);
  PTR(road_par_off, int32_t);
  PTR(road_par_idx, int32_t);
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
  if (!h->alive_blob) {  // the tick's alive list (large batches) + its two counters
    const size_t n = (size_t)h->cfg.num_envs * h->cfg.num_vehicles + 2;
    SMX_HIP(hipMalloc((void**)&h->alive_blob, n * sizeof(int32_t)));
    SMX_HIP(hipMemset(h->alive_blob, 0, n * sizeof(int32_t)));
  }
  if (!h->slow_blob) {
    const size_t n = 4 * (size_t)h->cfg.num_envs * h->cfg.num_vehicles + 8;
    SMX_HIP(hipMalloc((void**)&h->slow_blob, n * sizeof(int32_t)));
    SMX_HIP(hipMemset(h->slow_blob, 0, n * sizeof(int32_t)));
  }
  if (!h->pending_blob) {
    const size_t n = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
    SMX_HIP(hipMalloc((void**)&h->pending_blob, n));
    SMX_HIP(hipMemset(h->pending_blob, 0, n));
  }
  if (!h->scan_carry) {  // all ones = NaN: nothing to start a seeded search from yet
    const size_t n = 6 * (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
    SMX_HIP(hipMalloc((void**)&h->scan_carry, n * sizeof(double)));
    SMX_HIP(hipMemset(h->scan_carry, 0xff, n * sizeof(double)));
  }
  // hand-off storage of the waypoints sensor's chain walks (the library's own: it never leaves the tick)
  if ((h->cfg.sensors & SMX_SENSOR_WAYPOINTS) && !h->knots_blob) {
    const size_t paths = (size_t)h->cfg.num_envs * h->cfg.num_vehicles * SMX_WP_LANES;
    const size_t off_D = (size_t)(SMX_WPK_CAP + 1) * paths * sizeof(int32_t);
    const size_t off_n = off_D + paths * sizeof(double), off_nk = off_n + paths * sizeof(int16_t);
    const size_t off_end = off_nk + paths * sizeof(int16_t), off_key = off_end + paths * sizeof(int32_t);
    const size_t off_cnt = off_key + 3 * paths * sizeof(int32_t), off_nk16 = off_cnt + paths;
    const size_t bytes = off_nk16 + paths;
    SMX_HIP(hipMalloc(&h->knots_blob, bytes));
    SMX_HIP(hipMemset(h->knots_blob, 0, bytes));
    SMX_HIP(hipMemset((char*)h->knots_blob + off_key, 0xff, 3 * paths * sizeof(int32_t)));  // no list is valid yet
    char* kb = (char*)h->knots_blob;
    h->knots.idx = (int32_t*)kb;
    h->knots.D = (double*)(kb + off_D);
    h->knots.n = (int16_t*)(kb + off_n);
    h->knots.nk = (int16_t*)(kb + off_nk);
    h->knots.end16 = (int32_t*)(kb + off_end);
    h->knots.key = (int32_t*)(kb + off_key);
    h->knots.cnt = (uint8_t*)(kb + off_cnt);
    h->knots.nk16 = (uint8_t*)(kb + off_nk16);
  }
  if ((h->cfg.sensors & SMX_SENSOR_WAYPOINTS) && !h->spill_blob) {
    // a region of (SMX_WPE_KNOTS + 1) knot records + lanes per column of every k_waypoints_emit workgroup: 1.6 KB per
    // vehicle (the outputs are 8 KB), touched only by the few workgroups of a tick whose paths outgrow the LDS pool
    const size_t tv = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
    const size_t groups = (tv + SMX_WPT_VEHICLES - 1) / SMX_WPT_VEHICLES;
    SMX_HIP(hipMalloc(&h->spill_blob, groups * SMX_WPE_SPILL_GROUP_BYTES));
  }
  if (!h->status_dev) {
    SMX_HIP(hipMalloc((void**)&h->status_dev, sizeof(int32_t)));
    SMX_HIP(hipMemset(h->status_dev, 0, sizeof(int32_t)));
  }
  if (!h->ctrl_blob) {
    const size_t tv = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
    const size_t path_bytes = (size_t)SMX_CTRL_WPS * 3 * tv * sizeof(double);
    SMX_HIP(hipMalloc(&h->ctrl_blob, path_bytes + tv * sizeof(int32_t)));
    SMX_HIP(hipMemset(h->ctrl_blob, 0, path_bytes + tv * sizeof(int32_t)));
    h->ctrl.path = (double*)h->ctrl_blob;
    h->ctrl.n = (int32_t*)((char*)h->ctrl_blob + path_bytes);
  }
  if (!h->side_ready) {
    // lowest priority: the caller's stream carries the tick's critical chain (control -> seeds -> waypoints);
    // the side kernels fill the chip around it instead of sharing it evenly
    int prio_least = 0, prio_greatest = 0;
    SMX_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    for (int i = 0; i < 3; ++i) {
      SMX_HIP(hipStreamCreateWithPriority(&h->side[i], hipStreamNonBlocking, ((SMX_SIDE_PRIO >> i) & 1) ? 0 : prio_least));
      SMX_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    SMX_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    SMX_HIP(hipEventCreateWithFlags(&h->ev_fork_grid, hipEventDisableTiming));
    h->side_ready = true;
  }
  h->map_loaded = true;
  return SMX_OK;
}

extern "C" int smx_set_vias(smx_handle h, const smx_via* vias_host, int32_t n, const int32_t* slot_off_host) {
  if (!h) return SMX_ERR_INVALID;
  if (n < 0 || (n > 0 && (!vias_host || !slot_off_host))) return fail(h, SMX_ERR_INVALID, "smx_set_vias: null table");
  if (n > 0 && h->cfg.via_max <= 0) return fail(h, SMX_ERR_INVALID, "smx_set_vias: cfg.via_max is 0");
  if (n > 0 && !h->map_loaded) return fail(h, SMX_ERR_STATE, "smx_set_vias needs the map (lane indices are checked)");
  const int nv = h->cfg.num_vehicles;
  if (n > 0) {
    if (slot_off_host[0] != 0 || slot_off_host[nv] != n) return fail(h, SMX_ERR_INVALID, "smx_set_vias: slot offsets");
    for (int s = 0; s < nv; ++s)
      if (slot_off_host[s + 1] < slot_off_host[s] || slot_off_host[s + 1] - slot_off_host[s] > 32)
        return fail(h, SMX_ERR_INVALID, "smx_set_vias: at most 32 vias per agent, offsets ascending");
    for (int i = 0; i < n; ++i)
      if (vias_host[i].lane < 0 || vias_host[i].lane >= h->map.n_lanes)
        return fail(h, SMX_ERR_INVALID, "smx_set_vias: lane index out of range");
  }
  SMX_HIP(hipSetDevice(h->device));
  if (h->vias_dev) (void)hipFree(h->vias_dev);
  if (h->via_off_dev) (void)hipFree(h->via_off_dev);
  h->vias_dev = nullptr;
  h->via_off_dev = nullptr;
  h->n_vias = 0;
  if (n == 0) return SMX_OK;
  SMX_HIP(hipMalloc(&h->vias_dev, (size_t)n * sizeof(smx_via)));
  SMX_HIP(hipMalloc(&h->via_off_dev, (size_t)(nv + 1) * sizeof(int32_t)));
  SMX_HIP(hipMemcpy(h->vias_dev, vias_host, (size_t)n * sizeof(smx_via), hipMemcpyHostToDevice));
  SMX_HIP(hipMemcpy(h->via_off_dev, slot_off_host, (size_t)(nv + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  h->n_vias = n;
  return SMX_OK;
}

extern "C" int smx_set_missions(smx_handle h, const smx_mission* missions_host, int32_t n_slots,
                                const int32_t* route_roads_host, int32_t n_route_roads) {
  if (!h) return SMX_ERR_INVALID;
  if (!h->map_loaded) return fail(h, SMX_ERR_STATE, "smx_set_missions needs the map (road indices are checked)");
  if (n_slots < 0 || n_route_roads < 0 || (n_slots > 0 && !missions_host) || (n_route_roads > 0 && !route_roads_host))
    return fail(h, SMX_ERR_INVALID, "smx_set_missions: null table");
  const int nv = h->cfg.num_vehicles, nr = h->map.n_roads;
  if (n_slots != 0 && n_slots != nv) return fail(h, SMX_ERR_INVALID, "smx_set_missions: one mission per vehicle slot (cfg.num_vehicles)");
  if ((size_t)nv * (size_t)nr > 0x7fffffffull) return fail(h, SMX_ERR_INVALID, "smx_set_missions: slots x roads too large");
  const int nl = h->map.n_lanes;
  std::vector<int16_t> pos((size_t)n_slots * nr, (int16_t)-1);
  std::vector<uint8_t> lane_ok((size_t)n_slots * nl, (uint8_t)0);
  std::vector<int32_t> last((size_t)n_slots, -1);
  std::vector<double> goal((size_t)n_slots * 3, 0.0);
  bool any = false;
  for (int s = 0; s < n_slots; ++s) {
    const smx_mission& ms = missions_host[s];
    if (ms.route_len == 0) continue;  // endless mission: empty route (plan.py:321-323)
    if (ms.route_len < 0 || ms.route_len > 32767 || ms.route_off < 0 || (int64_t)ms.route_off + ms.route_len > n_route_roads)
      return fail(h, SMX_ERR_INVALID, "smx_set_missions: route range outside route_roads (at most 32767 roads)");
    if (!(ms.goal_radius >= 0.0) || !std::isfinite(ms.goal_x) || !std::isfinite(ms.goal_y))
      return fail(h, SMX_ERR_INVALID, "smx_set_missions: a fixed route needs a PositionalGoal (finite position, radius >= 0)");
    for (int k = 0; k < ms.route_len; ++k) {
      const int road = route_roads_host[ms.route_off + k];
      if (road < 0 || road >= nr) return fail(h, SMX_ERR_INVALID, "smx_set_missions: road index out of range");
      int16_t& p = pos[(size_t)s * nr + road];
      if (p < 0) p = (int16_t)k;  // first occurrence: `min` over the route keeps the first minimum
    }
    last[s] = route_roads_host[ms.route_off + ms.route_len - 1];
    // lanepoints.py:666-683 per lane (the rule lane_allowed evaluates for the short in-junction lists): on a road
    // of the route, and — unless that is the route's last road — leading on to a road of the route
    const int16_t* on = &pos[(size_t)s * nr];
    for (int lane = 0; lane < nl; ++lane) {
      const int road = h->host_lane_road[lane];
      bool ok = on[road] >= 0;
      if (ok && road != last[s]) {
        bool any = false;
        for (int k = h->host_lane_out_off[lane]; k < h->host_lane_out_off[lane + 1]; ++k)
          any = any || on[h->host_lane_road[h->host_lane_out_idx[k]]] >= 0;
        ok = any;
      }
      lane_ok[(size_t)s * nl + lane] = ok ? 1 : 0;
    }
    goal[3 * s] = ms.goal_x;
    goal[3 * s + 1] = ms.goal_y;
    goal[3 * s + 2] = ms.goal_radius;
    any = true;
  }
  SMX_HIP(hipSetDevice(h->device));
  SMX_HIP(hipDeviceSynchronize());  // launches in flight still read the old table
  if (h->missions_blob) (void)hipFree(h->missions_blob);
  h->missions_blob = nullptr;
  h->missions = MissionsDev{nullptr, nullptr};
  h->map.route_pos = nullptr;
  h->map.route_lane_ok = nullptr;
  // the knot lists of the previous tick were walked under the old routes
  if (h->knots_blob && h->knots.key)
    SMX_HIP(hipMemset(h->knots.key, 0xff, 3 * (size_t)h->cfg.num_envs * nv * SMX_WP_LANES * sizeof(int32_t)));
  if (!any) return SMX_OK;
  const size_t goal_bytes = goal.size() * sizeof(double), last_bytes = last.size() * sizeof(int32_t),
               pos_bytes = pos.size() * sizeof(int16_t);
  const size_t off_last = goal_bytes, off_pos = (goal_bytes + last_bytes + 7) & ~(size_t)7;
  const size_t off_lane = (off_pos + pos_bytes + 7) & ~(size_t)7;
  SMX_HIP(hipMalloc(&h->missions_blob, off_lane + lane_ok.size()));
  char* base = (char*)h->missions_blob;
  SMX_HIP(hipMemcpy(base, goal.data(), goal_bytes, hipMemcpyHostToDevice));
  SMX_HIP(hipMemcpy(base + off_last, last.data(), last_bytes, hipMemcpyHostToDevice));
  SMX_HIP(hipMemcpy(base + off_pos, pos.data(), pos_bytes, hipMemcpyHostToDevice));
  h->missions.goal = (const double*)base;
  h->missions.route_last = (const int32_t*)(base + off_last);
  SMX_HIP(hipMemcpy(base + off_lane, lane_ok.data(), lane_ok.size(), hipMemcpyHostToDevice));
  h->map.route_pos = (const int16_t*)(base + off_pos);
  h->map.route_lane_ok = (const uint8_t*)(base + off_lane);
  return SMX_OK;
}

extern "C" int smx_set_lidar_rays(smx_handle h, const double* rays_dev, int32_t n_rays) {
  if (!h) return SMX_ERR_INVALID;
  if (n_rays != h->cfg.lidar_rays) return fail(h, SMX_ERR_INVALID, "n_rays != cfg.lidar_rays");
  h->lidar_rays = rays_dev;
  return SMX_OK;
}

// ---- entry check of every smx_reset / smx_step* (and smx_check_buffers, which needs no device) ----
namespace {
struct BufSpec {
  const char* name;
  const void* ptr;
  uint64_t have;   // elements the caller declared
  uint8_t dtype;   // SMX_DT_* the caller declared
  uint64_t need;   // elements the configuration implies
  uint8_t want;    // SMX_DT_* of the ABI
  bool required;   // NULL is an error
};
const char* dtype_name(int d) {
  static const char* n[] = {"none", "f64", "f32", "i32", "i16", "i8", "u8", "u64"};
  return (d >= 0 && d <= SMX_DT_U64) ? n[d] : "?";
}
bool check_spec(const BufSpec& b, std::string& err) {
  if (!b.ptr) {
    if (!b.required) return true;
    err = std::string(b.name) + " is NULL but the configuration needs it";
    return false;
  }
  if (b.dtype != b.want) {
    err = std::string(b.name) + ": declared dtype " + dtype_name(b.dtype) + ", the ABI says " + dtype_name(b.want);
    return false;
  }
  if (b.have < b.need) {
    err = std::string(b.name) + ": " + std::to_string(b.have) + " elements declared, the configuration needs " +
          std::to_string(b.need) + " (a short buffer would be an out-of-bounds device write)";
    return false;
  }
  return true;
}
}  // namespace

static int check_buffers_impl(const smx_config& c, bool has_vias, bool need_lidar_rays_set, const smx_state* st,
                              const smx_spawns* sp, const smx_outputs* o, std::string& err) {
  if (!st || !sp || !o) {
    err = "null state / spawns / outputs";
    return SMX_ERR_INVALID;
  }
  (void)need_lidar_rays_set;
  const uint64_t E = (uint64_t)c.num_envs, T = E * (uint64_t)c.num_vehicles;
  const bool wp = (c.sensors & SMX_SENSOR_WAYPOINTS) != 0, nb = (c.sensors & SMX_SENSOR_NEIGHBORS) != 0;
  const bool ogm = (c.sensors & SMX_SENSOR_OGM) != 0, dagm = (c.sensors & SMX_SENSOR_DAGM) != 0;
  const bool lidar = (c.sensors & SMX_SENSOR_LIDAR) != 0, vias = c.via_max > 0 && has_vias;
  const bool rw = (c.sensors & SMX_SENSOR_ROAD_WAYPOINTS) != 0;
  const uint64_t RWL = rw ? (uint64_t)c.rw_lanes : 0, RWP = rw ? (uint64_t)c.rw_paths : 0, RWR = rw ? 2 * (uint64_t)c.rw_horizon + 1 : 0;
  const uint64_t PW = (uint64_t)c.wp_paths * c.wp_len, K = (uint64_t)c.nb_max, R = (uint64_t)c.lidar_rays;
#define ST(field, idx, need, want, req) \
  {"state." #field, st->field, st->count[idx], st->dtype[idx], (uint64_t)(need), want, req}
#define OUT(field, idx, need, want, req) \
  {"out." #field, o->field, o->count[idx], o->dtype[idx], (uint64_t)(need), want, req}
  const BufSpec specs[] = {
      ST(f64, SMX_ST_F64, SMX_S_COUNT * T, SMX_DT_F64, true),
      ST(flags, SMX_ST_FLAGS, T, SMX_DT_I32, true),
      ST(steps, SMX_ST_STEPS, T, SMX_DT_I32, true),
      ST(env_ticks, SMX_ST_ENV_TICKS, E, SMX_DT_I32, true),
      ST(env_done_count, SMX_ST_ENV_DONE_COUNT, E, SMX_DT_I32, true),
      ST(env_episode, SMX_ST_ENV_EPISODE, E, SMX_DT_I32, true),
      ST(driven_path, SMX_ST_DRIVEN_PATH, T * SMX_DRIVEN_PATH_LEN, SMX_DT_F64, (c.done_criteria & SMX_DONE_NOT_MOVING) != 0),
      ST(seed_cache, SMX_ST_SEED_CACHE, SMX_SEED_COUNT * T, SMX_DT_I32, true),
      ST(facts_i32, SMX_ST_FACTS_I32, SMX_FACT_I_COUNT * T, SMX_DT_I32, true),
      ST(facts_f64, SMX_ST_FACTS_F64, SMX_FACT_F_COUNT * T, SMX_DT_F64, true),
      ST(env_reset_pending, SMX_ST_ENV_RESET_PENDING, E, SMX_DT_I32, true),
      {"spawns.pose", sp->pose, sp->pose_count, SMX_DT_F64, (uint64_t)(sp->episodes > 0 ? sp->episodes : 0) * T * 4, SMX_DT_F64, true},
      {"spawns.social", sp->social, sp->social_count, SMX_DT_F64, (uint64_t)(sp->episodes > 0 ? sp->episodes : 0) * T * 2, SMX_DT_F64,
       c.num_social > 0},
      OUT(ego_pos, SMX_OUT_EGO_POS, 3 * T, SMX_DT_F64, true),
      OUT(ego_f32, SMX_OUT_EGO_F32, SMX_EGO_F32_COUNT * T, SMX_DT_F32, true),
      OUT(ego_lane, SMX_OUT_EGO_LANE, 2 * T, SMX_DT_I16, true),
      OUT(events, SMX_OUT_EVENTS, SMX_EV_COUNT * T, SMX_DT_U8, true),
      OUT(reward, SMX_OUT_REWARD, T, SMX_DT_F64, true),
      OUT(dist, SMX_OUT_DIST, T, SMX_DT_F64, true),
      OUT(done, SMX_OUT_DONE, T, SMX_DT_U8, true),
      OUT(active, SMX_OUT_ACTIVE, T, SMX_DT_U8, true),
      OUT(env_done, SMX_OUT_ENV_DONE, E, SMX_DT_U8, true),
      OUT(via_near, SMX_OUT_VIA_NEAR, T * (uint64_t)(c.via_max > 0 ? c.via_max : 0), SMX_DT_I8, vias),
      OUT(via_near_count, SMX_OUT_VIA_NEAR_COUNT, T, SMX_DT_U8, vias),
      OUT(via_hit, SMX_OUT_VIA_HIT, T, SMX_DT_I32, vias),
      OUT(learner, SMX_OUT_LEARNER, 2 * T, SMX_DT_F32, false),
      OUT(wp_pos, SMX_OUT_WP_POS, T * PW * 3, SMX_DT_F64, wp),
      OUT(wp_heading, SMX_OUT_WP_HEADING, T * PW, SMX_DT_F32, wp),
      OUT(wp_lane_width, SMX_OUT_WP_LANE_WIDTH, T * PW, SMX_DT_F32, wp),
      OUT(wp_speed_limit, SMX_OUT_WP_SPEED_LIMIT, T * PW, SMX_DT_F32, wp),
      OUT(wp_lane_index, SMX_OUT_WP_LANE_INDEX, T * PW, SMX_DT_I8, wp),
      OUT(wp_lane_id, SMX_OUT_WP_LANE_ID, T * PW, SMX_DT_I16, wp),
      OUT(wp_count, SMX_OUT_WP_COUNT, T * (uint64_t)(c.wp_paths + 1), SMX_DT_U8, wp),
      OUT(nb_pos, SMX_OUT_NB_POS, T * K * 3, SMX_DT_F64, nb),
      OUT(nb_box, SMX_OUT_NB_BOX, T * K * 3, SMX_DT_F32, nb),
      OUT(nb_heading, SMX_OUT_NB_HEADING, T * K, SMX_DT_F32, nb),
      OUT(nb_speed, SMX_OUT_NB_SPEED, T * K, SMX_DT_F32, nb),
      OUT(nb_lane_index, SMX_OUT_NB_LANE_INDEX, T * K, SMX_DT_I8, nb),
      OUT(nb_lane_id, SMX_OUT_NB_LANE_ID, T * K, SMX_DT_I16, nb),
      OUT(nb_slot, SMX_OUT_NB_SLOT, T * K, SMX_DT_I8, nb),
      OUT(nb_count, SMX_OUT_NB_COUNT, T, SMX_DT_U8, nb),
      OUT(ogm, SMX_OUT_OGM, T * (uint64_t)c.ogm_width * c.ogm_height, SMX_DT_U8, ogm),
      OUT(lidar_hit, SMX_OUT_LIDAR_HIT, T * R, SMX_DT_U8, lidar),
      OUT(lidar_point, SMX_OUT_LIDAR_POINT, T * R * 3, SMX_DT_F64, lidar),
      OUT(dagm, SMX_OUT_DAGM, T * (uint64_t)c.dagm_width * c.dagm_height, SMX_DT_U8, dagm),
      OUT(collidees, SMX_OUT_COLLIDEES, T, SMX_DT_U64, false),
      OUT(rw_lane_count, SMX_OUT_RW_LANE_COUNT, T, SMX_DT_U8, rw),
      OUT(rw_lane, SMX_OUT_RW_LANE, T * RWL, SMX_DT_I16, rw),
      OUT(rw_path_count, SMX_OUT_RW_PATH_COUNT, T * RWL, SMX_DT_I16, rw),
      OUT(rw_count, SMX_OUT_RW_COUNT, T * RWL * RWP, SMX_DT_U8, rw),
      OUT(rw_pos, SMX_OUT_RW_POS, T * RWL * RWP * RWR * 3, SMX_DT_F64, rw),
      OUT(rw_heading, SMX_OUT_RW_HEADING, T * RWL * RWP * RWR, SMX_DT_F32, rw),
      OUT(rw_lane_width, SMX_OUT_RW_LANE_WIDTH, T * RWL * RWP * RWR, SMX_DT_F32, rw),
      OUT(rw_speed_limit, SMX_OUT_RW_SPEED_LIMIT, T * RWL * RWP * RWR, SMX_DT_F32, rw),
      OUT(rw_lane_index, SMX_OUT_RW_LANE_INDEX, T * RWL * RWP * RWR, SMX_DT_I8, rw),
      OUT(rw_lane_id, SMX_OUT_RW_LANE_ID, T * RWL * RWP * RWR, SMX_DT_I16, rw),
      OUT(final_ego_pos, SMX_OUT_FINAL_EGO_POS, 3 * T, SMX_DT_F64, false),
      OUT(final_ego_f32, SMX_OUT_FINAL_EGO_F32, SMX_EGO_F32_COUNT * T, SMX_DT_F32, false),
      OUT(final_ego_lane, SMX_OUT_FINAL_EGO_LANE, 2 * T, SMX_DT_I16, false),
      OUT(final_events, SMX_OUT_FINAL_EVENTS, SMX_EV_COUNT * T, SMX_DT_U8, false),
      OUT(final_dist, SMX_OUT_FINAL_DIST, T, SMX_DT_F64, false),
  };
#undef ST
#undef OUT
  if (sp->episodes < 1) {
    err = "spawn table is empty (episodes < 1)";
    return SMX_ERR_INVALID;
  }
  for (const BufSpec& b : specs)
    if (!check_spec(b, err)) return SMX_ERR_INVALID;
  const int finals = (o->final_ego_pos != nullptr) + (o->final_ego_f32 != nullptr) + (o->final_ego_lane != nullptr) +
                     (o->final_events != nullptr) + (o->final_dist != nullptr);
  if (finals != 0 && finals != 5) {
    err = "out.final_*: give all five buffers or none";
    return SMX_ERR_INVALID;
  }
  return SMX_OK;
}

extern "C" int smx_check_buffers(const smx_config* cfg, int has_vias, const smx_state* st, const smx_spawns* sp,
                                 const smx_outputs* out, char* err, uint64_t err_len) {
  std::string msg;
  int rc = cfg ? check_buffers_impl(*cfg, has_vias != 0, false, st, sp, out, msg) : SMX_ERR_INVALID;
  if (!cfg) msg = "null config";
  if (err && err_len > 0) {
    const size_t n = std::min<size_t>(msg.size(), (size_t)err_len - 1);
    memcpy(err, msg.data(), n);
    err[n] = 0;
  }
  return rc;
}

static int check_buffers(smx_handle h, const smx_state* st, const smx_spawns* sp, const smx_outputs* o) {
  std::string msg;
  const int rc = check_buffers_impl(h->cfg, h->n_vias > 0, true, st, sp, o, msg);
  if (rc != SMX_OK) return fail(h, rc, msg);
  if ((h->cfg.sensors & SMX_SENSOR_LIDAR) && !h->lidar_rays)
    return fail(h, SMX_ERR_STATE, "lidar sensor enabled but smx_set_lidar_rays has not been called");
  return SMX_OK;
}

static int enqueue(smx_handle h, bool is_step, const int8_t* actions, const float* actions_f32, const double* traj,
                   const int32_t* traj_n, const uint8_t* mask, const smx_state* st,
                   const smx_spawns* sp, const smx_outputs* out, void* stream_) {
  if (!h) return SMX_ERR_INVALID;
  if (!h->map_loaded) return fail(h, SMX_ERR_STATE, "smx_load_map has not been called");
  int rc = check_buffers(h, st, sp, out);
  if (rc != SMX_OK) return rc;
  if (is_step) {
    const int sp_ = h->cfg.action_space;
    const bool ok = sp_ == SMX_ACTION_SPACE_LANE ? actions != nullptr
                    : sp_ == SMX_ACTION_SPACE_TRAJECTORY ? (traj != nullptr && traj_n != nullptr)
                                                         : actions_f32 != nullptr;
    if (!ok)
      return fail(h, SMX_ERR_INVALID,
                  "actions do not match cfg.action_space (smx_step: Lane, smx_step_trajectory: Trajectory, "
                  "smx_step_continuous: the float spaces)");
  }
  hipStream_t stream = (hipStream_t)stream_;
  const smx_config& c = h->cfg;
  KernelArgs a;
  a.cfg = c;
  a.map = h->map;
  a.st = *st;
  a.sp = *sp;
  a.out = *out;
  a.actions = actions;
  a.actions_f32 = actions_f32;
  a.traj = traj;
  a.traj_n = traj_n;
  a.env_mask = mask;
  a.lidar_rays = h->lidar_rays;
  a.vias = h->n_vias > 0 ? h->vias_dev : nullptr;
  a.missions = h->missions;
  a.via_slot_off = h->via_off_dev;
  a.first_only = 0;
  a.keep_reward_done = 0;
  a.reset_all = 0;
  a.heading_gain_pos = h->heading_gain_pos;
  a.nb_d2_max = h->nb_d2_max;
  a.walk_new = 0;
  a.wp_spill = (char*)h->spill_blob;
  a.wp_pool_limit = h->wp_pool_limit;
  a.lateral_gain_pos = h->lateral_gain_pos;
  a.debug_skip = h->debug_skip;
  a.knots = h->knots;
  a.status = h->status_dev;
  a.alive_list = nullptr;
  a.alive_count = nullptr;
  const size_t total = (size_t)c.num_envs * c.num_vehicles;
  a.slow_list = nullptr;
  a.slow_count = nullptr;
  a.seed_pending = nullptr;
  a.seeds_carry = h->scan_carry;
  a.facts_carry = h->scan_carry ? h->scan_carry + 4 * total : nullptr;
  const int veh_blocks = (int)((total + SMX_BLOCK - 1) / SMX_BLOCK);
  // Small batches are bound by one wavefront's latency, so independent work is spread over more
  // workgroups (k_scan halves as separate roles: 54 vs 70 us at 8 k vehicles; the OGM role inside
  // k_sensors); large batches are bound by throughput, where the same tricks cost occupancy
  // (131 k vehicles: k_scan 0.69 vs 0.52 ms split vs back-to-back, OGM inside k_sensors +6 %).
  const bool small_batch = h->launch_strategy == SMX_LAUNCH_SMALL ||
                           (h->launch_strategy == SMX_LAUNCH_AUTO && total <= SMX_LARGE_BATCH_VEHICLES) || SMX_SKIP(*h, 131072);
  const int scan_split = (small_batch || SMX_SKIP(*h, 65536)) ? 1 : 0;
  const int scan_blocks = (scan_split ? 2 : 1) * (int)((total * SMX_TEAM + SMX_BLOCK - 1) / SMX_BLOCK);
  const int vpb = SMX_BLOCK / SMX_WP_LANES;
  const int wp_blocks = (int)((total + vpb - 1) / vpb);
  const int epb = SMX_BLOCK / c.num_vehicles;
  const int obs_blocks = (c.num_envs + epb - 1) / epb;
  const int env_blocks = (c.num_envs + SMX_BLOCK - 1) / SMX_BLOCK;
  const bool timed = h->timing && is_step && h->ev_used < 65536;
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
  // phase timing (smx_set_timing level 2): one boundary event after every kernel of the tick
  const bool phased = h->phase_timing && is_step && h->ph_used < 16384;
  hipEvent_t* ph = nullptr;
  if (phased) {
    const size_t need = (h->ph_used + 1) * (SMX_PHASE_COUNT + 1);
    while (h->ph_pool.size() < need) {
      hipEvent_t e;
      SMX_HIP(hipEventCreate(&e));
      h->ph_pool.push_back(e);
    }
    ph = &h->ph_pool[h->ph_used * (SMX_PHASE_COUNT + 1)];
    SMX_HIP(hipEventRecord(ph[0], stream));
  }
#define SMX_PHASE_END(p) \
  if (phased) SMX_HIP(hipEventRecord(ph[(p) + 1], stream))
  const int lidar_blocks = (c.sensors & SMX_SENSOR_LIDAR) ? (int)total : 0;
  // the OGM tile is dynamic LDS: on small batches, up to 16 KiB, it rides along in k_sensors; otherwise
  // the OGM role gets its own launch
  const size_t ogm_bytes = (c.sensors & SMX_SENSOR_OGM) ? (size_t)c.ogm_width * c.ogm_height : 0;
  // (from SMX_OGM_ENV_MIN_VEHICLES on, the per-env kernel of the large form beats the per-observer role inside
  // k_sensors even as a launch of its own on the same stream: a third of the instructions per tile)
  const bool ogm_env_small = small_batch && total >= SMX_OGM_ENV_MIN_VEHICLES && c.num_vehicles <= 32 && ogm_bytes * SMX_OGM_WAVES * 2 <= 64 * 1024;
  const bool ogm_inline = small_batch && ogm_bytes > 0 && ogm_bytes <= 16 * 1024 && !ogm_env_small;
  const bool ogm_alone = ogm_bytes > 0 && !ogm_inline;
  const size_t dagm_bytes = (c.sensors & SMX_SENSOR_DAGM) ? (size_t)c.dagm_width * c.dagm_height : 0;
  a.dagm_reach = h->dagm_reach;
  a.wp_blocks = wp_blocks;
  a.obs_blocks = obs_blocks;
  a.lidar_blocks = lidar_blocks;
  const unsigned sensor_blocks = (unsigned)(wp_blocks + obs_blocks + lidar_blocks + (ogm_inline ? (int)total : 0));
  const size_t sensor_lds = ogm_inline ? ogm_bytes : 0;
  // one pass of scan + sensors + commit (the tick's, then the reset pass restricted to new vehicles)
  // OGM tiles on their own: per env (four wavefronts share the env's poses) on large batches while four
  // tiles fit a workgroup's LDS, else per observer
  auto launch_ogm = [&](hipStream_t st_, const KernelArgs& k) {
    if ((!small_batch || ogm_env_small) && c.num_vehicles <= 32 && ogm_bytes * SMX_OGM_WAVES * 2 <= 64 * 1024)
      hipLaunchKernelGGL(k_ogm_env<2>, dim3((unsigned)c.num_envs), dim3(SMX_OGM_WAVES * 64), ogm_bytes * SMX_OGM_WAVES * 2, st_, k);
    else if (!small_batch && ogm_bytes * SMX_OGM_WAVES <= 64 * 1024)
      hipLaunchKernelGGL(k_ogm_env<1>, dim3((unsigned)c.num_envs), dim3(SMX_OGM_WAVES * 64), ogm_bytes * SMX_OGM_WAVES, st_, k);
    else
      hipLaunchKernelGGL(k_ogm, dim3((unsigned)total), dim3(SMX_BLOCK), ogm_bytes, st_, k);
  };
  const bool routed = h->missions.route_last != nullptr;  // some slot has a fixed route: the scan instance that knows them
  bool fast_scan = false;  // the tick of a large batch (not its reset pass): one-lane scan kernels + teams over their slow lists
  int slow_parity = 0;
  auto observation_pass = [&](const KernelArgs& k, bool phases) {
    // Large batches, no per-kernel timing asked: the grid maps and the lidar (which read poses only) leave on
    // side stream 0 at once and overlap the scan — kernels bound by their own write stream beside one bound by
    // arithmetic and load latency; observe goes to side stream 1 after the scan, the waypoint kernels stay on
    // the caller's stream; all are joined before k_commit.
    const bool fork = !small_batch && !phased && h->side_ready;
    hipStream_t s_grid = stream, s_obs = stream;
    KernelArgs kf = k, ks = k, kwp = k;  // (kwp: the waypoint kernels, which pass over the slow chain's vehicles)
    bool slow_chain_forked = false, slow_chain_pending = false, seeds_fork_recorded = false;
    if (fork) {
      (void)hipEventRecord(h->ev_fork_grid, stream);
      (void)hipStreamWaitEvent(h->side[0], h->ev_fork_grid, 0);
      s_grid = h->side[0];
      s_obs = h->side[1];
      if (ogm_alone) launch_ogm(s_grid, k);
      if (dagm_bytes) hipLaunchKernelGGL(k_dagm, dim3((unsigned)total), dim3(SMX_BLOCK), dagm_bytes, s_grid, k);
      if (lidar_blocks) hipLaunchKernelGGL(k_lidar, dim3((unsigned)lidar_blocks), dim3(SMX_BLOCK), 0, s_grid, k);
    }
    if (!small_batch) {
      // the scan's halves as two launches, on two streams when forked: path seeds (-> waypoint kernels) on the
      // caller's, road facts (-> observe) on side 1
      const unsigned half_blocks = (unsigned)((total * SMX_TEAM_LARGE + SMX_BLOCK - 1) / SMX_BLOCK);
      // (where the searches are long — lanes that split and cross: 4lane 2048 x 16 0.430 -> 0.393 ms; on loop the four-lane
      // teams stay: 32 768 vehicles 0.260 either way, 65 536 0.371 against 0.378)
      const bool wide_teams = h->map_junctions && total <= SMX_SCAN_WIDE_MAX_VEHICLES;
      const unsigned wide_blocks = (unsigned)((total * SMX_TEAM + SMX_BLOCK - 1) / SMX_BLOCK);
      const bool fast = fast_scan && k.alive_list != nullptr && !k.first_only && h->slow_blob && !SMX_SCAN_UNSEEDED;
      const unsigned fast_blocks = (unsigned)((total + SMX_BLOCK - 1) / SMX_BLOCK);
      kf = k;
      ks = k;  // facts / seeds: each half appends to its own slow list
      if (fast) {
        int32_t* slow_counters = h->slow_blob + 4 * total + 4 * slow_parity;
        kf.slow_list = h->slow_blob;
        kf.slow_count = slow_counters;
        ks.slow_list = h->slow_blob + total;
        ks.slow_count = slow_counters + 1;
      }
      // path seeds without the ten-nearest list: agents with a route object and no fixed route, waypoints sensor on
      const bool fast_seeds = !SMX_WP_STAGED && fast && !routed && (c.sensors & SMX_SENSOR_WAYPOINTS) && h->pending_blob && one_lane_seeds(h);
      if (fast_seeds) {
        ks.seed_pending = h->pending_blob;
        kwp.seed_pending = h->pending_blob;
        hipLaunchKernelGGL(k_scan_fast<1>, dim3(fast_blocks), dim3(SMX_BLOCK), 0, stream, ks);
        if (fork) {
          // the slow chain — the few vehicles whose seeds take the searches from scratch, then their walks and rows by the
          // serial emitter — runs beside the main chain (k_wp_walk -> k_waypoints_emit pass over those vehicles)
          // (one event after the seeds half serves this fork and the facts half's below: every record on the caller's
          // stream is a packet its next kernel waits behind, ten microseconds of the tick's longest chain)
          (void)hipEventRecord(h->ev_fork, stream);
          seeds_fork_recorded = true;
          (void)hipStreamWaitEvent(h->side[2], h->ev_fork, 0);
          if (h->map_junctions)
            hipLaunchKernelGGL(k_scan_listed<1>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, h->side[2], ks);
          else
            hipLaunchKernelGGL((k_scan_listed<1, false, SMX_TEAM>), dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, h->side[2], ks);
          if (!h->map_junctions && h->knots_blob) {
            hipLaunchKernelGGL(k_waypoints_walk_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, h->side[2], ks);
          } else {
            hipLaunchKernelGGL(k_waypoints_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, h->side[2], ks);
            if (h->knots_blob) hipLaunchKernelGGL(k_wp_walk_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, h->side[2], ks);
          }
          slow_chain_forked = true;
        } else {
          slow_chain_pending = true;  // (one stream: after the main waypoint kernels, below)
        }
      } else if (routed)
        hipLaunchKernelGGL((k_scan_half<1, true>), dim3(half_blocks), dim3(SMX_BLOCK), 0, stream, k);
      else if (wide_teams)
        hipLaunchKernelGGL((k_scan_half<1, false, SMX_TEAM>), dim3(wide_blocks), dim3(SMX_BLOCK), 0, stream, k);
      else
        hipLaunchKernelGGL(k_scan_half<1>, dim3(half_blocks), dim3(SMX_BLOCK), 0, stream, k);
      // the facts half (-> observe) has slack, the seeds half heads the tick's longest chain (-> walk -> rows): the
      // facts half starts when the seeds half is done and then fills the chip beside the waypoint kernels, whose two
      // wavefronts per SIMD leave it half empty (C4, ticks 20-220: 0.815 -> 0.792 ms)
      // (late in a run, with 40 % of the agents alive, starting both halves together is 1.5 % faster; with 80 % alive
      // it is 4 % slower)
      // (at 32 768 vehicles — a quarter of the headline batch, one rank's shard at four GPUs — the chains are short and
      // both halves start together: 0.241 against 0.249 ms; at 65 536 the order above wins, 0.294 against 0.300)
      if (fork && total <= SMX_FACTS_EARLY_MAX) {
        (void)hipStreamWaitEvent(h->side[1], h->ev_fork_grid, 0);
      } else if (fork) {
        if (!seeds_fork_recorded) (void)hipEventRecord(h->ev_fork, stream);
        (void)hipStreamWaitEvent(h->side[1], h->ev_fork, 0);
      }
      if (fast) {
        hipLaunchKernelGGL(k_scan_fast<0>, dim3(fast_blocks), dim3(SMX_BLOCK), 0, s_obs, kf);
        hipLaunchKernelGGL(k_scan_listed<0>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, s_obs, kf);
      } else if (wide_teams)
        hipLaunchKernelGGL((k_scan_half<0, false, SMX_TEAM>), dim3(wide_blocks), dim3(SMX_BLOCK), 0, s_obs, k);
      else
        hipLaunchKernelGGL(k_scan_half<0>, dim3(half_blocks), dim3(SMX_BLOCK), 0, s_obs, k);  // (the facts half seeds no path)
      // (holding the grid kernels back as well was slower: 0.81 -> 0.85 ms; they overlap the seeds half)
    } else if (scan_split) {
      if (routed)
        hipLaunchKernelGGL((k_scan<true, true>), dim3(scan_blocks), dim3(SMX_BLOCK), 0, stream, k);
      else
        hipLaunchKernelGGL(k_scan<true>, dim3(scan_blocks), dim3(SMX_BLOCK), 0, stream, k);
    } else {
      if (routed)
        hipLaunchKernelGGL((k_scan<false, true>), dim3(scan_blocks), dim3(SMX_BLOCK), 0, stream, k);
      else
        hipLaunchKernelGGL(k_scan<false>, dim3(scan_blocks), dim3(SMX_BLOCK), 0, stream, k);
    }
    if (phases && phased) (void)hipEventRecord(ph[SMX_PHASE_SCAN + 1], stream);
    if (!fork) {
      if (ogm_alone) launch_ogm(stream, k);
      if (dagm_bytes) hipLaunchKernelGGL(k_dagm, dim3((unsigned)total), dim3(SMX_BLOCK), dagm_bytes, stream, k);
    }
    if (phases && phased) (void)hipEventRecord(ph[SMX_PHASE_OGM + 1], stream);
    if (small_batch) {
      hipLaunchKernelGGL(k_sensors, dim3(sensor_blocks), dim3(SMX_BLOCK), sensor_lds, stream, k);
    } else {
      // staged rows whenever the waypoints sensor is on and the rows fit that form
      if ((c.sensors & SMX_SENSOR_WAYPOINTS) && c.wp_paths <= SMX_WPT_MAX_PATHS && h->knots_blob) {
        hipLaunchKernelGGL(k_wp_walk, dim3((unsigned)((total * SMX_WP_LANES + SMX_BLOCK - 1) / SMX_BLOCK)), dim3(SMX_BLOCK), 0, stream, kwp);
#if SMX_WP_STAGED  // developer variant: round 2's staged rows
        const size_t stage_bytes = std::max((size_t)c.wp_len * SMX_BLOCK * 16, (size_t)SMX_MAX_KNOTS * SMX_BLOCK * sizeof(int));
        hipLaunchKernelGGL(k_waypoints_tables, dim3(wp_blocks), dim3(SMX_BLOCK), stage_bytes, stream, k);
#else
        if (fast_scan && k.alive_list != nullptr && h->slow_blob) {
          KernelArgs kw = kwp;
          kw.slow_list = h->slow_blob + 3 * total;
          kw.slow_count = h->slow_blob + 4 * total + 4 * slow_parity + 3;
          hipLaunchKernelGGL(k_waypoints_emit, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, kw);
          hipLaunchKernelGGL(k_waypoints_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, kw);
          if (slow_chain_pending) {
            hipLaunchKernelGGL(k_scan_listed<1>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ks);
            hipLaunchKernelGGL(k_waypoints_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ks);
            if (h->knots_blob) hipLaunchKernelGGL(k_wp_walk_listed, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ks);
            slow_chain_pending = false;
          }
        } else {
          const size_t stage_bytes = std::max((size_t)c.wp_len * SMX_BLOCK * 16, (size_t)SMX_MAX_KNOTS * SMX_BLOCK * sizeof(int));
          hipLaunchKernelGGL(k_waypoints_tables, dim3(wp_blocks), dim3(SMX_BLOCK), stage_bytes, stream, k);
        }
#endif
      } else
        hipLaunchKernelGGL(k_waypoints, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, k);
      hipLaunchKernelGGL(k_observe, dim3(obs_blocks), dim3(SMX_BLOCK), 0, s_obs, k);
      if (lidar_blocks && !fork) hipLaunchKernelGGL(k_lidar, dim3((unsigned)lidar_blocks), dim3(SMX_BLOCK), 0, stream, k);
    }
    if (fork) {
      // (each side stream joins the caller's directly: chaining side 0 through side 1 puts one more hop behind the
      // last kernel when k_observe ends the tick — 1 % late in a run)
      for (int i = 0; i < (slow_chain_forked ? 3 : 2); ++i) {
        (void)hipEventRecord(h->ev_join[i], h->side[i]);
        (void)hipStreamWaitEvent(stream, h->ev_join[i], 0);
      }
    }
    if (c.sensors & SMX_SENSOR_ROAD_WAYPOINTS)  // (poses are the tick's new ones; flags still those of its start)
      hipLaunchKernelGGL(k_road_waypoints, dim3((unsigned)((total * SMX_RW_LANE_CAP + SMX_BLOCK - 1) / SMX_BLOCK)), dim3(SMX_BLOCK), 0,
                         stream, k);
    if (phases && phased) (void)hipEventRecord(ph[SMX_PHASE_SENSORS + 1], stream);
    hipLaunchKernelGGL(k_commit, dim3(obs_blocks), dim3(SMX_BLOCK), 0, stream, k);
    if (phases && phased) (void)hipEventRecord(ph[SMX_PHASE_COMMIT + 1], stream);
  };
  // the LDS-path form of k_control fits one wavefront per SIMD: only while the batch needs no more
  const bool lds_path = small_batch && total * SMX_WP_LANES <= (size_t)1024 * 64;
  if (is_step && c.num_social > 0 && c.social_model == SMX_SOCIAL_IDM)
    hipLaunchKernelGGL(k_social, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a);
  const bool two_launch_control = is_step && !small_batch && h->ctrl_blob;
  if (is_step && !small_batch && h->alive_blob) {
    int32_t* counters = h->alive_blob + total;
    int32_t* slow_counters = h->slow_blob + 4 * total;  // [parity][4]: facts, seeds, control, waypoint rows
    hipLaunchKernelGGL(k_alive_list, dim3((unsigned)((total + SMX_ALIVE_BLOCK - 1) / SMX_ALIVE_BLOCK)), dim3(SMX_ALIVE_BLOCK), 0, stream, a, h->alive_blob, counters + h->alive_parity,
                       counters + (h->alive_parity ^ 1), slow_counters + 4 * (h->alive_parity ^ 1));
    a.alive_list = h->alive_blob;
    a.alive_count = counters + h->alive_parity;
    // One lane per vehicle + slow lists where the slow lists stay short: a map whose lanes never split (loop: under 1 %
    // of the vehicles).  Where lanes branch and cross, a third of the vehicles would take the lists' serial forms
    // (minicity, 262 144 vehicles: 1.40 ms a tick against 0.9x with round 2's team kernels for everybody), so those
    // maps keep the team kernels.
    fast_scan = one_lane_cut(h);
    slow_parity = h->alive_parity;
    h->alive_parity ^= 1;
  }
  if (two_launch_control) {
    // large batches: candidate paths by teams of four, then law + physics with one lane per vehicle
    const CtrlHandoff ho = h->ctrl;
    KernelArgs ac = a;  // the controller's slow list
    if (fast_scan && h->slow_blob) {
      ac.slow_list = h->slow_blob + 2 * total;
      ac.slow_count = h->slow_blob + 4 * total + 4 * slow_parity + 2;
    }
    switch (c.action_space) {
      case SMX_ACTION_SPACE_LANE:
        if (fast_scan && h->slow_blob) {  // one lane per vehicle; the rest through the slow list (k_control_fast)
          hipLaunchKernelGGL(k_control_fast<SMX_ACTION_SPACE_LANE>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, ac);
          hipLaunchKernelGGL(k_control_paths_listed<SMX_ACTION_SPACE_LANE>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ac, ho);
          hipLaunchKernelGGL(k_control_law_listed<SMX_ACTION_SPACE_LANE>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ac, ho);
        } else {
          hipLaunchKernelGGL(k_control_paths<SMX_ACTION_SPACE_LANE>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
          hipLaunchKernelGGL(k_control_law<SMX_ACTION_SPACE_LANE>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
        }
        break;
      case SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED:
        if (fast_scan && h->slow_blob) {
          hipLaunchKernelGGL(k_control_fast<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, ac);
          hipLaunchKernelGGL(k_control_paths_listed<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ac, ho);
          hipLaunchKernelGGL(k_control_law_listed<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3((unsigned)h->slow_blocks), dim3(SMX_BLOCK), 0, stream, ac, ho);
        } else {
          hipLaunchKernelGGL(k_control_paths<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
          hipLaunchKernelGGL(k_control_law<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
        }
        break;
      case SMX_ACTION_SPACE_CONTINUOUS:
        hipLaunchKernelGGL(k_control_law<SMX_ACTION_SPACE_CONTINUOUS>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
        break;
      case SMX_ACTION_SPACE_ACTUATOR_DYNAMIC:
        hipLaunchKernelGGL(k_control_law<SMX_ACTION_SPACE_ACTUATOR_DYNAMIC>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
        break;
      default:
        hipLaunchKernelGGL(k_control_law<SMX_ACTION_SPACE_TRAJECTORY>, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, a, ho);
        break;
    }
  }
  if (is_step) {
    if (!two_launch_control) switch (c.action_space) {
      case SMX_ACTION_SPACE_LANE:
        if (lds_path)
          hipLaunchKernelGGL((k_control<SMX_ACTION_SPACE_LANE, true>), dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a);
        else
          hipLaunchKernelGGL(k_control<SMX_ACTION_SPACE_LANE>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a);
        break;
      case SMX_ACTION_SPACE_CONTINUOUS:
        hipLaunchKernelGGL(k_control<SMX_ACTION_SPACE_CONTINUOUS>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a);
        break;
      case SMX_ACTION_SPACE_ACTUATOR_DYNAMIC:
        hipLaunchKernelGGL(k_control<SMX_ACTION_SPACE_ACTUATOR_DYNAMIC>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a);
        break;
      case SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED:
        if (lds_path)
          hipLaunchKernelGGL((k_control<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED, true>), dim3(wp_blocks),
                             dim3(SMX_BLOCK), 0, stream, a);
        else
          hipLaunchKernelGGL(k_control<SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED>, dim3(wp_blocks), dim3(SMX_BLOCK), 0,
                             stream, a);
        break;
      default:
        hipLaunchKernelGGL(k_control<SMX_ACTION_SPACE_TRAJECTORY>, dim3(wp_blocks), dim3(SMX_BLOCK), 0, stream, a);
        break;
    }
    SMX_PHASE_END(SMX_PHASE_CONTROL);
    observation_pass(a, true);
  }
  if (!is_step || c.auto_reset) {
    KernelArgs r = a;
    r.alive_list = nullptr;
    r.alive_count = nullptr;
    r.first_only = 1;
    r.keep_reward_done = is_step ? 1 : 0;
    r.reset_all = (!is_step && mask == nullptr) ? 1 : 0;
    r.env_mask = is_step ? nullptr : mask;
    if (!is_step) {  // in a step, k_commit has already respawned the envs that ended
      hipLaunchKernelGGL(k_reset, dim3(veh_blocks), dim3(SMX_BLOCK), 0, stream, r);
      hipLaunchKernelGGL(k_reset_env, dim3(env_blocks), dim3(SMX_BLOCK), 0, stream, r);
    }
    const unsigned sweep_blocks = (unsigned)((total + SMX_BLOCK - 1) / SMX_BLOCK);
    if (ogm_bytes) hipLaunchKernelGGL(k_grid_first<false>, dim3(sweep_blocks), dim3(SMX_BLOCK), ogm_bytes, stream, r);
    if (dagm_bytes) hipLaunchKernelGGL(k_grid_first<true>, dim3(sweep_blocks), dim3(SMX_BLOCK), dagm_bytes, stream, r);
    if (c.sensors & SMX_SENSOR_ROAD_WAYPOINTS)  // before k_first clears SMX_F_FIRST
      hipLaunchKernelGGL(k_road_waypoints, dim3((unsigned)((total * SMX_RW_LANE_CAP + SMX_BLOCK - 1) / SMX_BLOCK)), dim3(SMX_BLOCK), 0,
                         stream, r);
    if (lidar_blocks && !small_batch) {
      // large batches: the new vehicles' lidar as a launch of its own instead of one after the other inside
      // k_first — at C5 an env restart brings 64 new vehicles, whose serial lidar roles made the reset pass 0.58 ms
      // of a 1.5 ms tick late in a run (many restarts per tick)
      hipLaunchKernelGGL(k_lidar_first, dim3((unsigned)std::min<size_t>(SMX_LIDAR_FIRST_BLOCKS, total)), dim3(SMX_BLOCK), 0, stream, r);
      r.lidar_blocks = 0;
    }
    // large batches: k_first also walks the new vehicles' knot lists, for the next tick's k_control_fast
    // (only k_control_fast reads them: not on the maps that keep the team kernels)
    r.walk_new = (!small_batch && one_lane_cut(h) && h->knots_blob && (c.sensors & SMX_SENSOR_WAYPOINTS) &&
                  c.wp_paths <= SMX_WPT_MAX_PATHS) ? 1 : 0;
    hipLaunchKernelGGL(k_first, dim3(obs_blocks), dim3(SMX_FIRST_BLOCK), 0, stream, r);
  }
  SMX_HIP(hipGetLastError());
  if (phased) {
    SMX_PHASE_END(SMX_PHASE_RESET);
    h->ph_used += 1;
  }
#undef SMX_PHASE_END
  if (timed) {
    SMX_HIP(hipEventRecord(h->ev_pool[2 * h->ev_used + 1], stream));
    h->ev_used += 1;
  }
  return SMX_OK;
}

extern "C" int smx_reset(smx_handle h, const uint8_t* env_mask_dev, const smx_state* st, const smx_spawns* sp,
                         const smx_outputs* out, void* hip_stream) {
  return enqueue(h, false, nullptr, nullptr, nullptr, nullptr, env_mask_dev, st, sp, out, hip_stream);
}

extern "C" int smx_step(smx_handle h, const int8_t* actions_dev, const smx_state* st, const smx_spawns* sp,
                        const smx_outputs* out, void* hip_stream) {
  return enqueue(h, true, actions_dev, nullptr, nullptr, nullptr, nullptr, st, sp, out, hip_stream);
}

extern "C" int smx_step_continuous(smx_handle h, const float* actions_dev, const smx_state* st, const smx_spawns* sp,
                                   const smx_outputs* out, void* hip_stream) {
  return enqueue(h, true, nullptr, actions_dev, nullptr, nullptr, nullptr, st, sp, out, hip_stream);
}

extern "C" int smx_step_trajectory(smx_handle h, const double* trajectories_dev, const int32_t* counts_dev,
                                   const smx_state* st, const smx_spawns* sp, const smx_outputs* out, void* hip_stream) {
  return enqueue(h, true, nullptr, nullptr, trajectories_dev, counts_dev, nullptr, st, sp, out, hip_stream);
}

extern "C" int smx_sync(smx_handle h, void* hip_stream) {
  if (!h) return SMX_ERR_INVALID;
  SMX_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  if (h->status_dev) {  // what the kernels could not report themselves
    int32_t bits = 0;
    SMX_HIP(hipMemcpy(&bits, h->status_dev, sizeof(bits), hipMemcpyDeviceToHost));
    if (bits) {
      SMX_HIP(hipMemset(h->status_dev, 0, sizeof(int32_t)));
      if (bits & SMX_DEVICE_BAD_LANE_ACTION)
        return fail(h, SMX_ERR_INVALID,
                    "a Lane action code outside -1..3 reached smx_step since the last smx_sync; the agents that sent one "
                    "were stepped as if they had sent no action");
    }
  }
  return SMX_OK;
}

// Developer diagnostics (not part of include/smx.h): the lengths of the last large-form tick's slow lists
// (scan facts, scan seeds, control, waypoint rows), after a synchronisation of the device.
extern "C" int smx_debug_slow_counts(smx_handle h, int32_t* out4) {
  if (!h || !out4) return SMX_ERR_INVALID;
  if (!h->slow_blob) return fail(h, SMX_ERR_STATE, "no slow lists (the map is not loaded)");
  SMX_HIP(hipDeviceSynchronize());
  const size_t total = (size_t)h->cfg.num_envs * h->cfg.num_vehicles;
  SMX_HIP(hipMemcpy(out4, h->slow_blob + 4 * total + 4 * (h->alive_parity ^ 1), 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SMX_OK;
}

// developer / tests: use only `records` of k_waypoints_emit's LDS knot pool (0 < records <= SMX_WPE_POOL), so that
// small batches reach its overflow area too (a full pool needs sixteen vehicles in a bend in one workgroup)
extern "C" int smx_debug_set_wp_pool(smx_handle h, int32_t records) {
  if (!h) return SMX_ERR_INVALID;
  if (records < 1 || records > SMX_WPE_POOL) return fail(h, SMX_ERR_INVALID, "smx_debug_set_wp_pool: 1 .. SMX_WPE_POOL records");
  h->wp_pool_limit = records;
  return SMX_OK;
}

extern "C" int smx_set_timing(smx_handle h, int level) {
  if (!h) return SMX_ERR_INVALID;
  h->timing = level == 1;
  h->phase_timing = level == 2;
  return SMX_OK;
}

extern "C" int smx_read_phase_ms(smx_handle h, float* ms, int32_t max_steps, int32_t* n) {
  if (!h || !ms || !n || max_steps < 0) return SMX_ERR_INVALID;
  int32_t count = 0;
  for (size_t i = 0; i < h->ph_used; ++i) {
    hipEvent_t* ph = &h->ph_pool[i * (SMX_PHASE_COUNT + 1)];
    SMX_HIP(hipEventSynchronize(ph[SMX_PHASE_COUNT]));
    if (count >= max_steps) continue;
    for (int p = 0; p < SMX_PHASE_COUNT; ++p) {
      float t = 0.f;
      SMX_HIP(hipEventElapsedTime(&t, ph[p], ph[p + 1]));
      ms[(size_t)count * SMX_PHASE_COUNT + p] = t;
    }
    ++count;
  }
  h->ph_used = 0;
  *n = count;
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

extern "C" const char* smx_last_error(smx_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" void smx_destroy(smx_handle h) {
  if (!h) return;
  if (h->map_blob) (void)hipFree(h->map_blob);
  if (h->knots_blob) (void)hipFree(h->knots_blob);
  if (h->alive_blob) (void)hipFree(h->alive_blob);
  if (h->scan_carry) (void)hipFree(h->scan_carry);
  if (h->slow_blob) (void)hipFree(h->slow_blob);
  if (h->pending_blob) (void)hipFree(h->pending_blob);
  if (h->spill_blob) (void)hipFree(h->spill_blob);
  if (h->ctrl_blob) (void)hipFree(h->ctrl_blob);
  if (h->status_dev) (void)hipFree(h->status_dev);
  if (h->side_ready) {
    for (int i = 0; i < 3; ++i) {
      (void)hipStreamSynchronize(h->side[i]);
      (void)hipStreamDestroy(h->side[i]);
      (void)hipEventDestroy(h->ev_join[i]);
    }
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_fork_grid);
  }
  if (h->vias_dev) (void)hipFree(h->vias_dev);
  if (h->via_off_dev) (void)hipFree(h->via_off_dev);
  if (h->missions_blob) (void)hipFree(h->missions_blob);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ph_pool) (void)hipEventDestroy(e);
  delete h;
}
