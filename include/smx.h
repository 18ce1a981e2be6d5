/*
 * smx.h — C-ABI of libsmarts_mi355x.so: the MI355X-native SMARTS hot path.
 *
 * One handle = one GPU = one shard of E independent environment instances x N
 * vehicle slots.  The boundary replaces the seam
 *     SMARTS.step(agent_actions) -> (obs, rewards, dones, scores)
 *                                   reference smarts/core/smarts.py:187-227, 236-327
 *     SMARTS.reset(scenario)     -> first observations
 *                                   reference smarts/core/smarts.py:365-460
 * for a batch of instances, the way ParallelEnv batches whole processes
 * (reference smarts/env/wrappers/parallel_env.py:214-233, 303-309 auto-reset).
 * The reference has no FFI on this path (it imports pybullet/sumolib directly), so
 * the entry points below are what a cffi/ctypes binding of that seam would bind;
 * INTEGRATION.md shows the binding.
 *
 * Conventions: every function returns 0 on success, a negative smx_status on
 * failure and never throws; smx_last_error() gives the text.  The caller owns
 * every device buffer (PyTorch-ROCm tensors handed over as raw pointers); the
 * library allocates only the map tables.  A handle is not thread-safe; distinct
 * handles are independent.  smx_step / smx_reset enqueue work on the given HIP
 * stream and do no host<->device copies and no synchronisation.
 */
#ifndef SMX_H
#define SMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smx_handle_s* smx_handle;

typedef enum smx_status {
  SMX_OK = 0,
  SMX_ERR_INVALID = -1,  /* bad argument / shape / null pointer */
  SMX_ERR_HIP = -2,      /* a HIP runtime call failed */
  SMX_ERR_STATE = -3,    /* call order (e.g. step before load_map) */
  SMX_ERR_NOMEM = -4
} smx_status;

/* ---- action space: reference smarts/core/controllers/__init__.py:42-54, 137-144 ---- */
enum {
  SMX_ACTION_KEEP_LANE = 0,         /* "keep_lane"         -> target 15.0 m/s, lane change 0  */
  SMX_ACTION_SLOW_DOWN = 1,         /* "slow_down"         -> target  0.0 m/s, lane change 0  */
  SMX_ACTION_CHANGE_LANE_LEFT = 2,  /* "change_lane_left"  -> target 12.5 m/s, lane change +1 */
  SMX_ACTION_CHANGE_LANE_RIGHT = 3, /* "change_lane_right" -> target 12.5 m/s, lane change -1 */
  SMX_ACTION_NONE = -1              /* agent sent no action this tick (smarts.py:1233-1240) */
};

/* ---- done criteria bits: reference smarts/core/agent_interface.py:186-206 ---- */
enum {
  SMX_DONE_COLLISION = 1 << 0,
  SMX_DONE_OFF_ROAD = 1 << 1,
  SMX_DONE_OFF_ROUTE = 1 << 2,
  SMX_DONE_ON_SHOULDER = 1 << 3,
  SMX_DONE_WRONG_WAY = 1 << 4,
  SMX_DONE_NOT_MOVING = 1 << 5
};

/* ---- event columns of smx_outputs.events, order of reference smarts/core/events.py:23-34 ---- */
enum {
  SMX_EV_COLLISIONS = 0,
  SMX_EV_OFF_ROAD = 1,
  SMX_EV_OFF_ROUTE = 2,
  SMX_EV_ON_SHOULDER = 3,
  SMX_EV_WRONG_WAY = 4,
  SMX_EV_NOT_MOVING = 5,
  SMX_EV_REACHED_GOAL = 6,
  SMX_EV_REACHED_MAX_EPISODE_STEPS = 7,
  SMX_EV_AGENTS_ALIVE_DONE = 8,
  SMX_EV_COUNT = 9
};

/* ---- sensor switches: reference smarts/core/agent_interface.py:209-297 ---- */
enum {
  SMX_SENSOR_WAYPOINTS = 1 << 0,
  SMX_SENSOR_NEIGHBORS = 1 << 1,
  SMX_SENSOR_ACCELEROMETER = 1 << 2,
  SMX_SENSOR_OGM = 1 << 3,
  SMX_SENSOR_LIDAR = 1 << 4,
  SMX_SENSOR_DAGM = 1 << 5,  /* drivable-area grid map (sensors.py:675-716) */
  SMX_SENSOR_ROAD_WAYPOINTS = 1 << 6 /* RoadWaypointsSensor (sensors.py:991-1040) */
};

enum { SMX_SOCIAL_CONSTANT = 0, SMX_SOCIAL_IDM = 1 };

/* ActionSpaceType (controllers/__init__.py:42-57) values on this path.  Lane takes int8 codes
 * (smx_step); the others take three floats per agent (smx_step_continuous):
 *   CONTINUOUS                 throttle, brake, steering                  (:94-99)
 *   ACTUATOR_DYNAMIC           throttle, brake, steering rate             (actuator_dynamic_controller.py:47-80)
 *   LANE_WITH_CONTINUOUS_SPEED target speed, lane change (-1 / 0 / +1), - (:113-124)
 * A NaN in the first float means "no action this tick". */
enum {
  SMX_ACTION_SPACE_LANE = 0,
  SMX_ACTION_SPACE_CONTINUOUS = 1,
  SMX_ACTION_SPACE_ACTUATOR_DYNAMIC = 2,
  SMX_ACTION_SPACE_LANE_WITH_CONTINUOUS_SPEED = 3,
  SMX_ACTION_SPACE_TRAJECTORY = 4 /* smx_step_trajectory: PD tracking (trajectory_tracking_controller.py:176-331) */
};

#define SMX_MAX_ALIVE_LISTS 4
typedef struct smx_config {
  int32_t num_envs;          /* E: environment instances in this shard            */
  int32_t num_vehicles;      /* N: vehicle slots per instance (<= 64)             */
  double dt;                 /* fixed_timestep_sec, reference hiway_env.py:113    */
  uint32_t sensors;          /* SMX_SENSOR_* bits                                 */
  uint32_t done_criteria;    /* SMX_DONE_* bits                                   */
  int32_t wp_lookahead;      /* Waypoints.lookahead (agent_interface.py:78), 32   */
  int32_t wp_paths;          /* dense rows kept per agent, format_obs.py:42 -> 4  */
  int32_t wp_len;            /* dense waypoints kept per path, format_obs.py:42 -> 20 */
  int32_t nb_max;            /* dense neighbour rows, format_obs.py:41 -> 10      */
  double nb_radius;          /* NeighborhoodVehicles.radius; < 0 = unlimited      */
  int32_t max_episode_steps; /* <= 0 = None                                       */
  double not_moving_time;    /* EventConfiguration, agent_interface.py:175-183    */
  double not_moving_distance;
  int32_t auto_reset;        /* ParallelEnv auto_reset, parallel_env.py:303-309   */
  int32_t reset_elapsed_steps; /* ticks the reference spends before the first ego
                                  observation exists (smarts.py:426-434)          */
  int32_t ogm_width, ogm_height; /* OGM sensor grid (agent_interface.py:42-51)    */
  double ogm_resolution;
  int32_t lidar_rays;        /* number of rays in smx_lidar_rays                  */
  double lidar_max_distance;
  int32_t action_space;      /* SMX_ACTION_SPACE_*                                */
  /* Scripted social traffic (stands in for the SUMO provider on fixed-vehicle-count scenarios,
   * smarts.py:868-921, chassis.py:187-320 BoxChassis): the LAST num_social slots of every env
   * are kinematic lane followers.  They move along their lane's centre line at
   * social_speed_factor x the lane's speed limit, continue onto outgoing lane
   * (slot + lanes crossed) mod #outgoing, are seen by every sensor and collision test, and
   * produce no observation of their own.  State reuse: SMX_S_MCL_X = lane, SMX_S_MCL_Y =
   * arclength offset, SMX_S_SPD_INT = lanes crossed. */
  int32_t num_social;
  double social_speed_factor;
  /* DoneCriteria.agents_alive (agent_interface.py:155-176, sensors.py:404-441): an agent is done
   * when fewer agents are left in its env than asked.  Counts are taken over the agents registered
   * at the start of the tick (agent_manager ids).  0 = not set; alive_list_mask[k] bit i = agent
   * slot i belongs to list k (agents_list), alive_list_min[k] = minimum_agents_alive_in_list. */
  int32_t via_max;           /* near-via rows kept per agent (<= 32); 0 = via sensor off */
  int32_t alive_min_ego;
  int32_t alive_min_total;
  int32_t alive_lists;          /* number of lists used, <= SMX_MAX_ALIVE_LISTS */
  int32_t alive_list_min[4];
  uint64_t alive_list_mask[4];
  /* drivable-area grid map (agent_interface.py:29-38): same view as the OGM (centred on the ego,
   * row 0 ahead), 255 where the pixel centre lies within half a lane width of a lane centre line */
  int32_t dagm_width, dagm_height;
  double dagm_resolution;
  /* Speed model of the scripted social vehicles: SMX_SOCIAL_CONSTANT = social_speed_factor x the speed
   * limit; SMX_SOCIAL_IDM = intelligent-driver car following towards that desired speed (accel 2.6,
   * decel 4.5, tau 1.0, min gap 2.5: SUMO's passenger defaults).  The leader is the nearest alive
   * vehicle of the env (agents included) less than 60 m ahead along the follower's heading and less
   * than 1.6 m to its side, read at the start of the tick; gap = centre distance - 3.68.  Parity with
   * SUMO's own car-following is unpinned. */
  int32_t social_model;
  /* RoadWaypointsSensor (agent_interface.py RoadWaypoints.horizon, sensors.py:991-1040): per lane of the nearest
   * lane's road, its parallel roads and the roads oncoming at the vehicle, the waypoint paths of lookahead
   * 2 x horizon that start `horizon` metres behind the vehicle (through incoming lanes where the lane is shorter).
   * Dense rows: the first rw_lanes lanes (<= SMX_RW_LANE_CAP) in the reference's order, the first rw_paths paths
   * of each, 2 x horizon + 1 waypoints per path; true counts are reported beside them. */
  int32_t rw_horizon;        /* 1 .. SMX_RW_HORIZON_MAX; read when SMX_SENSOR_ROAD_WAYPOINTS is set */
  int32_t rw_lanes;
  int32_t rw_paths;
} smx_config;
#define SMX_RW_LANE_CAP 8
#define SMX_RW_HORIZON_MAX 64

/* ---- packed map records (smarts_amd.map_compiler.pack_tables) ---- */
typedef struct smx_lp_rec {   /* one lanepoint = one 64-byte line (LanePoint + LinkedLanePoint, lanepoints.py:46-70) */
  double x, y, heading;       /* pose (heading as Pose.heading yields it, coordinates.py:394-403) */
  double dirx, diry;          /* radians_to_vec(heading) (math.py:247-253), host libm */
  int32_t lane;
  int32_t next_off;           /* first successor record in succ_rec */
  int32_t next0;              /* first successor lanepoint, -1 none */
  uint16_t n_next;
  uint8_t inferred;           /* LinkedLanePoint.is_inferred */
  uint8_t flags;              /* bit 0: interpolated points down the chain have consecutive indices */
  int32_t knot_next;          /* next non-inferred lanepoint following successor 0 */
  int32_t knot_hops;          /* hops from here to knot_next */
} smx_lp_rec;
typedef struct smx_succ_rec { /* one entry of LinkedLanePoint.nexts */
  int32_t idx;                /* the successor lanepoint */
  int32_t lane;               /* its lane (route filter of lanepoints.py:666-683) */
  int32_t knot;               /* first non-inferred lanepoint down that branch */
  int32_t hops;               /* hops from the branching point to it */
} smx_succ_rec;
typedef struct smx_shape_rec { /* one centre-line vertex of a lane (Lane shape, sumo_road_network.py:287-296) */
  double x, y;
  double cum;                 /* arclength from the lane's first vertex, summed vertex by vertex */
  double len;                 /* length of the segment to the next vertex (0 on the last one)   */
} smx_shape_rec;
typedef struct smx_pt_rec { double x, y; int32_t idx, lane; } smx_pt_rec;   /* lanepoint grid member */
typedef struct smx_seg_rec {  /* centre-line segment grid member */
  double x1, y1, x2, y2;
  double thr;                 /* 0.5 * lane width + 0.1 (road_with_point, sumo_road_network.py:707) */
  double len, cum;            /* shape_rec[v0].len / .cum: the segment's length and its arclength along the lane */
  int32_t lane;
  int32_t v0;                 /* index of the segment's first vertex in shape_rec (its second one is v0 + 1) */
} smx_seg_rec;

/* Host-side map tables produced by smarts_amd.map_compiler (compile_map + pack_tables).
 * smx_load_map copies them to the device.  Lane order = sumolib _allLanes order. */
typedef struct smx_map_tables {
  int32_t n_lanes, n_roads, n_lanepoints, n_shape_pts, n_succ;
  const int32_t* lane_road;
  const int32_t* lane_index;
  const double* lane_width;
  const double* lane_speed;
  const double* lane_length;     /* the net.xml length attribute (sumo_road_network.py:283-285) */
  const uint8_t* lane_in_junction;
  const int32_t* lane_shape_off; /* n_lanes + 1 */
  const double* shape_x;
  const double* shape_y;
  const smx_shape_rec* shape_rec; /* n_shape_pts, same order as shape_x / shape_y */
  const int32_t* lane_out_off;   /* n_lanes + 1 */
  const int32_t* lane_out_idx;   /* Lane.outgoing_lanes (sumo_road_network.py:350-358) */
  const int32_t* road_lane_off;  /* n_roads + 1 */
  const int32_t* road_lanes;
  const uint8_t* road_is_junction;
  const int32_t* road_out_road;
  const smx_lp_rec* lp_rec;      /* n_lanepoints, the reference's global lanepoint order */
  const smx_succ_rec* succ_rec;  /* n_succ */
  double lpg_x0, lpg_y0, lpg_cell; /* uniform grid over lanepoints (replaces scipy KD-trees) */
  int32_t lpg_nx, lpg_ny;
  const int32_t* lpg_off;        /* lpg_nx * lpg_ny + 1 */
  const smx_pt_rec* lpg_pts;     /* members by value, contiguous per cell */
  double sg_x0, sg_y0, sg_cell;  /* uniform grid over centre-line segments (replaces the rtree) */
  int32_t sg_nx, sg_ny;
  const int32_t* sg_off;         /* sg_nx * sg_ny + 1 */
  const smx_seg_rec* sg_rec;
  double default_lane_width;     /* sumo_road_network.py:80 */
  /* RoadWaypointsSensor's neighbourhood (sensors.py:991-1040) */
  const int32_t* lane_in_off;    /* n_lanes + 1 */
  const int32_t* lane_in_idx;    /* Lane.incoming_lanes (sumo_road_network.py:342-348), sumolib's order */
  const int32_t* road_par_off;   /* n_roads + 1 */
  const int32_t* road_par_idx;   /* Road.parallel_roads (sumo_road_network.py:607-618) */
} smx_map_tables;

/* ---- simulation state, caller-owned device memory, struct-of-arrays over (E, N) ---- */
enum {
  SMX_S_X = 0, SMX_S_Y, SMX_S_HEADING,   /* pose of the base frame (chassis.py:493-505) */
  SMX_S_U, SMX_S_V, SMX_S_R,             /* body-frame velocity + yaw rate               */
  SMX_S_DELTA,                            /* steer joint angle (chassis.py:510-525)       */
  SMX_S_LAT_INT, SMX_S_SPD_INT, SMX_S_STEER, SMX_S_THROTTLE, SMX_S_SPD_ERR,
  SMX_S_MCL_X, SMX_S_MCL_Y,              /* LaneFollowingControllerState (lane_following_controller.py:34-49) */
  SMX_S_TRIP_X, SMX_S_TRIP_Y, SMX_S_TRIP_H, SMX_S_DIST, /* TripMeterSensor (sensors.py:880-947) */
  SMX_S_LV0_LONG, SMX_S_LV0_LAT, SMX_S_AV0_Z,           /* AccelerometerSensor history (sensors.py:1046-1087) */
  SMX_S_LV1_LONG, SMX_S_LV1_LAT, SMX_S_AV1_Z,
  SMX_S_PATH_SUM,                        /* DrivenPathSensor running window length (sensors.py:855-877) */
  SMX_S_PREV_X, SMX_S_PREV_Y,            /* position at the previous observation (driven-path segment) */
  SMX_S_COUNT
};
enum {
  SMX_F_ALIVE = 1 << 0,
  SMX_F_MCL_SET = 1 << 1,
  SMX_F_RESERVED2 = 1 << 2, /* (was the trip-meter bit; now facts_i32[SMX_FI_TRIP_HAS_WP]) */
  SMX_F_HIST_SHIFT = 3, /* bits 3-4: accelerometer samples held (0..2) */
  SMX_F_FIRST = 1 << 5, /* vehicle was just (re)created: its next observation is a reset observation */
  SMX_F_SOCIAL = 1 << 6 /* scripted social vehicle (no controller, no observation) */
};

#define SMX_DRIVEN_PATH_LEN 500 /* DrivenPathSensor deque, sensors.py:838 */

/* Element types of the caller-owned buffers (the dtype the caller declares for each pointer). */
enum { SMX_DT_NONE = 0, SMX_DT_F64, SMX_DT_F32, SMX_DT_I32, SMX_DT_I16, SMX_DT_I8, SMX_DT_U8, SMX_DT_U64 };
enum { /* indices into smx_state.count / .dtype: the pointers in declaration order */
  SMX_ST_F64 = 0, SMX_ST_FLAGS, SMX_ST_STEPS, SMX_ST_ENV_TICKS, SMX_ST_ENV_DONE_COUNT, SMX_ST_ENV_EPISODE,
  SMX_ST_DRIVEN_PATH, SMX_ST_SEED_CACHE, SMX_ST_FACTS_I32, SMX_ST_FACTS_F64, SMX_ST_ENV_RESET_PENDING,
  SMX_ST_BUFFERS
};
typedef struct smx_state {
  double* f64;        /* [SMX_S_COUNT][E*N]                                    */
  int32_t* flags;     /* [E*N]  SMX_F_* bits                                   */
  int32_t* steps;     /* [E*N]  SensorState._step (sensors.py:609-637)         */
  int32_t* env_ticks; /* [E]    ticks since reset (elapsed_sim_time = ticks*dt)*/
  int32_t* env_done_count; /* [E] agents that have ever been done (hiway_env.py:258-261) */
  int32_t* env_episode;    /* [E] episodes completed (indexes the spawn table) */
  double* driven_path; /* [E*N][SMX_DRIVEN_PATH_LEN] ring of step lengths, or NULL
                         when SMX_DONE_NOT_MOVING tracking is not wanted        */
  int32_t* seed_cache; /* [SMX_SEED_COUNT][E*N]: start road / route filter / start lanepoints
                          found by the last observation at the vehicle's current pose; the
                          next tick's controller queries the map at that same pose
                          (lane_following_controller.py:96-98) and reuses them     */
  int32_t* facts_i32;  /* [SMX_FACT_I_COUNT][E*N] per-tick map facts of each vehicle (scan kernel
                          -> observe kernels): nearest lane, road flags, trip-meter seed   */
  double* facts_f64;   /* [SMX_FACT_F_COUNT][E*N]: nearest-lane distance, lane heading there  */
  int32_t* env_reset_pending; /* [E] set by the observe kernel when auto_reset fires        */
  /* what the caller allocated: element count and SMX_DT_* of each buffer above, in declaration order
   * (SMX_ST_*); checked against the sizes smx_config implies on every entry (smx_check_buffers) */
  uint64_t count[SMX_ST_BUFFERS];
  uint8_t dtype[SMX_ST_BUFFERS + 5]; /* (+5: keeps the struct a multiple of 8 bytes) */
} smx_state;
#define SMX_SEED_COUNT 9 /* road, filter n, filter roads x2, lane count, start lanepoint x4 */
enum {
  SMX_FI_LANE = 0, SMX_FI_FLAGS, SMX_FI_TRIP_START, SMX_FI_OBS_START,
  SMX_FI_TRIP_HAS_WP, /* trip meter holds a waypoint (TripMeterSensor._wps_for_distance non-empty); persists across ticks */
  SMX_FI_FLAGS_NEXT,  /* the flags word after this tick's observation, applied by the commit kernel */
  SMX_FI_VIA_CONSUMED, /* bit v: via v of the agent's list was hit in this episode (ViaSensor._consumed_via_points) */
  SMX_FACT_I_COUNT
};
enum { SMX_FF_LANE_DIST = 0, SMX_FF_LANE_HEADING /* lane heading at the nearest centre-line point */, SMX_FACT_F_COUNT };
enum { SMX_FACT_ON_ROAD = 1 << 0, SMX_FACT_CORNER_SHIFT = 1 /* bits 1-4: corner q on road */ };

/* Spawn table: episode k of env e starts from row (k mod episodes).
 * x, y = vehicle centre, heading in reference convention, speed m/s. */
typedef struct smx_spawns {
  int32_t episodes;
  const double* pose; /* device, [episodes][E*N][4] = x, y, heading, speed */
  const double* social; /* device, [episodes][E*N][2] = lane, arclength offset (social slots only; NULL if none) */
  uint64_t pose_count, social_count; /* float64 elements the caller allocated for each table */
} smx_spawns;

/* ---- per-tick outputs, caller-owned device memory, dense StdObs layout
 *      (reference smarts/env/wrappers/format_obs.py:40-42, 313-373, 401-603) ---- */
enum { /* columns of smx_outputs.ego_f32 */
  SMX_EGO_HEADING = 0, SMX_EGO_SPEED, SMX_EGO_STEERING, SMX_EGO_YAW_RATE,
  SMX_EGO_LIN_VEL = 4,  /* 3 */
  SMX_EGO_ANG_VEL = 7,  /* 3 */
  SMX_EGO_LIN_ACC = 10, /* 3 */
  SMX_EGO_ANG_ACC = 13, /* 3 */
  SMX_EGO_LIN_JERK = 16,/* 3 */
  SMX_EGO_ANG_JERK = 19,/* 3 */
  SMX_EGO_BOX = 22,     /* 3: length, width, height */
  SMX_EGO_F32_COUNT = 25
};

enum { /* indices into smx_outputs.count / .dtype: the pointers in declaration order */
  SMX_OUT_EGO_POS = 0, SMX_OUT_EGO_F32, SMX_OUT_EGO_LANE, SMX_OUT_EVENTS, SMX_OUT_REWARD, SMX_OUT_DIST, SMX_OUT_DONE,
  SMX_OUT_ACTIVE, SMX_OUT_ENV_DONE, SMX_OUT_VIA_NEAR, SMX_OUT_VIA_NEAR_COUNT, SMX_OUT_VIA_HIT, SMX_OUT_LEARNER,
  SMX_OUT_WP_POS, SMX_OUT_WP_HEADING, SMX_OUT_WP_LANE_WIDTH, SMX_OUT_WP_SPEED_LIMIT, SMX_OUT_WP_LANE_INDEX,
  SMX_OUT_WP_LANE_ID, SMX_OUT_WP_COUNT, SMX_OUT_NB_POS, SMX_OUT_NB_BOX, SMX_OUT_NB_HEADING, SMX_OUT_NB_SPEED,
  SMX_OUT_NB_LANE_INDEX, SMX_OUT_NB_LANE_ID, SMX_OUT_NB_SLOT, SMX_OUT_NB_COUNT, SMX_OUT_OGM, SMX_OUT_LIDAR_HIT,
  SMX_OUT_LIDAR_POINT, SMX_OUT_DAGM, SMX_OUT_COLLIDEES,
  SMX_OUT_RW_LANE_COUNT, SMX_OUT_RW_LANE, SMX_OUT_RW_PATH_COUNT, SMX_OUT_RW_COUNT, SMX_OUT_RW_POS, SMX_OUT_RW_HEADING,
  SMX_OUT_RW_LANE_WIDTH, SMX_OUT_RW_SPEED_LIMIT, SMX_OUT_RW_LANE_INDEX, SMX_OUT_RW_LANE_ID,
  SMX_OUT_FINAL_EGO_POS, SMX_OUT_FINAL_EGO_F32, SMX_OUT_FINAL_EGO_LANE, SMX_OUT_FINAL_EVENTS, SMX_OUT_FINAL_DIST,
  SMX_OUT_BUFFERS
};
typedef struct smx_outputs {
  double* ego_pos;       /* [E*N][3]                                          */
  float* ego_f32;        /* [E*N][SMX_EGO_F32_COUNT]                          */
  int16_t* ego_lane;     /* [E*N][2] = lane id (table index, -1 none), lane_index */
  uint8_t* events;       /* [E*N][SMX_EV_COUNT]                               */
  double* reward;        /* [E*N] trip-meter increment (agent_manager.py:233) */
  double* dist;          /* [E*N] distance_travelled / score                  */
  uint8_t* done;         /* [E*N]                                             */
  uint8_t* active;       /* [E*N] 1 while the agent has a vehicle after this tick */
  uint8_t* env_done;     /* [E]   dones["__all__"] (hiway_env.py:258-261)     */
  /* via sensor (if smx_set_vias gave any): the near vias (within the 40 m lane acquisition range,
   * vehicle.py:553-557) as indices into the agent's list, nearest first, -1 padded; bit v of
   * via_hit = via v was hit this tick */
  int8_t* via_near;      /* [E*N][via_max]                                    */
  uint8_t* via_near_count; /* [E*N]                                           */
  int32_t* via_hit;      /* [E*N]                                             */
  /* optional learner-facing block [2][E*N] float32: row 0 = reward, row 1 = done, rewritten whole
   * every tick (absent agents read 0) — what a multi-GPU job gathers per tick;
   * the caller may alternate buffers between ticks.  NULL if unused. */
  float* learner;
  /* waypoints sensor, [E*N][wp_paths][wp_len] */
  double* wp_pos;        /* ...[3], z = 0 (format_obs.py:594)                 */
  float* wp_heading;
  float* wp_lane_width;
  float* wp_speed_limit;
  int8_t* wp_lane_index;
  int16_t* wp_lane_id;   /* extra: lane table index of each waypoint          */
  uint8_t* wp_count;     /* [E*N][wp_paths + 1]: total #paths, then #waypoints per kept path */
  /* neighbourhood sensor, [E*N][nb_max] */
  double* nb_pos;        /* ...[3]                                            */
  float* nb_box;         /* ...[3]                                            */
  float* nb_heading;
  float* nb_speed;
  int8_t* nb_lane_index; /* -1 when no lane within the observer's length      */
  int16_t* nb_lane_id;
  int8_t* nb_slot;       /* vehicle slot of each neighbour, -1 = padding      */
  uint8_t* nb_count;     /* [E*N] neighbours found (may exceed nb_max)        */
  /* occupancy grid sensor [E*N][ogm_height][ogm_width], NULL if unused       */
  uint8_t* ogm;
  /* lidar sensor [E*N][lidar_rays]: hit flag + point, NULL if unused         */
  uint8_t* lidar_hit;
  double* lidar_point;   /* ...[3]                                            */
  /* drivable-area grid map [E*N][dagm_height][dagm_width], NULL if unused     */
  uint8_t* dagm;
  /* collisions (smarts.py:1270-1291, sensors.py:206-211): bit j = the agent's chassis touches the vehicle in
   * slot j of its env this tick — one Collision(collidee_id) per set bit; events[SMX_EV_COLLISIONS] = any bit */
  uint64_t* collidees;   /* [E*N]                                             */
  /* road waypoints (RoadWaypoints.lanes, sensors.py:999-1012), NULL if unused; L = rw_lanes, P = rw_paths,
   * R = 2 * rw_horizon + 1.  Rows beyond a count are not written. */
  uint8_t* rw_lane_count;   /* [E*N]          lanes the sensor reports (may exceed L)              */
  int16_t* rw_lane;         /* [E*N][L]       their lane ids in the reference's dict order, -1 none */
  int16_t* rw_path_count;   /* [E*N][L]       paths of that lane (may exceed P; saturates at 32767)  */
  uint8_t* rw_count;        /* [E*N][L][P]    waypoints of the kept path (0 = no such path)         */
  double* rw_pos;           /* [E*N][L][P][R][3]                                                    */
  float* rw_heading;        /* [E*N][L][P][R]                                                       */
  float* rw_lane_width;
  float* rw_speed_limit;
  int8_t* rw_lane_index;
  int16_t* rw_lane_id;
  /* auto_reset only, all five or none (NULL): the low-dimensional rows of the FINISHING tick of an env that restarts
   * inside the launch — the observation the reference hands back as info[agent]["env_obs"]
   * (smarts/env/wrappers/parallel_env.py:303-309, smarts/env/hiway_env.py:243-246) — copied by k_commit before the
   * reset pass overwrites ego_pos / ego_f32 / ego_lane / events / dist with the next episode's first observation.  Written
   * only for the slots of an env whose env_done is raised this tick; same layouts as their namesakes. */
  double* final_ego_pos;
  float* final_ego_f32;
  int16_t* final_ego_lane;
  uint8_t* final_events;
  double* final_dist;
  /* what the caller allocated: element count and SMX_DT_* of each buffer above, in declaration order
   * (SMX_OUT_*); 0 / SMX_DT_NONE for a NULL pointer */
  uint64_t count[SMX_OUT_BUFFERS];
  uint8_t dtype[(SMX_OUT_BUFFERS + 7) & ~7]; /* (rounded up: keeps the struct a multiple of 8 bytes) */
} smx_outputs;

/* ---- entry points ---- */
/* On failure no handle is left behind (*out = NULL); smx_last_error(NULL) then gives the reason. */
int smx_create(const smx_config* cfg, int device, smx_handle* out);
/* The entry check of smx_reset / smx_step*, callable on its own and without a device: every buffer the
 * configuration needs is non-NULL, declared with the expected SMX_DT_* and at least as many elements as
 * the configuration implies (a short buffer would be an out-of-bounds device write).  `has_vias`: vias
 * were given (smx_set_vias).  Returns SMX_OK or SMX_ERR_INVALID with the reason in err[err_len]. */
int smx_check_buffers(const smx_config* cfg, int has_vias, const smx_state* st, const smx_spawns* sp,
                      const smx_outputs* out, char* err, uint64_t err_len);
int smx_load_map(smx_handle h, const smx_map_tables* map);
/* Via points of the agents' missions (plan.py:180-188; ViaSensor sensors.py:1090-1149).  One list per
 * agent slot, shared by every env: vias[slot_off[s] .. slot_off[s+1]) belong to slot s (at most 32
 * each).  Host pointers; the library keeps a device copy.  n = 0 clears. */
typedef struct smx_via {
  double x, y;            /* Via.position */
  double hit_distance;
  double required_speed;
  int32_t lane;           /* Via.lane_id as a lane table index */
  int32_t pad;
} smx_via;
int smx_set_vias(smx_handle h, const smx_via* vias_host, int32_t n, const int32_t* slot_off_host);
/* Missions of the agent slots (plan.py:196-222, 316-349), shared by every env like the vias: n_slots = 0
 * clears, else n_slots = cfg.num_vehicles.  route_len = 0: endless mission with an empty route (plan.py:321-323,
 * what every slot has until this is called).  Otherwise a fixed route: roads[route_off .. route_off + route_len)
 * are road table indices in route order (RoadMap.Route.roads as sumo_road_network.py:711-765 generates them:
 * junction-internal roads included) and the goal is a PositionalGoal (plan.py:86-120).  With a fixed route
 * the waypoint paths of the controller and the waypoints sensor follow the route (sumo_road_network.py:822-829,
 * 862-882; lanepoints.py:666-683), off_route / reached_goal are live (sensors.py:491-496, 527-578) and the trip
 * meter counts only waypoints on the route (sensors.py:908-913).  Host pointers; the library keeps a device
 * copy.  Waits for the device; not to be called between smx_step and the use of its outputs. */
typedef struct smx_mission {
  double goal_x, goal_y, goal_radius;
  int32_t route_off, route_len;
} smx_mission;
int smx_set_missions(smx_handle h, const smx_mission* missions_host, int32_t n_slots, const int32_t* route_roads_host,
                     int32_t n_route_roads);
/* Base ray directions (device, [lidar_rays][3]); reference lidar.py:89-113 */
int smx_set_lidar_rays(smx_handle h, const double* rays_dev, int32_t n_rays);
/* Re-initialise the envs whose mask byte is non-zero (NULL = all) from the spawn
 * table and produce their first observations. */
int smx_reset(smx_handle h, const uint8_t* env_mask_dev, const smx_state* st, const smx_spawns* sp,
              const smx_outputs* out, void* hip_stream);
/* One tick for every env: actions[E*N] are SMX_ACTION_* (int8, device). */
int smx_step(smx_handle h, const int8_t* actions_dev, const smx_state* st, const smx_spawns* sp,
             const smx_outputs* out, void* hip_stream);
/* The same tick for the float action spaces: actions[E*N][3] (float32, device). */
int smx_step_continuous(smx_handle h, const float* actions_dev, const smx_state* st, const smx_spawns* sp,
                        const smx_outputs* out, void* hip_stream);
/* The same tick for ActionSpaceType.Trajectory.  trajectories[E*N][4][SMX_TRAJ_COLS] (float64, device):
 * rows x, y, heading, speed; columns 0..9 = the first ten points, column 10 = the LAST point of the
 * trajectory — all the PD controller reads; counts[E*N] (int32, device) = the trajectory's true
 * length, 0 = no action this tick. */
#define SMX_TRAJ_COLS 11
int smx_step_trajectory(smx_handle h, const double* trajectories_dev, const int32_t* counts_dev, const smx_state* st,
                        const smx_spawns* sp, const smx_outputs* out, void* hip_stream);
/* Waits for the stream, then reports what only the kernels could see: SMX_ERR_INVALID if a Lane action code
 * outside -1..3 was met since the last smx_sync (such an agent is stepped as if it had sent no action). */
int smx_sync(smx_handle h, void* hip_stream);
/* Device-side timing: while enabled, every smx_step is bracketed by a hipEvent pair recorded on
 * the stream it is launched on (no synchronisation).  smx_read_step_ms waits for the recorded
 * launches, writes their durations (milliseconds, oldest first, at most `max`) and clears the
 * log; *n receives the count.  smx_last_step_ms is the single-launch convenience form.
 *
 * Levels: 0 = off; 1 = one event pair around the whole smx_step (the bench's timed region uses
 * this); 2 = a boundary event after every kernel of the tick, read back per phase with
 * smx_read_phase_ms as ms[n][SMX_PHASE_COUNT] (a phase whose sensor is disabled reads ~0). */
enum {
  SMX_PHASE_CONTROL = 0, /* k_control: controllers + vehicle dynamics (a1-a6)            */
  SMX_PHASE_SCAN,        /* k_scan: road facts + lanepoint seeds (a8, a9 front half)      */
  SMX_PHASE_OGM,         /* k_ogm (a14) and k_dagm                                        */
  SMX_PHASE_SENSORS,     /* k_sensors: waypoints | observe | lidar roles (a7, a9-a13, a15) */
  SMX_PHASE_COMMIT,      /* k_commit: flags, env done count, auto-reset respawn           */
  SMX_PHASE_RESET,       /* auto-reset pass (parallel_env.py:303-309), all kernels        */
  SMX_PHASE_COUNT
};
int smx_set_timing(smx_handle h, int level);
int smx_read_phase_ms(smx_handle h, float* ms, int32_t max_steps, int32_t* n);
int smx_read_step_ms(smx_handle h, float* ms, int32_t max, int32_t* n);
int smx_last_step_ms(smx_handle h, float* ms);
const char* smx_last_error(smx_handle h);
/* sizeof() of the ABI structs as compiled (0 config, 1 map tables, 2 state, 3 spawns, 4 outputs):
 * lets a foreign-language binding verify its mirror of the layouts. */
uint64_t smx_struct_size(int which);
/* Lateral gains of the lane-following controller for target_speed > 0
 * (lane_following_controller.py:420-430); defaults are the sedan's clip bounds (0.04, 3.4). */
int smx_set_controller_gains(smx_handle h, double heading_gain, double lateral_gain);
/* How a tick is cut into launches.  The same role functions run either way and the results are the same;
 * the forms differ in what bounds them.  SMALL: few launches whose workgroups take different roles (a batch
 * that cannot fill the chip is bound by one wavefront's latency); LARGE: one launch per role, waypoint rows
 * emitted in memory order from LDS knot tables (bound by throughput).  AUTO picks by vehicle count. */
enum { SMX_LAUNCH_AUTO = 0, SMX_LAUNCH_SMALL = 1, SMX_LAUNCH_LARGE = 2,
       SMX_LAUNCH_LARGE_ONE_LANE = 3, SMX_LAUNCH_LARGE_TEAMS = 4 /* a cut of LARGE forced, whatever the map and the size (smx_launch_form) */ };
int smx_set_launch_strategy(smx_handle h, int strategy);
/* The form the next tick will run in (after smx_load_map): the LARGE form comes in two cuts — one lane per vehicle
 * seeded by last tick's answers with the rare cases on device-side slow lists, where the map's lanes never split
 * (no lanepoint with several successors: loop), and teams of lanes per vehicle for everybody where they do
 * (intersections, minicity: a third of the vehicles would be "rare cases").  Inside the one-lane cut the path-seeds
 * search is the one-lane kernel from 114 688 vehicles on and the team kernel below (its slow chain's latency would
 * end the tick of a smaller batch).  Same results either way. */
enum { SMX_FORM_SMALL = 0, SMX_FORM_LARGE_TEAMS = 1, SMX_FORM_LARGE_ONE_LANE = 2 };
int smx_launch_form(smx_handle h);
const char* smx_version(void);
void smx_destroy(smx_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SMX_H */
