"""ctypes binding of include/smx.h (libsmarts_mi355x.so).

There is no CPU fallback: if the HIP extension is missing or does not load, every
use of the product path raises.  (The CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import LIB_PATH

# ---- enums (include/smx.h) ----
ACTION_KEEP_LANE, ACTION_SLOW_DOWN, ACTION_CHANGE_LANE_LEFT, ACTION_CHANGE_LANE_RIGHT = 0, 1, 2, 3
ACTION_NONE = -1
DONE_COLLISION, DONE_OFF_ROAD, DONE_OFF_ROUTE = 1, 2, 4
DONE_ON_SHOULDER, DONE_WRONG_WAY, DONE_NOT_MOVING = 8, 16, 32
(EV_COLLISIONS, EV_OFF_ROAD, EV_OFF_ROUTE, EV_ON_SHOULDER, EV_WRONG_WAY, EV_NOT_MOVING, EV_REACHED_GOAL,
 EV_REACHED_MAX_EPISODE_STEPS, EV_AGENTS_ALIVE_DONE, EV_COUNT) = range(10)
EVENT_NAMES = ["collisions", "off_road", "off_route", "on_shoulder", "wrong_way", "not_moving", "reached_goal",
               "reached_max_episode_steps", "agents_alive_done"]
EV = {name.upper(): i for i, name in enumerate(EVENT_NAMES)}
ACTION_SPACES = {"Lane": 0, "Continuous": 1, "ActuatorDynamic": 2, "LaneWithContinuousSpeed": 3, "Trajectory": 4}
TRAJ_COLS = 11
SOCIAL_MODELS = {"constant": 0, "idm": 1}
PHASES = ["control", "scan", "ogm", "sensors", "commit", "reset"]
SENSOR_WAYPOINTS, SENSOR_NEIGHBORS, SENSOR_ACCELEROMETER, SENSOR_OGM, SENSOR_LIDAR, SENSOR_DAGM = 1, 2, 4, 8, 16, 32
SENSOR_ROAD_WAYPOINTS = 64
STATE_FIELDS = ["X", "Y", "HEADING", "U", "V", "R", "DELTA", "LAT_INT", "SPD_INT", "STEER", "THROTTLE", "SPD_ERR",
                "MCL_X", "MCL_Y", "TRIP_X", "TRIP_Y", "TRIP_H", "DIST", "LV0_LONG", "LV0_LAT", "AV0_Z", "LV1_LONG",
                "LV1_LAT", "AV1_Z", "PATH_SUM", "PREV_X", "PREV_Y"]
S = {name: i for i, name in enumerate(STATE_FIELDS)}
S_COUNT = len(STATE_FIELDS)
F_ALIVE, F_MCL_SET, F_TRIP_HAS_WP, F_HIST_SHIFT, F_FIRST, F_SOCIAL = 1, 2, 4, 3, 32, 64
FACT_I_COUNT, FACT_F_COUNT = 7, 2
DRIVEN_PATH_LEN = 500
SEED_COUNT = 9
EGO = dict(HEADING=0, SPEED=1, STEERING=2, YAW_RATE=3, LIN_VEL=4, ANG_VEL=7, LIN_ACC=10, ANG_ACC=13, LIN_JERK=16,
           ANG_JERK=19, BOX=22)
EGO_F32_COUNT = 25

_p = C.c_void_p
_i32 = C.c_int32
_f64 = C.c_double


class SmxConfig(C.Structure):
    _fields_ = [
        ("num_envs", _i32), ("num_vehicles", _i32), ("dt", _f64), ("sensors", C.c_uint32),
        ("done_criteria", C.c_uint32), ("wp_lookahead", _i32), ("wp_paths", _i32), ("wp_len", _i32),
        ("nb_max", _i32), ("nb_radius", _f64), ("max_episode_steps", _i32), ("not_moving_time", _f64),
        ("not_moving_distance", _f64), ("auto_reset", _i32), ("reset_elapsed_steps", _i32),
        ("ogm_width", _i32), ("ogm_height", _i32), ("ogm_resolution", _f64), ("lidar_rays", _i32),
        ("lidar_max_distance", _f64), ("action_space", _i32), ("num_social", _i32), ("social_speed_factor", _f64), ("via_max", _i32),
        ("alive_min_ego", _i32), ("alive_min_total", _i32), ("alive_lists", _i32), ("alive_list_min", _i32 * 4),
        ("alive_list_mask", C.c_uint64 * 4),
        ("dagm_width", _i32), ("dagm_height", _i32), ("dagm_resolution", _f64),
        ("social_model", _i32), ("rw_horizon", _i32), ("rw_lanes", _i32), ("rw_paths", _i32),
    ]


class SmxMapTables(C.Structure):
    _fields_ = [
        ("n_lanes", _i32), ("n_roads", _i32), ("n_lanepoints", _i32), ("n_shape_pts", _i32), ("n_succ", _i32),
        ("lane_road", _p), ("lane_index", _p), ("lane_width", _p), ("lane_speed", _p), ("lane_length", _p),
        ("lane_in_junction", _p), ("lane_shape_off", _p), ("shape_x", _p), ("shape_y", _p), ("shape_rec", _p),
        ("lane_out_off", _p), ("lane_out_idx", _p), ("road_lane_off", _p), ("road_lanes", _p),
        ("road_is_junction", _p), ("road_out_road", _p),
        ("lp_rec", _p), ("succ_rec", _p),
        ("lpg_x0", _f64), ("lpg_y0", _f64), ("lpg_cell", _f64), ("lpg_nx", _i32), ("lpg_ny", _i32),
        ("lpg_off", _p), ("lpg_pts", _p),
        ("sg_x0", _f64), ("sg_y0", _f64), ("sg_cell", _f64), ("sg_nx", _i32), ("sg_ny", _i32),
        ("sg_off", _p), ("sg_rec", _p), ("default_lane_width", _f64),
        ("lane_in_off", _p), ("lane_in_idx", _p), ("road_par_off", _p), ("road_par_idx", _p),
    ]


# SMX_DT_* (include/smx.h): the dtype the caller declares for each buffer
DT_NONE, DT_F64, DT_F32, DT_I32, DT_I16, DT_I8, DT_U8, DT_U64 = range(8)
STATE_BUFFERS = ["f64", "flags", "steps", "env_ticks", "env_done_count", "env_episode", "driven_path", "seed_cache",
                 "facts_i32", "facts_f64", "env_reset_pending"]


class SmxState(C.Structure):
    _fields_ = [(name, _p) for name in STATE_BUFFERS] + [
        ("count", C.c_uint64 * len(STATE_BUFFERS)), ("dtype", C.c_uint8 * (len(STATE_BUFFERS) + 5))]


class SmxVia(C.Structure):
    _fields_ = [("x", _f64), ("y", _f64), ("hit_distance", _f64), ("required_speed", _f64), ("lane", _i32), ("pad", _i32)]


class SmxMission(C.Structure):
    _fields_ = [("goal_x", _f64), ("goal_y", _f64), ("goal_radius", _f64), ("route_off", _i32), ("route_len", _i32)]


class SmxSpawns(C.Structure):
    _fields_ = [("episodes", _i32), ("pose", _p), ("social", _p), ("pose_count", C.c_uint64), ("social_count", C.c_uint64)]


OUTPUT_FIELDS = [
    "ego_pos", "ego_f32", "ego_lane", "events", "reward", "dist", "done", "active", "env_done", "learner",
    "wp_pos", "wp_heading", "wp_lane_width", "wp_speed_limit", "wp_lane_index", "wp_lane_id", "wp_count",
    "nb_pos", "nb_box", "nb_heading", "nb_speed", "nb_lane_index", "nb_lane_id", "nb_slot", "nb_count",
    "ogm", "lidar_hit", "lidar_point", "dagm", "collidees",
    "rw_lane_count", "rw_lane", "rw_path_count", "rw_count", "rw_pos", "rw_heading", "rw_lane_width", "rw_speed_limit",
    "rw_lane_index", "rw_lane_id",
    "final_ego_pos", "final_ego_f32", "final_ego_lane", "final_events", "final_dist",
]
OUTPUT_FIELDS.insert(OUTPUT_FIELDS.index("learner"), "via_hit")
OUTPUT_FIELDS.insert(OUTPUT_FIELDS.index("via_hit"), "via_near_count")
OUTPUT_FIELDS.insert(OUTPUT_FIELDS.index("via_near_count"), "via_near")


class SmxOutputs(C.Structure):
    _fields_ = [(name, _p) for name in OUTPUT_FIELDS] + [
        ("count", C.c_uint64 * len(OUTPUT_FIELDS)), ("dtype", C.c_uint8 * ((len(OUTPUT_FIELDS) + 7) & ~7))]


def torch_dtype_code(t) -> int:
    """SMX_DT_* of a torch tensor (the dtype enum checked on entry, include/smx.h)."""
    import torch

    return {torch.float64: DT_F64, torch.float32: DT_F32, torch.int32: DT_I32, torch.int16: DT_I16, torch.int8: DT_I8,
            torch.uint8: DT_U8, torch.int64: DT_U64, torch.uint64: DT_U64}[t.dtype]


def bind_buffer(struct, names, name, tensor):
    """Point field `name` of an SmxState / SmxOutputs at `tensor` (None = NULL) and declare its extent."""
    k = names.index(name)
    setattr(struct, name, tensor.data_ptr() if tensor is not None else None)
    struct.count[k] = int(tensor.numel()) if tensor is not None else 0
    struct.dtype[k] = torch_dtype_code(tensor) if tensor is not None else DT_NONE


EXPORTS = [
    "smx_create", "smx_load_map", "smx_set_vias", "smx_set_missions", "smx_step_continuous", "smx_step_trajectory", "smx_read_phase_ms", "smx_set_lidar_rays", "smx_reset", "smx_step", "smx_sync", "smx_last_step_ms",
    "smx_set_timing", "smx_last_error", "smx_version", "smx_destroy", "smx_set_controller_gains", "smx_struct_size", "smx_read_step_ms",
    "smx_check_buffers", "smx_set_launch_strategy", "smx_launch_form",
]
LAUNCH_FORMS = {0: "small", 1: "large_teams", 2: "large_one_lane"}
LAUNCH_STRATEGIES = {"auto": 0, "small": 1, "large": 2, "large_one_lane": 3, "large_teams": 4}
OGM_ENV_MIN_VEHICLES = 8192  # small form: OGM tiles by k_ogm_env from this many vehicles on (SMX_OGM_ENV_MIN_VEHICLES)
LARGE_BATCH_VEHICLES = 16384  # SMX_LAUNCH_AUTO: the LARGE form above this many vehicles (smx_kernels.hip)

_lib: Optional[C.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load the HIP extension; raises NativeLibraryError (never falls back)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("SMX_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} not found: build it with `python -m smarts_amd.build` (hipcc --offload-arch=gfx950); "
            "smarts_amd has no CPU fallback"
        )
    # PyTorch-ROCm ships its own HIP runtime; the device buffers come from it, so it has to be the
    # one this library binds to.  Loading torch first makes the dynamic linker resolve
    # libamdhip64 to torch's copy (loading this library first would bring up a second runtime that
    # sees no device).
    import torch  # noqa: F401

    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the box
        raise NativeLibraryError(f"cannot load {path}: {e}") from e
    h = C.c_void_p
    lib.smx_create.argtypes = [C.POINTER(SmxConfig), C.c_int, C.POINTER(h)]
    lib.smx_create.restype = C.c_int
    lib.smx_load_map.argtypes = [h, C.POINTER(SmxMapTables)]
    lib.smx_load_map.restype = C.c_int
    lib.smx_set_lidar_rays.argtypes = [h, _p, _i32]
    lib.smx_set_lidar_rays.restype = C.c_int
    lib.smx_reset.argtypes = [h, _p, C.POINTER(SmxState), C.POINTER(SmxSpawns), C.POINTER(SmxOutputs), _p]
    lib.smx_reset.restype = C.c_int
    lib.smx_step.argtypes = [h, _p, C.POINTER(SmxState), C.POINTER(SmxSpawns), C.POINTER(SmxOutputs), _p]
    lib.smx_step.restype = C.c_int
    lib.smx_sync.argtypes = [h, _p]
    lib.smx_sync.restype = C.c_int
    lib.smx_last_step_ms.argtypes = [h, C.POINTER(C.c_float)]
    lib.smx_last_step_ms.restype = C.c_int
    lib.smx_read_step_ms.argtypes = [h, C.POINTER(C.c_float), _i32, C.POINTER(_i32)]
    lib.smx_read_step_ms.restype = C.c_int
    lib.smx_step_continuous.argtypes = [h, _p, C.POINTER(SmxState), C.POINTER(SmxSpawns), C.POINTER(SmxOutputs), _p]
    lib.smx_step_continuous.restype = C.c_int
    lib.smx_set_vias.argtypes = [h, C.POINTER(SmxVia), _i32, C.POINTER(_i32)]
    lib.smx_set_vias.restype = C.c_int
    lib.smx_set_missions.argtypes = [h, C.POINTER(SmxMission), _i32, C.POINTER(_i32), _i32]
    lib.smx_set_missions.restype = C.c_int
    lib.smx_step_trajectory.argtypes = [h, _p, _p, C.POINTER(SmxState), C.POINTER(SmxSpawns), C.POINTER(SmxOutputs), _p]
    lib.smx_step_trajectory.restype = C.c_int
    lib.smx_read_phase_ms.argtypes = [h, C.POINTER(C.c_float), _i32, C.POINTER(_i32)]
    lib.smx_read_phase_ms.restype = C.c_int
    lib.smx_set_timing.argtypes = [h, C.c_int]
    lib.smx_set_timing.restype = C.c_int
    lib.smx_set_controller_gains.argtypes = [h, _f64, _f64]
    lib.smx_set_controller_gains.restype = C.c_int
    lib.smx_last_error.argtypes = [h]
    lib.smx_last_error.restype = C.c_char_p
    lib.smx_version.argtypes = []
    lib.smx_version.restype = C.c_char_p
    lib.smx_destroy.argtypes = [h]
    lib.smx_destroy.restype = None
    lib.smx_check_buffers.argtypes = [C.POINTER(SmxConfig), C.c_int, C.POINTER(SmxState), C.POINTER(SmxSpawns),
                                      C.POINTER(SmxOutputs), C.c_char_p, C.c_uint64]
    lib.smx_check_buffers.restype = C.c_int
    lib.smx_set_launch_strategy.argtypes = [h, C.c_int]
    lib.smx_set_launch_strategy.restype = C.c_int
    lib.smx_launch_form.argtypes = [h]
    lib.smx_launch_form.restype = C.c_int
    lib.smx_struct_size.argtypes = [C.c_int]
    lib.smx_struct_size.restype = C.c_uint64
    for which, mirror in enumerate((SmxConfig, SmxMapTables, SmxState, SmxSpawns, SmxOutputs)):
        if lib.smx_struct_size(which) != C.sizeof(mirror):
            raise NativeLibraryError(
                f"ABI mismatch: {mirror.__name__} is {C.sizeof(mirror)} bytes here, "
                f"{lib.smx_struct_size(which)} in {path} (stale build?)"
            )
    _lib = lib
    return lib


class SmxError(RuntimeError):
    pass


def check(lib, handle, rc: int, what: str):
    if rc != 0:
        msg = lib.smx_last_error(handle)
        raise SmxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
