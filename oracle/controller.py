"""Oracle: action dispatch + lane-following controller (test infrastructure).

Follows reference ``smarts/core/controllers/__init__.py:61-152`` (dispatch),
``smarts/core/controllers/lane_following_controller.py`` (whole file) and
``smarts/core/controllers/trajectory_tracking_controller.py:444-473``
(curvature).  ``scipy.signal.place_poles`` is the same call the reference makes
(``lane_following_controller.py:420``).
"""
import math

import numpy as np
from scipy import signal

from . import ref_math as rm

METER_PER_SECOND_TO_KM_PER_HR = 3.6

# controllers/__init__.py:137-144: Lane action strings -> (target_speed, lane_change)
LANE_ACTIONS = {
    "keep_lane": (15, 0),
    "slow_down": (0, 0),
    "change_lane_left": (12.5, 1),
    "change_lane_right": (12.5, -1),
}
LANE_ACTION_NAMES = ["keep_lane", "slow_down", "change_lane_left", "change_lane_right"]

# desired closed-loop poles (lane_following_controller.py:58-61)
DESIRED_POLES = np.array([-35, -15, -2, -3])


class LaneFollowingControllerState:
    """lane_following_controller.py:34-49."""

    def __init__(self, target_lane_id=None):
        self.target_lane_id = target_lane_id
        self.target_speed = None
        self.heading_error_gain = None
        self.lateral_error_gain = None
        self.lateral_integral_error = 0
        self.integral_speed_error = 0
        self.steering_state = 0
        self.throttle_state = 0
        self.speed_error = 0
        self.min_curvature_location = (None, None)


def curvature_calculation(trajectory, offset=0, num_points=5):
    """trajectory_tracking_controller.py:444-473."""
    relative_heading_sum, relative_distant_sum = 0, 0
    if len(trajectory[2]) <= num_points + offset:
        return 1e20
    for i in range(num_points):
        relative_heading_sum += rm.min_angles_difference_signed(
            trajectory[2][i + 1 + offset], trajectory[2][i + offset]
        )
        relative_distant_sum += abs(
            math.sqrt(
                (trajectory[0][i + offset] - trajectory[0][i + offset + 1]) ** 2
                + (trajectory[1][i + offset] - trajectory[1][i + offset + 1]) ** 2
            )
        )
    if relative_heading_sum == 0:
        return 1e20
    return relative_distant_sum / relative_heading_sum


def lateral_gains(target_speed, half_vehicle_len, vehicle_mass, vehicle_inertia_z, road_stiffness):
    """lane_following_controller.py:376-437 -> (heading_error_gain, lateral_error_gain)."""
    if target_speed > 0:
        state_matrix = np.array(
            [
                [0, target_speed, 0, target_speed],
                [0, 0, 1, 0],
                [0, 0, -(2 * road_stiffness * (half_vehicle_len ** 2)) / (target_speed * vehicle_inertia_z), 0],
                [0, 0, -1, -2 * road_stiffness / (vehicle_mass * target_speed)],
            ]
        )
        input_matrix = np.array(
            [
                [0],
                [0],
                [half_vehicle_len * road_stiffness / vehicle_inertia_z],
                [road_stiffness / (vehicle_mass * target_speed)],
            ]
        )
        fsf1 = signal.place_poles(state_matrix, input_matrix, DESIRED_POLES, method="KNV0")
        return (
            np.clip(fsf1.gain_matrix[0][1], 0.02, 0.04),
            np.clip(fsf1.gain_matrix[0][0], 3.4, 4.1),
        )
    return 0.01, 0.36


def find_current_lane(wp_paths, vehicle_position):
    """lane_following_controller.py:367-374."""
    rel = [np.linalg.norm(wp_paths[idx][0].pos - vehicle_position[0:2]) for idx in range(len(wp_paths))]
    return np.argmin(rel)


def perform_lane_following(road_map, veh, state, dt, target_speed=12.5, lane_change=0, route=()):
    """lane_following_controller.py:63-365.

    ``veh`` exposes what the reference reads from ``vehicle`` / ``vehicle.chassis``:
    ``position`` (3,), ``heading``, ``speed``, ``lateral_speed``
    (= ``chassis.longitudinal_lateral_speed[1]``), ``yaw_rate_z``
    (= ``chassis.velocity_vectors[1][2]``), ``length``, ``max_steering_wheel``,
    ``mass``, ``inertia_z``, ``road_stiffness``.
    Returns ``(throttle, brake, steering)`` as handed to ``vehicle.control``.
    """
    wp_paths = road_map.waypoint_paths(veh.position, veh.heading, lookahead=16, route=route)
    assert wp_paths, "no waypoints found.  not near lane?"
    current_lane = find_current_lane(wp_paths, veh.position)
    wp_path = wp_paths[np.clip(current_lane + lane_change, 0, len(wp_paths) - 1)]

    ewma_road_curviness = 0.0
    for wp_a, wp_b in reversed(list(zip(wp_path, wp_path[1:]))):
        ewma_road_curviness = rm.lerp(
            ewma_road_curviness, math.degrees(abs(wp_a.relative_heading(wp_b.heading))), 0.03
        )
    road_curviness = np.clip(ewma_road_curviness / 2.5, 0, 1)

    num_trajectory_points = min([10, len(wp_path)])
    trajectory = [
        [wp_path[i].pos[0] for i in range(num_trajectory_points)],
        [wp_path[i].pos[1] for i in range(num_trajectory_points)],
        [wp_path[i].heading for i in range(num_trajectory_points)],
    ]
    look_ahead_curvature = abs(curvature_calculation(trajectory, 4))
    min_curvature = 2
    if look_ahead_curvature <= min_curvature:
        state.min_curvature_location = (wp_path[4].pos[0], wp_path[4].pos[1])

    look_ahead_wp_num = 3 if road_curviness > 0.5 else 4
    look_ahead_wp_num = min(look_ahead_wp_num, len(wp_path) - 1)

    reference_heading = wp_path[0].heading
    look_ahead_wp = wp_path[look_ahead_wp_num]
    look_ahead_dist = look_ahead_wp.dist_to(veh.position)
    vehicle_look_ahead_pt = [
        veh.position[0] - look_ahead_dist * math.sin(veh.heading),
        veh.position[1] + look_ahead_dist * math.cos(veh.heading),
    ]

    if road_curviness < 0.3:
        raw_throttle = -METER_PER_SECOND_TO_KM_PER_HR * 1.8 * (veh.speed - target_speed)
    elif road_curviness > 0.3 and road_curviness < 0.8:
        raw_throttle = -0.6 * METER_PER_SECOND_TO_KM_PER_HR * (veh.speed - np.clip(target_speed, 0, 6.94))
    else:
        raw_throttle = -0.6 * METER_PER_SECOND_TO_KM_PER_HR * (veh.speed - np.clip(target_speed, 0, 5.56))

    speed_error = veh.speed - target_speed
    state.integral_speed_error += speed_error * dt
    velocity_error_damping_term = (speed_error - state.speed_error) / dt
    lateral_force_coefficient = 1.5
    if veh.speed < 8 or target_speed < 6:
        lateral_force_coefficient = 0
    raw_throttle += (
        -0.2 * velocity_error_damping_term
        - 0.1 * state.integral_speed_error
        + abs(lateral_force_coefficient * math.sin(state.steering_state * veh.max_steering_wheel))
    )
    state.speed_error = speed_error

    if (state.min_curvature_location != (None, None)) and math.sqrt(
        (veh.position[0] - state.min_curvature_location[0]) ** 2
        + (veh.position[1] - state.min_curvature_location[1]) ** 2
    ) < 2:
        reference_heading = wp_path[look_ahead_wp_num].heading

    if state.target_speed != target_speed:
        state.target_speed = target_speed
        state.heading_error_gain, state.lateral_error_gain = lateral_gains(
            target_speed, veh.length / 2, veh.mass, veh.inertia_z, veh.road_stiffness
        )

    controller_lat_error = wp_path[look_ahead_wp_num].signed_lateral_error(vehicle_look_ahead_pt)

    curvature_radius = curvature_calculation(trajectory)
    brake_norm = 0
    if raw_throttle < 0:
        brake_norm = np.clip(-raw_throttle, 0, 1)
        throttle_norm = 0
    else:
        if veh.speed > 70 / 3.6 and abs(curvature_radius) <= 1e3:
            traction_gain = 4.5
        elif 40 / 3.6 <= veh.speed <= 70 / 3.6 and abs(curvature_radius) <= 3:
            traction_gain = 2.5
        else:
            traction_gain = 0.5
        throttle_norm = np.clip(
            raw_throttle - traction_gain * METER_PER_SECOND_TO_KM_PER_HR * abs(veh.lateral_speed), 0, 1
        )

    state.lateral_integral_error += dt * controller_lat_error
    steering_feed_forward_gain = 0.15
    if abs(curvature_radius) < 7:
        steering_feed_forward_gain = 0.45
    steering_controller_feed_forward = 1 * steering_feed_forward_gain * (1 / curvature_radius) * (veh.speed) ** 2
    normalized_speed = np.clip(veh.speed * 3.6 / 100, 0, 1)
    heading_speed_gain = -rm.lerp(0.5, 14, normalized_speed)
    yaw_rate_speed_gain = rm.lerp(5.75, 11.75, normalized_speed)
    lateral_speed_gain = np.clip(rm.lerp(-1, 14, normalized_speed), 1, 2)

    max_steering_normalized = 1
    if abs(curvature_radius) > 1e7 and lane_change != 0:
        heading_speed_gain = -4.95
        yaw_rate_speed_gain = 1
        lateral_speed_gain = 0.22
        max_steering_normalized = 0.12

    z_yaw = veh.yaw_rate_z
    heading_error = rm.min_angles_difference_signed((veh.heading % (2 * math.pi)), reference_heading)
    steering_norm = np.clip(
        -heading_speed_gain * math.degrees(state.heading_error_gain) * heading_error
        + lateral_speed_gain * state.lateral_error_gain * (controller_lat_error)
        + yaw_rate_speed_gain * z_yaw
        + 0.3 * state.lateral_integral_error
        - steering_controller_feed_forward,
        -max_steering_normalized,
        max_steering_normalized,
    )
    state.steering_state = rm.low_pass_filter(steering_norm, state.steering_state, 5.5, dt)
    state.throttle_state = rm.low_pass_filter(throttle_norm, state.throttle_state, 2, dt, lower_bound=0)
    return state.throttle_state, brake_norm, state.steering_state


# ``_update_target_lane_if_reached_end_of_lane`` (lane_following_controller.py:439-473) is
# deliberately NOT restated: ``target_lane_id`` is write-only state (nothing on the
# path ever reads it back), and its update goes through the road map's shared
# ``_WaypointsCache`` keyed by lane *index* (sumo_road_network.py:1229-1276), which
# makes it depend on which other agent queried last.  See DESIGN.md "Deviations".


# ---------------------------------------------------------------------------------------------
# ActionSpaceType.Trajectory: TrajectoryTrackingController.perform_trajectory_tracking_PD
# (trajectory_tracking_controller.py:230-331) with its helpers (:333-441) and the sedan's
# controller parameters (models/controller_parameters.yaml:1-16).
# ---------------------------------------------------------------------------------------------
SEDAN_CONTROL = dict(
    final_heading_gain=0.15, final_lateral_gain=4.65, final_steering_filter_constant=23.5, throttle_filter_constant=22.5,
    velocity_gain=5.1, velocity_integral_gain=0, traction_gain=6, final_lateral_error_derivative_gain=0.3,
    final_heading_error_derivative_gain=3.1, initial_look_ahead_distant=6, derivative_activation=1,
    speed_reduction_activation=1, velocity_damping_gain=0.001, windup_gain=0.01,
)


class TrajectoryTrackingControllerState:
    """trajectory_tracking_controller.py:117-129."""

    def __init__(self):
        self.heading_error_gain = None
        self.lateral_error_gain = None
        self.heading_error = 0
        self.lateral_error = 0
        self.velocity_error = 0
        self.integral_velocity_error = 0
        self.integral_windup_error = 0
        self.steering_state = 0
        self.throttle_state = 0


def _heading_lateral_error(veh, trajectory, initial_look_ahead_distant, speed_reduction_activation):
    """:398-441."""
    heading_error = rm.min_angles_difference_signed((veh.heading % (2 * math.pi)), trajectory[2][0])
    look_ahead_points = initial_look_ahead_distant
    if abs(curvature_calculation(trajectory, 4)) < 30 and speed_reduction_activation:
        initial_look_ahead_distant = 1
        look_ahead_points = 1
    k = min([look_ahead_points, len(trajectory[2]) - 1])
    path_vector = rm.radians_to_vec(trajectory[2][k])
    look_ahead_pt = [
        veh.position[0] - initial_look_ahead_distant * math.sin(veh.heading),
        veh.position[1] + initial_look_ahead_distant * math.cos(veh.heading),
    ]
    lateral_error = rm.signed_dist_to_line(look_ahead_pt, [trajectory[0][k], trajectory[1][k]], path_vector)
    return heading_error, lateral_error


def _raw_throttle_feedback(veh, state, trajectory, velocity_gain, velocity_integral_gain, integral_velocity_error,
                           velocity_damping_gain, windup_gain, traction_gain, speed_reduction_activation,
                           throttle_filter_constant, dt_sec):
    """:333-395."""
    desired_speed = trajectory[3][-1]
    absolute_ahead_curvature = abs(curvature_calculation(trajectory, 4))
    if absolute_ahead_curvature < 30 and speed_reduction_activation:
        desired_speed = np.clip(0.8 * desired_speed, 0, 8.3)
    elif absolute_ahead_curvature < 100 and speed_reduction_activation:
        desired_speed *= 0.8
    velocity_error = veh.speed - desired_speed
    velocity_error_damping_term = (velocity_error - state.velocity_error) / dt_sec
    raw_throttle = 3.6 * (
        -0.5 * velocity_gain * velocity_error
        - velocity_integral_gain * (integral_velocity_error + windup_gain * state.integral_windup_error)
        - velocity_damping_gain * velocity_error_damping_term
    )
    state.velocity_error = velocity_error
    state.integral_windup_error = np.clip(raw_throttle, -1, 1) - raw_throttle
    state.throttle_state = rm.low_pass_filter(
        raw_throttle, state.throttle_state, throttle_filter_constant, dt_sec,
        raw_value=-traction_gain * abs(veh.longitudinal_lateral_speed[1]),
    )
    return state.throttle_state, desired_speed


def perform_trajectory_tracking_pd(trajectory, veh, state, dt_sec, params=SEDAN_CONTROL):
    """:230-331.  ``trajectory`` = (xs, ys, headings, speeds); returns (throttle, brake, steering)."""
    final_steering_filter_constant = params["final_steering_filter_constant"]
    throttle_filter_constant = params["throttle_filter_constant"]
    lateral_gain = 0.61
    heading_gain = 0.01
    lateral_error_derivative_gain = 0.15
    heading_error_derivative_gain = 0.5
    normalized_speed = np.clip((3.6 * trajectory[3][0] - 20) / (80 - 20), 0, 1)
    if veh.speed > 70 / 3.6:  # adjust_gains_for_normalized_speed is False in the reference (:262)
        lateral_gain = 1.51
        heading_error_derivative_gain = 0.1
    steering_filter_constant = rm.lerp(12, final_steering_filter_constant, normalized_speed)
    if abs(rm.min_angles_difference_signed(trajectory[2][-1], trajectory[2][0])) > 2:
        throttle_filter_constant = 2.5
    if abs(curvature_calculation(trajectory, 0, num_points=3)) < 150:
        heading_gain = 0.05
        lateral_error_derivative_gain = 0.015
        heading_error_derivative_gain = 0.05
    heading_error, lateral_error = _heading_lateral_error(
        veh, trajectory, params["initial_look_ahead_distant"], params["speed_reduction_activation"])
    curvature_radius = curvature_calculation(trajectory)
    z_yaw = veh.angular_velocity[2]
    derivative_term = (+heading_error_derivative_gain * z_yaw
                       + lateral_error_derivative_gain * (lateral_error - state.lateral_error) / dt_sec)
    steering_feed_forward_term = 0.1 * (1 / curvature_radius) * (veh.speed) ** 2
    steering_raw = np.clip(
        params["derivative_activation"] * derivative_term + math.degrees(heading_gain * (heading_error))
        + 1 * lateral_gain * lateral_error - steering_feed_forward_term, -1, 1)
    state.steering_state = rm.low_pass_filter(steering_raw, state.steering_state, steering_filter_constant, dt_sec)
    raw_throttle, desired_speed = _raw_throttle_feedback(
        veh, state, trajectory, params["velocity_gain"], params["velocity_integral_gain"], state.integral_velocity_error,
        params["velocity_damping_gain"], params["windup_gain"], params["traction_gain"],
        params["speed_reduction_activation"], throttle_filter_constant, dt_sec)
    if raw_throttle > 0:
        brake_norm, throttle_norm = 0, np.clip(raw_throttle, 0, 1)
    else:
        brake_norm, throttle_norm = np.clip(-raw_throttle, 0, 1), 0
    state.heading_error = heading_error
    state.lateral_error = lateral_error
    state.integral_velocity_error += (veh.speed - desired_speed) * dt_sec
    return float(throttle_norm), float(brake_norm), float(state.steering_state)


class PackedTrajectory:
    """The form a trajectory travels in (include/smx.h smx_step_trajectory): the controller reads
    points 0..9, the last point and the length, nothing else.  Indexing mirrors a full list."""

    def __init__(self, row, n):
        self.row, self.n = row, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        if i == self.n - 1:
            return float(self.row[10]) if self.n > 10 else float(self.row[i])
        if i >= 10:
            raise IndexError(f"point {i} of a packed trajectory is not carried")
        return float(self.row[i])


def unpack_trajectory(packed, n):
    """``packed``: [4][11] (x, y, heading, speed rows; column 10 = the last point)."""
    return [PackedTrajectory(packed[r], n) for r in range(4)]
