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
