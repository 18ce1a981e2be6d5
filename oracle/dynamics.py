"""Oracle: vehicle body — control application, integration, state read-back (test infrastructure).

What is restated from the reference, and what is a MODEL SUBSTITUTION
----------------------------------------------------------------------
Restated line by line (in-tree Python):

* ``AckermannChassis.control`` / ``_apply_steering`` / ``_apply_throttle`` /
  ``_apply_brake`` (``smarts/core/chassis.py:678-794``): brake suppression below
  1/36 m/s, steer target ``-steering * max_steering / gear_ratio``, wheel torque
  ``throttle * max_torque - brake * max_btorque`` on all four wheels;
* the state read-back ``pose / speed / steering / yaw_rate / velocity_vectors /
  longitudinal_lateral_speed`` (``chassis.py:493-566``) incl. the
  ``vec_to_radians(omega[:2])`` yaw-rate quirk;
* ``Vehicle.bounding_box`` (``vehicle.py:315-332``);
* the substep schedule of ``SMARTS._step_pybullet`` / ``_setup_bullet_client``
  (``smarts.py:602-613, 923-931``): ``int(dt * 240)`` internal substeps of 1/240 s.

Substituted (pybullet==3.0.6 is not in the reference tree and cannot run here;
**pose trajectories are parity-unpinned**, SURVEY.md §7 "Hard parts"): the
Featherstone multibody + LCP wheel contact solve is replaced by a planar
single-track ("bicycle") model of the same URDF (``models/vehicle.urdf``:
wheelbase 3.0 m with the base frame mid-wheelbase, chassis 2356 kg / izz 2681.95
+ 4 x 15 kg wheels + 3 x 1 kg links, steer limits +-0.8727 rad, wheel radius
0.31265 m).  Lateral tyre forces follow the linear law that the reference's own
lane-following controller design linearises (``lane_following_controller.py:
391-419``: cornering stiffness = ``sim.road_stiffness`` = 1e5 N/rad per axle),
saturated at the friction limit of ``models/plane.urdf`` (mu = 3).  Below 2 m/s
the model degenerates to the no-slip kinematic bicycle (the slip-angle law is
singular at rest).  The steer joints follow pybullet's default POSITION_CONTROL
motor (position gain 0.1 per internal substep).  Brake torque is dissipative (it
stops at zero speed instead of reversing the car; the reference guards the same
effect with the 1/36 m/s rule, chassis.py:695-701).  Integration is explicit
Euler at the 240 Hz substep.
"""
import math

import numpy as np

from . import ref_math as rm

# models/controller_parameters.yaml:17-25 ("sedan" chassis block)
WHEEL_RADIUS = 0.31265
MAX_TORQUE = 1600.0
MAX_BTORQUE = 1400.0
MAX_STEERING = 12.56
STEERING_GEAR_RATIO = 17.4
# models/vehicle.urdf
CHASSIS_MASS = 2356.0
CHASSIS_INERTIA_Z = 2681.95008628
TOTAL_MASS = 2356.0 + 4 * 15.0 + 2 * 1.0 + 1.0
AXLE_DIST = 1.5  # both axles sit 1.5 m from the base frame
TRACK_HALF = 0.5
TOTAL_INERTIA_Z = CHASSIS_INERTIA_Z + (4 * 15.0 + 2 * 1.0) * (AXLE_DIST ** 2 + TRACK_HALF ** 2)
INV_TOTAL_MASS = 1.0 / TOTAL_MASS
INV_TOTAL_INERTIA_Z = 1.0 / TOTAL_INERTIA_Z
WHEELBASE = 3.0
REAR_AXLE_TO_BASE = 1.5
CORNERING_STIFFNESS = 100000.0  # = ROAD_STIFFNESS, the controller's design value
GROUND_FRICTION = 3.0  # models/plane.urdf lateral_friction
GRAVITY = 9.8  # smarts.py:615
KINEMATIC_BELOW_SPEED = 2.0
STEER_LIMIT = 0.8727
CHASSIS_LENGTH = 3.68
CHASSIS_WIDTH = 1.47
CHASSIS_HEIGHT = 1.0
BASE_HEIGHT = WHEEL_RADIUS - 0.3  # wheel centres sit 0.3 m above the base frame
# models/plane.urdf contact stiffness, read by SMARTS.road_stiffness (smarts.py:773-776)
ROAD_STIFFNESS = 100000.0
# smarts.py:67
MAX_PYBULLET_FREQ = 240
STEER_POSITION_GAIN = 0.1  # pybullet POSITION_CONTROL default positionGain


class VehicleBody:
    """One Ackermann vehicle (planar state)."""

    def __init__(self, x, y, heading, speed):
        self.x = float(x)
        self.y = float(y)
        self.z = BASE_HEIGHT
        self.heading = rm.wrap_heading(heading)
        # body-frame velocity of the base frame (u forward, v to the left), yaw rate, steer joint angle
        self.u = float(speed)
        self.v = 0.0
        self.delta = 0.0
        self.yaw_rate_z = 0.0
        self._last_control = (0.0, 0.0, 0.0)
        # constants read by the controller
        self.length = CHASSIS_LENGTH
        self.width = CHASSIS_WIDTH
        self.height = CHASSIS_HEIGHT
        self.max_steering_wheel = MAX_STEERING / STEERING_GEAR_RATIO  # chassis.py:611-614
        self.mass = CHASSIS_MASS  # getDynamicsInfo(chassis link)
        self.inertia_z = CHASSIS_INERTIA_Z
        self.road_stiffness = ROAD_STIFFNESS

    # ---- read-back (chassis.py:493-566) ----
    @property
    def position(self):
        return np.array([self.x, self.y, self.z])

    @property
    def lateral_body_speed(self):
        return self.v

    @property
    def world_velocity(self):
        h = self.heading
        fwd = (-math.sin(h), math.cos(h))
        left = (-math.cos(h), -math.sin(h))
        v = self.lateral_body_speed
        return np.array([self.u * fwd[0] + v * left[0], self.u * fwd[1] + v * left[1], 0.0])

    @property
    def speed(self):
        velocity = self.world_velocity
        return math.sqrt(velocity.dot(velocity))

    @property
    def longitudinal_lateral_speed(self):
        velocity = self.world_velocity
        heading = self.heading
        return (
            (velocity[1] * math.cos(heading) - velocity[0] * math.sin(heading)),
            (velocity[1] * math.sin(heading) + velocity[0] * math.cos(heading)),
        )

    @property
    def lateral_speed(self):
        return self.longitudinal_lateral_speed[1]

    @property
    def steering(self):
        return -self.delta  # chassis.py:510-525 (mean of the two steer joints, sign flipped)

    @property
    def yaw_rate(self):
        # chassis.py:552-556: an *angle* of (omega_x, omega_y); planar motion has both zero
        return rm.vec_to_radians((0.0, 0.0))

    @property
    def linear_velocity(self):
        return np.array(self.longitudinal_lateral_speed + (0,))  # chassis.py:536-541

    @property
    def angular_velocity(self):
        return np.array([0.0, 0.0, self.yaw_rate_z])

    @property
    def bounding_box(self):
        """vehicle.py:315-332."""
        origin = self.position[:2]
        dimensions = np.array([self.width, self.length])
        corners = np.array([(-1, 1), (1, 1), (1, -1), (-1, -1)]) / 2
        return [
            rm.rotate_around_point(point=origin + corner * dimensions, radians=self.heading, origin=origin)
            for corner in corners
        ]

    # ---- control (chassis.py:678-718) ----
    def control(self, throttle=0, brake=0, steering=0):
        assert 0 <= throttle <= 1 and 0 <= brake <= 1 and -1 <= steering <= 1
        if brake > 0 and self.longitudinal_lateral_speed[0] < 1 / 36:
            brake = 0
        self._last_control = (float(throttle), float(brake), float(steering))

    # ---- integration (model substitution, see module docstring) ----
    def step(self, dt):
        throttle, brake, steering = self._last_control
        substeps = max(1, int(dt * MAX_PYBULLET_FREQ))
        h = dt / substeps
        delta_target = -steering * MAX_STEERING * (1 / STEERING_GEAR_RATIO)
        drive_accel = 4.0 * (throttle * MAX_TORQUE) / WHEEL_RADIUS / TOTAL_MASS
        brake_decel = 4.0 * (brake * MAX_BTORQUE) / WHEEL_RADIUS / TOTAL_MASS
        f_max = 0.5 * GROUND_FRICTION * TOTAL_MASS * GRAVITY
        for _ in range(substeps):
            self.delta += STEER_POSITION_GAIN * (delta_target - self.delta)
            self.delta = min(max(self.delta, -STEER_LIMIT), STEER_LIMIT)
            u, v, r = self.u, self.v, self.yaw_rate_z
            u_new = u + h * (drive_accel + v * r)
            if brake_decel > 0.0 and u_new > 0.0:
                # a brake opposes motion; it does not push the car backwards
                u_new = max(0.0, u_new - h * brake_decel)
            if u_new >= KINEMATIC_BELOW_SPEED:
                # one reciprocal per substep and reciprocal constants: the model is this module's own
                # (a substitution), and the device evaluates exactly this form
                inv_u = 1.0 / u_new
                alpha_f = self.delta - (v + AXLE_DIST * r) * inv_u
                alpha_r = -(v - AXLE_DIST * r) * inv_u
                f_f = min(max(CORNERING_STIFFNESS * alpha_f, -f_max), f_max)
                f_r = min(max(CORNERING_STIFFNESS * alpha_r, -f_max), f_max)
                v_new = v + h * ((f_f + f_r) * INV_TOTAL_MASS - u_new * r)
                r_new = r + h * (AXLE_DIST * (f_f - f_r) * INV_TOTAL_INERTIA_Z)
            else:
                r_new = u_new * math.tan(self.delta) / WHEELBASE
                v_new = r_new * REAR_AXLE_TO_BASE
            hd = self.heading
            self.x += h * (-u_new * math.sin(hd) - v_new * math.cos(hd))
            self.y += h * (u_new * math.cos(hd) - v_new * math.sin(hd))
            self.heading = hd + h * r_new
            self.u, self.v, self.yaw_rate_z = u_new, v_new, r_new
        self.heading = rm.wrap_heading(self.heading)
