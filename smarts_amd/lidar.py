"""Lidar sensor parameters and base-ray table (host side).

Mirrors ``SensorParams`` / ``Lidar._compute_rays`` (reference ``smarts/core/lidar_sensor_params.py:24-66``,
``smarts/core/lidar.py:89-113``): one base ray per (laser elevation, azimuth step), computed once
on the host and uploaded to the device; the per-tick kernel only adds the sensor origin.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np


@dataclass(frozen=True)
class SensorParams:
    """lidar_sensor_params.py:24-34."""

    start_angle: float
    end_angle: float
    laser_angles: Tuple[float, ...]
    angle_resolution: float
    max_distance: float
    noise_mu: float = 0.0
    noise_sigma: float = 0.0


# lidar_sensor_params.py:37-55 — the two presets of the reference
VelodyneHDL32E = SensorParams(
    start_angle=0.0,
    end_angle=2 * np.pi,
    laser_angles=tuple(np.linspace(-np.radians(30.67), np.radians(10.67), 24)),
    angle_resolution=0.1728,
    max_distance=100.0,
    noise_mu=0.0,
    noise_sigma=0.078,
)

# default of AgentInterface(lidar=True) (agent_interface.py:132-135): 50 rings x 6 azimuths
BasicLidar = SensorParams(
    start_angle=0.0,
    end_angle=2 * np.pi,
    laser_angles=tuple(np.linspace(-np.radians(4), np.radians(10), 50)),
    angle_resolution=1.0,
    max_distance=20.0,
    noise_mu=0.0,
    noise_sigma=0.078,
)

# BASELINE.json configs[4]: one planar ring of 100 rays.  int(2*pi / (2*pi/100)) truncates to 99
# (lidar.py:91), so the ring asks for a slightly finer step to get exactly 100 rays.
Planar100 = SensorParams(
    start_angle=0.0, end_angle=2 * np.pi, laser_angles=(0.0,), angle_resolution=2 * np.pi / 100.5, max_distance=20.0
)


def ray_count(p: SensorParams) -> int:
    return int((p.end_angle - p.start_angle) / p.angle_resolution) * len(p.laser_angles)


def base_rays(p: SensorParams) -> np.ndarray:
    """[R, 3] ray vectors of length ``max_distance``; ring-major order (lidar.py:96-99).

    Each ray is the +y unit vector scaled by max_distance and rotated by the quaternion the
    reference builds from euler (roll = azimuth step, pitch = 0, yaw = -elevation) — the reference
    hands pybullet's (x, y, z, w) quaternion to a routine that reads it as (w, x, y, z)
    (lidar.py:100-106); the table reproduces exactly that."""
    n_az = int((p.end_angle - p.start_angle) / p.angle_resolution)
    yaw = -np.asarray(p.laser_angles, dtype=np.float64)[:, None]
    roll = (np.arange(n_az, dtype=np.float64) * p.angle_resolution)[None, :]
    roll, yaw = np.broadcast_arrays(roll, yaw)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    # (x, y, z, w) of the euler quaternion with pitch = 0 ...
    ex, ey, ez, ew = sr * cy, sr * sy, cr * sy, cr * cy
    # ... consumed as (w, x, y, z) and normalised before rotating v = (0, d, 0)
    qw, qx, qy, qz = ex, ey, ez, ew
    n = np.sqrt(qw * qw + qx * qx + qy * qy + qz * qz)
    qw, qx, qy, qz = qw / n, qx / n, qy / n, qz / n
    d = p.max_distance
    # R(q) applied to (0, d, 0): second column of the rotation matrix
    rx = 2.0 * (qx * qy - qw * qz) * d
    ry = (1.0 - 2.0 * (qx * qx + qz * qz)) * d
    rz = 2.0 * (qy * qz + qw * qx) * d
    return np.ascontiguousarray(np.stack([rx, ry, rz], axis=-1).reshape(-1, 3))
