"""Oracle: occupancy grid map and lidar sensors (test infrastructure).

OGM — reference ``OGMSensor`` (``smarts/core/sensors.py:719-758``) over the Panda3D offscreen
camera (``smarts/core/renderer.py:325-395``): an orthographic top-down render, centred on the
vehicle, rotated with its heading (``sensors.py:670-672``), film size ``width*res x height*res``
(``renderer.py:384-385``), alpha channel = 255 where a vehicle mesh covers the pixel, road hidden
(``renderer.py:223``), rows flipped so that row 0 is ahead of the vehicle (``sensors.py:748``).
Panda3D (an OpenGL rasteriser, not in the reference tree) is substituted by its sampling rule on
the vehicles' footprints: a pixel is set iff its *centre* lies inside some vehicle's oriented
chassis rectangle.  Pinned only by the reference's ±2 px checks (``test_observations.py:133-152``).

Lidar — reference ``Lidar._compute_rays`` (``smarts/core/lidar.py:89-113``) restated line by line
(including the quaternion-order quirk) and pinned by
``tests/golden/lidar_rays.npz``; ``_trace_rays`` (``lidar.py:115-134``) = pybullet
``rayTestBatch`` is substituted by exact ray / oriented-box and ray / ground-plane intersection
(chassis box 1.47 x 3.68 x 1.0 centred 0.6 m above the base frame, ``models/vehicle.urdf``);
the emitting vehicle itself is not hit (the ray starts inside its own box).
"""
import itertools
import math

import numpy as np

from . import ref_math as rm
from .dynamics import BASE_HEIGHT, CHASSIS_HEIGHT, CHASSIS_LENGTH, CHASSIS_WIDTH

CHASSIS_BOX_Z = 0.6  # collision box origin above the base frame (models/vehicle.urdf)


def ogm(ego, vehicles, width, height, resolution):
    """(height, width) uint8 grid for `ego` (a VehicleBody); `vehicles` = all alive bodies (ego included)."""
    grid = np.zeros((height, width), dtype=np.uint8)
    h = ego.heading
    right = np.array([math.cos(h), math.sin(h)])
    fwd = np.array([-math.sin(h), math.cos(h)])
    cols = (np.arange(width) + 0.5 - width / 2) * resolution   # x to the right of the vehicle
    rows = (height / 2 - (np.arange(height) + 0.5)) * resolution  # y ahead of the vehicle
    X, Y = np.meshgrid(cols, rows)
    ce, se = math.cos(h), math.sin(h)
    for v in vehicles:
        d = np.array([v.x - ego.x, v.y - ego.y])
        ex, ey = float(d @ right), float(d @ fwd)
        # the vehicle's axes in the ego frame, from each vehicle's own cos / sin of its heading (one pair per
        # vehicle and tick serves every observer; smx_kernels.hip k_ogm_env): forward = (-sin vh, cos vh) and
        # right = (cos vh, sin vh) projected on the ego's right / forward axes
        cm, sm = math.cos(v.heading), math.sin(v.heading)
        vf = (cm * se - sm * ce, sm * se + cm * ce)
        vr = (cm * ce + sm * se, sm * ce - cm * se)
        qx, qy = X - ex, Y - ey
        inside = (np.abs(qx * vf[0] + qy * vf[1]) <= 0.5 * v.length) & (np.abs(qx * vr[0] + qy * vr[1]) <= 0.5 * v.width)
        grid[inside] = 255
    return grid


def dagm(ego, lanes, width, height, resolution):
    """Drivable-area grid map (DrivableAreaGridMapSensor, sensors.py:675-716) for `ego`: (height, width)
    uint8, the OGM's camera (sensors.py:651-672: centred on the vehicle, up = its heading, np.flipud).
    The reference renders the road mesh — every lane's centre line buffered by half its width
    (sumo_road_network.py:986-1019) — with Panda3D, which is absent; restated as: a pixel is 255 when its
    centre is within half the lane width of a segment of that lane's centre line.  PARITY UNPINNED
    beyond the reference's own check (non-zero within +-2 px of on-road vehicles, test_observations.py:150-153).

    `lanes`: iterable of (shape [(x, y), ...], width) — every lane of the map, internal lanes included."""
    grid = np.zeros((height, width), dtype=np.uint8)
    h = ego.heading
    rx, ry = math.cos(h), math.sin(h)
    fx, fy = -math.sin(h), math.cos(h)
    for shape, lane_width in lanes:
        hw = 0.5 * lane_width
        for (x1, y1), (x2, y2) in zip(shape[:-1], shape[1:]):
            d1x, d1y, d2x, d2y = x1 - ego.x, y1 - ego.y, x2 - ego.x, y2 - ego.y
            ax, ay = d1x * rx + d1y * ry, d1x * fx + d1y * fy
            bx, by = d2x * rx + d2y * ry, d2x * fx + d2y * fy
            # bounding box of the band in pixels (a pure speed-up: pixels outside cannot pass the test)
            c0 = int(math.floor((min(ax, bx) - hw) / resolution + 0.5 * width - 0.5)) - 1
            c1 = int(math.ceil((max(ax, bx) + hw) / resolution + 0.5 * width - 0.5)) + 1
            r0 = int(math.floor(0.5 * height - 0.5 - (max(ay, by) + hw) / resolution)) - 1
            r1 = int(math.ceil(0.5 * height - 0.5 - (min(ay, by) - hw) / resolution)) + 1
            c0, r0, c1, r1 = max(c0, 0), max(r0, 0), min(c1, width - 1), min(r1, height - 1)
            if c0 > c1 or r0 > r1:
                continue
            px = (np.arange(c0, c1 + 1) + 0.5 - 0.5 * width) * resolution
            py = (0.5 * height - (np.arange(r0, r1 + 1) + 0.5)) * resolution
            X, Y = np.meshgrid(px, py)
            # squared point / segment distance, the arithmetic of sim._seg_point_dist2
            dx, dy = bx - ax, by - ay
            ll = dx * dx + dy * dy
            t = np.zeros_like(X) if ll == 0.0 else ((X - ax) * dx + (Y - ay) * dy) / ll
            t = np.minimum(1.0, np.maximum(0.0, t))
            ex, ey = ax + t * dx - X, ay + t * dy - Y
            inside = ex * ex + ey * ey <= hw * hw
            grid[r0:r1 + 1, c0:c1 + 1][inside] = 255
    return grid


def quaternion_from_euler(roll, pitch, yaw):
    """pybullet.getQuaternionFromEuler -> (x, y, z, w)."""
    cr, sr = math.cos(roll * 0.5), math.sin(roll * 0.5)
    cp, sp = math.cos(pitch * 0.5), math.sin(pitch * 0.5)
    cy, sy = math.cos(yaw * 0.5), math.sin(yaw * 0.5)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy)


def base_rays(start_angle, end_angle, laser_angles, angle_resolution, max_distance):
    """``Lidar._compute_rays`` (lidar.py:89-108): direction vectors of the base rays."""
    n_rays = int((end_angle - start_angle) / angle_resolution)
    yaws = -1 * np.asarray(laser_angles)
    rolls = np.arange(n_rays) * angle_resolution
    out = []
    for yaw, roll in itertools.product(yaws, rolls):
        rot = quaternion_from_euler(roll, 0, yaw)
        out.append(rm.rotate_quat(np.asarray(rot, dtype=float), np.asarray((0, max_distance, 0), dtype=float)))
    return np.array(out)


def _ray_box(origin, direction, body):
    """Smallest t in [0, 1] at which origin + t*direction enters the chassis box of `body`, or None."""
    h = body.heading
    f = np.array([-math.sin(h), math.cos(h), 0.0])
    r = np.array([math.cos(h), math.sin(h), 0.0])
    u = np.array([0.0, 0.0, 1.0])
    c = np.array([body.x, body.y, BASE_HEIGHT + CHASSIS_BOX_Z])
    half = (0.5 * CHASSIS_LENGTH, 0.5 * CHASSIS_WIDTH, 0.5 * CHASSIS_HEIGHT)
    rel = origin - c
    tmin, tmax = 0.0, 1.0
    for axis, hw in zip((f, r, u), half):
        o = float(rel @ axis)
        d = float(direction @ axis)
        if d == 0.0:
            if abs(o) > hw:
                return None
            continue
        t1, t2 = (-hw - o) / d, (hw - o) / d
        if t1 > t2:
            t1, t2 = t2, t1
        tmin, tmax = max(tmin, t1), min(tmax, t2)
        if tmin > tmax:
            return None
    return tmin


def lidar(ego, others, rays, lidar_offset=(0.0, 0.0, 1.0)):
    """Point cloud of one vehicle: (points [R,3] with inf on miss, hits [R] bool).

    Origin = vehicle position + (0, 0, 1) (sensors.py:805-821); rays do not rotate with the
    vehicle (lidar.py:109-113)."""
    origin = np.array([ego.x, ego.y, BASE_HEIGHT]) + np.asarray(lidar_offset)
    pts = np.full((len(rays), 3), np.inf)
    hits = np.zeros(len(rays), dtype=bool)
    for i, d in enumerate(rays):
        best = None
        if d[2] < 0.0:  # ground plane z = 0
            t = -origin[2] / d[2]
            if 0.0 <= t <= 1.0:
                best = t
        for b in others:
            t = _ray_box(origin, d, b)
            if t is not None and (best is None or t < best):
                best = t
        if best is not None:
            hits[i] = True
            pts[i] = origin + best * d
    return pts, hits


# ---------------------------------------------------------------------------------------------
# RoadWaypointsSensor (sensors.py:991-1040)
# ---------------------------------------------------------------------------------------------
def road_waypoints(rmap, position, heading, horizon=32, route=None):
    """``RoadWaypointsSensor.__call__`` -> ``{lane_id: [waypoint paths]}`` in the reference's insertion order
    (the lanes of the nearest lane's road, of its parallel roads, of the roads oncoming at the point).
    ``route``: the plan's road ids ([] / None = the endless mission's empty route: no filter)."""
    lane = rmap.nearest_lane(position)
    if not lane:
        return {}
    road = lane.road
    point = (position[0], position[1], position[2] if len(position) > 2 else 0.0)
    lane_paths = {}
    for croad in [road] + road.parallel_roads + road.oncoming_roads_at_point(point):
        for ln in croad.lanes:
            lane_paths[ln.lane_id] = _paths_for_lane(rmap, ln, point, horizon, route, None)
    return lane_paths


def _paths_for_lane(rmap, lane, point, horizon, route, overflow_offset):
    """sensors.py:1014-1040 (the waypoint spacing is assumed to be 1 m there too)."""
    if overflow_offset is None:
        offset = lane.offset_along_lane(point)
        start_offset = offset - horizon
    else:
        start_offset = lane.length + overflow_offset
    incoming_lanes = lane.incoming_lanes
    if start_offset < 0 and len(incoming_lanes) > 0:
        paths = []
        for lane_in in incoming_lanes:
            paths += _paths_for_lane(rmap, lane_in, point, horizon, route, start_offset)
        return paths
    start_offset = max(0, start_offset)
    wp_start = lane.from_lane_coord(start_offset)
    # Pose.from_center(wp_start, heading).as_position2d() -> the start point; lookahead = 2 x horizon
    return rmap.lane_waypoint_paths_at(lane, wp_start[:2], horizon * 2, list(route) if route else None)
