"""Oracle: angle / vector / polyline arithmetic (test infrastructure, see oracle/__init__.py).

Follows reference ``smarts/core/utils/math.py`` and ``smarts/core/coordinates.py``
function by function; each docstring names the lines restated.
"""
import math

import numpy as np

TWO_PI = 2 * math.pi


def wrap_heading(value):
    """``Heading.__new__`` (coordinates.py:175-184): wrap to (-pi, pi]."""
    value = float(value) % TWO_PI
    if value > math.pi:
        value -= TWO_PI
    return value


def heading_relative_to(h, other):
    """``Heading.relative_to`` (coordinates.py:227-239): wrapped twice."""
    return wrap_heading(wrap_heading(h - other))


def vec_to_radians(v):
    """math.py:256-277 (0 rad = +y, counter-clockwise, result in [0, 2pi))."""
    x, y = v
    r = math.atan2(abs(y), abs(x))
    if x < 0:
        if y < 0:
            return (r + 0.5 * math.pi) % TWO_PI
        return (0.5 * math.pi - r) % TWO_PI
    elif y < 0:
        return (1.5 * math.pi - r) % TWO_PI
    return (r - 0.5 * math.pi) % TWO_PI


def radians_to_vec(radians):
    """math.py:247-253."""
    angle = (radians + math.pi * 0.5) % TWO_PI
    return np.array((math.cos(angle), math.sin(angle)))


def quat_from_angle(angle):
    """``fast_quaternion_from_angle`` math.py:97-106 -> (x, y, z, w)."""
    half = angle * 0.5
    return np.array([0, 0, math.sin(half), math.cos(half)])


def yaw_from_quat(q):
    """math.py:78-94, (x, y, z, w) order."""
    siny_cosp = 2 * (q[0] * q[1] + q[3] * q[2])
    cosy_cosp = q[3] ** 2 + q[0] ** 2 - q[1] ** 2 - q[2] ** 2
    return np.arctan2(siny_cosp, cosy_cosp)


def pose_heading_from_angle(angle):
    """What ``Pose(orientation=fast_quaternion_from_angle(a)).heading`` evaluates to
    (coordinates.py:394-403): the angle goes through the quaternion and back."""
    return wrap_heading(yaw_from_quat(quat_from_angle(angle)))


def min_angles_difference_signed(first, second):
    """math.py:447-449."""
    return ((first - second) + math.pi) % TWO_PI - math.pi


def signed_dist_to_line(point, line_point, line_dir_vec):
    """math.py:163-185 (negative = right of the directed line)."""
    p = np.array(point[:2])
    p1 = line_point
    p2 = line_point + line_dir_vec
    u = abs(line_dir_vec[1] * p[0] - line_dir_vec[0] * p[1] + p2[0] * p1[1] - p2[1] * p1[0])
    d = u / np.linalg.norm(line_dir_vec)
    line_normal = np.array([-line_dir_vec[1], line_dir_vec[0]])
    _sign = np.sign(np.dot(p - p1, line_normal))
    return d * _sign


def lerp(a, b, p):
    """math.py:206-216."""
    assert 0 <= p <= 1
    return a * (1.0 - p) + b * p


def low_pass_filter(input_value, previous_filter_state, filter_constant, time_step, lower_bound=-1, raw_value=0):
    """math.py:219-244."""
    previous_filter_state += time_step * filter_constant * (input_value - previous_filter_state)
    previous_filter_state = np.clip(previous_filter_state + raw_value, lower_bound, 1)
    return previous_filter_state


def inplace_unwrap(wp_array):
    """math.py:537-550 (numpy unwrap without the copy)."""
    p = np.asarray(wp_array)
    dd = np.subtract(p[1:], p[:-1])
    ddmod = np.mod(dd + math.pi, TWO_PI) - math.pi
    np.copyto(ddmod, math.pi, where=(ddmod == -math.pi) & (dd > 0))
    ph_correct = ddmod - dd
    np.copyto(ph_correct, 0, where=abs(dd) < math.pi)
    p[1:] += ph_correct.cumsum(axis=-1)
    return p


def rotate_around_point(point, radians, origin=(0, 0)):
    """math.py:436-444 (note: clockwise-positive formula)."""
    x, y = point
    ox, oy = origin
    qx = ox + math.cos(radians) * (x - ox) + math.sin(radians) * (y - oy)
    qy = oy + -math.sin(radians) * (x - ox) + math.cos(radians) * (y - oy)
    return np.array([qx, qy])


def mult_quat(q1, q2):
    """math.py:109-119 (index 0 treated as the scalar part)."""
    q3 = np.copy(q1)
    q3[0] = q1[0] * q2[0] - q1[1] * q2[1] - q1[2] * q2[2] - q1[3] * q2[3]
    q3[1] = q1[0] * q2[1] + q1[1] * q2[0] + q1[2] * q2[3] - q1[3] * q2[2]
    q3[2] = q1[0] * q2[2] - q1[1] * q2[3] + q1[2] * q2[0] + q1[3] * q2[1]
    q3[3] = q1[0] * q2[3] + q1[1] * q2[2] - q1[2] * q2[1] + q1[3] * q2[0]
    return q3


def rotate_quat(quat, vect):
    """math.py:122-133."""
    vect = np.append([0], vect)
    norm_vect = np.linalg.norm(vect)
    vect /= norm_vect
    quat_ = np.append(quat[0], -quat[1:])
    res = mult_quat(quat, mult_quat(vect, quat_)) * norm_vect
    return res[1:]


def position_to_ego_frame(position, ego_position, ego_heading):
    """math.py:464-487."""
    m = np.eye(3)
    m[0, 0] = np.cos(-ego_heading)
    m[0, 1] = -np.sin(-ego_heading)
    m[1, 0] = np.sin(-ego_heading)
    m[1, 1] = np.cos(-ego_heading)
    rel = np.asarray(position) - np.asarray(ego_position)
    return np.matmul(m, rel.T).T.tolist()


def round_param_for_dt(dt):
    """math.py:553-563."""
    strep = np.format_float_positional(dt)
    decimal = strep.find(".")
    if decimal >= len(strep) - 1:
        return 1 - decimal
    return len(strep) - decimal - 1


# ---- polyline helpers: in-tree twins of sumolib.geomhelper (math.py:280-433) ----
def is_close(a, b, rel_tol=1e-09, abs_tol=0.0):
    return abs(a - b) <= max(rel_tol * max(abs(a), abs(b)), abs_tol)


def euclidean_distance(p1, p2):
    dx = p1[0] - p2[0]
    dy = p1[1] - p2[1]
    return math.sqrt(dx * dx + dy * dy)


def position_at_offset(p1, p2, offset):
    """math.py:300-314."""
    if is_close(offset, 0.0):
        return p1
    dist = euclidean_distance(p1, p2)
    if is_close(dist, offset):
        return p2
    return p1[0] + (p2[0] - p1[0]) * (offset / dist), p1[1] + (p2[1] - p1[1]) * (offset / dist)


def position_at_shape_offset(shape, offset):
    """math.py:333-345."""
    seen_length = 0
    curr = shape[0]
    for next_p in shape[1:]:
        next_length = euclidean_distance(curr, next_p)
        if seen_length + next_length > offset:
            return position_at_offset(curr, next_p, offset - seen_length)
        seen_length += next_length
        curr = next_p
    return shape[-1]


def line_offset_with_minimum_distance_to_point(point, line_start, line_end, perpendicular=False):
    """math.py:348-367."""
    p, p1, p2 = point, line_start, line_end
    d = euclidean_distance(p1, p2)
    u = ((p[0] - p1[0]) * (p2[0] - p1[0])) + ((p[1] - p1[1]) * (p2[1] - p1[1]))
    if d == 0.0 or u < 0.0 or u > d * d:
        if perpendicular:
            return -1
        if u < 0.0:
            return 0.0
        return d
    return u / d


def polygon_offset_with_minimum_distance_to_point(point, polygon):
    """math.py:370-390."""
    p, s = point, polygon
    seen = 0
    min_dist = 1e400
    min_offset = -1
    for i in range(len(s) - 1):
        p_offset = line_offset_with_minimum_distance_to_point(p, s[i], s[i + 1])
        dist = min_dist if p_offset == -1 else euclidean_distance(p, position_at_offset(s[i], s[i + 1], p_offset))
        if dist < min_dist:
            min_dist = dist
            min_offset = p_offset + seen
        seen += euclidean_distance(s[i], s[i + 1])
    return min_offset


def distance_point_to_line(point, line_start, line_end, perpendicular=False):
    """math.py:393-411."""
    p1, p2 = line_start, line_end
    offset = line_offset_with_minimum_distance_to_point(point, line_start, line_end, perpendicular)
    if offset == -1:
        return -1
    if offset == 0:
        return euclidean_distance(point, p1)
    u = offset / euclidean_distance(line_start, line_end)
    intersection = (p1[0] + u * (p2[0] - p1[0]), p1[1] + u * (p2[1] - p1[1]))
    return euclidean_distance(point, intersection)


def distance_point_to_polygon(point, polygon, perpendicular=False):
    """math.py:414-433."""
    p, s = point, polygon
    min_dist = None
    for i in range(len(s) - 1):
        dist = distance_point_to_line(p, s[i], s[i + 1], perpendicular)
        if dist == -1 and perpendicular and i != 0:
            dist = euclidean_distance(point, s[i])
        if dist != -1:
            if min_dist is None or dist < min_dist:
                min_dist = dist
    if min_dist is not None:
        return min_dist
    return -1
