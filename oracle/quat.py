"""Quaternion (wxyz) helpers in fp32 numpy -- oracle restatement of
add_gym/util/torch_util.py (reference file:line cited per function).  Test infrastructure."""
import numpy as np

F = np.float32


def _f(x):
    return np.asarray(x, dtype=F)


def normalize(x, eps=1e-9):
    # torch_util.py:12-14: x / clamp(||x||, min=eps)
    x = _f(x)
    n = np.sqrt(np.sum(x * x, axis=-1, dtype=F)).astype(F)
    return x / np.maximum(n, F(eps))[..., None]


def quat_conjugate(q):
    # torch_util.py:35-36
    q = _f(q)
    return np.concatenate([q[..., 0:1], -q[..., 1:]], axis=-1)


def quat_pos(q):
    # torch_util.py:40-44: flip sign when w < 0
    q = _f(q)
    z = (q[..., 0:1] < 0).astype(F)
    return (F(1) - F(2) * z) * q


def quat_mul(a, b):
    # torch_util.py:48-61
    a, b = _f(a), _f(b)
    w1, x1, y1, z1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    w2, x2, y2, z2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2
    z = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2
    return np.stack([w, x, y, z], axis=-1).astype(F)


def _cross(a, b):
    return np.stack(
        [
            a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
            a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
            a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0],
        ],
        axis=-1,
    ).astype(F)


def quat_rotate(q, v):
    # torch_util.py:65-70: v + w*t + qv x t, t = 2*(qv x v)
    q, v = _f(q), _f(v)
    qw, qv = q[..., 0:1], q[..., 1:]
    t = F(2) * _cross(qv, v)
    return (v + qw * t + _cross(qv, t)).astype(F)


def quat_to_axis_angle(q):
    # torch_util.py:74-94
    q = quat_pos(q)
    length = np.sqrt(np.sum(q[..., 1:] * q[..., 1:], axis=-1, dtype=F)).astype(F)
    angle = F(2) * np.arctan2(length, q[..., 0]).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = q[..., 1:] / length[..., None]
    mask = length > F(1e-5)
    angle = np.where(mask, angle, F(0)).astype(F)
    default_axis = np.zeros_like(axis)
    default_axis[..., 2] = 1
    axis = np.where(mask[..., None], axis, default_axis).astype(F)
    return axis, angle


def quat_to_exp_map(q):
    # torch_util.py:205-210
    axis, angle = quat_to_axis_angle(q)
    return (angle[..., None] * axis).astype(F)


def axis_angle_to_quat(axis, angle):
    # torch_util.py:186-195: [cos(t/2), normalize(axis)*sin(t/2)] then unit-normalised
    axis, angle = _f(axis), _f(angle)
    theta = (angle / F(2))[..., None]
    xyz = normalize(axis) * np.sin(theta).astype(F)
    w = np.cos(theta).astype(F)
    return normalize(np.concatenate([w, xyz], axis=-1))


def quat_to_tan_norm(q):
    # torch_util.py:231-242: [rot(q, x), rot(q, z)]
    q = _f(q)
    ex = np.zeros(q.shape[:-1] + (3,), F)
    ex[..., 0] = 1
    ez = np.zeros(q.shape[:-1] + (3,), F)
    ez[..., 2] = 1
    return np.concatenate([quat_rotate(q, ex), quat_rotate(q, ez)], axis=-1)


def quat_diff(q0, q1):
    # torch_util.py:275-278
    return quat_mul(q1, quat_conjugate(q0))


def quat_diff_angle(q0, q1):
    # torch_util.py:281-285
    return quat_to_axis_angle(quat_diff(q0, q1))[1]


def quat_normalize(q):
    # torch_util.py:294-297
    return normalize(quat_pos(q))


def slerp(q0, q1, t):
    # torch_util.py:300-323 (t has one dim fewer than q)
    q0, q1, t = _f(q0), _f(q1), _f(t)
    cos_h = np.sum(q0 * q1, axis=-1, dtype=F).astype(F)
    q1 = np.where((cos_h < 0)[..., None], -q1, q1)
    cos_h = np.abs(cos_h)[..., None]
    with np.errstate(invalid="ignore", divide="ignore"):
        half = np.arccos(cos_h).astype(F)
        sin_h = np.sqrt(F(1) - cos_h * cos_h).astype(F)
        tt = t[..., None]
        ra = np.sin((F(1) - tt) * half).astype(F) / sin_h
        rb = np.sin(tt * half).astype(F) / sin_h
        new_q = ra * q0 + rb * q1
    new_q = np.where(np.abs(sin_h) < F(0.001), F(0.5) * q0 + F(0.5) * q1, new_q)
    new_q = np.where(np.abs(cos_h) >= 1, q0, new_q)
    return new_q.astype(F)


def calc_heading(q):
    # torch_util.py:326-334
    q = _f(q)
    ex = np.zeros(q.shape[:-1] + (3,), F)
    ex[..., 0] = 1
    d = quat_rotate(q, ex)
    return np.arctan2(d[..., 1], d[..., 0]).astype(F)


def calc_heading_quat_inv(q):
    # torch_util.py:348-356
    h = calc_heading(q)
    axis = np.zeros(np.shape(h) + (3,), F)
    axis[..., 2] = 1
    return axis_angle_to_quat(axis, -h)


def quat_twist_angle(q, twist_axis):
    # torch_util.py:386-406
    q, twist_axis = _f(q), _f(twist_axis)
    p = np.sum(twist_axis * q[..., 1:], axis=-1, dtype=F).astype(F)
    twist = q.copy()
    twist[..., 1:] = p[..., None] * twist_axis
    twist = quat_normalize(twist)
    axis, angle = quat_to_axis_angle(twist)
    dot = np.sum(twist_axis * axis, axis=-1, dtype=F)
    return np.where(dot < 0, -angle, angle).astype(F)
