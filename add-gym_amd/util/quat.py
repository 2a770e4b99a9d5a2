"""Host-side (torch, CPU or GPU tensors) quaternion helpers, wxyz, used only by the one-time
clip ingest (anim/).  Same primitive torch ops as the reference's util/torch_util.py (cited) so
that the step tables come out bit-identical to the reference's CPU path."""
import torch


def unit(x, eps=1e-9):
    # torch_util.py:12-14
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def conj(q):
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)  # torch_util.py:35-36


def positive(q):
    return (1 - 2 * (q[..., :1] < 0).float()) * q  # torch_util.py:40-44


def mul(a, b):
    # torch_util.py:48-61
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def axis_angle(q):
    # torch_util.py:74-94
    q = positive(q)
    length = torch.norm(q[..., 1:], dim=-1, p=2)
    angle = 2.0 * torch.atan2(length, q[..., 0])
    axis = q[..., 1:] / length.unsqueeze(-1)
    fallback = torch.zeros_like(axis)
    fallback[..., -1] = 1
    ok = length > 1e-5
    return torch.where(ok.unsqueeze(-1), axis, fallback), torch.where(ok, angle, torch.zeros_like(angle))


def exp_map(q):
    axis, angle = axis_angle(q)
    return angle.unsqueeze(-1) * axis  # torch_util.py:198-210


def from_axis_angle(axis, angle):
    # torch_util.py:186-195
    half = (angle / 2).unsqueeze(-1)
    return unit(torch.cat([half.cos(), unit(axis) * half.sin()], dim=-1))


def normalized(q):
    return unit(positive(q))  # torch_util.py:294-297


def slerp(q0, q1, t):
    # torch_util.py:300-323
    c = torch.sum(q0 * q1, dim=-1)
    q1 = torch.where((c < 0).unsqueeze(-1), -q1, q1)
    c = torch.abs(c).unsqueeze(-1)
    half = torch.acos(c)
    s = torch.sqrt(1.0 - c * c)
    t = t.unsqueeze(-1)
    out = (torch.sin((1 - t) * half) / s) * q0 + (torch.sin(t * half) / s) * q1
    out = torch.where(torch.abs(s) < 0.001, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(torch.abs(c) >= 1, q0, out)


def twist_angle(q, axis):
    # torch_util.py:375-406
    proj = torch.sum(axis * q[..., 1:], dim=-1)
    tw = q.clone()
    tw[..., 1:] = proj.unsqueeze(-1) * axis
    ax, ang = axis_angle(normalized(tw))
    ang = ang.clone()
    ang[torch.sum(axis * ax, dim=-1) < 0] *= -1
    return ang
