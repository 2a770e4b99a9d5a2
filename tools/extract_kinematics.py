#!/usr/bin/env python3
"""Derive add-gym_amd/assets/g1_29_kinematics.xml from the reference's G1 MJCF (assets/g1_description/g1_29.xml): the data
the hot path and the rigid-body engine read, and nothing else.

  kinematics  body tree, hinge names / axes / ranges, body offsets                      (hot path, motion ingest)
  dynamics    <inertial> of every body; per joint: damping, armature (class defaults resolved) and the actuator force range
  collision   the MJCF's own primitive collision geoms (the four 5 mm spheres under each foot, the shoulder cylinders as
              two cap spheres) and, for bodies whose collision geom is a mesh, a handful of proxy points: the mesh's support
              points along 14 directions (+-x, +-y, +-z and the 8 diagonals), i.e. the vertices a ground plane can touch
              first (taken over the union of the body's collision meshes); merged within 1 cm.  Ground-plane contact only needs those.

Meshes, visual geoms, actuators, sensors and sites are dropped.  Build-container tooling (reads /root/reference, incl. the STL
files); the output is a data asset."""
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/g1_description/g1_29.xml"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "add-gym_amd", "assets", "g1_29_kinematics.xml")
mesh_dir = os.path.join(os.path.dirname(src), "meshes")

root = ET.parse(src).getroot()
joint_defaults = {}
for d in root.find("default").findall("default"):
    j = d.find("joint")
    if j is not None:
        joint_defaults[d.attrib["class"]] = dict(j.attrib)
mesh_files = {m.attrib["name"]: m.attrib["file"] for m in root.find("asset").findall("mesh")}


def read_stl(path):
    with open(path, "rb") as f:
        raw = f.read()
    (n,) = struct.unpack_from("<I", raw, 80)
    assert len(raw) == 84 + 50 * n, f"{path}: not a binary STL"
    tri = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return tri["v"].reshape(-1, 3).astype(np.float64)


def quat_rot(q, v):
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return v @ R.T


DIRS = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]] +
                [[a, b, c] for a in (1, -1) for b in (1, -1) for c in (1, -1)], np.float64)
DIRS /= np.linalg.norm(DIRS, axis=1, keepdims=True)


def fvec(s, n=None):
    v = np.array([float(x) for x in s.split()], np.float64)
    assert n is None or len(v) == n
    return v


def collision_points(node):
    """[(pos3, radius)] in the body frame."""
    pts, verts = [], []
    for g in node.findall("geom"):
        if g.attrib.get("contype") == "0" and g.attrib.get("conaffinity") == "0":
            continue  # visual only
        typ = g.attrib.get("type", "sphere")
        pos = fvec(g.attrib.get("pos", "0 0 0"), 3)
        quat = fvec(g.attrib.get("quat", "1 0 0 0"), 4)
        if typ == "sphere":
            pts.append((pos, float(g.attrib["size"].split()[0])))
        elif typ == "cylinder":  # size = radius, half length along the geom's z axis
            r, hl = fvec(g.attrib["size"], 2)
            for s in (-1.0, 1.0):
                pts.append((pos + quat_rot(quat, np.array([0, 0, s * hl])), r))
        elif typ == "mesh":
            verts.append(quat_rot(quat, read_stl(os.path.join(mesh_dir, mesh_files[g.attrib["mesh"]]))) + pos)
        else:
            raise ValueError(f"unsupported collision geom type {typ}")
    if verts:  # support points of the union of the body's collision meshes
        v = np.concatenate(verts, axis=0)
        pts += [(p, 0.0) for p in v[np.argmax(v @ DIRS.T, axis=0)]]
    merged = []
    for p, r in pts:
        if not any(np.linalg.norm(p - q) < 0.01 and abs(r - s) < 1e-9 for q, s in merged):
            merged.append((p, r))
    return merged


def fmt(v):
    return " ".join(f"{x:.9g}" for x in np.atleast_1d(v))


def conv(node):
    attrs = {k: node.attrib[k] for k in ("name", "pos", "quat") if k in node.attrib}
    out = ET.Element("body", attrs)
    ine = node.find("inertial")
    ET.SubElement(out, "inertial", {k: ine.attrib[k] for k in ("pos", "quat", "mass", "diaginertia") if k in ine.attrib})
    for j in node.findall("joint"):
        a = {k: j.attrib[k] for k in ("name", "type", "axis", "range") if k in j.attrib}
        if j.attrib.get("type", "hinge") == "hinge":
            d = joint_defaults.get(j.attrib.get("class", ""), {})
            a["damping"] = j.attrib.get("damping", d.get("damping", "0"))
            a["armature"] = j.attrib.get("armature", d.get("armature", "0"))
            a["frictionloss"] = j.attrib.get("frictionloss", d.get("frictionloss", "0"))
            a["actuatorfrcrange"] = j.attrib["actuatorfrcrange"]
        ET.SubElement(out, "joint", a)
    for p, r in collision_points(node):
        ET.SubElement(out, "geom", {"type": "sphere", "size": fmt(r), "pos": fmt(p)})
    for c in node.findall("body"):
        out.append(conv(c))
    return out


m = ET.Element("mujoco", {"model": "g1_29dof_kinematics"})
wb = ET.SubElement(m, "worldbody")
wb.append(conv(root.find("worldbody").find("body")))
ET.indent(m)
ET.ElementTree(m).write(dst)
npts = len(m.findall(".//geom"))
print("wrote", dst, f"({npts} collision points)")
