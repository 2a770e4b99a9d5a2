#!/usr/bin/env python3
"""Derive add-gym_amd/assets/g1_29_kinematics.xml from the reference's G1 MJCF: only the
kinematic data the hot path reads (body tree, hinge names/axes/ranges, body offsets).  Meshes,
inertias, geoms, actuators and sensors are dropped -- physics is out of scope.  Build-container
tooling (reads /root/reference); the output is a data asset."""
import os
import sys
import xml.etree.ElementTree as ET

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/g1_description/g1_29.xml"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "add-gym_amd", "assets", "g1_29_kinematics.xml")


def conv(node):
    attrs = {k: node.attrib[k] for k in ("name", "pos", "quat") if k in node.attrib}
    out = ET.Element("body", attrs)
    for j in node.findall("joint"):
        ET.SubElement(out, "joint", {k: j.attrib[k] for k in ("name", "type", "axis", "range") if k in j.attrib})
    for c in node.findall("body"):
        out.append(conv(c))
    return out


root = ET.parse(src).getroot()
m = ET.Element("mujoco", {"model": "g1_29dof_kinematics"})
wb = ET.SubElement(m, "worldbody")
wb.append(conv(root.find("worldbody").find("body")))
ET.indent(m)
ET.ElementTree(m).write(dst)
print("wrote", dst)
