"""Constant tables of addhip_rigid_step (include/addhip.h: addhip_rigid_model_t) from the robot's MJCF asset: body tree,
inertials, joint dynamics and collision spheres.  Host-side, one-time (numpy); the tables are uploaded once and only the PD
gain columns change afterwards (BaseEntity.set_dofs_kp / set_dofs_kv, base_engine.py:216-240).

Links and dofs keep the engine API's breadth-first numbering; the kernel walks the tree depth-first, so the rows of `body` /
`topo` are in depth-first pre-order and carry the breadth-first indices they map back to."""
import xml.etree.ElementTree as ET

import numpy as np

from .. import _lib as L


def _quat_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


class RigidModelTables:
    MAX_BRANCH = 4

    def __init__(self, xml_path):
        body0 = ET.parse(xml_path).getroot().find("worldbody").find("body")
        nodes, parent = [], []
        todo = [(body0, -1)]
        while todo:  # breadth first == link / dof order of the engine API (kin_char_model.py)
            node, par = todo.pop(0)
            me = len(nodes)
            nodes.append(node)
            parent.append(par)
            todo += [(c, me) for c in node.findall("body")]
        nb = len(nodes)
        if nb != L.NUM_DOF + 1:
            raise ValueError(f"the packed state rows hold {L.NUM_DOF} hinge dofs; {xml_path} has {nb - 1}")
        children = [[j for j in range(nb) if parent[j] == i] for i in range(nb)]
        order = []

        def visit(i):
            order.append(i)
            for c in children[i]:
                visit(c)

        visit(0)
        pos_of = {b: k for k, b in enumerate(order)}  # breadth-first index -> traversal index
        self.names = [n.attrib["name"] for n in nodes]
        self.bfs_of_traversal = order
        body = np.zeros((nb, L.RIGID_BODY_W), np.float32)
        topo = np.zeros((nb, L.RIGID_TOPO_W), np.int32)
        points = []
        slots = 0
        f = lambda s: np.array([float(x) for x in s.split()], np.float64)
        for k, b in enumerate(order):
            node = nodes[b]
            r = f(node.attrib.get("pos", "0 0 0"))
            R = _quat_to_mat(f(node.attrib.get("quat", "1 0 0 0")))
            ine = node.find("inertial")
            mass = float(ine.attrib["mass"])
            com = f(ine.attrib.get("pos", "0 0 0"))
            Rq = _quat_to_mat(f(ine.attrib.get("quat", "1 0 0 0")))
            Ic = Rq @ np.diag(f(ine.attrib["diaginertia"])) @ Rq.T
            cx = _skew(com)
            Io = Ic + mass * (cx @ cx.T)  # about the body origin
            row = body[k]
            row[0:3] = r
            row[3:12] = R.reshape(-1)
            row[12] = mass
            row[13:16] = mass * com
            row[16:22] = [Io[0, 0], Io[0, 1], Io[0, 2], Io[1, 1], Io[1, 2], Io[2, 2]]
            axis = 0
            if b > 0:
                hinges = [j for j in node.findall("joint") if j.attrib.get("type", "hinge") == "hinge"]
                if len(hinges) != 1:
                    raise ValueError(f"body {self.names[b]}: exactly one hinge joint per body is supported")
                j = hinges[0]
                a = f(j.attrib["axis"])
                if sorted(np.abs(a)) != [0, 0, 1] or a.max() != 1:
                    raise ValueError(f"joint {j.attrib['name']}: hinge axes must be +x, +y or +z of the body frame")
                axis = int(np.argmax(a))
                lo, hi = f(j.attrib["range"])
                frc = max(abs(x) for x in f(j.attrib.get("actuatorfrcrange", "-1e9 1e9")))
                row[22:27] = [lo, hi, float(j.attrib.get("damping", 0)), float(j.attrib.get("armature", 0)), frc]
            nch = len(children[b])
            slot = -1
            if nch > 1:
                slot = slots
                slots += 1
            elif nch == 1 and pos_of[children[b][0]] != k + 1:
                raise AssertionError("depth-first order broken")
            pts = [(f(g.attrib["pos"]), float(g.attrib["size"].split()[0])) for g in node.findall("geom")]
            row[29] = max([float(np.linalg.norm(p)) + rad for p, rad in pts], default=0.0)  # bounding radius of the collision spheres
            topo[k] = [pos_of[parent[b]] if b > 0 else -1, axis, b - 1, nch, slot, len(points), len(pts), b]
            points += [[p[0], p[1], p[2], rad] for p, rad in pts]
        if slots > self.MAX_BRANCH:
            raise ValueError(f"at most {self.MAX_BRANCH} bodies with more than one child are supported")
        self.body, self.topo = body, topo
        self.points = np.asarray(points, np.float32).reshape(-1, 4)
        self.num_bodies, self.num_points = nb, len(points)
        self.total_mass = float(body[:, 12].sum())

    MAX_CHAIN_STEPS = 10

    def chain_table(self):
        """The depth-first body order cut into chains (maximal runs whose parent is the previous body) for the four-lanes-per-env
        kernel: int32 [4][16] = len, start step, attach lane (-1: the root), body indices -- or None when the tree needs more than
        four chains or more than MAX_CHAIN_STEPS steps (the one-lane kernel handles any tree).  G1: left leg | right leg |
        waist + left arm | right arm."""
        runs = []
        for k in range(1, self.num_bodies):
            if runs and self.topo[k, 0] == k - 1 and runs[-1][-1] == k - 1:
                runs[-1].append(k)
            else:
                runs.append([k])
        if len(runs) > 4:
            return None
        tab = np.zeros((4, 16), np.int32)
        tab[:, 2] = -1
        where = {}  # body -> (lane, step)
        for lane, run in enumerate(runs):
            par = int(self.topo[run[0], 0])
            if par == 0:
                start, attach = 0, -1
            else:
                if par not in where:
                    return None
                attach, pstep = where[par]
                start = pstep + 1
            if start + len(run) > self.MAX_CHAIN_STEPS:
                return None
            tab[lane, 0], tab[lane, 1], tab[lane, 2] = len(run), start, attach
            tab[lane, 3:3 + len(run)] = run
            for i, b in enumerate(run):
                where[b] = (lane, start + i)
        return tab

    def set_gains(self, kp, kv):
        """kp / kv per dof in breadth-first dof order (29 values) -> columns 27 / 28 of the traversal-ordered rows."""
        for k in range(1, self.num_bodies):
            dof = self.topo[k, 2]
            self.body[k, 27], self.body[k, 28] = float(kp[dof]), float(kv[dof])

    def link_mask(self, link_indices):
        m = 0
        for b in link_indices:
            m |= 1 << int(b)
        return m
