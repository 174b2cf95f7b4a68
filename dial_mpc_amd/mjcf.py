"""MJCF -> flat model constants ("model compiler").

The reference obtains its model through ``brax.io.mjcf.load`` -> MuJoCo's C compiler
(``dial_mpc/envs/base_env.py:15-29``, ``dial_mpc/envs/unitree_go2_env.py:95-99``).  MuJoCo is
not a dependency here, so this module implements the subset of the MJCF compiler the hot path
needs: ``<include>``, nested ``<default>`` classes / ``childclass``,
``<compiler angle autolimits>``, ``<option>`` + ``<flag>``, bodies / inertials / joints (free,
hinge, slide) / geoms (plane, sphere, capsule incl. ``fromto``) / sites / actuators (motor,
position) / keyframes / ``<contact><exclude>``, the static contact list (plane-sphere, plane-capsule,
sphere-capsule, capsule-capsule) with MuJoCo's parameter mixing rules (priority, condim, solmix,
friction), pyramidal and elliptic cones, and the quantities MuJoCo derives at ``qpos0``: ``body_invweight0``,
``dof_invweight0`` and ``stat.meaninertia``.

Everything is fp64 NumPy on the host; the result is a dict whose keys are the field names of
``struct dial_model`` (include/dial_mpc.h) plus name tables under ``"names"``.
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from typing import Any, Dict, List, Optional

import numpy as np

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX = 0, 2, 3, 6
CON_PLANE_SPHERE, CON_PLANE_CAPSULE_P, CON_PLANE_CAPSULE_N, CON_SPHERE_CAPSULE, CON_CAPSULE_CAPSULE = 0, 1, 2, 3, 4
CON_PLANE_BOX, CON_SPHERE_BOX, CON_CAPSULE_BOX, CON_BOX_BOX = 5, 6, 7, 8
MJ_MINVAL = 1e-15

_GEOM_TYPES = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4,
               "cylinder": 5, "box": 6, "mesh": 7}


# ------------------------------------------------------------------ small math (fp64)
def quat_mul(u, v):
    return np.array([
        u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3],
        u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2],
        u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1],
        u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0],
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def rotate(v, q):
    return quat_to_mat(q) @ np.asarray(v, dtype=np.float64)


def _normalize(q):
    q = np.asarray(q, dtype=np.float64)
    n = np.linalg.norm(q)
    return q / n if n > 0 else q


def _z2quat(vec):
    """Quaternion rotating the z axis onto ``vec`` (MuJoCo's mjuu_z2quat)."""
    vec = _normalize(vec)
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    return np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)])


def _floats(s: str) -> np.ndarray:
    return np.array([float(t) for t in s.split()], dtype=np.float64)


# ------------------------------------------------------------------ XML loading
def _load_tree(path: str) -> ET.Element:
    root = ET.parse(path).getroot()
    base = os.path.dirname(os.path.abspath(path))

    def expand(elem: ET.Element):
        out = []
        for ch in list(elem):
            if ch.tag == "include":
                inc = _load_tree(os.path.join(base, ch.attrib["file"]))
                out.extend(list(inc))
            else:
                expand(ch)
                out.append(ch)
        elem[:] = out

    expand(root)
    return root


class _Defaults:
    """Nested <default> classes: class name -> {element tag -> attribute dict}."""

    _ACT_TAGS = ("motor", "position", "general", "velocity")

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}

    def _ingest(self, elem: ET.Element, name: str, parent: Optional[str]):
        cur = {t: dict(a) for t, a in self.classes.get(parent, {}).items()} if parent else {}
        cur = {t: dict(a) for t, a in cur.items()}
        if name in self.classes and parent is None:
            for t, a in self.classes[name].items():
                cur.setdefault(t, {}).update(a)
        for ch in elem:
            if ch.tag == "default":
                continue
            cur.setdefault(ch.tag, {}).update(ch.attrib)
        self.classes[name] = cur
        for ch in elem:
            if ch.tag == "default":
                self._ingest(ch, ch.attrib["class"], name)

    def load(self, root: ET.Element):
        for d in root.findall("default"):
            self._ingest(d, d.attrib.get("class", "main"), None)

    def resolve(self, elem: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass or "main")
        if cls not in self.classes:
            raise ValueError(f"unknown default class {cls!r}")
        tag = "joint" if elem.tag == "freejoint" else elem.tag
        attrs = dict(self.classes[cls].get(tag, {}))
        if elem.tag == "freejoint":
            attrs = {}
        attrs.update({k: v for k, v in elem.attrib.items() if k != "class"})
        return attrs


def _frame_quat(attrs: Dict[str, str], angle_scale: float) -> np.ndarray:
    if "quat" in attrs:
        return _normalize(_floats(attrs["quat"]))
    if "euler" in attrs:
        e = _floats(attrs["euler"]) * angle_scale
        q = np.array([1.0, 0, 0, 0])
        for i, ax in enumerate(np.eye(3)):  # default eulerseq "xyz" (intrinsic)
            qi = np.concatenate([[math.cos(e[i] / 2)], ax * math.sin(e[i] / 2)])
            q = quat_mul(q, qi)
        return q
    for bad in ("axisangle", "xyaxes", "zaxis"):
        if bad in attrs:
            raise NotImplementedError(f"orientation attribute {bad!r} is not supported")
    return np.array([1.0, 0.0, 0.0, 0.0])


# ------------------------------------------------------------------ the compiler
def _read_stl(path: str, scale: np.ndarray) -> np.ndarray:
    """Binary STL -> (ntri, 3, 3) float64 triangle vertices (float32 on disk)."""
    with open(path, "rb") as f:
        raw = f.read()
    ntri = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
    if len(raw) < 84 + 50 * ntri:
        raise ValueError(f"{path}: not a binary STL")
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]),
                        count=ntri, offset=84)
    return rec["v"].astype(np.float64) * scale


def _polyhedron_mass_props(tris: np.ndarray):
    """Volume, centre of mass and inertia tensor about the COM (unit density) of a closed,
    outward-oriented triangle surface, by signed tetrahedra against the origin."""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    com = ((a + b + c) * det[:, None]).sum(0) / (24.0 * vol)
    # second moments  int x_i x_j dV  of each tetrahedron (0,a,b,c):  det/120 * (sum_pq v_p v_q^T + sum_p v_p v_p^T)
    S = a + b + c
    C = (np.einsum("ti,tj->ij", S * det[:, None], S)
         + np.einsum("ti,tj->ij", a * det[:, None], a)
         + np.einsum("ti,tj->ij", b * det[:, None], b)
         + np.einsum("ti,tj->ij", c * det[:, None], c)) / 120.0
    C -= vol * np.outer(com, com)  # shift to the COM
    inertia = np.trace(C) * np.eye(3) - C
    return vol, com, inertia


def _mesh_mass_props(path: str, scale: np.ndarray, mode: str = "convex"):
    """Mass properties MuJoCo infers from a mesh geom at unit density.  ``convex`` integrates over the
    convex hull of the vertices (the mesh/inertia default of current MuJoCo releases), ``exact`` over
    the mesh surface itself (needs a watertight, consistently oriented mesh)."""
    tris = _read_stl(path, scale)
    if mode == "convex":
        from scipy.spatial import ConvexHull
        pts = np.unique(tris.reshape(-1, 3), axis=0)
        hull = ConvexHull(pts, qhull_options="Qt")
        t = pts[hull.simplices]
        # orient every facet outward (away from an interior point)
        inside = pts[hull.vertices].mean(0)
        nrm = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        flip = np.einsum("ij,ij->i", nrm, t[:, 0] - inside) < 0
        t[flip] = t[flip][:, ::-1]
        tris = t
    elif mode != "exact":
        raise ValueError(f"unknown mesh inertia mode {mode!r}")
    return _polyhedron_mass_props(tris)



def compile_mjcf(path: str, mesh_inertia: str = "convex") -> Dict[str, Any]:
    root = _load_tree(path)
    comp = {}
    for c in root.findall("compiler"):
        comp.update(c.attrib)
    angle_scale = 1.0 if comp.get("angle", "degree") == "radian" else math.pi / 180.0
    autolimits = comp.get("autolimits", "true") == "true"

    opt = {"timestep": 0.002, "gravity": "0 0 -9.81", "iterations": 100, "ls_iterations": 50,
           "tolerance": 1e-8, "ls_tolerance": 0.01, "impratio": 1.0, "cone": "pyramidal",
           "integrator": "Euler", "solver": "Newton"}
    flags = {"eulerdamp": "enable"}
    for o in root.findall("option"):
        opt.update(o.attrib)
        for f in o.findall("flag"):
            flags.update(f.attrib)
    if opt["integrator"] != "Euler" or opt["solver"] != "Newton":
        raise NotImplementedError("only the Euler integrator and Newton solver are supported")

    dfl = _Defaults()
    dfl.load(root)

    meshes: Dict[str, Dict[str, Any]] = {}
    meshdir = os.path.join(os.path.dirname(os.path.abspath(path)), comp.get("meshdir", ""))
    for asset in root.findall("asset"):
        for me in asset.findall("mesh"):
            f = me.attrib.get("file")
            if f is None:
                continue
            name = me.attrib.get("name", os.path.splitext(os.path.basename(f))[0])
            meshes[name] = dict(file=os.path.join(meshdir, f),
                                scale=_floats(me.attrib["scale"]) if "scale" in me.attrib else np.ones(3),
                                inertia=me.attrib.get("inertia", mesh_inertia))

    bodies: List[Dict[str, Any]] = [dict(name="world", parent=0, pos=np.zeros(3),
                                         quat=np.array([1.0, 0, 0, 0]), ipos=np.zeros(3),
                                         iquat=np.array([1.0, 0, 0, 0]), mass=0.0,
                                         inertia=np.zeros(3), depth=0)]
    joints: List[Dict[str, Any]] = []
    geoms: List[Dict[str, Any]] = []
    sites: List[Dict[str, Any]] = []

    def limited(attrs, key_range, key_limited):
        lim = attrs.get(key_limited, "auto")
        if lim == "true":
            return True
        if lim == "false":
            return False
        return autolimits and key_range in attrs

    def add_geom(e, bid, childclass):
        a = dfl.resolve(e, childclass)
        gtype = _GEOM_TYPES[a.get("type", "sphere")]
        contype = int(a.get("contype", 1))
        conaff = int(a.get("conaffinity", 1))
        size = np.zeros(3)
        if "size" in a:
            s = _floats(a["size"])
            size[: len(s)] = s
        pos = _floats(a["pos"]) if "pos" in a else np.zeros(3)
        quat = _frame_quat(a, angle_scale)
        if "fromto" in a:
            ft = _floats(a["fromto"])
            vec = ft[0:3] - ft[3:6]
            size[1] = np.linalg.norm(vec) / 2
            pos = (ft[0:3] + ft[3:6]) / 2
            quat = _z2quat(vec)
        fr = np.array([1.0, 0.005, 0.0001])
        if "friction" in a:
            f = _floats(a["friction"])
            fr[: len(f)] = f
        solref = np.array([0.02, 1.0])
        if "solref" in a:
            s = _floats(a["solref"])
            solref[: len(s)] = s
        solimp = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
        if "solimp" in a:
            s = _floats(a["solimp"])
            solimp[: len(s)] = s
        geoms.append(dict(name=a.get("name", ""), type=gtype, body=bid, contype=contype,
                          conaffinity=conaff, condim=int(a.get("condim", 3)), size=size, pos=pos,
                          quat=quat, friction=fr, solref=solref, solimp=solimp,
                          margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)),
                          priority=int(a.get("priority", 0)), solmix=float(a.get("solmix", 1)),
                          has_mass=("mass" in a and float(a["mass"]) > 0),
                          mesh=a.get("mesh"), density=float(a.get("density", 1000.0)),
                          mass=float(a["mass"]) if "mass" in a else None,
                          group=int(a.get("group", 0))))

    def walk(elem: ET.Element, parent_id: int, childclass: Optional[str], depth: int):
        for e in elem:
            if e.tag == "geom":
                add_geom(e, parent_id, childclass)
            elif e.tag == "site":
                a = dfl.resolve(e, childclass)
                sites.append(dict(name=a.get("name", ""), body=parent_id,
                                  pos=_floats(a["pos"]) if "pos" in a else np.zeros(3),
                                  quat=_frame_quat(a, angle_scale)))
            elif e.tag == "body":
                cc = e.attrib.get("childclass", childclass)
                bid = len(bodies)
                b = dict(name=e.attrib.get("name", f"body{bid}"), parent=parent_id,
                         pos=_floats(e.attrib["pos"]) if "pos" in e.attrib else np.zeros(3),
                         quat=_frame_quat(e.attrib, angle_scale), depth=depth + 1,
                         ipos=np.zeros(3), iquat=np.array([1.0, 0, 0, 0]), mass=0.0,
                         inertia=np.zeros(3), has_inertial=False)
                bodies.append(b)
                for ch in e:
                    if ch.tag == "inertial":
                        ia = ch.attrib
                        b["has_inertial"] = True
                        b["ipos"] = _floats(ia["pos"])
                        b["mass"] = float(ia["mass"])
                        if "diaginertia" in ia:
                            b["iquat"] = _frame_quat(ia, angle_scale)
                            b["inertia"] = _floats(ia["diaginertia"])
                        elif "fullinertia" in ia:
                            f = _floats(ia["fullinertia"])  # xx yy zz xy xz yz
                            I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                            w, v = np.linalg.eigh(I)
                            order = np.argsort(-w)
                            w, v = w[order], v[:, order]
                            if np.linalg.det(v) < 0:
                                v[:, 2] *= -1
                            b["inertia"] = w
                            b["iquat"] = _mat_to_quat(v)
                        else:
                            raise ValueError("inertial needs diaginertia or fullinertia")
                    elif ch.tag in ("joint", "freejoint"):
                        a = dfl.resolve(ch, cc)
                        jt = "free" if ch.tag == "freejoint" else a.get("type", "hinge")
                        jtype = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE,
                                 "hinge": JNT_HINGE}[jt]
                        if jtype == JNT_BALL:
                            raise NotImplementedError("ball joints are not supported")
                        rng = _floats(a["range"]) if "range" in a else np.zeros(2)
                        if jtype == JNT_HINGE:
                            rng = rng * angle_scale
                        solref = np.array([0.02, 1.0])
                        if "solreflimit" in a:
                            s = _floats(a["solreflimit"])
                            solref[: len(s)] = s
                        solimp = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
                        if "solimplimit" in a:
                            s = _floats(a["solimplimit"])
                            solimp[: len(s)] = s
                        joints.append(dict(
                            name=a.get("name", ""), type=jtype, body=bid,
                            pos=_floats(a["pos"]) if "pos" in a else np.zeros(3),
                            axis=_normalize(_floats(a["axis"])) if "axis" in a
                            else np.array([0.0, 0.0, 1.0]),
                            range=rng,
                            limited=(jtype in (JNT_HINGE, JNT_SLIDE)) and limited(a, "range", "limited"),
                            damping=float(a.get("damping", 0)), armature=float(a.get("armature", 0)),
                            stiffness=float(a.get("stiffness", 0)),
                            frictionloss=float(a.get("frictionloss", 0)),
                            ref=float(a.get("ref", 0)) * (angle_scale if jtype == JNT_HINGE else 1.0),
                            margin=float(a.get("margin", 0)), solref=solref, solimp=solimp))
                        for key, dflt in (("solreffriction", [0.02, 1.0]), ("solimpfriction", [0.9, 0.95, 0.001, 0.5, 2.0])):
                            val = np.array(dflt)
                            if key in a:
                                x = _floats(a[key])
                                val[: len(x)] = x
                            joints[-1][key] = val
                        if joints[-1]["stiffness"] != 0:
                            raise NotImplementedError("joint stiffness not supported")
                        if joints[-1]["frictionloss"] != 0 and jtype in (JNT_FREE, JNT_BALL):
                            # (one friction row per dof of the joint in MuJoCo; fri_dof below records one dof per joint)
                            raise NotImplementedError("frictionloss is supported on slide / hinge joints only")
                walk(e, bid, cc, depth + 1)

    wb = root.find("worldbody")
    # MuJoCo numbers bodies depth-first in document order; joints/geoms/sites are grouped by body.
    walk(wb, 0, None, 0)
    # worldbody may appear several times (scene + included model)
    for extra in root.findall("worldbody")[1:]:
        walk(extra, 0, None, 0)

    nbody = len(bodies)
    # joints must be grouped by body id in body order (they are, by construction of walk()
    # only when siblings don't interleave; sort to be safe and stable)
    joints.sort(key=lambda j: j["body"])
    geoms_all = sorted(geoms, key=lambda g: g["body"])
    sites.sort(key=lambda s: s["body"])
    for bi, b in enumerate(bodies[1:], start=1):
        if b.get("has_inertial", False):
            continue
        # compiler inertiafromgeom="auto": a body without <inertial> takes the mass properties of its
        # geoms (inertiagrouprange 0..5, i.e. visual geoms count).  Mesh geoms only.
        mass, mcom, parts = 0.0, np.zeros(3), []
        for g in geoms_all:
            if g["body"] != bi or not (0 <= g["group"] <= 5):
                continue
            if g["mass"] is not None and g["mass"] == 0.0:
                continue                      # e.g. the Allegro collision primitives (`mass="0"`)
            if g["type"] == GEOM_BOX:
                hx, hy, hz = g["size"]
                vol, com = 8.0 * hx * hy * hz, np.zeros(3)
                I = vol / 3.0 * np.diag([hy * hy + hz * hz, hx * hx + hz * hz, hx * hx + hy * hy])
            elif g["type"] != _GEOM_TYPES["mesh"] or g["mesh"] not in meshes:
                raise NotImplementedError(
                    f"body {b['name']!r}: inertia-from-geom is only implemented for mesh and box geoms")
            else:
                me = meshes[g["mesh"]]
                vol, com, I = _mesh_mass_props(me["file"], me["scale"], me["inertia"])
            gm = g["mass"] if g["mass"] is not None else g["density"] * vol
            if gm <= 0:
                continue
            R = quat_to_mat(g["quat"])
            parts.append((gm, g["pos"] + R @ com, R @ (I * (gm / vol)) @ R.T))
            mass += gm
            mcom += gm * parts[-1][1]
        if mass <= 0:
            raise NotImplementedError(f"body {b['name']!r} has no <inertial> and no massive geoms")
        mcom /= mass
        Ib = np.zeros((3, 3))
        for gm, c, I in parts:
            d = c - mcom
            Ib += I + gm * (d @ d * np.eye(3) - np.outer(d, d))
        w, v = np.linalg.eigh(Ib)
        order = np.argsort(-w)
        w, v = w[order], v[:, order]
        if np.linalg.det(v) < 0:
            v[:, 2] *= -1
        b.update(has_inertial=True, ipos=mcom, mass=mass, inertia=w, iquat=_mat_to_quat(v))

    # ---- qpos / dof addressing
    nq = nv = 0
    for j in joints:
        j["qposadr"], j["dofadr"] = nq, nv
        nq += 7 if j["type"] == JNT_FREE else 1
        nv += 6 if j["type"] == JNT_FREE else 1
    njnt = len(joints)
    body_jntadr = [-1] * nbody
    body_jntnum = [0] * nbody
    body_dofadr = [-1] * nbody
    body_dofnum = [0] * nbody
    for ji, j in enumerate(joints):
        b = j["body"]
        if body_jntnum[b] == 0:
            body_jntadr[b] = ji
            body_dofadr[b] = j["dofadr"]
        body_jntnum[b] += 1
        body_dofnum[b] += 6 if j["type"] == JNT_FREE else 1

    dof_bodyid, dof_jntid, dof_parentid = [], [], []
    dof_armature, dof_damping = [], []
    last_dof_of_body = [-1] * nbody
    for ji, j in enumerate(joints):
        b = j["body"]
        n = 6 if j["type"] == JNT_FREE else 1
        for k in range(n):
            d = j["dofadr"] + k
            if last_dof_of_body[b] >= 0:
                par = last_dof_of_body[b]
            else:
                p = bodies[b]["parent"]
                while p > 0 and last_dof_of_body[p] < 0:
                    p = bodies[p]["parent"]
                par = last_dof_of_body[p] if p > 0 else -1
            dof_parentid.append(par)
            dof_bodyid.append(b)
            dof_jntid.append(ji)
            dof_armature.append(j["armature"])
            dof_damping.append(j["damping"])
            last_dof_of_body[b] = d

    qpos0 = np.zeros(nq)
    for j in joints:
        a = j["qposadr"]
        if j["type"] == JNT_FREE:
            b = bodies[j["body"]]
            qpos0[a:a + 3] = b["pos"]
            qpos0[a + 3:a + 7] = b["quat"]
        else:
            qpos0[a] = j["ref"]

    # subtree end (bodies are in DFS order) and root ids
    subtree_end = list(range(1, nbody + 1))
    for b in range(nbody - 1, 0, -1):
        p = bodies[b]["parent"]
        subtree_end[p] = max(subtree_end[p], subtree_end[b])
    rootid = [0] * nbody
    for b in range(1, nbody):
        p = bodies[b]["parent"]
        rootid[b] = b if p == 0 else rootid[p]

    # ---- collision geoms and the static contact list
    excludes = set()
    names_b = [b["name"] for b in bodies]
    for c in root.findall("contact"):
        for ex in c.findall("exclude"):
            b1, b2 = names_b.index(ex.attrib["body1"]), names_b.index(ex.attrib["body2"])
            excludes.add((min(b1, b2), max(b1, b2)))
    cgeoms = [g for g in geoms_all if (g["contype"] | g["conaffinity"]) != 0 and g["type"] != 7]
    contacts: List[Dict[str, Any]] = []
    pairs = []
    # bodies welded to the world (no joint between them and the world): MuJoCo never collides two of those
    static_body = [True] * nbody
    for b in range(1, nbody):
        static_body[b] = static_body[bodies[b]["parent"]] and body_jntnum[b] == 0
    for i in range(len(cgeoms)):
        for k in range(i + 1, len(cgeoms)):
            g1, g2 = cgeoms[i], cgeoms[k]
            if not ((g1["contype"] & g2["conaffinity"]) | (g2["contype"] & g1["conaffinity"])):
                continue
            b1, b2 = g1["body"], g2["body"]
            if b1 == b2 or (static_body[b1] and static_body[b2]):
                continue
            if b1 != 0 and b2 != 0 and (bodies[b1]["parent"] == b2 or bodies[b2]["parent"] == b1):
                continue
            if (min(b1, b2), max(b1, b2)) in excludes:
                continue
            pairs.append((k, i) if g1["type"] > g2["type"] else (i, k))
    elliptic = opt["cone"] == "elliptic"

    def pair_condim(p):
        g1, g2 = cgeoms[p[0]], cgeoms[p[1]]
        if g1["priority"] != g2["priority"]:          # the higher-priority geom decides (mj_contactParam)
            return (g1 if g1["priority"] > g2["priority"] else g2)["condim"]
        return max(g1["condim"], g2["condim"])

    # MJX groups contacts by collision function, then condim; within a group geom-pair order.  With elliptic
    # cones make_constraint emits the rows condim by condim (1, 3, 4, 6): contacts are listed in that order so
    # that contact order = row order.
    def pair_key(p):
        g1, g2 = cgeoms[p[0]], cgeoms[p[1]]
        if elliptic:
            return (pair_condim(p), g1["type"], g2["type"])
        return (g1["type"], g2["type"], pair_condim(p))
    pairs.sort(key=pair_key)
    # (kind, sub) of the candidate contacts one geom pair yields (a fixed number, like MJX's collision functions)
    kinds = {(GEOM_PLANE, GEOM_SPHERE): ((CON_PLANE_SPHERE, 0),),
             (GEOM_PLANE, GEOM_CAPSULE): ((CON_PLANE_CAPSULE_P, 0), (CON_PLANE_CAPSULE_N, 0)),
             (GEOM_SPHERE, GEOM_CAPSULE): ((CON_SPHERE_CAPSULE, 0),), (GEOM_CAPSULE, GEOM_CAPSULE): ((CON_CAPSULE_CAPSULE, 0),),
             (GEOM_PLANE, GEOM_BOX): tuple((CON_PLANE_BOX, k) for k in range(4)),
             (GEOM_SPHERE, GEOM_BOX): ((CON_SPHERE_BOX, 0),),
             (GEOM_CAPSULE, GEOM_BOX): ((CON_CAPSULE_BOX, 0), (CON_CAPSULE_BOX, 1)),
             (GEOM_BOX, GEOM_BOX): tuple((CON_BOX_BOX, k) for k in range(4))}
    for (i1, i2) in pairs:
        g1, g2 = cgeoms[i1], cgeoms[i2]
        if (g1["type"], g2["type"]) not in kinds:
            raise NotImplementedError(
                f"collision pair type ({g1['type']},{g2['type']}) is not supported")
        if g1["priority"] == g2["priority"]:
            mix = g1["solmix"] / (g1["solmix"] + g2["solmix"])
            fr = np.maximum(g1["friction"], g2["friction"])
            if g1["solref"][0] > 0 and g2["solref"][0] > 0:
                solref = mix * g1["solref"] + (1 - mix) * g2["solref"]
            else:
                solref = np.minimum(g1["solref"], g2["solref"])
            solimp = mix * g1["solimp"] + (1 - mix) * g2["solimp"]
        else:
            gp = g1 if g1["priority"] > g2["priority"] else g2
            fr, solref, solimp = gp["friction"], gp["solref"], gp["solimp"]
        condim = pair_condim((i1, i2))
        if condim not in ((3, 6) if elliptic else (3,)):
            raise NotImplementedError(f"condim {condim} contacts are not supported with the {opt['cone']} cone")
        margin = max(g1["margin"], g2["margin"])
        gap = max(g1["gap"], g2["gap"])
        base = dict(geom1=i1, geom2=i2, body1=g1["body"], body2=g2["body"], dim=condim,
                    friction=np.array([fr[0], fr[0], fr[1], fr[2], fr[2]]), solref=solref,
                    solimp=solimp, margin=margin - gap)
        for kind, sub in kinds[(g1["type"], g2["type"])]:
            contacts.append(dict(base, kind=kind, sub=sub))

    lim_jnt = [ji for ji, j in enumerate(joints) if j["limited"]]
    fri_jnt = [ji for ji, j in enumerate(joints) if j["frictionloss"] > 0]

    # ---- actuators
    acts = []
    jnames = [j["name"] for j in joints]
    for aroot in root.findall("actuator"):
        for e in aroot:
            a = dfl.resolve(e, None)
            if e.tag not in ("motor", "position"):
                raise NotImplementedError(f"actuator type {e.tag!r} is not supported")
            j = joints[jnames.index(a["joint"])]
            gear = _floats(a["gear"])[0] if "gear" in a else 1.0
            lim = limited(a, "ctrlrange", "ctrllimited")
            cr = _floats(a["ctrlrange"]) if "ctrlrange" in a else np.zeros(2)
            if not lim:
                # brax.io.mjcf.load_model rewrites unlimited ranges to (-inf, inf) in place
                # [UPSTREAM-MEMORY, SURVEY B.1]; base_env.py:29,63-65 then clips with them.
                cr = np.array([-np.inf, np.inf])
            acts.append(dict(name=a.get("name", ""), dofadr=j["dofadr"], qposadr=j["qposadr"],
                             gear=gear, ctrllimited=lim, ctrlrange=cr,
                             isposition=(e.tag == "position"), kp=float(a.get("kp", 1.0))))
    nu = len(acts)

    keys = {}
    for kroot in root.findall("keyframe"):
        for k in kroot.findall("key"):
            keys[k.attrib["name"]] = _floats(k.attrib["qpos"])

    m: Dict[str, Any] = dict(
        nq=nq, nv=nv, nu=nu, nbody=nbody, njnt=njnt, ngeom=len(cgeoms), nsite=len(sites),
        ncon=len(contacts), nlim=len(lim_jnt),
        nfri=len(fri_jnt),
        nefc=len(lim_jnt) + len(fri_jnt) + (sum(c["dim"] for c in contacts) if elliptic else 4 * len(contacts)),
        iterations=int(opt["iterations"]), ls_iterations=int(opt["ls_iterations"]),
        eulerdamp=0 if flags.get("eulerdamp", "enable") == "disable" else 1,
        cone=0 if opt["cone"] == "pyramidal" else 1,
        # line-search bracket rule (include/dial_mpc.h): `_in_bracket`, the rule of every MJX release that can run ALL of the
        # reference's envs (its Allegro env needs elliptic cones, i.e. MJX >= 3.1.4; tools/reference_env.txt pins 3.2.7).
        # DIAL_LS_SWAP (MJX <= 3.1.3) stays selectable per model: the tests use it where rollouts are compared one by one
        ls_rule=1,
        timestep=float(opt["timestep"]), gravity=_floats(str(opt["gravity"])),
        tolerance=float(opt["tolerance"]), ls_tolerance=float(opt["ls_tolerance"]),
        impratio=float(opt["impratio"]), meaninertia=0.0,
        body_parent=np.array([b["parent"] for b in bodies]),
        body_jntadr=np.array(body_jntadr), body_jntnum=np.array(body_jntnum),
        body_dofadr=np.array(body_dofadr), body_dofnum=np.array(body_dofnum),
        body_depth=np.array([b["depth"] for b in bodies]),
        body_subtree_end=np.array(subtree_end), body_rootid=np.array(rootid),
        body_pos=np.array([b["pos"] for b in bodies]),
        body_quat=np.array([b["quat"] for b in bodies]),
        body_ipos=np.array([b["ipos"] for b in bodies]),
        body_iquat=np.array([b["iquat"] for b in bodies]),
        body_mass=np.array([b["mass"] for b in bodies]),
        body_inertia=np.array([b["inertia"] for b in bodies]),
        body_invweight0=np.zeros((nbody, 2)),
        jnt_type=np.array([j["type"] for j in joints]),
        jnt_qposadr=np.array([j["qposadr"] for j in joints]),
        jnt_dofadr=np.array([j["dofadr"] for j in joints]),
        jnt_bodyid=np.array([j["body"] for j in joints]),
        jnt_limited=np.array([int(j["limited"]) for j in joints]),
        jnt_pos=np.array([j["pos"] for j in joints]),
        jnt_axis=np.array([j["axis"] for j in joints]),
        jnt_range=np.array([j["range"] for j in joints]),
        jnt_solref=np.array([j["solref"] for j in joints]),
        jnt_solimp=np.array([j["solimp"] for j in joints]),
        jnt_margin=np.array([j["margin"] for j in joints]),
        qpos0=qpos0, key_qpos=qpos0.copy(),
        dof_bodyid=np.array(dof_bodyid), dof_jntid=np.array(dof_jntid),
        dof_parentid=np.array(dof_parentid), dof_armature=np.array(dof_armature),
        dof_damping=np.array(dof_damping), dof_invweight0=np.zeros(nv),
        geom_mjid=np.array([next(k for k, h in enumerate(geoms_all) if h is g) for g in cgeoms], dtype=np.int64),   # MuJoCo / MJX geom id of every collision geom
        geom_type=np.array([g["type"] for g in cgeoms]),
        geom_bodyid=np.array([g["body"] for g in cgeoms]),
        geom_pos=np.array([g["pos"] for g in cgeoms]).reshape(-1, 3),
        geom_quat=np.array([g["quat"] for g in cgeoms]).reshape(-1, 4),
        geom_size=np.array([g["size"] for g in cgeoms]).reshape(-1, 3),
        site_bodyid=np.array([s["body"] for s in sites], dtype=np.int64),
        site_pos=np.array([s["pos"] for s in sites]).reshape(-1, 3),
        site_quat=np.array([s["quat"] for s in sites]).reshape(-1, 4),
        con_kind=np.array([c["kind"] for c in contacts], dtype=np.int64),
        con_geom1=np.array([c["geom1"] for c in contacts], dtype=np.int64),
        con_geom2=np.array([c["geom2"] for c in contacts], dtype=np.int64),
        con_body1=np.array([c["body1"] for c in contacts], dtype=np.int64),
        con_body2=np.array([c["body2"] for c in contacts], dtype=np.int64),
        con_dim=np.array([c["dim"] for c in contacts], dtype=np.int64),
        con_sub=np.array([c["sub"] for c in contacts], dtype=np.int64),
        con_friction=np.array([c["friction"] for c in contacts]).reshape(-1, 5),
        con_solref=np.array([c["solref"] for c in contacts]).reshape(-1, 2),
        con_solimp=np.array([c["solimp"] for c in contacts]).reshape(-1, 5),
        con_margin=np.array([c["margin"] for c in contacts]),
        lim_jnt=np.array(lim_jnt, dtype=np.int64),
        fri_dof=np.array([joints[ji]["dofadr"] for ji in fri_jnt], dtype=np.int64),
        fri_loss=np.array([joints[ji]["frictionloss"] for ji in fri_jnt], dtype=np.float64),
        fri_solref=np.array([joints[ji]["solreffriction"] for ji in fri_jnt], dtype=np.float64).reshape(-1, 2),
        fri_solimp=np.array([joints[ji]["solimpfriction"] for ji in fri_jnt], dtype=np.float64).reshape(-1, 5),
        act_dofadr=np.array([a["dofadr"] for a in acts], dtype=np.int64),
        act_qposadr=np.array([a["qposadr"] for a in acts], dtype=np.int64),
        act_ctrllimited=np.array([int(a["ctrllimited"]) for a in acts], dtype=np.int64),
        act_isposition=np.array([int(a["isposition"]) for a in acts], dtype=np.int64),
        act_gear=np.array([a["gear"] for a in acts]),
        act_kp=np.array([a["kp"] for a in acts]),
        act_ctrlrange=np.array([a["ctrlrange"] for a in acts]).reshape(-1, 2),
    )
    m["names"] = dict(body=[b["name"] for b in bodies], joint=jnames,
                      site=[s["name"] for s in sites], geom=[g["name"] for g in cgeoms],
                      actuator=[a["name"] for a in acts])
    m["keyframes"] = {k: v.tolist() for k, v in keys.items()}
    _set_const(m)
    return m


# ---------------------------------------------------------------- the contact ARRAY as data
# The order of the static contact list (and how many candidates a geom pair contributes) is MJX-internal -- pair grouping by
# collision function, per-function candidate counts that changed between releases -- and the reference's crate envs read
# that array by POSITION (unitree_go2_env.py:750, unitree_h1_env.py:469-470, 522-523).  The list is therefore data, not
# code: `contact_slots(model)` names the geom pair of every position, `reorder_contacts(model, slots)` rebuilds the list in
# any given order / multiplicity -- e.g. the per-slot geom ids that tools/export_reference_vectors.py records from a
# reference run -- without touching the compiler, the oracle or a kernel (all of them consume dial_model.con_*).
_CON_KEYS = ("con_kind", "con_geom1", "con_geom2", "con_body1", "con_body2", "con_dim", "con_sub", "con_friction", "con_solref",
             "con_solimp", "con_margin")
# candidate contacts one pair of that kind can be asked for (con_sub < this; candidates the geometry does not produce are parked
# at dist = 1: box_box ranks the <= 8 points of the clipped face, plane_box the 8 vertices, capsule_box has its two spheres)
_MAX_SUB = {5: 8, 6: 1, 7: 2, 8: 8}


def contact_slots(m: Dict[str, Any], ids: str = "index") -> List[tuple]:
    """(geom1, geom2) of every position of the model's contact list -- as indices into the model's collision-geom list
    (ids="index"), as MuJoCo / MJX geom ids (ids="mujoco": what a reference run's contact.geom holds) or as names."""
    g = m["names"]["geom"] if ids == "name" else (np.asarray(m["geom_mjid"]).tolist() if ids == "mujoco" else list(range(int(m["ngeom"]))))
    return [(g[int(a)], g[int(b)]) for a, b in zip(np.asarray(m["con_geom1"]), np.asarray(m["con_geom2"]))]


def reorder_contacts(m: Dict[str, Any], slots, ids: str = "index") -> Dict[str, Any]:
    """A copy of the compiled model whose contact list follows `slots`: one (geom1, geom2) pair -- collision-geom indices,
    unique names, or (ids="mujoco") MuJoCo / MJX geom ids as a reference run records them; in either order -- per position of
    the wanted contact ARRAY.  The k-th occurrence of a pair takes the pair's k-th
    candidate (con_sub = k); a pair may be listed MORE often than this compiler emits it (box kinds: up to 8, the extra
    candidates carry the next con_sub and come out parked when the geometry has no such point).  Pairs the compiler emits but
    `slots` does not mention keep their candidates, appended after the listed ones in their old order (the physics is unchanged
    by any reordering: rows are summed in another order, nothing else)."""
    names = m["names"]["geom"]
    ncon = int(m["ncon"])
    arr = {k: np.asarray(m[k]) for k in _CON_KEYS}
    mjid = np.asarray(m["geom_mjid"]).tolist() if ids == "mujoco" else None
    idx = lambda g: names.index(g) if isinstance(g, str) else (mjid.index(int(g)) if mjid is not None else int(g))  # noqa: E731
    by_pair: Dict[tuple, List[int]] = {}
    for c in range(ncon):
        by_pair.setdefault((int(arr["con_geom1"][c]), int(arr["con_geom2"][c])), []).append(c)
    for v in by_pair.values():
        v.sort(key=lambda c: int(arr["con_sub"][c]) * 4 + (int(arr["con_kind"][c]) == 2))   # plane-capsule: the +axis end first
    used: Dict[tuple, int] = {}
    order: List[tuple] = []                         # (source contact, con_sub override or None)
    for a, b in slots:
        ia, ib = idx(a), idx(b)
        key = (ia, ib) if (ia, ib) in by_pair else (ib, ia)
        if key not in by_pair:
            raise ValueError(f"contact slot ({a}, {b}): the compiled model has no contact between these geoms")
        k = used.get(key, 0)
        used[key] = k + 1
        cands = by_pair[key]
        if k < len(cands):
            order.append((cands[k], None))
        else:
            kind = int(arr["con_kind"][cands[0]])
            if k >= _MAX_SUB.get(kind, len(cands)):
                raise ValueError(f"contact slot ({a}, {b}): occurrence {k + 1} exceeds what contact kind {kind} can produce")
            order.append((cands[0], k))
    listed = {c for c, sub in order if sub is None}
    order += [(c, None) for c in range(ncon) if c not in listed]
    out = dict(m)
    for key in _CON_KEYS:
        out[key] = np.stack([arr[key][c] for c, _ in order]) if arr[key].ndim > 1 else np.array([arr[key][c] for c, _ in order])
    out["con_sub"] = np.array([int(arr["con_sub"][c]) if sub is None else sub for c, sub in order], dtype=np.int64)
    out["ncon"] = len(order)
    out["nefc"] = int(m["nefc"]) + (int(np.sum(out["con_dim"])) - int(np.sum(arr["con_dim"])) if int(m.get("cone", 0)) == 1
                                    else 4 * (len(order) - ncon))
    return out


def _mat_to_quat(R):
    w = math.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = math.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = math.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = math.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x = math.copysign(x, R[2, 1] - R[1, 2])
    y = math.copysign(y, R[0, 2] - R[2, 0])
    z = math.copysign(z, R[1, 0] - R[0, 1])
    return _normalize(np.array([w, x, y, z]))


# ------------------------------------------------------------------ fp64 host kinematics / CRB
def host_kinematics(m: Dict[str, Any], qpos: np.ndarray) -> Dict[str, np.ndarray]:
    """Forward kinematics + COM quantities at ``qpos``, fp64."""
    nb, nv = m["nbody"], m["nv"]
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0, 0] = 1
    xanchor = np.zeros((m["njnt"], 3))
    xaxis = np.zeros((m["njnt"], 3))
    for b in range(1, nb):
        p = m["body_parent"][b]
        pos = xpos[p] + rotate(m["body_pos"][b], xquat[p])
        quat = quat_mul(xquat[p], m["body_quat"][b])
        for ji in range(m["body_jntadr"][b], m["body_jntadr"][b] + m["body_jntnum"][b]):
            qa = m["jnt_qposadr"][ji]
            if m["jnt_type"][ji] == JNT_FREE:
                pos = qpos[qa:qa + 3].copy()
                quat = _normalize(qpos[qa + 3:qa + 7])
                xanchor[ji] = pos
                xaxis[ji] = [0, 0, 1]
            else:
                anchor = rotate(m["jnt_pos"][ji], quat) + pos
                axis = rotate(m["jnt_axis"][ji], quat)
                xanchor[ji], xaxis[ji] = anchor, axis
                if m["jnt_type"][ji] == JNT_HINGE:
                    ang = qpos[qa] - m["qpos0"][qa]
                    qloc = np.concatenate([[math.cos(ang / 2)], m["jnt_axis"][ji] * math.sin(ang / 2)])
                    quat = quat_mul(quat, qloc)
                    pos = anchor - rotate(m["jnt_pos"][ji], quat)
                else:  # slide
                    pos = pos + axis * (qpos[qa] - m["qpos0"][qa])
        xpos[b], xquat[b] = pos, quat
    xmat = np.array([quat_to_mat(q) for q in xquat])
    xipos = np.array([xpos[b] + xmat[b] @ m["body_ipos"][b] for b in range(nb)])
    ximat = np.array([quat_to_mat(quat_mul(xquat[b], m["body_iquat"][b])) for b in range(nb)])
    # subtree COM
    mass = m["body_mass"].astype(np.float64).copy()
    mpos = xipos * mass[:, None]
    for b in range(nb - 1, 0, -1):
        p = m["body_parent"][b]
        mass[p] += mass[b]
        mpos[p] += mpos[b]
    subtree_com = np.where(mass[:, None] < MJ_MINVAL, xipos, mpos / np.maximum(mass, MJ_MINVAL)[:, None])
    root_com = subtree_com[m["body_rootid"]]
    # cinert as 6x6 spatial inertia about root_com ([ang; lin] ordering)
    spat = np.zeros((nb, 6, 6))
    for b in range(1, nb):
        d = xipos[b] - root_com[b]
        I = ximat[b] @ np.diag(m["body_inertia"][b]) @ ximat[b].T
        mb = m["body_mass"][b]
        I = I + mb * (d @ d * np.eye(3) - np.outer(d, d))
        hx = _skew(mb * d)
        spat[b, :3, :3] = I
        spat[b, :3, 3:] = hx
        spat[b, 3:, :3] = hx.T
        spat[b, 3:, 3:] = mb * np.eye(3)
    cdof = np.zeros((nv, 6))
    for ji in range(m["njnt"]):
        b = m["jnt_bodyid"][ji]
        da = m["jnt_dofadr"][ji]
        off = root_com[b] - xanchor[ji]
        if m["jnt_type"][ji] == JNT_FREE:
            cdof[da:da + 3, 3:] = np.eye(3)
            for k in range(3):
                ax = xmat[b][:, k]
                cdof[da + 3 + k] = np.concatenate([ax, np.cross(ax, off)])
        elif m["jnt_type"][ji] == JNT_HINGE:
            cdof[da] = np.concatenate([xaxis[ji], np.cross(xaxis[ji], off)])
        else:
            cdof[da] = np.concatenate([np.zeros(3), xaxis[ji]])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat,
                subtree_com=subtree_com, root_com=root_com, spat=spat, cdof=cdof,
                xanchor=xanchor, xaxis=xaxis)


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def host_mass_matrix(m: Dict[str, Any], kin: Dict[str, np.ndarray]) -> np.ndarray:
    """Composite-rigid-body mass matrix incl. armature, fp64."""
    nb, nv = m["nbody"], m["nv"]
    crb = kin["spat"].copy()
    for b in range(nb - 1, 0, -1):
        p = m["body_parent"][b]
        if p > 0:
            crb[p] += crb[b]
    M = np.zeros((nv, nv))
    for i in range(nv):
        f = crb[m["dof_bodyid"][i]] @ kin["cdof"][i]
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = kin["cdof"][j] @ f
            j = m["dof_parentid"][j]
    M += np.diag(m["dof_armature"])
    return M


def host_jac(m: Dict[str, Any], kin: Dict[str, np.ndarray], body: int, point: np.ndarray):
    """Translational / rotational Jacobian (3 x nv each) of ``body`` at world ``point``."""
    nv = m["nv"]
    jacp, jacr = np.zeros((3, nv)), np.zeros((3, nv))
    if body == 0:
        return jacp, jacr
    off = point - kin["root_com"][body]
    b = body
    while b > 0 and m["body_dofnum"][b] == 0:
        b = m["body_parent"][b]
    if b == 0:
        return jacp, jacr
    i = m["body_dofadr"][b] + m["body_dofnum"][b] - 1
    while i >= 0:
        cd = kin["cdof"][i]
        jacr[:, i] = cd[:3]
        jacp[:, i] = cd[3:] + np.cross(cd[:3], off)
        i = m["dof_parentid"][i]
    return jacp, jacr


def _set_const(m: Dict[str, Any]) -> None:
    """MuJoCo's mj_setConst subset: meaninertia, body_invweight0, dof_invweight0 at qpos0."""
    nv, nb = m["nv"], m["nbody"]
    kin = host_kinematics(m, m["qpos0"])
    M = host_mass_matrix(m, kin)
    Minv = np.linalg.inv(M)
    m["meaninertia"] = float(np.mean(np.diag(M))) if nv else 1.0
    biw = np.zeros((nb, 2))
    for b in range(1, nb):
        jp_, jr_ = host_jac(m, kin, b, kin["xipos"][b])
        biw[b, 0] = np.trace(jp_ @ Minv @ jp_.T) / 3
        biw[b, 1] = np.trace(jr_ @ Minv @ jr_.T) / 3
    m["body_invweight0"] = biw
    diw = np.zeros(nv)
    for ji in range(m["njnt"]):
        da = m["jnt_dofadr"][ji]
        if m["jnt_type"][ji] == JNT_FREE:
            d = np.diag(Minv)[da:da + 6]
            diw[da:da + 3] = d[:3].mean()
            diw[da + 3:da + 6] = d[3:].mean()
        else:
            diw[da] = Minv[da, da]
    m["dof_invweight0"] = diw


# ------------------------------------------------------------------ (de)serialisation
def model_to_json(m: Dict[str, Any]) -> str:
    def conv(v):
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, (np.floating, np.integer)):
            return v.item()
        return v
    return json.dumps({k: conv(v) for k, v in m.items()}, indent=1)


def model_from_json(text: str) -> Dict[str, Any]:
    raw = json.loads(text)
    out: Dict[str, Any] = {}
    for k, v in raw.items():
        if isinstance(v, list):
            out[k] = np.array(v)
        else:
            out[k] = v
    return out


def set_keyframe(m: Dict[str, Any], name: str) -> None:
    m["key_qpos"] = np.array(m["keyframes"][name], dtype=np.float64)
