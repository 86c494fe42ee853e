"""ReaK XML archives (`.rkx`) of planning scenes: writer + reader for exactly the classes a scene needs.

Format: `xml_oarchive` / `xml_iarchive` of the reference (R/core/serialization/xml_archiver.cpp:425-706: header
`<reak_serialization version="2">`, one element per field named by RK_SERIAL_SAVE_WITH_NAME, primitive values in double
quotes, objects as `<name type_ID="id.id.0" version="v" object_ID="n" is_external="false">` -- a shared object is written
once and referred to by its object_ID afterwards, a null pointer is `type_ID="0" version="0" object_ID="0"`; plain
serializables carry `type_ID` and `version` only).  Field names and type ids follow the classes' own save() functions:

  geometry     named_object (name) -> geometry_3D (mAnchor, mPose) -> shape_3D -> plane / box (mDimensions), sphere
               (mRadius), capped_cylinder / cylinder (mLength, mRadius)            R/geometry/shapes/*.cpp
               proxy_query_model_3D (mShapeList)                                   R/geometry/proximity/proxy_query_model.hpp:181
  kinetostatics pose_3D (Parent, Position, Quat), frame_3D (+ Velocity .. Torque), gen_coord (q, q_dot, q_ddot, f),
               vect<double,N> (N x "q"), quaternion ("q[0]" .. "q[3]"), mat<double,symmetric> (q, rowCount)
  KTE chain    kte_map_chain (mKTEs) of revolute_joint_3D (mAngle, mAxis, mBase, mEnd, mJacobian), rigid_link_3D (mBase,
               mEnd, mPoseOffset), inertia_3D (mCenterOfMass = joint_dependent_frame_3D, mMass, mInertiaTensor),
               inertia_gen (mCenterOfMass = joint_dependent_gen_coord, mMass), driving_actuator_gen (mFrame, mJoint,
               mDriveForce)                                                        R/ctrl/mbd_kte/*.hpp
Not covered: the rendering model (colored_model_3D) the obstacle-course builder also writes, Jacobian objects (written
null; ReaK rebuilds them), the joint-dependency maps of inertia elements (std::map<gen_coord ptr, jacobian ptr>: only the
entry COUNT is written in ReaK's field, the set of upstream coordinates travels in `mUpStreamMask`, a field of THIS writer
that ReaK's classes do not have), 2D classes, flexible_beam_3D, protobuf / binary archives.

One deviation, on purpose: ReaK streams doubles with the default ostream precision (6 significant digits), which does not
round-trip; this writer emits 17 significant digits (`repr`), which ReaK's reader (operator>> on a double) accepts.
Nothing here touches the GPU; the scene a file describes becomes a `scenarios.Scenario` and goes through the C-ABI as usual.
The reference ships no `.rkx` scene (its only archives are integrator records): files written here are checked against
this reader and against the class definitions cited above -- parity unpinned beyond that."""
import math
import re

import numpy as np

from . import types as T

HEADER = '<?xml version="1.0" encoding="UTF-8" standalone="yes" ?>\n<!DOCTYPE reak_serialization>\n<reak_serialization version="2">\n'
FOOTER = "</reak_serialization>\n"

# rtti ids (RK_RTTI_MAKE_*): class -> (type id chain, version)
TYPE_IDS = {
    "vect3": ([0x11, 4, 3], 1), "vect2": ([0x11, 4, 2], 1), "quaternion": ([0x0A, 4], 1), "pose_3D": ([0x1E, 4], 1),
    "frame_3D": ([0x20, 4], 1), "gen_coord": ([0x0F, 4], 1), "mat_sym": ([0x12, 4, 3, 1], 1),
    "plane": ([0xC310000F], 1), "sphere": ([0xC3100010], 1), "capped_cylinder": ([0xC3100011], 1),
    "cylinder": ([0xC3100012], 1), "box": ([0xC3100013], 1), "proxy_query_model_3D": ([0xC320001B], 1),
    "kte_map_chain": ([0xC2100002], 1), "revolute_joint_3D": ([0xC2100004], 1), "rigid_link_3D": ([0xC2100009], 1),
    "inertia_gen": ([0xC210000A], 1), "inertia_3D": ([0xC210000C], 1), "driving_actuator_gen": ([0xC2100023], 1),
    "joint_dependent_gen_coord": ([0xC2000002], 1), "joint_dependent_frame_3D": ([0xC2000004], 1),
}
ID_TO_CLASS = {".".join(str(i) for i in ids) + ".0": name for name, (ids, _) in TYPE_IDS.items()}
SHAPE_CLASS = {T.SHAPE_SPHERE: "sphere", T.SHAPE_BOX: "box", T.SHAPE_CCYLINDER: "capped_cylinder", T.SHAPE_PLANE: "plane",
               T.SHAPE_CYLINDER: "cylinder"}
CLASS_SHAPE = {v: k for k, v in SHAPE_CLASS.items()}


# ------------------------------------------------------------------------------------------------ object model
class Obj:
    """One serialized object: class name + ordered fields (name -> float | int | str | bool | Obj | None | list)."""

    def __init__(self, cls, **fields):
        self.cls = cls
        self.fields = dict(fields)

    def __getitem__(self, k):
        return self.fields[k]


def vect(v):
    return Obj("vect%d" % len(v), q=[float(x) for x in v])


def quaternion(q):
    return Obj("quaternion", **{"q[%d]" % i: float(q[i]) for i in range(4)})


def pose_3d(pose, parent=None):
    return Obj("pose_3D", Parent=parent, Position=vect(pose.pos), Quat=quaternion(pose.quat))


def frame_3d(pose=None, parent=None, acceleration=(0.0, 0.0, 0.0)):
    pose = pose if pose is not None else T.make_pose()
    z = (0.0, 0.0, 0.0)
    return Obj("frame_3D", Parent=parent, Position=vect(pose.pos), Quat=quaternion(pose.quat), Velocity=vect(z),
               AngVelocity=vect(z), Acceleration=vect(acceleration), AngAcceleration=vect(z), Force=vect(z), Torque=vect(z))


def gen_coord():
    return Obj("gen_coord", q=0.0, q_dot=0.0, q_ddot=0.0, f=0.0)


# ------------------------------------------------------------------------------------------------ writer
class Writer:
    def __init__(self):
        self.out = [HEADER]
        self.tab = 0
        self.ids = {}  # id(Obj) -> object_ID (the archive's mObjRegistry; id 0 is reserved for null)
        self.alive = []
        self.next_id = 1

    def _line(self, s):
        self.out.append("    " * self.tab + s + "\n")

    @staticmethod
    def _num(v):
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, int):
            return str(v)
        return repr(float(v))  # shortest string that round-trips (<= 17 significant digits)

    def prim(self, name, v):
        self._line(f'<{name}>"{v if isinstance(v, str) else self._num(v)}"</{name}>')

    def _fields(self, obj):
        for k, v in obj.fields.items():
            if isinstance(v, list) and obj.cls.startswith("vect"):  # vect<T,N>: N fields called "q"
                for x in v:
                    self.prim("q", x)
            elif isinstance(v, list):  # std::vector<T>: name_count, then name_q[i]
                self.prim(k + "_count", len(v))
                for i, x in enumerate(v):
                    self.item(f"{k}_q[{i}]", x)
            else:
                self.item(k, v)

    def item(self, name, v):
        if isinstance(v, Obj) and v.cls in ("vect3", "vect2", "quaternion", "pose_3D", "mat_sym") and not getattr(v, "shared", False):
            ids, ver = TYPE_IDS[v.cls]
            self._line(f'<{name} type_ID="{".".join(str(i) for i in ids)}.0" version="{ver}">')
            self.tab += 1
            self._fields(v)
            self.tab -= 1
            self._line(f"</{name}>")
        elif isinstance(v, Obj):  # a shared pointer
            ids, ver = TYPE_IDS[v.cls]
            known = id(v) in self.ids
            if not known:
                self.ids[id(v)] = self.next_id
                self.alive.append(v)  # (the registry is keyed by identity: a freed temporary's id could be re-used)
                self.next_id += 1
            self._line(f'<{name} type_ID="{".".join(str(i) for i in ids)}.0" version="{ver}" object_ID="{self.ids[id(v)]}" '
                       f'is_external="false">')
            if not known:
                self.tab += 1
                self._fields(v)
                self.tab -= 1
            self._line(f"</{name}>")
        elif v is None:
            self._line(f'<{name} type_ID="0" version="0" object_ID="0" is_external="false">')
            self._line(f"</{name}>")
        else:
            self.prim(name, v)

    def text(self):
        return "".join(self.out) + FOOTER


# ------------------------------------------------------------------------------------------------ reader
_TOKEN = re.compile(r'<(/?)([^\s>]+)((?:\s+[A-Za-z_]+="[^"]*")*)\s*>|"([^"]*)"')


def _parse(text):
    """-> list of top-level (name, node); node = str (primitive) | dict(attrs=..., children=[(name, node), ...])."""
    body = text[text.index("<reak_serialization"):]
    toks = list(_TOKEN.finditer(body))
    stack = [{"attrs": {}, "children": []}]
    names = ["reak_serialization"]
    pending = None
    for m in toks[1:]:
        close, name, attrs, val = m.group(1), m.group(2), m.group(3), m.group(4)
        if val is not None:
            pending = val
        elif close:
            node = stack.pop()
            nm = names.pop()
            if nm == "reak_serialization":
                return node["children"]
            assert nm == name, (nm, name)
            stack[-1]["children"].append((nm, pending if (pending is not None and not node["children"] and not node["attrs"]) else node))
            pending = None
        else:
            stack.append({"attrs": dict(re.findall(r'([A-Za-z_]+)="([^"]*)"', attrs or "")), "children": []})
            names.append(name)
            pending = None
    raise ValueError("unterminated archive")


class Reader:
    def __init__(self, text):
        self.top = _parse(text)
        self.registry = {}

    def obj(self, node):
        """dict node -> Obj (shared objects resolved through the object_ID registry), or None for a null pointer."""
        a = node["attrs"]
        if a.get("type_ID") == "0":
            return None
        oid = int(a.get("object_ID", "0"))
        if oid and oid in self.registry and not node["children"]:
            return self.registry[oid]
        cls = ID_TO_CLASS.get(a["type_ID"])
        if cls is None:
            raise ValueError("class of type_ID %s is not covered by this reader" % a["type_ID"])
        o = Obj(cls)
        if oid:
            self.registry[oid] = o
        ch = node["children"]
        if cls.startswith("vect"):
            o.fields["q"] = [float(v) for k, v in ch]
            return o
        i = 0
        while i < len(ch):
            k, v = ch[i]
            if k.endswith("_count") and isinstance(v, str):
                n, base = int(v), k[: -len("_count")]
                if all(i + 1 + j < len(ch) and ch[i + 1 + j][0] == f"{base}_q[{j}]" for j in range(n)) and (
                        n > 0 or base in ("mShapeList", "mKTEs", "q")):
                    o.fields[base] = [self.value(ch[i + 1 + j][1]) for j in range(n)]
                    i += 1 + n
                    continue
            o.fields[k] = self.value(v)
            i += 1
        return o

    def value(self, node):
        if isinstance(node, dict):
            return self.obj(node)
        if node in ("true", "false"):
            return node == "true"
        try:
            return int(node)
        except ValueError:
            try:
                return float(node)
            except ValueError:
                return node


# ------------------------------------------------------------------------------------------------ scene <-> objects
def _pose_from(o):
    return T.make_pose(o["Position"]["q"], [o["Quat"]["q[%d]" % i] for i in range(4)])


def shape_object(shape, frames, name):
    """rkh_shape -> the ReaK shape object (anchor = a frame of the chain, or null for a world shape)."""
    cls = SHAPE_CLASS[shape.kind]
    o = Obj(cls, name=name, mAnchor=frames[shape.anchor] if shape.anchor >= 0 else None, mPose=pose_3d(shape.pose))
    d = shape.dims
    if cls == "box":
        o.fields["mDimensions"] = vect(d[:3])
    elif cls == "plane":
        o.fields["mDimensions"] = vect(d[:2])
    elif cls == "sphere":
        o.fields["mRadius"] = float(d[0])
    else:
        o.fields["mLength"], o.fields["mRadius"] = float(d[0]), float(d[1])
    return o


def shape_from_object(o, frame_index):
    s = T.Shape(kind=CLASS_SHAPE[o.cls], anchor=frame_index.get(id(o["mAnchor"]), -1) if o["mAnchor"] is not None else -1)
    s.pose = _pose_from(o["mPose"])
    if o.cls == "box":
        s.dims[:] = o["mDimensions"]["q"]
    elif o.cls == "plane":
        s.dims[:] = list(o["mDimensions"]["q"]) + [0.0]
    elif o.cls == "sphere":
        s.dims[:] = [o["mRadius"], 0.0, 0.0]
    else:
        s.dims[:] = [o["mLength"], o["mRadius"], 0.0]
    return s


def chain_objects(ops, base, n_frames, n_coords):
    """The op list of a serial / branching chain -> kte_map_chain object graph (frames and coordinates are shared objects)."""
    frames = [frame_3d(base.pose, None, base.acceleration)] + [frame_3d() for _ in range(n_frames - 1)]
    for f in frames:
        f.shared = True
    coords = [gen_coord() for _ in range(n_coords)]
    ktes = []
    for k, op in enumerate(ops):
        if op.kind == T.KTE_DRIVING_ACTUATOR_GEN:
            o = Obj("driving_actuator_gen", name=f"actuator_{op.coord}", mFrame=coords[op.coord], mJoint=("op", op.joint_op),
                    mDriveForce=0.0)
        elif op.kind == T.KTE_INERTIA_GEN:
            dep = Obj("joint_dependent_gen_coord", mFrame=coords[op.coord], mUpStreamJoints_count=bin(op.upstream).count("1"),
                      mUpStreamMask=int(op.upstream))
            o = Obj("inertia_gen", name=f"rotor_inertia_{op.coord}", mCenterOfMass=dep, mMass=float(op.mass))
        elif op.kind == T.KTE_REVOLUTE_JOINT_3D:
            o = Obj("revolute_joint_3D", name=f"joint_{op.coord}", mAngle=coords[op.coord], mAxis=vect(op.axis),
                    mBase=frames[op.base_frame], mEnd=frames[op.end_frame], mJacobian=None)
        elif op.kind == T.KTE_RIGID_LINK_3D:
            o = Obj("rigid_link_3D", name=f"link_{k}", mBase=frames[op.base_frame], mEnd=frames[op.end_frame],
                    mPoseOffset=pose_3d(op.offset))
        elif op.kind == T.KTE_INERTIA_3D:
            dep = Obj("joint_dependent_frame_3D", mFrame=frames[op.end_frame], mUpStreamJoints_count=bin(op.upstream).count("1"),
                      mUpStreamMask=int(op.upstream))
            a = op.inertia  # a11 a12 a13 a22 a23 a33 = the symmetric matrix's storage order (mat_alg_symmetric.hpp)
            o = Obj("inertia_3D", name=f"link_inertia_{k}", mCenterOfMass=dep, mMass=float(op.mass),
                    mInertiaTensor=Obj("mat_sym", q=[float(v) for v in a], rowCount=3))
        else:
            raise ValueError("KTE kind %d is not covered by the .rkx writer" % op.kind)
        ktes.append(o)
    for o in ktes:  # the actuator's joint pointer: the joint object itself
        if o.cls == "driving_actuator_gen":
            o.fields["mJoint"] = ktes[o["mJoint"][1]]
    return Obj("kte_map_chain", name="chain", mKTEs=ktes), frames, coords


def write_scene(scn, start=None, goal=None):
    """Scenario -> archive text: chain, robot + environment proxy models, start and goal vectors."""
    n_coords = scn.n_dof
    chain, frames, _ = chain_objects(scn.ops, scn.base, scn.n_frames, n_coords)
    robot = [shape_object(s, frames, f"robot_shape_{i}") for i, s in enumerate(scn.shapes) if s.anchor >= 0]
    env = [shape_object(s, frames, f"env_shape_{i}") for i, s in enumerate(scn.shapes) if s.anchor < 0]
    w = Writer()
    w.item("Item", chain)
    w.item("Item", Obj("proxy_query_model_3D", name="robot_proxy", mShapeList=robot))
    w.item("Item", Obj("proxy_query_model_3D", name="environment_proxy", mShapeList=env))
    st = scn.start if start is None else start
    gl = scn.goal if goal is None else goal
    w.prim("start_count", len(st))
    for i, v in enumerate(st):
        w.prim(f"start_q[{i}]", float(v))
    w.prim("goal_count", len(gl))
    for i, v in enumerate(gl):
        w.prim(f"goal_q[{i}]", float(v))
    return w.text()


def read_scene(text, template):
    """Archive text -> Scenario (dyn space / meta are taken from `template`: an archive of this kind holds the models,
    not the planner's options)."""
    from .scenarios import Scenario

    r = Reader(text)
    objs = [(k, r.value(v)) for k, v in r.top]
    chain = next(o for k, o in objs if isinstance(o, Obj) and o.cls == "kte_map_chain")
    models = [o for k, o in objs if isinstance(o, Obj) and o.cls == "proxy_query_model_3D"]
    prims = {k: o for k, o in objs if not isinstance(o, Obj)}
    # frames / coordinates in order of first appearance along the chain (base of the first joint first)
    frame_index, coord_index = {}, {}

    def fidx(f):
        return frame_index.setdefault(id(f), len(frame_index))

    def cidx(c):
        return coord_index.setdefault(id(c), len(coord_index))

    ktes = chain["mKTEs"]
    kte_index = {id(o): i for i, o in enumerate(ktes)}
    base_frame = None
    for o in ktes:  # pass 1: number the frames like serial_chain_ops does (joint base, joint end, link end)
        if o.cls in ("revolute_joint_3D", "rigid_link_3D"):
            if base_frame is None:
                base_frame = o["mBase"]
            fidx(o["mBase"])
            fidx(o["mEnd"])
    ops = []
    for o in ktes:
        if o.cls == "driving_actuator_gen":
            ops.append(T.KteOp(kind=T.KTE_DRIVING_ACTUATOR_GEN, coord=cidx(o["mFrame"]), base_frame=-1, end_frame=-1,
                               joint_op=kte_index[id(o["mJoint"])]))
        elif o.cls == "inertia_gen":
            dep = o["mCenterOfMass"]
            ops.append(T.KteOp(kind=T.KTE_INERTIA_GEN, coord=cidx(dep["mFrame"]), base_frame=-1, end_frame=-1, joint_op=-1,
                               upstream=int(dep["mUpStreamMask"]), mass=float(o["mMass"])))
        elif o.cls == "revolute_joint_3D":
            op = T.KteOp(kind=T.KTE_REVOLUTE_JOINT_3D, coord=cidx(o["mAngle"]), base_frame=fidx(o["mBase"]),
                         end_frame=fidx(o["mEnd"]), joint_op=-1)
            op.axis[:] = o["mAxis"]["q"]
            ops.append(op)
        elif o.cls == "rigid_link_3D":
            op = T.KteOp(kind=T.KTE_RIGID_LINK_3D, coord=-1, base_frame=fidx(o["mBase"]), end_frame=fidx(o["mEnd"]), joint_op=-1)
            op.offset = _pose_from(o["mPoseOffset"])
            ops.append(op)
        elif o.cls == "inertia_3D":
            dep = o["mCenterOfMass"]
            op = T.KteOp(kind=T.KTE_INERTIA_3D, coord=-1, base_frame=-1, end_frame=fidx(dep["mFrame"]), joint_op=-1,
                         upstream=int(dep["mUpStreamMask"]), mass=float(o["mMass"]))
            op.inertia[:] = o["mInertiaTensor"]["q"]
            ops.append(op)
    base = T.ChainBase()
    base.pose = _pose_from(base_frame)
    base.acceleration[:] = base_frame["Acceleration"]["q"]
    shapes = []
    for m in models:
        shapes += [shape_from_object(s, frame_index) for s in m["mShapeList"]]
    start = np.array([prims[f"start_q[{i}]"] for i in range(int(prims["start_count"]))], dtype=np.float64)
    goal = np.array([prims[f"goal_q[{i}]"] for i in range(int(prims["goal_count"]))], dtype=np.float64)
    return Scenario(name=template.name + "_rkx", ops=ops, base=base, shapes=shapes, dyn=template.dyn, n_dof=len(coord_index),
                    n_frames=len(frame_index), start=start, goal=goal, meta=dict(template.meta))


# ------------------------------------------------------------------------------------------------ obstacle courses
def obstacle_course(which):
    """The two environments of R/examples/misc/build_X8_obstacle_courses.cpp:35-157 restated as data: shape list (plane
    `floor` + boxes, every pose and dimension as written there) and the start / end positions of the vehicle.
    which: "one_building" (:35-66) or "window_crossing" (:69-152)."""
    xrot_pi = (math.cos(math.pi / 2), math.sin(math.pi / 2), 0.0, 0.0)  # quaternion::xrot(M_PI)
    if which == "one_building":
        items = [("floor", T.SHAPE_PLANE, (2.5, 2.5, 0.0), xrot_pi, (5.0, 5.0, 0.0)),
                 ("building", T.SHAPE_BOX, (2.5, 2.5, -2.5), (1, 0, 0, 0), (1.0, 1.0, 5.0))]
        start, end = (0.75, 0.75, -1.0), (4.25, 4.25, -3.0)
    elif which == "window_crossing":
        items = [("floor", T.SHAPE_PLANE, (5.0, 5.0, 0.0), xrot_pi, (10.0, 10.0, 0.0)),
                 ("wall1", T.SHAPE_BOX, (3.0, 1.5, -5.0), (1, 0, 0, 0), (0.2, 3.0, 10.0)),
                 ("wall2", T.SHAPE_BOX, (3.0, 4.5, -2.5), (1, 0, 0, 0), (0.2, 3.0, 5.0)),
                 ("wall3", T.SHAPE_BOX, (3.0, 4.5, -9.0), (1, 0, 0, 0), (0.2, 3.0, 2.0)),
                 ("wall4", T.SHAPE_BOX, (3.0, 8.0, -5.0), (1, 0, 0, 0), (0.2, 4.0, 10.0)),
                 ("wall5", T.SHAPE_BOX, (7.0, 2.0, -5.0), (1, 0, 0, 0), (0.2, 4.0, 10.0)),
                 ("wall6", T.SHAPE_BOX, (7.0, 5.5, -1.0), (1, 0, 0, 0), (0.2, 3.0, 2.0)),
                 ("wall7", T.SHAPE_BOX, (7.0, 5.5, -7.5), (1, 0, 0, 0), (0.2, 3.0, 5.0)),
                 ("wall8", T.SHAPE_BOX, (7.0, 8.5, -5.0), (1, 0, 0, 0), (0.2, 3.0, 5.0))]
        start, end = (0.75, 1.0, -1.0), (9.0, 3.0, -7.0)
    else:
        raise ValueError(which)
    shapes, names = [], []
    for name, kind, pos, quat, dims in items:
        s = T.Shape(kind=kind, anchor=-1)
        s.pose = T.make_pose(pos, quat)
        s.dims[:] = dims
        shapes.append(s)
        names.append(name)
    return shapes, names, np.array(start), np.array(end)


def write_obstacle_course(which):
    """The `<which>_proxy` model + start / end positions as build_X8_obstacle_courses.cpp streams them
    (`out << ..._proxy << start_position << end_position`; the rendering model is not covered)."""
    shapes, names, start, end = obstacle_course(which)
    w = Writer()
    w.item("Item", Obj("proxy_query_model_3D", name=which + "_proxy",
                       mShapeList=[shape_object(s, [], n) for s, n in zip(shapes, names)]))
    w.item("Item", vect(start))
    w.item("Item", vect(end))
    return w.text()


def read_obstacle_course(text):
    r = Reader(text)
    objs = [r.value(v) for k, v in r.top]
    model = next(o for o in objs if o.cls == "proxy_query_model_3D")
    vecs = [o for o in objs if o.cls == "vect3"]
    return ([shape_from_object(s, {}) for s in model["mShapeList"]], [s["name"] for s in model["mShapeList"]],
            np.array(vecs[0]["q"]), np.array(vecs[1]["q"]))
