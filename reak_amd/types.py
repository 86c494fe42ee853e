"""ctypes mirrors of include/rkh_types.h (the POD scene description that crosses the C-ABI)."""
import ctypes as C
import math

import numpy as np

RKH_MAX_DOF = 16
RKH_MAX_STATE = 2 * RKH_MAX_DOF

# rkh_kte_kind
KTE_DRIVING_ACTUATOR_GEN = 1
KTE_INERTIA_GEN = 2
KTE_REVOLUTE_JOINT_3D = 3
KTE_RIGID_LINK_3D = 4
KTE_INERTIA_3D = 5
KTE_FLEXIBLE_BEAM_3D = 6
KTE_REVOLUTE_JOINT_2D = 7
KTE_RIGID_LINK_2D = 8
KTE_INERTIA_2D = 9

# rkh_shape_kind
SHAPE_SPHERE = 1
SHAPE_BOX = 2
SHAPE_CCYLINDER = 3
SHAPE_CIRCLE = 4
SHAPE_RECTANGLE = 5
SHAPE_CRECT = 6
SHAPE_PLANE = 7      # dims = (x extent, y extent); normal = local z
SHAPE_CYLINDER = 8   # dims = (length, radius); flat ends, axis = local z
SHAPE_MESH = 9       # convex vertex set: dims = (first vertex in the scene's pool, vertex count)


class Pose(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("quat", C.c_double * 4)]


class KteOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("coord", C.c_int32),
        ("base_frame", C.c_int32),
        ("end_frame", C.c_int32),
        ("joint_op", C.c_int32),
        ("upstream", C.c_uint32),
        ("axis", C.c_double * 3),
        ("offset", Pose),
        ("mass", C.c_double),
        ("inertia", C.c_double * 6),
    ]


class ChainBase(C.Structure):
    _fields_ = [("pose", Pose), ("acceleration", C.c_double * 3)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("anchor", C.c_int32), ("pose", Pose), ("dims", C.c_double * 3)]


class DynSpace(C.Structure):
    _fields_ = [
        ("n_dof", C.c_int32),
        ("steps_per_edge", C.c_int32),
        ("dt", C.c_double),
        ("kp", C.c_double),
        ("kd", C.c_double),
        ("u_max", C.c_double),
        ("goal_tol", C.c_double),
        ("lower", C.c_double * RKH_MAX_STATE),
        ("upper", C.c_double * RKH_MAX_STATE),
    ]


class QsSpace(C.Structure):
    _fields_ = [("n_dof", C.c_int32), ("pad", C.c_int32), ("min_interval", C.c_double),
                ("lower", C.c_double * RKH_MAX_DOF), ("upper", C.c_double * RKH_MAX_DOF),
                ("speed_limits", C.c_double * RKH_MAX_DOF)]


class RrtParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32),
        ("max_vertices", C.c_uint32),
        ("max_results", C.c_uint32),
        ("steer_tol", C.c_double),
        ("conn_tol", C.c_double),
        ("start", C.c_double * RKH_MAX_STATE),
        ("goal", C.c_double * RKH_MAX_STATE),
    ]


class PrmParams(C.Structure):
    _fields_ = [("base", RrtParams), ("sampling_radius", C.c_double), ("expand_probability", C.c_double)]


def make_pose(pos=(0.0, 0.0, 0.0), quat=(1.0, 0.0, 0.0, 0.0)):
    p = Pose()
    p.pos[:] = [float(v) for v in pos]
    p.quat[:] = [float(v) for v in quat]
    return p


def make_pose_2d(pos=(0.0, 0.0), angle=0.0):
    """pose_2D carried in rkh_pose: pos[0..1] = Position, quat[0..1] = rot_mat_2D::q = (cos, sin)."""
    p = Pose()
    p.pos[:] = [float(pos[0]), float(pos[1]), 0.0]
    if angle == 0.0:
        p.quat[:] = [1.0, 0.0, 0.0, 0.0]  # rot_mat_2D() default: identity
    else:
        p.quat[:] = [math.cos(angle), math.sin(angle), 0.0, 0.0]
    return p


def as_array(items, ctype):
    arr = (ctype * len(items))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr


def dptr(a):
    """double* view of a C-contiguous float64 numpy array."""
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def u32ptr(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))
