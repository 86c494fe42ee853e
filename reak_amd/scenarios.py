"""Synthetic planning scenarios: KTE chains, obstacle worlds, dynamic spaces.

Everything is generated from a 64-bit seed with numpy's PCG64 (no files from the reference).
The chain follows the blueprint of ReaK's 6-R dynamic model
(examples/robot_airship/old/CRS_A465_models.cpp:318-790): per joint, in kte_map_chain order,
driving_actuator_gen -> inertia_gen (rotor) -> revolute_joint_3D -> rigid_link_3D -> inertia_3D.
"""
from dataclasses import dataclass, field

import numpy as np

from . import types as T


@dataclass
class Scenario:
    name: str
    ops: list
    base: T.ChainBase
    shapes: list
    dyn: T.DynSpace
    n_dof: int
    n_frames: int
    start: np.ndarray
    goal: np.ndarray
    meta: dict = field(default_factory=dict)
    mesh_vertices: np.ndarray = None  # [n][3] vertex pool of the SHAPE_MESH shapes (dims = first vertex, count)

    @property
    def D(self):
        return 2 * self.n_dof

    def ops_array(self):
        return T.as_array(self.ops, T.KteOp)

    def shapes_array(self):
        return T.as_array(self.shapes, T.Shape)

    def rrt_params(self, seed=1, max_vertices=1000, max_results=1 << 30, steer_tol=0.1, conn_tol=0.05):
        p = T.RrtParams()
        p.seed = seed
        p.max_vertices = max_vertices
        p.max_results = max_results
        p.steer_tol = steer_tol
        p.conn_tol = conn_tol
        for i in range(len(self.start)):
            p.start[i] = float(self.start[i])
            p.goal[i] = float(self.goal[i])
        return p


    def prm_params(self, sampling_radius=1.0, expand_probability=0.2, **kw):
        p = T.PrmParams()
        p.base = self.rrt_params(**kw)
        p.sampling_radius = sampling_radius
        p.expand_probability = expand_probability
        return p


def serial_chain_ops(axes, link_offsets, link_masses, link_inertias, joint_inertias):
    """Flatten a serial revolute chain into kte_map_chain order.  Frame 0 is the base;
    joint j: revolute base=2j, end=2j+1 ; link base=2j+1, end=2j+2 ; inertia_3D on frame 2j+2."""
    ops = []
    n = len(axes)
    for j in range(n):
        rev_index = 5 * j + 2
        a = T.KteOp(kind=T.KTE_DRIVING_ACTUATOR_GEN, coord=j, base_frame=-1, end_frame=-1, joint_op=rev_index)
        ops.append(a)
        g = T.KteOp(kind=T.KTE_INERTIA_GEN, coord=j, base_frame=-1, end_frame=-1, joint_op=-1,
                    upstream=(1 << j), mass=float(joint_inertias[j]))
        ops.append(g)
        r = T.KteOp(kind=T.KTE_REVOLUTE_JOINT_3D, coord=j, base_frame=2 * j, end_frame=2 * j + 1, joint_op=-1)
        r.axis[:] = [float(v) for v in axes[j]]
        ops.append(r)
        l = T.KteOp(kind=T.KTE_RIGID_LINK_3D, coord=-1, base_frame=2 * j + 1, end_frame=2 * j + 2, joint_op=-1)
        l.offset = T.make_pose(link_offsets[j])
        ops.append(l)
        i3 = T.KteOp(kind=T.KTE_INERTIA_3D, coord=-1, base_frame=-1, end_frame=2 * j + 2, joint_op=-1,
                     upstream=(1 << (j + 1)) - 1, mass=float(link_masses[j]))
        i3.inertia[:] = [float(v) for v in link_inertias[j]]
        ops.append(i3)
    return ops


def _random_unit_quat(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return q


def _dist_point_segment(p, a, b):
    ab = b - a
    t = np.clip(np.dot(p - a, ab) / np.dot(ab, ab), 0.0, 1.0)
    return np.linalg.norm(p - (a + t * ab))


def crs_like_chain():
    """6-R arm, CRS-A465-like geometry (all links along local z)."""
    axes = [(0, 0, 1), (0, -1, 0), (0, -1, 0), (0, 0, 1), (0, -1, 0), (0, 0, 1)]
    lengths = [0.33, 0.305, 0.15, 0.18, 0.076, 0.06]
    offsets = [(0.0, 0.0, L) for L in lengths]
    masses = [5.0, 4.0, 3.0, 2.0, 1.5, 1.0]
    inertias = [  # (a11,a12,a13,a22,a23,a33)
        (0.10, 0, 0, 0.10, 0, 0.05),
        (0.08, 0, 0, 0.08, 0, 0.02),
        (0.05, 0, 0, 0.05, 0, 0.01),
        (0.03, 0, 0, 0.03, 0, 0.01),
        (0.02, 0, 0, 0.02, 0, 0.01),
        (0.01, 0, 0, 0.01, 0, 0.01),
    ]
    joint_inertias = [1.0] * 6
    return axes, lengths, offsets, masses, inertias, joint_inertias


def flexible_beam_op(anchor_frame, world_pose, rest_length, stiffness, torsion_stiffness):
    """flexible_beam_3D (ctrl/mbd_kte/flexible_beam.cpp:155-193, no object frame) between a chain frame and an anchor fixed
    in the world."""
    op = T.KteOp(kind=T.KTE_FLEXIBLE_BEAM_3D, coord=-1, base_frame=anchor_frame, end_frame=-1, joint_op=-1, upstream=0)
    op.axis[:] = [rest_length, stiffness, torsion_stiffness]
    op.offset = world_pose
    return op


def make_c2(world_seed=1, n_obstacles=50, capsule_radius=0.05, steps_per_edge=20, dt=1e-3, tether=None, floor=None,
            n_cylinders=0, tool_sphere=None):
    """BASELINE config C2: 6-DOF revolute KTE chain, RK4 dt=1e-3 x 20 steps/edge, 50 convex obstacles.
    tether = (rest_length, stiffness, torsion_stiffness): a flexible_beam_3D from the end effector to a world anchor
    above the base (the beam parameters of BASELINE config C4: k = 1e4 N/m, k_theta = 1e2).
    floor = z: an 8 m x 8 m `plane` at that height (normal +z); n_cylinders: flat-ended `cylinder` obstacles (the
    reference only tests them against planes and spheres); tool_sphere = r: a `sphere` on the end effector."""
    rng = np.random.Generator(np.random.PCG64(world_seed))
    axes, lengths, offsets, masses, inertias, joint_inertias = crs_like_chain()
    n = len(axes)
    ops = serial_chain_ops(axes, offsets, masses, inertias, joint_inertias)
    if tether is not None:
        anchor = T.make_pose((0.3, 0.2, sum(lengths) + 0.1), (0.9238795325112867, 0.0, 0.3826834323650898, 0.0))
        ops.append(flexible_beam_op(2 * n, anchor, *tether))
    base = T.ChainBase()
    base.pose = T.make_pose()
    base.acceleration[:] = [0.0, 0.0, 9.81]  # gravity as base acceleration (mbd_kte/test_bm.cpp:52)

    shapes = []
    # robot model: one capsule per link, anchored on the joint's end frame, centred half-way along the link
    for j in range(n):
        s = T.Shape(kind=T.SHAPE_CCYLINDER, anchor=2 * j + 1)
        s.pose = T.make_pose((0.0, 0.0, 0.5 * lengths[j]))
        s.dims[:] = [lengths[j], capsule_radius, 0.0]
        shapes.append(s)
    # environment model: spheres / boxes / capsules in a 2 m cube, kept clear of the start pose
    reach_top = np.array([0.0, 0.0, sum(lengths)])
    origin = np.zeros(3)
    kinds = []
    while len(kinds) < n_obstacles:
        kind = [T.SHAPE_SPHERE, T.SHAPE_BOX, T.SHAPE_CCYLINDER][int(rng.integers(0, 3))]
        c = rng.uniform([-1.0, -1.0, -0.4], [1.0, 1.0, 1.6])
        s = T.Shape(kind=kind, anchor=-1)
        if kind == T.SHAPE_SPHERE:
            r = rng.uniform(0.05, 0.2)
            dims, brad, quat = [r, 0.0, 0.0], r, (1.0, 0.0, 0.0, 0.0)
        elif kind == T.SHAPE_BOX:
            d = rng.uniform(0.1, 0.4, size=3)
            dims, brad, quat = list(d), 0.5 * np.linalg.norm(d), _random_unit_quat(rng)
        else:
            L, r = rng.uniform(0.1, 0.4), rng.uniform(0.03, 0.1)
            dims, brad, quat = [L, r, 0.0], 0.5 * L + r, _random_unit_quat(rng)
        if _dist_point_segment(c, origin, reach_top) < brad + capsule_radius + 0.25:
            continue  # keep-out around the start pose (arm straight up) and the base column
        s.pose = T.make_pose(c, quat)
        s.dims[:] = [float(v) for v in dims]
        shapes.append(s)
        kinds.append(kind)

    if tool_sphere is not None:  # a spherical tool at the tip of the last link (robot model)
        s = T.Shape(kind=T.SHAPE_SPHERE, anchor=2 * (n - 1) + 1)
        s.pose = T.make_pose((0.0, 0.0, lengths[n - 1] + tool_sphere))
        s.dims[:] = [tool_sphere, 0.0, 0.0]
        shapes.insert(n, s)
    if floor is not None:
        s = T.Shape(kind=T.SHAPE_PLANE, anchor=-1)
        s.pose = T.make_pose((0.0, 0.0, float(floor)))
        s.dims[:] = [8.0, 8.0, 0.0]
        shapes.append(s)
    erng = np.random.Generator(np.random.PCG64(7000 + world_seed))
    placed = 0
    while placed < n_cylinders:
        c = erng.uniform([-0.9, -0.9, 0.0], [0.9, 0.9, 1.3])
        L, r = erng.uniform(0.15, 0.4), erng.uniform(0.05, 0.15)
        if _dist_point_segment(c, origin, reach_top) < float(np.hypot(r, 0.5 * L)) + capsule_radius + 0.25:
            continue
        s = T.Shape(kind=T.SHAPE_CYLINDER, anchor=-1)
        s.pose = T.make_pose(c, _random_unit_quat(erng))
        s.dims[:] = [L, r, 0.0]
        shapes.append(s)
        placed += 1

    dyn = T.DynSpace()
    dyn.n_dof = n
    dyn.steps_per_edge = steps_per_edge
    dyn.dt = dt
    dyn.kp, dyn.kd, dyn.u_max = 50.0, 10.0, 50.0
    dyn.goal_tol = 1e-3
    for j in range(n):
        dyn.lower[2 * j], dyn.upper[2 * j] = -np.pi, np.pi
        dyn.lower[2 * j + 1], dyn.upper[2 * j + 1] = -2.0, 2.0
    start = np.zeros(2 * n)
    goal = np.zeros(2 * n)
    goal[0::2] = [1.5, 0.9, -0.8, 0.5, 0.7, -0.3]
    return Scenario(name="C2", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=2 * n + 1,
                    start=start, goal=goal,
                    meta={"world_seed": world_seed, "n_obstacles": n_obstacles, "obstacle_kinds": kinds})


def make_pendulum(length=0.5, mass=1.0):
    """1-link pendulum (3D restatement of ctrl/mbd_kte/test_bm.cpp:45-77): revolute about -y at the
    origin, link of `length` along x, point mass at the tip, gravity as base acceleration +z."""
    ops = serial_chain_ops([(0, -1, 0)], [(length, 0.0, 0.0)], [mass], [(0, 0, 0, 0, 0, 0)], [0.0])
    base = T.ChainBase()
    base.pose = T.make_pose()
    base.acceleration[:] = [0.0, 0.0, 9.81]
    dyn = T.DynSpace()
    dyn.n_dof, dyn.steps_per_edge, dyn.dt = 1, 20, 1e-3
    dyn.kp, dyn.kd, dyn.u_max, dyn.goal_tol = 0.0, 0.0, 1e9, 0.0
    dyn.lower[0], dyn.upper[0], dyn.lower[1], dyn.upper[1] = -1e9, 1e9, -1e9, 1e9
    return Scenario(name="pendulum", ops=ops, base=base, shapes=[], dyn=dyn, n_dof=1, n_frames=3,
                    start=np.zeros(2), goal=np.zeros(2))


def make_c1(world_seed=1, n_obstacles=10, min_interval=0.05):
    """BASELINE config C1: 3-DOF planar arm, quasi-static RRT, 10 box obstacles, 5k nodes.
    The planar 3R arm of ctrl/kte_models/manip_3R_arm.cpp:45-152 (link lengths 0.5, 0.5, 0.3) is realised with the 3D
    KTEs (three revolute joints about z, links along x) so that it runs on the same kernels; links are capped cylinders
    of radius 0.04, obstacles are boxes extruded along z.  (The reference's 2D shapes / prox_crect_rectangle are not
    restated yet -- DESIGN.md section 7.)"""
    rng = np.random.Generator(np.random.PCG64(1000 + world_seed))
    lengths = [0.5, 0.5, 0.3]
    axes = [(0, 0, 1)] * 3
    offsets = [(L, 0.0, 0.0) for L in lengths]
    ops = serial_chain_ops(axes, offsets, [2.0, 1.5, 1.0], [(0.01, 0, 0, 0.05, 0, 0.05)] * 3, [0.1] * 3)
    base = T.ChainBase()
    base.pose = T.make_pose()
    n = 3
    shapes = []
    qy90 = (np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4), 0.0)  # capsule axis z -> link direction x
    for j in range(n):
        s = T.Shape(kind=T.SHAPE_CCYLINDER, anchor=2 * j + 1)
        s.pose = T.make_pose((0.5 * lengths[j], 0.0, 0.0), qy90)
        s.dims[:] = [lengths[j], 0.04, 0.0]
        shapes.append(s)
    placed = 0
    while placed < n_obstacles:
        c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 0.0])
        d = np.array([rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.5), 1.0])
        yaw = rng.uniform(-np.pi, np.pi)
        # keep the start pose (arm stretched along +x) and the goal pose (stretched along +y) clear
        brad = 0.5 * np.hypot(d[0], d[1])
        if _dist_point_segment(c, np.zeros(3), np.array([1.3, 0, 0])) < brad + 0.1:
            continue
        if _dist_point_segment(c, np.zeros(3), np.array([0, 1.3, 0])) < brad + 0.1:
            continue
        s = T.Shape(kind=T.SHAPE_BOX, anchor=-1)
        s.pose = T.make_pose(c, (np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)))
        s.dims[:] = [float(v) for v in d]
        shapes.append(s)
        placed += 1
    dyn = T.DynSpace()
    dyn.n_dof = n
    scn = Scenario(name="C1", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=2 * n + 1,
                   start=np.zeros(n), goal=np.array([np.pi / 2, 0.0, 0.0]),
                   meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi), "min_interval": min_interval})
    return scn


def planar_chain_ops(lengths, dynamics=False, masses=None, moments=None, joint_inertias=None):
    """kte_map_chain of a planar serial arm (manip_3R_arm.cpp:75-150): per joint {revolute_joint_2D, rigid_link_2D}, and
    with dynamics {driving_actuator_gen, inertia_gen, revolute_joint_2D, rigid_link_2D, inertia_2D at the link's end}."""
    ops = []
    for j, length in enumerate(lengths):
        if dynamics:
            ops.append(T.KteOp(kind=T.KTE_DRIVING_ACTUATOR_GEN, coord=j, base_frame=-1, end_frame=-1, joint_op=5 * j + 2))
            ops.append(T.KteOp(kind=T.KTE_INERTIA_GEN, coord=j, base_frame=-1, end_frame=-1, joint_op=-1, upstream=(1 << j),
                               mass=float(joint_inertias[j])))
        ops.append(T.KteOp(kind=T.KTE_REVOLUTE_JOINT_2D, coord=j, base_frame=2 * j, end_frame=2 * j + 1, joint_op=-1))
        l = T.KteOp(kind=T.KTE_RIGID_LINK_2D, coord=-1, base_frame=2 * j + 1, end_frame=2 * j + 2, joint_op=-1)
        l.offset = T.make_pose_2d((length, 0.0))
        ops.append(l)
        if dynamics:
            i2 = T.KteOp(kind=T.KTE_INERTIA_2D, coord=-1, base_frame=-1, end_frame=2 * j + 2, joint_op=-1,
                         upstream=(1 << (j + 1)) - 1, mass=float(masses[j]))
            i2.inertia[0] = float(moments[j])
            ops.append(i2)
    return ops


def make_c1_planar(world_seed=1, n_obstacles=10, min_interval=0.05, link_width=0.08, dynamics=False):
    """BASELINE config C1 with the reference's own 2D classes: the planar 3R arm of ctrl/kte_models/manip_3R_arm.cpp:45-152
    (revolute_joint_2D / rigid_link_2D, link lengths 0.5, 0.5, 0.3), links = capped_rectangle of width 0.08 anchored on
    the joints' end frames, obstacles = rectangle (0.2-0.5 m sides, centres uniform in [-1.5, 1.5]^2, uniform yaw),
    rejected if they come near the start (stretched along +x) or goal (along +y) configuration."""
    rng = np.random.Generator(np.random.PCG64(1000 + world_seed))
    lengths = [0.5, 0.5, 0.3]
    n = 3
    shapes = []
    ops = planar_chain_ops(lengths, dynamics=dynamics, masses=[3.0, 2.0, 1.0], moments=[0.06, 0.04, 0.01],
                           joint_inertias=[0.05, 0.04, 0.02])
    for j in range(n):
        s = T.Shape(kind=T.SHAPE_CRECT, anchor=2 * j + 1)
        s.pose = T.make_pose_2d((0.5 * lengths[j], 0.0))
        s.dims[:] = [lengths[j], link_width, 0.0]
        shapes.append(s)
    base = T.ChainBase()
    base.pose = T.make_pose_2d()
    placed = 0
    while placed < n_obstacles:
        c = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 0.0])
        d = np.array([rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.5)])
        yaw = rng.uniform(-np.pi, np.pi)
        brad = 0.5 * np.hypot(d[0], d[1])
        if _dist_point_segment(c, np.zeros(3), np.array([1.3, 0, 0])) < brad + 0.1:
            continue
        if _dist_point_segment(c, np.zeros(3), np.array([0, 1.3, 0])) < brad + 0.1:
            continue
        s = T.Shape(kind=T.SHAPE_RECTANGLE, anchor=-1)
        s.pose = T.make_pose_2d(c[:2], yaw)
        s.dims[:] = [float(d[0]), float(d[1]), 0.0]
        shapes.append(s)
        placed += 1
    dyn = T.DynSpace()
    dyn.n_dof = n
    start, goal = np.zeros(n), np.array([np.pi / 2, 0.0, 0.0])
    if dynamics:  # the steerable dynamic space over the planar arm (states (q, qd)), gravity along -y of the plane
        base.acceleration[:] = [0.0, 9.81, 0.0]
        dyn.steps_per_edge, dyn.dt = 20, 1e-3
        dyn.kp, dyn.kd, dyn.u_max, dyn.goal_tol = 50.0, 10.0, 50.0, 1e-3
        for j in range(n):
            dyn.lower[2 * j], dyn.upper[2 * j] = -np.pi, np.pi
            dyn.lower[2 * j + 1], dyn.upper[2 * j + 1] = -2.0, 2.0
        s0, g0 = np.zeros(2 * n), np.zeros(2 * n)
        s0[0::2], g0[0::2] = start, goal
        start, goal = s0, g0
    return Scenario(name="C1-planar", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=2 * n + 1,
                    start=start, goal=goal,
                    meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi), "min_interval": min_interval})


def make_planar_mixed(seed=1, n=4, n_obstacles=24):
    """A planar n-joint arm with mixed 2D shapes on both sides (circles, capped rectangles and rectangles on the links;
    all three kinds in the environment) and non-trivial link / shape poses: exercises every planar pair routine."""
    rng = np.random.Generator(np.random.PCG64(9100 + seed))
    ops, shapes = [], []
    lengths = list(rng.uniform(0.25, 0.45, size=n))
    for j in range(n):
        ops.append(T.KteOp(kind=T.KTE_REVOLUTE_JOINT_2D, coord=j, base_frame=2 * j, end_frame=2 * j + 1, joint_op=-1))
        l = T.KteOp(kind=T.KTE_RIGID_LINK_2D, coord=-1, base_frame=2 * j + 1, end_frame=2 * j + 2, joint_op=-1)
        l.offset = T.make_pose_2d((lengths[j], rng.uniform(-0.03, 0.03)), rng.uniform(-0.2, 0.2))
        ops.append(l)
        kind = [T.SHAPE_CRECT, T.SHAPE_CIRCLE, T.SHAPE_RECTANGLE][j % 3]
        s = T.Shape(kind=kind, anchor=2 * j + 1)
        s.pose = T.make_pose_2d((0.5 * lengths[j], 0.0), 0.0 if kind == T.SHAPE_CRECT else rng.uniform(-0.3, 0.3))
        s.dims[:] = {T.SHAPE_CRECT: [lengths[j], 0.06, 0.0], T.SHAPE_CIRCLE: [0.07, 0.0, 0.0],
                     T.SHAPE_RECTANGLE: [lengths[j] * 0.8, 0.07, 0.0]}[kind]
        shapes.append(s)
    base = T.ChainBase()
    base.pose = T.make_pose_2d((0.05, -0.02), 0.3)
    placed = 0
    while placed < n_obstacles:
        c = rng.uniform(-1.6, 1.6, size=2)
        if np.hypot(*c) < 0.55:
            continue
        kind = [T.SHAPE_RECTANGLE, T.SHAPE_CIRCLE, T.SHAPE_CRECT][placed % 3]
        s = T.Shape(kind=kind, anchor=-1)
        s.pose = T.make_pose_2d(c, rng.uniform(-np.pi, np.pi))
        s.dims[:] = {T.SHAPE_RECTANGLE: [rng.uniform(0.15, 0.4), rng.uniform(0.15, 0.4), 0.0],
                     T.SHAPE_CIRCLE: [rng.uniform(0.06, 0.18), 0.0, 0.0],
                     T.SHAPE_CRECT: [rng.uniform(0.15, 0.4), rng.uniform(0.06, 0.16), 0.0]}[kind]
        shapes.append(s)
        placed += 1
    dyn = T.DynSpace()
    dyn.n_dof = n
    goal = rng.uniform(-1.2, 1.2, size=n)
    return Scenario(name=f"planar{n}", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=2 * n + 1,
                    start=np.zeros(n), goal=goal,
                    meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi), "min_interval": 0.05})


def make_c3(world_seed=1, min_interval=0.05):
    """BASELINE config C3/C5: the 6-DOF chain and 50-obstacle world of C2, planned in the quasi-static joint space
    (manip_quasi_static_env) with RRT* and star_neighborhood k-NN rewiring."""
    c2 = make_c2(world_seed=world_seed)
    n = c2.n_dof
    scn = Scenario(name="C3", ops=c2.ops, base=c2.base, shapes=c2.shapes, dyn=c2.dyn, n_dof=n, n_frames=c2.n_frames,
                   start=np.zeros(n), goal=np.array([1.5, 0.9, -0.8, 0.5, 0.7, -0.3]),
                   meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi), "min_interval": min_interval})
    return scn


def random_convex_mesh(rng, n_vertices, radius):
    """A random convex vertex set: points on an ellipsoid (every point of an ellipsoid's surface is an extreme point of
    the set) with semi-axes in [0.5, 1] x radius, so the bounding radius about the local origin is <= radius."""
    axes = radius * rng.uniform(0.5, 1.0, size=3)
    axes[int(rng.integers(0, 3))] = radius
    u = rng.normal(size=(n_vertices, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    return u * axes[None, :]


def box_as_mesh(dims):
    """The eight corners of a box (for checking the GJK mesh path against the reference's box routines)."""
    hx, hy, hz = 0.5 * dims[0], 0.5 * dims[1], 0.5 * dims[2]
    return np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])


def make_c4(world_seed=1, n_obstacles=200, capsule_radius=0.05, mount_y=0.35, min_interval=0.05, beam=None, meshes=False):
    """BASELINE config C4 (kinematic part): 12-DOF dual arm -- two CRS-like 6-R arms on fixed mounts (rigid links from
    the chain base) at y = -/+ mount_y -- and 200 convex obstacles, planned in the quasi-static joint space (PRM).
    meshes=False: the obstacles are the reference's own shapes (spheres, boxes, capped cylinders; closed forms).
    meshes=True: C4 as BASELINE.json words it -- 200 convex MESH obstacles (random convex vertex sets of 12-32 vertices,
    bounding radius 0.05-0.25 m) evaluated by batched GJK; the reference has no such shape, the build's
    definition is in reak_amd/csrc/gjk_device.h.  The flexible beam of C4 is a force element and plays no role in a
    quasi-static space."""
    rng = np.random.Generator(np.random.PCG64(4000 + world_seed))
    axes, lengths, offsets, masses, inertias, joint_inertias = crs_like_chain()
    n1 = len(axes)
    ops, shapes = [], []
    frame = 1  # frame 0 = chain base
    for arm in range(2):
        mount = T.KteOp(kind=T.KTE_RIGID_LINK_3D, coord=-1, base_frame=0, end_frame=frame, joint_op=-1)
        mount.offset = T.make_pose((0.0, (-1.0 if arm == 0 else 1.0) * mount_y, 0.0))
        ops.append(mount)
        prev = frame
        frame += 1
        c0 = arm * n1
        for j in range(n1):
            c = c0 + j
            k0 = len(ops)
            ops.append(T.KteOp(kind=T.KTE_DRIVING_ACTUATOR_GEN, coord=c, base_frame=-1, end_frame=-1, joint_op=k0 + 2))
            ops.append(T.KteOp(kind=T.KTE_INERTIA_GEN, coord=c, base_frame=-1, end_frame=-1, joint_op=-1, upstream=(1 << c),
                               mass=float(joint_inertias[j])))
            r = T.KteOp(kind=T.KTE_REVOLUTE_JOINT_3D, coord=c, base_frame=prev, end_frame=frame, joint_op=-1)
            r.axis[:] = [float(v) for v in axes[j]]
            ops.append(r)
            l = T.KteOp(kind=T.KTE_RIGID_LINK_3D, coord=-1, base_frame=frame, end_frame=frame + 1, joint_op=-1)
            l.offset = T.make_pose(offsets[j])
            ops.append(l)
            i3 = T.KteOp(kind=T.KTE_INERTIA_3D, coord=-1, base_frame=-1, end_frame=frame + 1, joint_op=-1,
                         upstream=((1 << (c + 1)) - 1) & ~((1 << c0) - 1), mass=float(masses[j]))
            i3.inertia[:] = [float(v) for v in inertias[j]]
            ops.append(i3)
            s = T.Shape(kind=T.SHAPE_CCYLINDER, anchor=frame)  # link capsule on the joint's end frame
            s.pose = T.make_pose((0.0, 0.0, 0.5 * lengths[j]))
            s.dims[:] = [lengths[j], capsule_radius, 0.0]
            shapes.append(s)
            prev = frame + 1
            frame += 2
    tips = []
    for op in ops:
        if op.kind == T.KTE_INERTIA_3D:
            tips.append(op.end_frame)
    if beam is not None:  # flexible_beam_3D between the two end effectors (BASELINE C4: k = 1e4 N/m, k_theta = 1e2)
        b = T.KteOp(kind=T.KTE_FLEXIBLE_BEAM_3D, coord=-1, base_frame=tips[n1 - 1], end_frame=tips[2 * n1 - 1], joint_op=-1,
                    upstream=0)
        b.axis[:] = list(beam)
        b.offset = T.make_pose()
        ops.append(b)
    base = T.ChainBase()
    base.pose = T.make_pose()
    base.acceleration[:] = [0.0, 0.0, 9.81]
    # environment: spheres / boxes / capsules around both arms, clear of the upright start poses
    tops = [np.array([0.0, -mount_y, sum(lengths)]), np.array([0.0, mount_y, sum(lengths)])]
    roots = [np.array([0.0, -mount_y, 0.0]), np.array([0.0, mount_y, 0.0])]
    kinds = []
    pool = []
    while meshes and len(kinds) < n_obstacles:
        center = rng.uniform([-1.0, -1.4, 0.0], [1.0, 1.4, 1.4])
        brad = rng.uniform(0.05, 0.25)
        if min(_dist_point_segment(center, roots[a], tops[a]) for a in range(2)) < brad + capsule_radius + 0.12:
            continue
        verts = random_convex_mesh(rng, int(rng.integers(12, 33)), brad)
        s = T.Shape(kind=T.SHAPE_MESH, anchor=-1)
        s.dims[:] = [float(sum(len(v) for v in pool)), float(len(verts)), 0.0]
        s.pose = T.make_pose(center, _random_unit_quat(rng))
        pool.append(verts)
        shapes.append(s)
        kinds.append(T.SHAPE_MESH)
    while len(kinds) < n_obstacles:
        kind = [T.SHAPE_SPHERE, T.SHAPE_BOX, T.SHAPE_CCYLINDER][int(rng.integers(0, 3))]
        center = rng.uniform([-1.0, -1.4, 0.0], [1.0, 1.4, 1.4])
        s = T.Shape(kind=kind, anchor=-1)
        if kind == T.SHAPE_SPHERE:
            s.dims[:] = [rng.uniform(0.05, 0.15), 0.0, 0.0]
            brad = s.dims[0]
        elif kind == T.SHAPE_BOX:
            s.dims[:] = list(rng.uniform(0.06, 0.24, size=3))
            brad = 0.5 * float(np.linalg.norm([s.dims[0], s.dims[1], s.dims[2]]))
        else:
            s.dims[:] = [rng.uniform(0.1, 0.3), rng.uniform(0.03, 0.08), 0.0]
            brad = 0.5 * s.dims[0] + s.dims[1]
        if min(_dist_point_segment(center, roots[a], tops[a]) for a in range(2)) < brad + capsule_radius + 0.12:
            continue
        s.pose = T.make_pose(center, _random_unit_quat(rng))
        shapes.append(s)
        kinds.append(kind)
    n = 2 * n1
    goal = np.array([1.2, 0.8, -0.7, 0.4, 0.6, -0.3, -1.0, 0.7, -0.9, -0.5, 0.5, 0.4])
    dyn = T.DynSpace()   # the steerable dynamic space of C2, for both arms
    dyn.n_dof, dyn.steps_per_edge, dyn.dt = n, 20, 1e-3
    dyn.kp, dyn.kd, dyn.u_max, dyn.goal_tol = 50.0, 10.0, 50.0, 1e-3
    for j in range(n):
        dyn.lower[2 * j], dyn.upper[2 * j] = -np.pi, np.pi
        dyn.lower[2 * j + 1], dyn.upper[2 * j + 1] = -2.0, 2.0
    return Scenario(name="C4", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=frame,
                    start=np.zeros(n), goal=goal,
                    meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi), "min_interval": min_interval,
                          "world_seed": world_seed, "n_obstacles": n_obstacles, "obstacle_kinds": kinds},
                    mesh_vertices=(np.concatenate(pool) if pool else None))


def make_random_chain(n, seed=1, n_obstacles=12):
    """A random n-joint serial chain (oblique joint axes, skew link offsets, full inertia tensors) with capsule links
    and a few obstacles: exercises the kernels' template instantiations beyond the BASELINE configurations."""
    rng = np.random.Generator(np.random.PCG64(7000 + 31 * n + seed))
    axes = [tuple(v / np.linalg.norm(v)) for v in rng.normal(size=(n, 3))]
    lengths = list(rng.uniform(0.12, 0.3, size=n))
    offsets = [tuple(np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), L])) for L in lengths]
    masses = list(rng.uniform(0.5, 4.0, size=n))
    inertias = []
    for _ in range(n):
        a = rng.uniform(-0.02, 0.02, size=(3, 3))
        m = a @ a.T + np.diag(rng.uniform(0.01, 0.05, size=3))
        inertias.append((m[0, 0], m[0, 1], m[0, 2], m[1, 1], m[1, 2], m[2, 2]))
    ops = serial_chain_ops(axes, offsets, masses, inertias, list(rng.uniform(0.2, 1.0, size=n)))
    base = T.ChainBase()
    base.pose = T.make_pose((0.1, -0.2, 0.05), _random_unit_quat(rng))
    base.acceleration[:] = [0.0, 0.0, 9.81]
    shapes = []
    for j in range(n):
        s = T.Shape(kind=T.SHAPE_CCYLINDER, anchor=2 * j + 1)
        s.pose = T.make_pose((0.0, 0.0, 0.5 * lengths[j]))
        s.dims[:] = [lengths[j], 0.04, 0.0]
        shapes.append(s)
    for _ in range(n_obstacles):
        kind = [T.SHAPE_SPHERE, T.SHAPE_BOX, T.SHAPE_CCYLINDER][int(rng.integers(0, 3))]
        s = T.Shape(kind=kind, anchor=-1)
        c = rng.uniform(-1.0, 1.0, size=3)
        c *= max(1.0, 0.9 / np.linalg.norm(c))  # keep a clear ball around the base
        s.pose = T.make_pose(c, _random_unit_quat(rng))
        s.dims[:] = {T.SHAPE_SPHERE: [rng.uniform(0.05, 0.15), 0, 0], T.SHAPE_BOX: list(rng.uniform(0.1, 0.3, size=3)),
                     T.SHAPE_CCYLINDER: [rng.uniform(0.1, 0.3), rng.uniform(0.03, 0.08), 0]}[kind]
        shapes.append(s)
    dyn = T.DynSpace()
    dyn.n_dof, dyn.steps_per_edge, dyn.dt = n, 20, 1e-3
    dyn.kp, dyn.kd, dyn.u_max, dyn.goal_tol = 40.0, 8.0, 40.0, 1e-3
    for j in range(n):
        dyn.lower[2 * j], dyn.upper[2 * j] = -np.pi, np.pi
        dyn.lower[2 * j + 1], dyn.upper[2 * j + 1] = -2.0, 2.0
    goal = np.zeros(2 * n)
    goal[0::2] = rng.uniform(-1.0, 1.0, size=n)
    return Scenario(name=f"chain{n}", ops=ops, base=base, shapes=shapes, dyn=dyn, n_dof=n, n_frames=2 * n + 1,
                    start=np.zeros(2 * n), goal=goal, meta={"lower": np.full(n, -np.pi), "upper": np.full(n, np.pi),
                                                            "min_interval": 0.05})


def make_hidim(D, min_interval=0.05):
    """The reference's own planner scenario (R/ctrl/path_planning/test_hidim_planners.cpp:151,195-212): the unit hypercube
    [0, 1]^D without obstacles, start 0.05 * 1, goal 0.95 * 1.  Here it is a quasi-static space over a D-joint chain that
    carries no shapes and an environment without shapes -- ZERO proximity pairs, so is_free is the hyperbox test alone
    and every edge walk reaches its target.  Not modelled: no_obstacle_space clips an edge at max_edge_length = 0.2
    sqrt(D) (no_obstacle_space.hpp:178-184) and reports targets beyond it as unreachable; manip_quasi_static_env, the
    space on the hot path, has no such cap, so edges run to their samples.  The reference holds no expected output for
    the scenario (it pins nothing); the tests compare the device with the oracle on it."""
    axes = [((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))[j % 3] for j in range(D)]
    offsets = [(0.0, 0.0, 0.1)] * D
    ops = serial_chain_ops(axes, offsets, [1.0] * D, [(0.01, 0, 0, 0.01, 0, 0.01)] * D, [0.1] * D)
    base = T.ChainBase()
    base.pose = T.make_pose()
    dyn = T.DynSpace()
    dyn.n_dof = D
    return Scenario(name=f"hidim{D}", ops=ops, base=base, shapes=[], dyn=dyn, n_dof=D, n_frames=2 * D + 1,
                    start=np.full(D, 0.05), goal=np.full(D, 0.95),
                    meta={"lower": np.zeros(D), "upper": np.ones(D), "min_interval": min_interval,
                          "max_edge_length": 0.2 * np.sqrt(D)})


def nlp_proximity_shapes():
    """The shapes and poses of R/geometry/proximity/test_nlp_proximity.cpp:40-58: six cylinders and six boxes at four poses
    (a1 .. a4).  The reference feeds them to an NLP solver and prints what it finds -- it holds no expected distances --
    and has no closed form for these pair kinds (cylinder-cylinder, cylinder-box, box-box: proxy_query_model.cpp:215-374
    creates no finder); here they are a vector set for the support-map (GJK) distance query.  Returns {name: Shape}."""
    s3 = np.sqrt(3.0) / 3.0
    poses = {"a1": ((0.0, 0.0, 0.0), (0.8, 0.0, 0.6, 0.0)), "a2": ((0.0, 3.0, 5.0), (0.8, -0.6, 0.0, 0.0)),
             "a3": ((10.0, -3.0, -2.0), (1.0, 0.0, 0.0, 0.0)), "a4": ((-3.0, -3.0, 6.0), (s3, 0.0, -s3, s3))}
    spec = {"cy1": ("a1", T.SHAPE_CYLINDER, (5.0, 0.5, 0.0)), "cy2": ("a1", T.SHAPE_CYLINDER, (10.0, 0.25, 0.0)),
            "cy3": ("a1", T.SHAPE_CYLINDER, (1.0, 2.0, 0.0)), "cy4": ("a2", T.SHAPE_CYLINDER, (5.0, 0.5, 0.0)),
            "cy5": ("a3", T.SHAPE_CYLINDER, (5.0, 0.5, 0.0)), "cy6": ("a4", T.SHAPE_CYLINDER, (5.0, 0.5, 0.0)),
            "bx1": ("a1", T.SHAPE_BOX, (1.0, 2.0, 1.0)), "bx2": ("a1", T.SHAPE_BOX, (4.0, 1.0, 10.0)),
            "bx3": ("a1", T.SHAPE_BOX, (4.0, 4.0, 1.0)), "bx4": ("a2", T.SHAPE_BOX, (4.0, 2.0, 2.0)),
            "bx5": ("a3", T.SHAPE_BOX, (4.0, 2.0, 2.0)), "bx6": ("a4", T.SHAPE_BOX, (4.0, 2.0, 2.0))}
    out = {}
    for name, (pose, kind, dims) in spec.items():
        s = T.Shape(kind=kind, anchor=-1)
        s.pose = T.make_pose(*poses[pose])
        s.dims[:] = dims
        out[name] = s
    return out


# the pairs test_nlp_proximity.cpp:211-233 queues: cylinder-cylinder, box-box, then box-cylinder and cylinder-box
NLP_PROXIMITY_PAIRS = [("cy1", "cy4"), ("cy1", "cy5"), ("cy1", "cy6"), ("cy2", "cy4"), ("cy3", "cy4"),
                       ("bx1", "bx4"), ("bx1", "bx5"), ("bx1", "bx6"), ("bx2", "bx4"), ("bx3", "bx4"),
                       ("bx1", "cy4"), ("bx1", "cy5"), ("bx1", "cy6"), ("bx2", "cy4"), ("bx3", "cy4"),
                       ("cy1", "bx4"), ("cy1", "bx5"), ("cy1", "bx6"), ("cy2", "bx4"), ("cy3", "bx4")]
