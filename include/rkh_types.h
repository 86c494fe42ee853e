/* rkh_types.h -- plain-old-data scene description shared across the C-ABI boundary.
 *
 * Everything here is a flat C struct (no pointers to C++ objects) so that a
 * ReaK-side adaptor can flatten its shared_ptr object graph once and hand it over.
 * Each struct names the ReaK class whose *state* it carries (paths relative to
 * /root/reference/src/ReaK/).
 */
#ifndef RKH_TYPES_H
#define RKH_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RKH_MAX_DOF 16
#define RKH_MAX_STATE (2 * RKH_MAX_DOF)

/* pose_3D<double> (core/kinetostatics/pose_3D.hpp): Position + Quat (w,x,y,z). Parent is implicit. */
typedef struct rkh_pose {
  double pos[3];
  double quat[4];
} rkh_pose;

/* KTE op-codes: one entry of kte_map_chain::mKTEs (ctrl/mbd_kte/kte_map_chain.hpp:49-112).
 * doMotion runs the ops in order, doForce in reverse order (kte_map_chain.hpp:71-83). */
enum rkh_kte_kind {
  RKH_KTE_DRIVING_ACTUATOR_GEN = 1, /* ctrl/mbd_kte/driving_actuator.cpp:31-39  : coord, joint_op            */
  RKH_KTE_INERTIA_GEN = 2,          /* ctrl/mbd_kte/inertia.cpp:47-54           : coord, mass                */
  RKH_KTE_REVOLUTE_JOINT_3D = 3,    /* ctrl/mbd_kte/revolute_joint.cpp:121-213  : coord, axis, base, end     */
  RKH_KTE_RIGID_LINK_3D = 4,        /* ctrl/mbd_kte/rigid_link.cpp:152-186      : base, end, pose offset     */
  RKH_KTE_INERTIA_3D = 5,           /* ctrl/mbd_kte/inertia.cpp:111-122         : frame(end), mass, tensor   */
  /* flexible_beam_3D without an object frame (ctrl/mbd_kte/flexible_beam.cpp:155-193): a linear + torsional spring
   * between two anchors.  base_frame = mAnchor1 (a chain frame); end_frame = mAnchor2 (a chain frame) or -1 for an
   * anchor fixed in the world at pose `offset`; axis[0] = mRestLength, axis[1] = mStiffness, axis[2] = mTorsionStiffness.
   * The HIP kernels support one beam, listed last in the chain, between link end frames (or to a world anchor). */
  RKH_KTE_FLEXIBLE_BEAM_3D = 6,
  /* Planar chains (position level: quasi-static free spaces).  Poses are pose_2D<double> carried in rkh_pose as
   * pos[0..1] = Position and quat[0..1] = rot_mat_2D::q = (cos, sin) (core/kinetostatics/rotations_2D.hpp:89);
   * a chain is planar when its joints are REVOLUTE_JOINT_2D, and then all its links and shapes must be 2D. */
  RKH_KTE_REVOLUTE_JOINT_2D = 7,    /* ctrl/mbd_kte/revolute_joint.cpp:30-97     : coord, base, end           */
  RKH_KTE_RIGID_LINK_2D = 8,        /* ctrl/mbd_kte/rigid_link.cpp:87-128        : base, end, pose offset     */
  RKH_KTE_INERTIA_2D = 9            /* ctrl/mbd_kte/inertia.cpp:60-87            : frame(end), mass, inertia[0] = mMomentOfInertia */
};

typedef struct rkh_kte_op {
  int32_t kind;        /* rkh_kte_kind */
  int32_t coord;       /* generalized coordinate index (gen_coord), or -1 */
  int32_t base_frame;  /* frame index (frame_3D), or -1 */
  int32_t end_frame;   /* frame index (frame_3D), or -1 */
  int32_t joint_op;    /* DRIVING_ACTUATOR_GEN: index of the joint op receiving applyReactionForce */
  uint32_t upstream;   /* INERTIA_*: bitmask of coords in mUpStreamJoints (jacobian_joint_map.hpp) */
  double axis[3];      /* REVOLUTE_JOINT_3D: mAxis */
  rkh_pose offset;     /* RIGID_LINK_3D: mPoseOffset */
  double mass;         /* INERTIA_GEN / INERTIA_3D: mMass */
  double inertia[6];   /* INERTIA_3D: mInertiaTensor, symmetric (a11,a12,a13,a22,a23,a33) */
} rkh_kte_op;

/* Base frame of the chain (frame index 0). Gravity enters as base Acceleration
 * (ctrl/mbd_kte/test_bm.cpp:52). The base frame has no Parent (global). */
typedef struct rkh_chain_base {
  rkh_pose pose;
  double acceleration[3];
} rkh_chain_base;

/* shape_3D subclasses (geometry/shapes/{sphere,box,capped_cylinder}.hpp) with their geometry_3D
 * anchor + pose (geometry/shapes/geometry_3D.cpp:32-51). */
enum rkh_shape_kind {
  RKH_SHAPE_SPHERE = 1,    /* dims[0] = radius                              */
  RKH_SHAPE_BOX = 2,       /* dims[0..2] = full side lengths (mDimensions)  */
  RKH_SHAPE_CCYLINDER = 3, /* dims[0] = length, dims[1] = radius (capped_cylinder, axis = local z) */
  /* shape_2D subclasses (geometry/shapes/{circle,rectangle,capped_rectangle}.hpp), pose = pose_2D (see above) */
  RKH_SHAPE_CIRCLE = 4,    /* dims[0] = radius                                                      */
  RKH_SHAPE_RECTANGLE = 5, /* dims[0..1] = full side lengths (mDimensions)                          */
  RKH_SHAPE_CRECT = 6,     /* capped_rectangle: dims[0] = length along local x, dims[1] = width = cap diameter */
  /* more shape_3D subclasses (geometry/shapes/{plane,cylinder}.hpp) */
  RKH_SHAPE_PLANE = 7,     /* dims[0..1] = mDimensions (x, y extents; they only enter the bounding radius of the cull:
                            * the enabled prox_plane_* routines treat the plane as infinite), normal = local z          */
  RKH_SHAPE_CYLINDER = 8,  /* dims[0] = length, dims[1] = radius (flat ends, axis = local z)                          */
  /* Convex vertex set ("mesh"): NOT a reference class (its proximity module has closed forms only, TODO_list.txt:230);
   * BASELINE config C4's obstacles.  dims[0] = index of the shape's first vertex in the scene's vertex pool, dims[1] =
   * number of vertices (the shape is their convex hull; local coordinates).  Distances through GJK (rkh.h). */
  RKH_SHAPE_MESH = 9
};

typedef struct rkh_shape {
  int32_t kind;    /* rkh_shape_kind */
  int32_t anchor;  /* frame index of the KTE chain the shape is anchored to, -1 = world (no anchor) */
  rkh_pose pose;   /* mPose relative to the anchor */
  double dims[3];
} rkh_shape;

/* Steerable dynamic free space over a KTE chain ("kte_dynamic_free_space").  There is no verbatim
 * reference class; the loop shape restates examples/misc/MEAQR_topology.hpp:503-565 with a PD law
 * held constant over each RK4 step (ctrl/interpolation/constant_trajectory.hpp:138-150).
 * State layout is kte_nl_system's: x = (q0, qd0, q1, qd1, ...) (ctrl/ctrl_sys/kte_nl_system.hpp:190-193). */
typedef struct rkh_dyn_space {
  int32_t n_dof;
  int32_t steps_per_edge; /* RK4 steps of one full steer (fraction 1.0) */
  double dt;              /* RK4 step (one runge_kutta4_integrate_impl call per step) */
  double kp, kd, u_max;   /* u_i = clamp(kp (q*_i - q_i) + kd (qd*_i - qd_i), +-u_max) */
  double goal_tol;        /* steering stops once distance(x, target) <= goal_tol */
  double lower[RKH_MAX_STATE]; /* hyperbox_topology lower_corner (interleaved q, qd) */
  double upper[RKH_MAX_STATE]; /* hyperbox_topology upper_corner */
} rkh_dyn_space;

/* manip_quasi_static_env (ctrl/topologies/manip_free_workspace.hpp:113-300) over a hyperbox joint space with linear
 * interpolation: points are joint positions (D = n_dof), edges are walked in min_interval steps.
 * speed_limits: the space may be the RATE-LIMITED joint space the reference's manipulator environments plan in
 * (Ndof_rl_space, ctrl/topologies/Ndof_spaces.hpp; joint_limits_collection::map_to_space, joint_space_limits.tpp:62-84):
 * a point holds reach times t_i = q_i / gen_speed_limits[i] (joint_space_limits_detail.hpp:1552-1561), hyperbox, metric
 * and interpolation live in those coordinates, and the model is evaluated at q_i = t_i * gen_speed_limits[i]
 * (:1868-1875) before every proximity test.  0 (or 1) = an ordinary joint space. */
typedef struct rkh_qs_space {
  int32_t n_dof;
  int32_t pad;
  double min_interval;
  double lower[RKH_MAX_DOF];
  double upper[RKH_MAX_DOF];
  double speed_limits[RKH_MAX_DOF];
} rkh_qs_space;

/* sample_based_planner options (ctrl/path_planning/motion_planner_base.hpp:400-422) and the
 * point-to-point query (ctrl/path_planning/p2p_planning_query.hpp:74-229). */
typedef struct rkh_rrt_params {
  uint32_t seed;            /* get_global_rng().seed(seed) */
  uint32_t max_vertices;    /* m_max_vertex_count */
  uint32_t max_results;     /* path_planning_p2p_query::max_num_results */
  double steer_tol;         /* m_steer_progress_tolerance (default 0.1) */
  double conn_tol;          /* m_connection_tolerance (default 0.05) */
  double start[RKH_MAX_STATE];
  double goal[RKH_MAX_STATE];
} rkh_rrt_params;

/* prm_planner (ctrl/path_planning/prm_path_planner.tpp:131-365): the options above plus the sampling radius used
 * by random_walk (planning_visitors.hpp:403-432) and prm_density_calculator (density_calculators.hpp:45-73), and the
 * expansion probability (fixed at 0.2 by the reference's generate_prm call, prm_path_planner.tpp:250-253). */
typedef struct rkh_prm_params {
  rkh_rrt_params base;
  double sampling_radius;     /* m_sampling_radius */
  double expand_probability;  /* 0.2 in the reference */
} rkh_prm_params;

#ifdef __cplusplus
}
#endif
#endif
