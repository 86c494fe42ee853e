// scene.hip -- C-ABI: scene flattening (KTE program -> serial-chain tables, proxy pair list) and the
// kernel-level entry points rkh_state_derivative / rkh_min_distance / rkh_propagate (include/rkh.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "proximity_device.h"
#include "rkh_internal.h"

using namespace rkh;

namespace rkh {

// quaternion::getRotMat (rotations_3D.hpp:986-999), host copy used once per scene for the constant link offsets
static void host_rotmat(const double* q, double* R) {
  const double t01 = 2.0 * q[0] * q[1], t02 = 2.0 * q[0] * q[2], t03 = 2.0 * q[0] * q[3];
  const double t11 = 2.0 * q[1] * q[1], t12 = 2.0 * q[1] * q[2], t13 = 2.0 * q[1] * q[3];
  const double t22 = 2.0 * q[2] * q[2], t23 = 2.0 * q[2] * q[3], t33 = 2.0 * q[3] * q[3];
  const double r[9] = {1.0 - t22 - t33, t12 - t03, t02 + t13, t12 + t03, 1.0 - t11 - t33, t23 - t01,
                       t13 - t02,       t01 + t23, 1.0 - t11 - t22};
  std::memcpy(R, r, sizeof(r));
}

static thread_local const double* g_mesh_vertices = nullptr;
static thread_local uint32_t g_n_mesh_vertices = 0;

static bool mesh_range_ok(const rkh_shape& s) {
  const double first = s.dims[0], cnt = s.dims[1];
  return first >= 0.0 && cnt >= 1.0 && first == std::floor(first) && cnt == std::floor(cnt) &&
         first + cnt <= double(g_n_mesh_vertices);
}

static double bounding_radius(const rkh_shape& s) {
  switch (s.kind) {
    case RKH_SHAPE_MESH: {  // about the local origin, like the reference's shapes
      double r2 = 0.0;
      for (int i = 0; i < int(s.dims[1]); ++i) {
        const double* v = g_mesh_vertices + 3 * (size_t(s.dims[0]) + i);
        r2 = std::max(r2, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      }
      return std::sqrt(r2);
    }
    case RKH_SHAPE_SPHERE: return s.dims[0];
    case RKH_SHAPE_BOX: {
      double acc = 0.0;
      for (int i = 0; i < 3; ++i) acc += s.dims[i] * s.dims[i];
      return std::sqrt(acc) * 0.5;
    }
    case RKH_SHAPE_CCYLINDER: return s.dims[0] * 0.5 + s.dims[1];
    case RKH_SHAPE_PLANE: {  // plane.cpp:31-33
      double acc = 0.0;
      for (int i = 0; i < 2; ++i) acc += s.dims[i] * s.dims[i];
      return std::sqrt(acc) * 0.5;
    }
    case RKH_SHAPE_CYLINDER: return std::sqrt(s.dims[1] * s.dims[1] + 0.25 * s.dims[0] * s.dims[0]);  // cylinder.cpp:33-35
  }
  return 0.0;
}

// createProxFinderList's cascade of kinds for one pair (the host twin of pair_routine() in proximity_device.h)
static int host_pair_routine(int ka, int kb, bool* a_is_shape1) {
  auto other = [&](int first_kind) { return (ka == first_kind) ? kb : ka; };
  *a_is_shape1 = true;
  if (ka == RKH_SHAPE_MESH || kb == RKH_SHAPE_MESH) {  // the build's GJK query (no reference finder)
    const int ko = other(RKH_SHAPE_MESH);
    return (ko == RKH_SHAPE_SPHERE || ko == RKH_SHAPE_CCYLINDER || ko == RKH_SHAPE_BOX || ko == RKH_SHAPE_MESH) ? 12 : 0;
  }
  if (ka == RKH_SHAPE_PLANE || kb == RKH_SHAPE_PLANE) {
    *a_is_shape1 = (ka == RKH_SHAPE_PLANE);
    switch (other(RKH_SHAPE_PLANE)) {
      case RKH_SHAPE_PLANE: return 6;
      case RKH_SHAPE_SPHERE: return 7;
      case RKH_SHAPE_CCYLINDER: return 8;
      case RKH_SHAPE_CYLINDER: return 9;
      case RKH_SHAPE_BOX: return 10;
    }
    return 0;
  }
  if (ka == RKH_SHAPE_SPHERE || kb == RKH_SHAPE_SPHERE) {
    *a_is_shape1 = (ka == RKH_SHAPE_SPHERE);
    switch (other(RKH_SHAPE_SPHERE)) {
      case RKH_SHAPE_SPHERE: return 1;
      case RKH_SHAPE_CCYLINDER: return 2;
      case RKH_SHAPE_CYLINDER: return 11;
      case RKH_SHAPE_BOX: return 3;
    }
    return 0;
  }
  if (ka == RKH_SHAPE_CCYLINDER || kb == RKH_SHAPE_CCYLINDER) {
    *a_is_shape1 = (ka == RKH_SHAPE_CCYLINDER);
    switch (other(RKH_SHAPE_CCYLINDER)) {
      case RKH_SHAPE_CCYLINDER: return 4;
      case RKH_SHAPE_BOX: return 5;
    }
    return 0;
  }
  return 0;  // cylinder-cylinder, cylinder-box, box-box: no finder in the reference
}

// The time loops of the steer pattern and of runge_kutta4_integrate_impl depend only on (fraction, dt,
// steps): evaluate them on the host in the same fp64 arithmetic and hand the kernel integer counts.
rkh_status build_dyn_dev(const rkh_dyn_space& sp, double fraction, DynDev* out) {
  if (sp.n_dof < 1 || sp.n_dof > kMaxDof || sp.steps_per_edge < 0 || sp.steps_per_edge > kMaxSteps || !(sp.dt > 0.0)) {
    set_error("rkh_dyn_space: n_dof / steps_per_edge / dt out of range");
    return RKH_ERR_BAD_ARG;
  }
  DynDev d;
  std::memset(&d, 0, sizeof(d));
  d.dt = sp.dt; d.kp = sp.kp; d.kd = sp.kd; d.u_max = sp.u_max; d.goal_tol = sp.goal_tol;
  for (int i = 0; i < 2 * sp.n_dof; ++i) {
    d.lower[i] = sp.lower[i];
    d.upper[i] = sp.upper[i];
  }
  // the schedule is evaluated over the whole step budget (edges with their own travel fraction, EdgeIO::frac, cut it
  // on the device with the same comparison); n_steps = its length for this call's fraction
  d.full_time = sp.steps_per_edge * sp.dt;
  const double T_goal = fraction * d.full_time;
  double current_time = 0.0;
  int k = 0, n = 0;
  while (k < kMaxSteps) {
    if (current_time < T_goal) n = k + 1;
    double t = current_time;
    const double end_time = current_time + sp.dt;
    int it = 0;
    while (t < end_time) {  // runge_kutta4_integrator_sys.hpp:73-96
      t += sp.dt * 0.5;
      t += sp.dt * 0.5;
      ++it;
    }
    d.inner[k] = int8_t(it);
    current_time += sp.dt;
    ++k;
  }
  k = n;
  d.n_steps = k;
  *out = d;
  return RKH_OK;
}

// diagnostic: GJK on n world-anchored pairs (rkh_diag_gjk_distance)
__global__ void gjk_pairs_kernel(const rkh_shape* __restrict__ a, const rkh_shape* __restrict__ b, uint32_t n,
                                 const double* __restrict__ pool, double* __restrict__ out) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  auto conv = [](const rkh_shape& s) {
    ShapeG g;
    g.kind = s.kind;
    g.pos = mk3(s.pose.pos[0], s.pose.pos[1], s.pose.pos[2]);
    g.q = d4{s.pose.quat[0], s.pose.quat[1], s.pose.quat[2], s.pose.quat[3]};
    g.d0 = s.dims[0]; g.d1 = s.dims[1]; g.d2 = s.dims[2];
    return g;
  };
  out[i] = gjk_distance(to_gjk(conv(a[i]), pool), to_gjk(conv(b[i]), pool));
}

}  // namespace rkh

static rkh_status upload_scene(rkh_ctx* ctx, rkh_scene* sc, const std::vector<PairDev>& pairs, rkh_scene** out) {
  SceneDev& S = sc->host;
  sc->n_pairs = int(pairs.size());
  if (sc->n_pairs_verdict < 0 || sc->n_pairs_verdict > sc->n_pairs) sc->n_pairs_verdict = sc->n_pairs;
  RKH_HIP(hipSetDevice(ctx->device));
  if (S.has_meshes && g_n_mesh_vertices > 0) {
    RKH_HIP(hipMalloc(&sc->d_mesh_verts, size_t(g_n_mesh_vertices) * 3 * sizeof(double)));
    RKH_HIP(hipMemcpy(sc->d_mesh_verts, g_mesh_vertices, size_t(g_n_mesh_vertices) * 3 * sizeof(double), hipMemcpyHostToDevice));
    S.mesh_verts = sc->d_mesh_verts;
  }
  RKH_HIP(hipMalloc(&sc->d_scene, sizeof(SceneDev)));
  RKH_HIP(hipMemcpy(sc->d_scene, &S, sizeof(SceneDev), hipMemcpyHostToDevice));
  RKH_HIP(hipMalloc(&sc->d_pairs, std::max<size_t>(1, pairs.size()) * sizeof(PairDev)));
  if (!pairs.empty()) RKH_HIP(hipMemcpy(sc->d_pairs, pairs.data(), pairs.size() * sizeof(PairDev), hipMemcpyHostToDevice));
  RKH_HIP(hipMalloc(&sc->d_err, sizeof(int)));
  RKH_HIP(hipMemset(sc->d_err, 0, sizeof(int)));
  if (S.planar) register_planar_scene(sc->d_scene);
  if (S.has_meshes) register_mesh_scene(sc->d_scene);
  *out = sc;
  return RKH_OK;
}

// Planar chain (manip_3R_arm.cpp:75-150 pattern): {revolute_joint_2D, rigid_link_2D} per joint, 2D shapes only.
// Position level: the scene serves the quasi-static entry points; the dynamics entry points refuse it.
static rkh_status create_planar_scene(rkh_ctx* ctx, const rkh_kte_op* prog, int n_ops, const rkh_chain_base* base,
                                      const rkh_shape* shapes, int n_shapes, rkh_scene** out) {
  // position level: {revolute_joint_2D, rigid_link_2D} per joint; with dynamics: {driving_actuator_gen, inertia_gen,
  // revolute_joint_2D, rigid_link_2D, inertia_2D on the link's end frame} per joint
  const bool dynamics = prog[0].kind == RKH_KTE_DRIVING_ACTUATOR_GEN;
  const int per = dynamics ? 5 : 2;
  const int n = n_ops / per;
  if (n_ops % per != 0 || n < 1 || n > kMaxDof) {
    set_error("rkh_scene_create: a planar chain is a sequence of {revolute_joint_2D, rigid_link_2D} pairs, or of "
              "{driving_actuator_gen, inertia_gen, revolute_joint_2D, rigid_link_2D, inertia_2D} groups");
    return RKH_ERR_UNSUPPORTED;
  }
  rkh_scene* sc = new rkh_scene();
  sc->ctx = ctx;
  SceneDev& S = sc->host;
  std::memset(&S, 0, sizeof(S));
  S.n_dof = n;
  S.planar = 1;
  S.planar_dynamics = dynamics ? 1 : 0;
  for (int i = 0; i < 2; ++i) {
    S.base_pos[i] = base->pose.pos[i];
    S.base_quat[i] = base->pose.quat[i];
    S.base_acc[i] = base->acceleration[i];
  }
  std::vector<int> joint_end_frame(n);
  int prev_end = 0;
  for (int j = 0; j < n; ++j) {
    const rkh_kte_op& rev = prog[per * j + (dynamics ? 2 : 0)], &lnk = prog[per * j + (dynamics ? 3 : 1)];
    if (rev.kind != RKH_KTE_REVOLUTE_JOINT_2D || lnk.kind != RKH_KTE_RIGID_LINK_2D || rev.coord != j ||
        rev.base_frame != prev_end || lnk.base_frame != rev.end_frame) {
      delete sc;
      set_error("rkh_scene_create: planar op group " + std::to_string(j) + " does not continue the serial chain");
      return RKH_ERR_UNSUPPORTED;
    }
    JointDev& J = S.joints[j];
    if (dynamics) {
      const rkh_kte_op& act = prog[per * j], &gen = prog[per * j + 1], &in2 = prog[per * j + 4];
      if (act.kind != RKH_KTE_DRIVING_ACTUATOR_GEN || act.coord != j || act.joint_op != per * j + 2 ||
          gen.kind != RKH_KTE_INERTIA_GEN || gen.coord != j || gen.upstream != (1u << j) ||
          in2.kind != RKH_KTE_INERTIA_2D || in2.end_frame != lnk.end_frame || in2.upstream != ((1u << (j + 1)) - 1u)) {
        delete sc;
        set_error("rkh_scene_create: planar op group " + std::to_string(j) +
                  " is not {actuator, inertia_gen, revolute_joint_2D, rigid_link_2D, inertia_2D} of one serial joint");
        return RKH_ERR_UNSUPPORTED;
      }
      J.joint_inertia = gen.mass;
      J.mass = in2.mass;
      J.inertia[0] = in2.inertia[0];
    }
    prev_end = lnk.end_frame;
    joint_end_frame[j] = rev.end_frame;
    for (int i = 0; i < 2; ++i) {
      J.off_pos[i] = lnk.offset.pos[i];
      J.off_quat[i] = lnk.offset.quat[i];
    }
    S.mount_quat[j][0] = 1.0;
  }
  for (int i = 0; i < n_shapes; ++i) {
    const rkh_shape& s = shapes[i];
    if (s.kind < RKH_SHAPE_CIRCLE || s.kind > RKH_SHAPE_CRECT) {
      delete sc;
      set_error("rkh_scene_create: a planar chain takes 2D shapes only (circle, rectangle, capped_rectangle)");
      return RKH_ERR_UNSUPPORTED;
    }
    ShapeDev d;
    std::memset(&d, 0, sizeof(d));
    d.kind = s.kind;
    for (int k = 0; k < 2; ++k) {
      d.pos[k] = s.pose.pos[k];
      d.quat[k] = s.pose.quat[k];
      d.dims[k] = s.dims[k];
    }
    // circle.cpp:31-33 ; rectangle.cpp:31-33 and capped_rectangle.cpp:31-33: norm_2(mDimensions) * 0.5
    d.brad = (s.kind == RKH_SHAPE_CIRCLE) ? s.dims[0] : std::sqrt((0.0 + s.dims[0] * s.dims[0]) + s.dims[1] * s.dims[1]) * 0.5;
    if (s.anchor >= 0) {
      int link = -1;
      for (int j = 0; j < n; ++j)
        if (joint_end_frame[j] == s.anchor) link = j;
      if (link < 0 || S.n_robot >= 2 * kMaxDof) {
        delete sc;
        set_error("rkh_scene_create: robot shapes must be anchored on a revolute joint's end frame");
        return RKH_ERR_UNSUPPORTED;
      }
      d.link = link;
      S.robot[S.n_robot++] = d;
    } else {
      if (S.n_env >= kMaxEnvShapes) {
        delete sc;
        set_error("rkh_scene_create: too many environment shapes");
        return RKH_ERR_CAPACITY;
      }
      d.link = -1;
      S.env[S.n_env++] = d;
    }
  }
  for (int r = 0; r < S.n_robot; ++r) S.robot_n_reach[r] = S.n_env;
  // proxy_query_pair_2D::createProxFinderList (proxy_query_model.cpp:75-161), kept in its i-major / j-minor order:
  // findMinimumDistance (:163-189) culls against the running minimum with a radius the capped rectangle's caps
  // reach beyond, so the result depends on the order and the device replays the sequence
  std::vector<PairDev> pairs;
  for (int i = 0; i < S.n_robot; ++i)
    for (int j = 0; j < S.n_env; ++j) {
      const int ki = S.robot[i].kind, kj = S.env[j].kind;
      PairDev p;
      std::memset(&p, 0, sizeof(p));
      p.robot = uint16_t(i);
      p.env = uint16_t(j);
      if (ki == RKH_SHAPE_CIRCLE || kj == RKH_SHAPE_CIRCLE) {
        p.s1_is_robot = (ki == RKH_SHAPE_CIRCLE) ? 1 : 0;  // the circle is shape1
        const int ko = p.s1_is_robot ? kj : ki;
        p.routine = (ko == RKH_SHAPE_CIRCLE) ? 11 : (ko == RKH_SHAPE_CRECT ? 12 : 13);
      } else if (ki == RKH_SHAPE_CRECT || kj == RKH_SHAPE_CRECT) {
        p.s1_is_robot = (ki == RKH_SHAPE_CRECT) ? 1 : 0;  // the capped rectangle is shape1
        const int ko = p.s1_is_robot ? kj : ki;
        p.routine = (ko == RKH_SHAPE_CRECT) ? 14 : 15;
      } else {
        p.s1_is_robot = 1;  // rectangle-rectangle: model1's shape is shape1
        p.routine = 16;
      }
      pairs.push_back(p);
    }
  return upload_scene(ctx, sc, pairs, out);
}

extern "C" {

rkh_status rkh_scene_create(rkh_ctx* ctx, const rkh_kte_op* prog, int n_ops, const rkh_chain_base* base,
                            const rkh_shape* shapes, int n_shapes, rkh_scene** out) {
  return rkh_scene_create_with_meshes(ctx, prog, n_ops, base, shapes, n_shapes, nullptr, 0, out);
}

rkh_status rkh_scene_create_with_meshes(rkh_ctx* ctx, const rkh_kte_op* prog, int n_ops, const rkh_chain_base* base,
                                        const rkh_shape* shapes, int n_shapes, const double* mesh_vertices,
                                        uint32_t n_mesh_vertices, rkh_scene** out) {
  if (!ctx || !prog || !base || !out || n_ops < 1 || (n_shapes > 0 && !shapes)) return RKH_ERR_BAD_ARG;
  if (n_mesh_vertices > 0 && !mesh_vertices) return RKH_ERR_BAD_ARG;
  g_mesh_vertices = mesh_vertices;  // for bounding_radius() / validation while the scene is built (one host thread per ctx)
  g_n_mesh_vertices = n_mesh_vertices;
  if (prog[0].kind == RKH_KTE_REVOLUTE_JOINT_2D ||
      (n_ops >= 3 && prog[0].kind == RKH_KTE_DRIVING_ACTUATOR_GEN && prog[2].kind == RKH_KTE_REVOLUTE_JOINT_2D))
    return create_planar_scene(ctx, prog, n_ops, base, shapes, n_shapes, out);
  const bool has_beam = n_ops > 1 && prog[n_ops - 1].kind == RKH_KTE_FLEXIBLE_BEAM_3D;
  if (has_beam) --n_ops;  // the beam is validated below, after the chain
  // parse: [optional mount link from frame 0] {actuator, inertia_gen, revolute, link, inertia_3D} ...
  struct Group { int first_op; int mount_op; };
  std::vector<Group> groups;
  {
    int k = 0;
    while (k < n_ops) {
      Group g{0, -1};
      if (prog[k].kind == RKH_KTE_RIGID_LINK_3D && prog[k].base_frame == 0) {
        g.mount_op = k;
        ++k;
      }
      g.first_op = k;
      if (k + 5 > n_ops) {
        k = -1;
        break;
      }
      groups.push_back(g);
      k += 5;
    }
    if (k < 0 || groups.empty() || int(groups.size()) > kMaxDof) {
      set_error("rkh_scene_create: KTE program is not a chain of [mount link] {actuator, inertia_gen, revolute, link, "
                "inertia_3D} groups (optionally followed by one flexible_beam_3D), or has too many joints");
      return RKH_ERR_UNSUPPORTED;
    }
  }
  const int n = int(groups.size());
  rkh_scene* sc = new rkh_scene();
  sc->ctx = ctx;
  SceneDev& S = sc->host;
  std::memset(&S, 0, sizeof(S));
  S.n_dof = n;
  for (int i = 0; i < 3; ++i) {
    S.base_pos[i] = base->pose.pos[i];
    S.base_acc[i] = base->acceleration[i];
  }
  for (int i = 0; i < 4; ++i) S.base_quat[i] = base->pose.quat[i];
  std::vector<int> joint_end_frame(n), link_end_frame(n);
  int prev_end = 0;  // frame 0 = chain base
  bool first = true;
  int branch_first = 0;  // first joint of the current branch
  for (int j = 0; j < n; ++j) {
    const int k0 = groups[j].first_op;
    const rkh_kte_op& act = prog[k0], &gen = prog[k0 + 1], &rev = prog[k0 + 2], &lnk = prog[k0 + 3], &ine = prog[k0 + 4];
    bool starts_branch = false;
    int expect_base = prev_end;
    if (groups[j].mount_op >= 0) {  // a rigid link from the chain base carries this joint
      const rkh_kte_op& mt = prog[groups[j].mount_op];
      starts_branch = true;
      expect_base = mt.end_frame;
      for (int i = 0; i < 3; ++i) S.mount_pos[j][i] = mt.offset.pos[i];
      for (int i = 0; i < 4; ++i) S.mount_quat[j][i] = mt.offset.quat[i];
    } else if (!first && rev.base_frame == 0) {  // a second chain sitting directly on the base
      starts_branch = true;
      expect_base = 0;
      S.mount_quat[j][0] = 1.0;
    } else {
      S.mount_quat[j][0] = 1.0;
    }
    if (starts_branch) {
      S.branch_start[j] = 1;
      ++S.n_branches;
      branch_first = j;
    }
    S.branch_first[j] = branch_first;
    const uint32_t branch_mask = ((j + 1 >= 32 ? 0xFFFFFFFFu : ((1u << (j + 1)) - 1u))) & ~((1u << branch_first) - 1u);
    const bool ok = act.kind == RKH_KTE_DRIVING_ACTUATOR_GEN && gen.kind == RKH_KTE_INERTIA_GEN &&
                    rev.kind == RKH_KTE_REVOLUTE_JOINT_3D && lnk.kind == RKH_KTE_RIGID_LINK_3D &&
                    ine.kind == RKH_KTE_INERTIA_3D && act.coord == j && gen.coord == j && rev.coord == j &&
                    act.joint_op == k0 + 2 && gen.upstream == (1u << j) &&
                    (first && groups[j].mount_op < 0 ? rev.base_frame == 0 : rev.base_frame == expect_base) &&
                    lnk.base_frame == rev.end_frame && ine.end_frame == lnk.end_frame && ine.upstream == branch_mask;
    if (!ok) {
      delete sc;
      set_error("rkh_scene_create: op group " + std::to_string(j) + " does not match the chain pattern");
      return RKH_ERR_UNSUPPORTED;
    }
    first = false;
    prev_end = lnk.end_frame;
    joint_end_frame[j] = rev.end_frame;
    link_end_frame[j] = lnk.end_frame;
    JointDev& J = S.joints[j];
    for (int i = 0; i < 3; ++i) J.axis[i] = rev.axis[i];
    {  // axis_angle ctor normalisation (rotations_3D.hpp:1961-1974)
      double acc = 0.0;
      for (int i = 0; i < 3; ++i) acc += rev.axis[i] * rev.axis[i];
      const double tmp = std::sqrt(acc);
      if (tmp > 0.0000001) {
        for (int i = 0; i < 3; ++i) J.axis_n[i] = rev.axis[i] / tmp;
      } else {
        J.axis_n[0] = 1.0; J.axis_n[1] = 0.0; J.axis_n[2] = 0.0;
      }
    }
    J.joint_inertia = gen.mass;
    for (int i = 0; i < 3; ++i) J.off_pos[i] = lnk.offset.pos[i];
    for (int i = 0; i < 4; ++i) J.off_quat[i] = lnk.offset.quat[i];
    host_rotmat(J.off_quat, J.off_R);
    J.mass = ine.mass;
    for (int i = 0; i < 6; ++i) J.inertia[i] = ine.inertia[i];
  }
  if (has_beam) {
    const rkh_kte_op& bm = prog[n_ops];
    int j1 = -1, j2 = -1;
    for (int j = 0; j < n; ++j) {
      if (link_end_frame[j] == bm.base_frame) j1 = j;
      if (bm.end_frame >= 0 && link_end_frame[j] == bm.end_frame) j2 = j;
    }
    if (j1 < 0 || (bm.end_frame >= 0 && j2 < 0) || j1 == j2) {
      delete sc;
      set_error("rkh_scene_create: the flexible beam's anchors must be link end frames of the chain (anchor 2 may be a "
                "world anchor, end_frame = -1)");
      return RKH_ERR_UNSUPPORTED;
    }
    S.beam_on = 1;
    S.beam_j1 = j1;
    S.beam_j2 = j2;
    S.beam_rest = bm.axis[0];
    S.beam_k = bm.axis[1];
    S.beam_kt = bm.axis[2];
    for (int i = 0; i < 3; ++i) S.beam_pos[i] = bm.offset.pos[i];
    for (int i = 0; i < 4; ++i) S.beam_quat[i] = bm.offset.quat[i];
  }
  // shapes: robot model (anchored) / environment model (world).  The environment shapes are stored nearest-to-the-base
  // first (see SceneDev::robot_n_reach; the verdict "some pair is closer than 0" does not depend on the pair order).
  std::vector<int> robot_src, env_src;
  std::vector<int> order(n_shapes);
  {
    std::vector<double> key(n_shapes, -INFINITY);  // robot shapes first, in their given order
    for (int i = 0; i < n_shapes; ++i) {
      order[i] = i;
      const rkh_shape& s = shapes[i];
      if (s.anchor >= 0) continue;
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) d2 += (s.pose.pos[k] - S.base_pos[k]) * (s.pose.pos[k] - S.base_pos[k]);
      key[i] = std::sqrt(d2) - bounding_radius(s);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
  }
  std::vector<double> env_key;
  for (int oi = 0; oi < n_shapes; ++oi) {
    const int i = order[oi];
    const rkh_shape& s = shapes[i];
    const bool kind_ok = (s.kind >= RKH_SHAPE_SPHERE && s.kind <= RKH_SHAPE_CCYLINDER) || s.kind == RKH_SHAPE_PLANE ||
                         s.kind == RKH_SHAPE_CYLINDER || (s.kind == RKH_SHAPE_MESH && mesh_range_ok(s));
    if (!kind_ok) {
      delete sc;
      set_error("rkh_scene_create: unsupported shape kind");
      return RKH_ERR_UNSUPPORTED;
    }
    if (s.kind == RKH_SHAPE_PLANE || s.kind == RKH_SHAPE_CYLINDER) S.has_ext_shapes = 1;
    if (s.kind == RKH_SHAPE_MESH) S.has_meshes = 1;
    ShapeDev d;
    std::memset(&d, 0, sizeof(d));
    d.kind = s.kind;
    for (int k = 0; k < 3; ++k) { d.pos[k] = s.pose.pos[k]; d.dims[k] = s.dims[k]; }
    for (int k = 0; k < 4; ++k) d.quat[k] = s.pose.quat[k];
    d.brad = bounding_radius(s);
    if (s.anchor >= 0) {
      int link = -1;
      for (int j = 0; j < n; ++j)
        if (joint_end_frame[j] == s.anchor) link = j;
      if (link < 0 || S.n_robot >= 2 * kMaxDof) {
        delete sc;
        set_error("rkh_scene_create: robot shapes must be anchored on a revolute joint's end frame");
        return RKH_ERR_UNSUPPORTED;
      }
      d.link = link;
      S.robot[S.n_robot++] = d;
      robot_src.push_back(i);
    } else {
      if (S.n_env >= kMaxEnvShapes) {
        delete sc;
        set_error("rkh_scene_create: too many environment shapes");
        return RKH_ERR_CAPACITY;
      }
      d.link = -1;
      for (int k = 0; k < 3; ++k) S.env_cull[S.n_env][k] = d.pos[k];
      S.env_cull[S.n_env][3] = d.brad;
      if (d.kind <= RKH_SHAPE_CCYLINDER)
        S.env_kind_mask[d.kind == RKH_SHAPE_SPHERE ? 0 : (d.kind == RKH_SHAPE_BOX ? 1 : 2)][S.n_env / 64] |= 1ull << (S.n_env % 64);
      for (int k = RKH_SHAPE_SPHERE; k <= RKH_SHAPE_CYLINDER; ++k) {  // (meshes do not reach the kernel that reads this)
        bool first;
        if (host_pair_routine(k, d.kind, &first) != 0) S.env_finder_mask[k][S.n_env / 64] |= 1ull << (S.n_env % 64);
      }
      S.env[S.n_env++] = d;
      env_src.push_back(i);
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) d2 += (d.pos[k] - S.base_pos[k]) * (d.pos[k] - S.base_pos[k]);
      env_key.push_back(std::sqrt(d2) - d.brad);
    }
  }
  for (int r = 0; r < S.n_robot; ++r) {
    S.robot_n_reach[r] = S.n_env;
    if (S.n_branches != 0 || S.planar) continue;  // serial 3D chains only: every joint hangs off the previous link
    double reach = 0.0;
    for (int i = 0; i < S.robot[r].link; ++i) {
      const double* o = S.joints[i].off_pos;
      reach += std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    }
    const double* lp = S.robot[r].pos;
    reach += std::sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]) + S.robot[r].brad;
    reach *= 1.0 + 1e-9;  // the sums above are rounded
    int cnt = 0;
    while (cnt < S.n_env && env_key[cnt] <= reach + 1e-6) ++cnt;
    S.robot_n_reach[r] = cnt;
  }
  // proxy_query_pair_3D::createProxFinderList (proxy_query_model.cpp:215-374), then grouped by routine so
  // that the lanes of a wave run the same closed form (the verdict does not depend on the pair order)
  std::vector<PairDev> pairs;
  // reach of every robot shape's CENTRE from the chain base (over all configurations), for the plane rule below
  std::vector<double> centre_reach(S.n_robot, 0.0);
  for (int r = 0; r < S.n_robot; ++r) {
    double reach = 0.0;
    for (int i = 0; i < n; ++i) {  // serial chain: the links below the shape's joint; branching chains: all of them
      if (S.n_branches == 0 && i >= S.robot[r].link) break;
      const double* o = S.joints[i].off_pos;
      reach += std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
      const double* mp = S.mount_pos[i];
      reach += std::sqrt(mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
    }
    const double* lp = S.robot[r].pos;
    centre_reach[r] = reach + std::sqrt(lp[0] * lp[0] + lp[1] * lp[1] + lp[2] * lp[2]);
  }
  for (int i = 0; i < S.n_robot; ++i)
    for (int j = 0; j < S.n_env; ++j) {
      const int ki = S.robot[i].kind, kj = S.env[j].kind;
      PairDev p;
      std::memset(&p, 0, sizeof(p));
      p.robot = uint16_t(i);
      p.env = uint16_t(j);
      bool robot_first = true;
      p.routine = uint8_t(host_pair_routine(ki, kj, &robot_first));
      if (p.routine == 0) continue;  // no finder in the reference
      p.s1_is_robot = robot_first ? 1 : 0;
      if (ki == RKH_SHAPE_PLANE || kj == RKH_SHAPE_PLANE) {
        // findMinimumDistance skips a finder whose bounding spheres are further apart than the running minimum
        // (proxy_query_model.cpp:384-389).  For bounded shapes that cannot change the minimum; a plane's bounding radius
        // (plane.cpp:31) is finite although the prox_plane_* routines treat it as infinite, so for a plane pair the skip
        // could.  The kernels evaluate every plane pair; that is the reference's result exactly when the pair can never
        // be skipped, i.e. the shapes' bounding spheres overlap in every configuration -- required here.
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) d2 += (S.env[j].pos[k] - S.base_pos[k]) * (S.env[j].pos[k] - S.base_pos[k]);
        if (std::sqrt(d2) + centre_reach[i] - S.env[j].brad - S.robot[i].brad > 0.0) {
          delete sc;
          set_error("rkh_scene_create: a plane must be large enough that its bounding sphere (plane.cpp:31) always overlaps "
                    "the robot shapes' (otherwise the reference's result depends on the finder order)");
          return RKH_ERR_UNSUPPORTED;
        }
      }
      pairs.push_back(p);
    }
  // Pairs a robot shape can never reach (environment shape beyond its static reach, SceneDev::robot_n_reach: the bounding
  // spheres have a positive gap in EVERY configuration) go to the end of the list.  A verdict query ("is some pair closer
  // than 0") skips every pair whose spheres do not overlap anyway, so the verdict kernels -- steer and edge walks -- only
  // scan the first n_pairs_verdict entries (C2: 130 of 300); a distance query (rkh_min_distance) scans them all.
  auto unreachable = [&](const PairDev& q) { return int(q.env) >= S.robot_n_reach[q.robot] ? 1 : 0; };
  std::stable_sort(pairs.begin(), pairs.end(), [&](const PairDev& a, const PairDev& b) {
    const int ua = unreachable(a), ub = unreachable(b);
    return ua != ub ? ua < ub : a.routine < b.routine;
  });
  int n_verdict = 0;
  for (const PairDev& q : pairs) n_verdict += unreachable(q) ? 0 : 1;
  sc->n_pairs_verdict = n_verdict;
  return upload_scene(ctx, sc, pairs, out);
}

rkh_status rkh_diag_gjk_distance(rkh_ctx* ctx, const rkh_shape* a, const rkh_shape* b, uint32_t n,
                                 const double* mesh_vertices, uint32_t n_mesh_vertices, double* dist) {
  if (!ctx || !a || !b || !dist) return RKH_ERR_BAD_ARG;
  if (n == 0) return RKH_OK;
  RKH_HIP(hipSetDevice(ctx->device));
  rkh_shape *da = nullptr, *db = nullptr;
  double *dv = nullptr, *dd = nullptr;
  RKH_HIP(hipMalloc(&da, n * sizeof(rkh_shape)));
  RKH_HIP(hipMalloc(&db, n * sizeof(rkh_shape)));
  RKH_HIP(hipMalloc(&dd, n * sizeof(double)));
  RKH_HIP(hipMalloc(&dv, std::max<size_t>(1, size_t(n_mesh_vertices) * 3) * sizeof(double)));
  RKH_HIP(hipMemcpy(da, a, n * sizeof(rkh_shape), hipMemcpyHostToDevice));
  RKH_HIP(hipMemcpy(db, b, n * sizeof(rkh_shape), hipMemcpyHostToDevice));
  if (n_mesh_vertices) RKH_HIP(hipMemcpy(dv, mesh_vertices, size_t(n_mesh_vertices) * 3 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(rkh::gjk_pairs_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, da, db, n, dv, dd);
  RKH_HIP(hipGetLastError());
  RKH_HIP(hipMemcpyAsync(dist, dd, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RKH_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dv); (void)hipFree(dd);
  return RKH_OK;
}

rkh_status rkh_scene_destroy(rkh_scene* scene) {
  if (!scene) return RKH_OK;
  forget_planar_scene(scene->d_scene);  // (a later allocation may get the same address)
  forget_mesh_scene(scene->d_scene);
  hipFree(scene->d_scene);
  hipFree(scene->d_pairs);
  if (scene->d_mesh_verts) hipFree(scene->d_mesh_verts);
  hipFree(scene->d_err);
  delete scene;
  return RKH_OK;
}
int rkh_scene_num_dof(const rkh_scene* scene) { return scene ? scene->host.n_dof : 0; }
int rkh_scene_num_pairs(const rkh_scene* scene) { return scene ? scene->n_pairs : 0; }

}  // extern "C"

namespace {
struct DevBuf {  // scoped device scratch
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};
// dynamics entry points: branching chains are fine (wave-per-edge kernels); planar chains are position level only
rkh_status reject_branches(const rkh_scene* scene) {
  if (scene->host.planar && !scene->host.planar_dynamics) {
    set_error("this planar (2D) chain was given at position level (no actuators / inertias): quasi-static spaces only");
    return RKH_ERR_UNSUPPORTED;
  }
  return RKH_OK;
}
rkh_status check_err_flag(rkh_scene* scene) {
  int flag = 0;
  RKH_HIP(hipMemcpy(&flag, scene->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flag != 0) {
    RKH_HIP(hipMemset(scene->d_err, 0, sizeof(int)));
    set_error("mass matrix is singular (Cholesky pivot < 1e-8)");
    return rkh_status(flag);
  }
  return RKH_OK;
}
}  // namespace

extern "C" {

rkh_status rkh_state_derivative(rkh_scene* scene, const double* x, const double* u, uint32_t B, double* pd, double* M,
                                double* f) {
  if (scene && reject_branches(scene) != RKH_OK) return RKH_ERR_UNSUPPORTED;
  if (!scene || !x || !u || !pd) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  const int n = scene->host.n_dof;
  hipStream_t s = scene->ctx->stream;
  DevBuf dx, du, dpd, dM, df;
  RKH_HIP(hipMalloc(&dx.p, size_t(B) * 2 * n * 8));
  RKH_HIP(hipMalloc(&du.p, size_t(B) * n * 8));
  RKH_HIP(hipMalloc(&dpd.p, size_t(B) * 2 * n * 8));
  RKH_HIP(hipMalloc(&dM.p, size_t(B) * n * n * 8));
  RKH_HIP(hipMalloc(&df.p, size_t(B) * n * 8));
  RKH_HIP(hipMemcpyAsync(dx.p, x, size_t(B) * 2 * n * 8, hipMemcpyHostToDevice, s));
  RKH_HIP(hipMemcpyAsync(du.p, u, size_t(B) * n * 8, hipMemcpyHostToDevice, s));
  rkh_status st = launch_state_derivative(s, n, scene->d_scene, dx.as<double>(), du.as<double>(), B, dpd.as<double>(),
                                          dM.as<double>(), df.as<double>(), scene->d_err);
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(pd, dpd.p, size_t(B) * 2 * n * 8, hipMemcpyDeviceToHost, s));
  if (M) RKH_HIP(hipMemcpyAsync(M, dM.p, size_t(B) * n * n * 8, hipMemcpyDeviceToHost, s));
  if (f) RKH_HIP(hipMemcpyAsync(f, df.p, size_t(B) * n * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return check_err_flag(scene);
}

rkh_status rkh_min_distance(rkh_scene* scene, const double* x, uint32_t B, double* dist) {
  if (!scene || !x || !dist) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  const int n = scene->host.n_dof;
  hipStream_t s = scene->ctx->stream;
  DevBuf dx, dd;
  RKH_HIP(hipMalloc(&dx.p, size_t(B) * 2 * n * 8));
  RKH_HIP(hipMalloc(&dd.p, size_t(B) * 8));
  RKH_HIP(hipMemcpyAsync(dx.p, x, size_t(B) * 2 * n * 8, hipMemcpyHostToDevice, s));
  rkh_status st = launch_min_distance(s, n, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs,
                                      dx.as<double>(), B, dd.as<double>());
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(dist, dd.p, size_t(B) * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

rkh_status rkh_propagate(rkh_scene* scene, const rkh_dyn_space* space, const double* a, const double* b, uint32_t B,
                         double fraction, double* x_out, uint32_t* steps_free, double* record) {
  if (scene && reject_branches(scene) != RKH_OK) return RKH_ERR_UNSUPPORTED;
  if (!scene || !space || !a || !b || !x_out || !steps_free) return RKH_ERR_BAD_ARG;
  if (space->n_dof != scene->host.n_dof) {
    set_error("rkh_propagate: rkh_dyn_space.n_dof does not match the scene");
    return RKH_ERR_BAD_ARG;
  }
  if (B == 0) return RKH_OK;
  DynDev dyn;
  rkh_status st = build_dyn_dev(*space, fraction, &dyn);
  if (st != RKH_OK) return st;
  const int n = scene->host.n_dof, D = 2 * n;
  const int rec_stride = space->steps_per_edge + 1;
  hipStream_t s = scene->ctx->stream;
  DevBuf da, db, dxo, dsf, drec;
  RKH_HIP(hipMalloc(&da.p, size_t(B) * D * 8));
  RKH_HIP(hipMalloc(&db.p, size_t(B) * D * 8));
  RKH_HIP(hipMalloc(&dxo.p, size_t(B) * D * 8));
  RKH_HIP(hipMalloc(&dsf.p, size_t(B) * 4));
  if (record) {
    RKH_HIP(hipMalloc(&drec.p, size_t(B) * rec_stride * D * 8));
    RKH_HIP(hipMemsetAsync(drec.p, 0, size_t(B) * rec_stride * D * 8, s));
  }
  RKH_HIP(hipMemcpyAsync(da.p, a, size_t(B) * D * 8, hipMemcpyHostToDevice, s));
  RKH_HIP(hipMemcpyAsync(db.p, b, size_t(B) * D * 8, hipMemcpyHostToDevice, s));
  EdgeIO io;
  io.src = da.as<double>();
  io.src_stride = D;
  io.tgt = db.as<double>();
  io.tgt_stride = D;
  io.B = B;
  io.x_out = dxo.as<double>();
  io.steps_free = dsf.as<uint32_t>();
  io.record = record ? drec.as<double>() : nullptr;
  io.record_stride = rec_stride;
  io.err_flag = scene->d_err;
  // RKH_LANES_PER_EDGE = 128 (two waves per edge) | 64 | 16 | 2 | 1 selects the kernel mapping (identical results); by
  // default a call of few edges -- the adaptors steer ONE edge per call -- takes the lowest-latency mapping
  int lanes = (B <= 512) ? 128 : 64;
  if (const char* ev = getenv("RKH_LANES_PER_EDGE"))
    lanes = (atoi(ev) == 1) ? 1 : (atoi(ev) == 2 ? 2 : (atoi(ev) == 16 ? 16 : (atoi(ev) == 128 ? 128 : 64)));
  if ((lanes == 1 || lanes == 2) && !(n <= 7 && scene_fits_lane_kernel(scene->host, lanes))) lanes = 64;  // not a scene for that mapping
  if (lanes == 16 && 2 * n > 16) lanes = 64;
  DevBuf dws;
  if (lanes == 1) RKH_HIP(hipMalloc(&dws.p, propagate_lanes_workspace_bytes(n, B, 0, 1)));
  if (lanes == 2) RKH_HIP(hipMalloc(&dws.p, propagate_pairs_workspace_bytes(n, B, 0, 1)));
  st = launch_propagate(s, n, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs_verdict, dyn, io, B, nullptr, 0,
                        lanes, nullptr, nullptr, 1, dws.as<double>());
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(x_out, dxo.p, size_t(B) * D * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipMemcpyAsync(steps_free, dsf.p, size_t(B) * 4, hipMemcpyDeviceToHost, s));
  if (record) RKH_HIP(hipMemcpyAsync(record, drec.p, size_t(B) * rec_stride * D * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return check_err_flag(scene);
}

rkh_status rkh_diag_feval_cycles(rkh_scene* scene, const double* x, const double* u, uint32_t B, int iters,
                                 uint64_t* cycles) {
  if (scene && reject_branches(scene) != RKH_OK) return RKH_ERR_UNSUPPORTED;
  if (!scene || !x || !u || !cycles || B == 0) return RKH_ERR_BAD_ARG;
  const int n = scene->host.n_dof;
  hipStream_t s = scene->ctx->stream;
  DevBuf dx, du, dout, dsink;
  RKH_HIP(hipMalloc(&dx.p, size_t(B) * 2 * n * 8));
  RKH_HIP(hipMalloc(&du.p, size_t(B) * n * 8));
  RKH_HIP(hipMalloc(&dout.p, size_t(B) * 8 * 8));
  RKH_HIP(hipMalloc(&dsink.p, size_t(B) * 8));
  RKH_HIP(hipMemcpyAsync(dx.p, x, size_t(B) * 2 * n * 8, hipMemcpyHostToDevice, s));
  RKH_HIP(hipMemcpyAsync(du.p, u, size_t(B) * n * 8, hipMemcpyHostToDevice, s));
  rkh_status st;
  const char* ev = getenv("RKH_LANES_PER_EDGE");
  if (ev && atoi(ev) == 1) {  // two-lanes-per-edge kernel: one record of 8 counters per wave of states
    RKH_HIP(hipMemsetAsync(dout.p, 0, size_t(B) * 8 * 8, s));
    st = launch_lane_cycles(s, n, scene->d_scene, dx.as<double>(), du.as<double>(), B, iters,
                            dout.as<unsigned long long>(), dsink.as<double>());
  } else if (ev && atoi(ev) == 2) {  // its second generation
    RKH_HIP(hipMemsetAsync(dout.p, 0, size_t(B) * 8 * 8, s));
    st = launch_pair_cycles(s, n, scene->d_scene, dx.as<double>(), du.as<double>(), B, iters,
                            dout.as<unsigned long long>(), dsink.as<double>());
  } else if (ev && atoi(ev) == 128) {  // two waves per edge: B / 2 states, rows 2 b / 2 b + 1 = the two waves' counters
    RKH_HIP(hipMemsetAsync(dout.p, 0, size_t(B) * 8 * 8, s));
    st = (B >= 2) ? launch_feval_cycles_duo(s, n, scene->host.n_env, scene->d_scene, dx.as<double>(), du.as<double>(), B, iters,
                                            dout.as<unsigned long long>(), dsink.as<double>())
                  : RKH_ERR_BAD_ARG;
  } else {
    st = launch_feval_cycles(s, n, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs, dx.as<double>(),
                             du.as<double>(), B, iters, dout.as<unsigned long long>(), dsink.as<double>());
  }
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(cycles, dout.p, size_t(B) * 8 * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

rkh_status rkh_diag_proximity_counts(rkh_scene* scene, const double* x, uint32_t B, uint64_t counts[8]) {
  if (scene && reject_branches(scene) != RKH_OK) return RKH_ERR_UNSUPPORTED;
  if (!scene || !x || !counts || B == 0) return RKH_ERR_BAD_ARG;
  const int n = scene->host.n_dof;
  if (!(n <= 7 && scene_fits_lane_kernel(scene->host, 2))) {
    set_error("proximity counts: a scene of the two-lanes steer mapping is needed");
    return RKH_ERR_UNSUPPORTED;
  }
  hipStream_t s = scene->ctx->stream;
  DevBuf dx, dout;
  RKH_HIP(hipMalloc(&dx.p, size_t(B) * 2 * n * 8));
  RKH_HIP(hipMalloc(&dout.p, 8 * 8));
  RKH_HIP(hipMemcpyAsync(dx.p, x, size_t(B) * 2 * n * 8, hipMemcpyHostToDevice, s));
  RKH_HIP(hipMemsetAsync(dout.p, 0, 8 * 8, s));
  const rkh_status st = launch_pair_counts(s, n, scene->d_scene, dx.as<double>(), B, dout.as<unsigned long long>());
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(counts, dout.p, 8 * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  // what the host knows: proxy pairs of the scene, and the ones inside the shapes' static reach
  counts[5] = uint64_t(scene->n_pairs);
  counts[6] = uint64_t(scene->n_pairs_verdict);
  counts[7] = 0;
  return RKH_OK;
}

rkh_status rkh_edge_check(rkh_scene* scene, const double* lower, const double* upper, double min_interval,
                          const double* a, const double* b, uint32_t B, double fraction, double* out,
                          uint32_t* n_checked) {
  if (!scene || !lower || !upper || !a || !b || !out || !n_checked || !(min_interval > 0.0)) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  const int n = scene->host.n_dof;
  QsDev qs;
  std::memset(&qs, 0, sizeof(qs));
  qs.min_interval = min_interval;
  qs.fraction = fraction;
  qs_set_speed(qs, nullptr, n);  // this entry point takes an ordinary joint space
  for (int i = 0; i < n; ++i) {
    qs.lower[i] = lower[i];
    qs.upper[i] = upper[i];
  }
  hipStream_t s = scene->ctx->stream;
  DevBuf da, db, dxo, dnc;
  RKH_HIP(hipMalloc(&da.p, size_t(B) * n * 8));
  RKH_HIP(hipMalloc(&db.p, size_t(B) * n * 8));
  RKH_HIP(hipMalloc(&dxo.p, size_t(B) * n * 8));
  RKH_HIP(hipMalloc(&dnc.p, size_t(B) * 4));
  RKH_HIP(hipMemcpyAsync(da.p, a, size_t(B) * n * 8, hipMemcpyHostToDevice, s));
  RKH_HIP(hipMemcpyAsync(db.p, b, size_t(B) * n * 8, hipMemcpyHostToDevice, s));
  EdgeIO io;
  io.src = da.as<double>();
  io.src_stride = n;
  io.tgt = db.as<double>();
  io.tgt_stride = n;
  io.B = B;
  io.x_out = dxo.as<double>();
  io.steps_free = dnc.as<uint32_t>();
  io.err_flag = scene->d_err;
  rkh_status st = launch_edge_check(s, n, scene->host.n_env, scene->d_scene, scene->d_pairs, scene->n_pairs_verdict, qs, io, B);
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(out, dxo.p, size_t(B) * n * 8, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipMemcpyAsync(n_checked, dnc.p, size_t(B) * 4, hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

}  // extern "C"
