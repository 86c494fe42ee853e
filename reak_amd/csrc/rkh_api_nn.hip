// rkh_api_nn.hip -- C-ABI: context + nearest-neighbour store (include/rkh.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>

#include "rkh_internal.h"

namespace rkh {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int nn_padded_dims(int D);
const char* nn_last_kernel_name();
}  // namespace rkh

using namespace rkh;

extern "C" {

#define RKH_STR2(x) #x
#define RKH_STR(x) RKH_STR2(x)
const char* rkh_last_error(void) { return rkh::g_err.c_str(); }
const char* rkh_version(void) { return "reak_amd/librkh 0.3 (gfx950), ABI " RKH_STR(RKH_ABI_VERSION); }
uint32_t rkh_abi_version(void) { return RKH_ABI_VERSION; }
/* The caller's view of the public PODs against the library's: a caller built against another header is refused here
 * instead of having a stats array overrun or garbage read as speed limits. */
rkh_status rkh_abi_check(uint32_t abi_version, size_t sizeof_dyn_space, size_t sizeof_qs_space, size_t sizeof_rrt_params,
                         size_t sizeof_prm_params, size_t sizeof_planner_stats, size_t sizeof_rrtstar_stats,
                         size_t sizeof_prm_stats, size_t sizeof_birrt_stats, size_t sizeof_shape, size_t sizeof_kte_op) {
  const bool ok = abi_version == RKH_ABI_VERSION && sizeof_dyn_space == sizeof(rkh_dyn_space) &&
                  sizeof_qs_space == sizeof(rkh_qs_space) && sizeof_rrt_params == sizeof(rkh_rrt_params) &&
                  sizeof_prm_params == sizeof(rkh_prm_params) && sizeof_planner_stats == sizeof(rkh_planner_stats) &&
                  sizeof_rrtstar_stats == sizeof(rkh_rrtstar_stats) && sizeof_prm_stats == sizeof(rkh_prm_stats) &&
                  sizeof_birrt_stats == sizeof(rkh_birrt_stats) && sizeof_shape == sizeof(rkh_shape) &&
                  sizeof_kte_op == sizeof(rkh_kte_op);
  if (!ok) {
    rkh::set_error("rkh_abi_check: the caller was built against another version of rkh.h / rkh_types.h (ABI " +
                   std::to_string(abi_version) + " vs " + std::to_string(RKH_ABI_VERSION) + ", or a struct size differs)");
    return RKH_ERR_BAD_ARG;
  }
  return RKH_OK;
}

rkh_status rkh_ctx_create(int device, rkh_ctx** out) {
  if (!out) return RKH_ERR_BAD_ARG;
  int count = 0;
  RKH_HIP(hipGetDeviceCount(&count));
  if (device < 0 || device >= count) {
    set_error("rkh_ctx_create: no such HIP device");
    return RKH_ERR_DEVICE;
  }
  RKH_HIP(hipSetDevice(device));
  rkh_ctx* c = new rkh_ctx();
  c->device = device;
  RKH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return RKH_OK;
}
rkh_status rkh_ctx_destroy(rkh_ctx* ctx) {
  if (!ctx) return RKH_OK;
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return RKH_OK;
}
rkh_status rkh_ctx_synchronize(rkh_ctx* ctx) {
  if (!ctx) return RKH_ERR_BAD_ARG;
  RKH_HIP(hipStreamSynchronize(ctx->stream));
  return RKH_OK;
}
void* rkh_ctx_stream(rkh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

rkh_status rkh_nn_create(rkh_ctx* ctx, int dims, uint64_t capacity, rkh_nn** out) {
  if (!ctx || !out || dims < 1 || capacity < 1) return RKH_ERR_BAD_ARG;
  const int DP = nn_padded_dims(dims);
  if (DP < 0) {
    set_error("rkh_nn_create: dims > 32 unsupported");
    return RKH_ERR_BAD_ARG;
  }
  rkh_nn* nn = new rkh_nn();
  nn->ctx = ctx;
  nn->st.D = dims;
  // round the capacity up to whole 256-row tiles so a sweep never reads past the allocation
  nn->st.capacity = (capacity + 255) / 256 * 256;
  RKH_HIP(hipSetDevice(ctx->device));
  RKH_HIP(hipMalloc(&nn->st.d_pos, nn->st.capacity * DP * sizeof(double)));
  *out = nn;
  return RKH_OK;
}
rkh_status rkh_nn_destroy(rkh_nn* nn) {
  if (!nn) return RKH_OK;
  hipFree(nn->st.d_pos);
  hipFree(nn->d_q);
  hipFree(nn->d_idx);
  hipFree(nn->d_dist);
  hipFree(nn->d_count);
  hipFree(nn->d_part_dist);
  hipFree(nn->d_part_idx);
  hipFree(nn->d_seed);
  hipFree(nn->d_knn_ws);
  delete nn;
  return RKH_OK;
}
rkh_status rkh_nn_clear(rkh_nn* nn) {
  if (!nn) return RKH_ERR_BAD_ARG;
  nn->n = 0;
  nn->max_abs_coord = 0.0;
  nn->removed.clear();
  nn->n_removed = 0;
  return RKH_OK;
}
uint64_t rkh_nn_size(const rkh_nn* nn) { return nn ? nn->n : 0; }
uint64_t rkh_nn_live_size(const rkh_nn* nn) { return nn ? nn->n - nn->n_removed : 0; }

rkh_status rkh_nn_remove(rkh_nn* nn, uint64_t index) {
  if (!nn) return RKH_ERR_BAD_ARG;
  if (index >= nn->n) {
    set_error("rkh_nn_remove: no such vertex");
    return RKH_ERR_BAD_ARG;
  }
  if (nn->removed.size() < nn->n) nn->removed.resize(nn->n, 0);
  if (nn->removed[index]) return RKH_OK;
  const int DP = nn_padded_dims(nn->st.D);
  double row[64];
  for (int d = 0; d < DP; ++d) row[d] = INFINITY;  // the padding columns too: rows past the end look the same to a sweep
  hipStream_t s = nn->ctx->stream;
  RKH_HIP(hipMemcpyAsync(nn->st.d_pos + index * DP, row, DP * sizeof(double), hipMemcpyHostToDevice, s));
  RKH_HIP(hipStreamSynchronize(s));
  nn->removed[index] = 1;
  ++nn->n_removed;
  return RKH_OK;
}

rkh_status rkh_nn_append(rkh_nn* nn, const double* pts, uint64_t n) {
  if (!nn || (!pts && n)) return RKH_ERR_BAD_ARG;
  if (nn->n + n > nn->st.capacity) {
    set_error("rkh_nn_append: capacity exceeded");
    return RKH_ERR_CAPACITY;
  }
  if (n == 0) return RKH_OK;
  const int D = nn->st.D, DP = nn_padded_dims(D);
  // the single-precision pre-filters are exact only if every |coordinate| <= coord_bound
  double amax = 0.0;
  for (uint64_t i = 0; i < n * uint64_t(D); ++i) {
    const double v = std::fabs(pts[i]);
    if (!(v <= amax)) amax = v;  // NaN ends up in amax
  }
  if (nn->coord_bound > 0.0 && !(amax <= nn->coord_bound)) {
    set_error("rkh_nn_append: a coordinate exceeds the bound given to rkh_nn_set_coord_bound");
    return RKH_ERR_BAD_ARG;
  }
  if (!(amax <= nn->max_abs_coord)) nn->max_abs_coord = amax;
  hipStream_t s = nn->ctx->stream;
  if (DP == D) {
    RKH_HIP(hipMemcpyAsync(nn->st.d_pos + nn->n * DP, pts, n * D * sizeof(double), hipMemcpyHostToDevice, s));
  } else {
    std::vector<double> tmp(n * DP, 0.0);
    for (uint64_t i = 0; i < n; ++i) std::memcpy(&tmp[i * DP], pts + i * D, D * sizeof(double));
    RKH_HIP(hipMemcpyAsync(nn->st.d_pos + nn->n * DP, tmp.data(), n * DP * sizeof(double), hipMemcpyHostToDevice, s));
    RKH_HIP(hipStreamSynchronize(s));
  }
  RKH_HIP(hipStreamSynchronize(s));
  nn->n += n;
  return RKH_OK;
}

static rkh_status ensure_partials(rkh_nn* nn, uint32_t B) {
  const uint32_t blocks = nn1_partial_blocks(nn->n, B);
  const uint64_t need = uint64_t(blocks) * B;
  if (need > nn->part_cap) {
    hipFree(nn->d_part_dist);
    hipFree(nn->d_part_idx);
    nn->d_part_dist = nullptr;
    nn->d_part_idx = nullptr;
    RKH_HIP(hipMalloc(&nn->d_part_dist, need * sizeof(double)));
    RKH_HIP(hipMalloc(&nn->d_part_idx, need * sizeof(uint32_t)));
    nn->part_cap = need;
  }
  nn->part_blocks = blocks;
  if (B > nn->seed_cap) {
    hipFree(nn->d_seed);
    nn->d_seed = nullptr;
    RKH_HIP(hipMalloc(&nn->d_seed, uint64_t(B) * sizeof(uint32_t)));
    RKH_HIP(hipMemsetAsync(nn->d_seed, 0xFF, uint64_t(B) * sizeof(uint32_t), nn->ctx->stream));
    nn->seed_cap = B;
  }
  return RKH_OK;
}

static rkh_status ensure_scratch(rkh_nn* nn, uint64_t q_elems, uint64_t res_elems) {
  if (q_elems > nn->q_cap) {
    hipFree(nn->d_q);
    nn->d_q = nullptr;
    RKH_HIP(hipMalloc(&nn->d_q, q_elems * sizeof(double)));
    nn->q_cap = q_elems;
  }
  if (res_elems > nn->res_cap) {
    hipFree(nn->d_idx);
    hipFree(nn->d_dist);
    hipFree(nn->d_count);
    nn->d_idx = nullptr;
    nn->d_dist = nullptr;
    nn->d_count = nullptr;
    RKH_HIP(hipMalloc(&nn->d_idx, res_elems * sizeof(uint32_t)));
    RKH_HIP(hipMalloc(&nn->d_dist, res_elems * sizeof(double)));
    RKH_HIP(hipMalloc(&nn->d_count, res_elems * sizeof(uint32_t)));
    nn->res_cap = res_elems;
  }
  return RKH_OK;
}

rkh_status rkh_nn_set_coord_bound(rkh_nn* nn, double bound) {
  if (!nn || !(bound >= 0.0)) return RKH_ERR_BAD_ARG;
  if (bound > 0.0 && !(nn->max_abs_coord <= bound)) {
    set_error("rkh_nn_set_coord_bound: rows already appended exceed the bound");
    return RKH_ERR_BAD_ARG;
  }
  nn->coord_bound = bound;
  return RKH_OK;
}

rkh_status rkh_nn_query1_async(rkh_nn* nn, const double* d_q, uint32_t B, uint32_t* d_idx, double* d_dist) {
  if (!nn || !d_q || !d_idx || !d_dist) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  rkh_status st = ensure_partials(nn, B);
  if (st != RKH_OK) return st;
  NnArgs a;
  a.pos = nn->st.d_pos;
  a.n = nn->n;
  a.q = d_q;
  a.B = B;
  a.part_dist = nn->d_part_dist;
  a.part_idx = nn->d_part_idx;
  a.idx = d_idx;
  a.dist = d_dist;
  a.seed = nn->d_seed;
  hipEvent_t e0 = nn->ev0, e1 = nn->ev1;
  nn->ev0 = nn->ev1 = nullptr;
  return launch_nn1(nn->ctx->stream, nn->st.D, a, nullptr, 1, nn->n, B, nn->part_blocks, e0, e1, nn->coord_bound);
}

rkh_status rkh_nn_query1(rkh_nn* nn, const double* q, uint32_t B, uint32_t* idx, double* dist) {
  if (!nn || !q || !idx || !dist) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  if (nn->coord_bound > 0.0) {
    for (uint64_t i = 0; i < uint64_t(B) * nn->st.D; ++i)
      if (!(std::fabs(q[i]) <= nn->coord_bound)) {
        set_error("rkh_nn_query1: a query coordinate exceeds the bound given to rkh_nn_set_coord_bound");
        return RKH_ERR_BAD_ARG;
      }
  }
  rkh_status st = ensure_scratch(nn, uint64_t(B) * nn->st.D, B);
  if (st != RKH_OK) return st;
  hipStream_t s = nn->ctx->stream;
  RKH_HIP(hipMemcpyAsync(nn->d_q, q, uint64_t(B) * nn->st.D * sizeof(double), hipMemcpyHostToDevice, s));
  st = rkh_nn_query1_async(nn, nn->d_q, B, nn->d_idx, nn->d_dist);
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(idx, nn->d_idx, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RKH_HIP(hipMemcpyAsync(dist, nn->d_dist, B * sizeof(double), hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  return RKH_OK;
}

rkh_status rkh_nn_queryk_async(rkh_nn* nn, const double* d_q, uint32_t B, uint32_t k, double radius, uint32_t* d_idx,
                               double* d_dist, uint32_t* d_count) {
  if (!nn || !d_q || !d_idx || !d_dist || !d_count || k == 0) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  KnnWorkspace ws;
  size_t bytes = 0;
  rkh_status st = knn_plan(nn->n, B, k, &ws, &bytes);
  if (st != RKH_OK) return st;
  if (bytes > nn->knn_ws_bytes) {
    (void)hipFree(nn->d_knn_ws);
    nn->d_knn_ws = nullptr;
    RKH_HIP(hipMalloc(&nn->d_knn_ws, bytes));
    nn->knn_ws_bytes = bytes;
  }
  knn_carve(nn->d_knn_ws, B, &ws);
  st = launch_nnk(nn->ctx->stream, nn->st, nn->n, d_q, B, k, radius, d_idx, d_dist, d_count, ws);
  if (st != RKH_OK) return st;
  return RKH_OK;
}

rkh_status rkh_nn_queryk(rkh_nn* nn, const double* q, uint32_t B, uint32_t k, double radius, uint32_t* idx,
                         double* dist, uint32_t* count) {
  if (!nn || !q || !idx || !dist || !count || k == 0) return RKH_ERR_BAD_ARG;
  if (B == 0) return RKH_OK;
  rkh_status st = ensure_scratch(nn, uint64_t(B) * nn->st.D, uint64_t(B) * k);
  if (st != RKH_OK) return st;
  hipStream_t s = nn->ctx->stream;
  RKH_HIP(hipMemcpyAsync(nn->d_q, q, uint64_t(B) * nn->st.D * sizeof(double), hipMemcpyHostToDevice, s));
  st = rkh_nn_queryk_async(nn, nn->d_q, B, k, radius, nn->d_idx, nn->d_dist, nn->d_count);
  if (st != RKH_OK) return st;
  RKH_HIP(hipMemcpyAsync(idx, nn->d_idx, uint64_t(B) * k * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RKH_HIP(hipMemcpyAsync(dist, nn->d_dist, uint64_t(B) * k * sizeof(double), hipMemcpyDeviceToHost, s));
  RKH_HIP(hipMemcpyAsync(count, nn->d_count, B * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  uint32_t overflow = 0;  // first word of the k-NN workspace
  RKH_HIP(hipMemcpyAsync(&overflow, nn->d_knn_ws, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  RKH_HIP(hipStreamSynchronize(s));
  if (overflow) {
    set_error("rkh_nn_queryk: candidate capacity exceeded (too many vertices within the bound; shrink the radius)");
    return RKH_ERR_CAPACITY;
  }
  return RKH_OK;
}

rkh_status rkh_nn_fill_uniform(rkh_nn* nn, uint64_t n, uint64_t seed) {
  if (!nn) return RKH_ERR_BAD_ARG;
  if (n > nn->st.capacity) {
    set_error("rkh_nn_fill_uniform: capacity exceeded");
    return RKH_ERR_CAPACITY;
  }
  rkh_status st = launch_fill_uniform(nn->ctx->stream, nn->st, n, seed);
  if (st != RKH_OK) return st;
  RKH_HIP(hipStreamSynchronize(nn->ctx->stream));
  nn->n = n;
  // every row was rewritten: no tombstone survives (a stale host copy would report a wrong live size and make a later
  // rkh_nn_remove of such an index a silent no-op)
  nn->removed.assign(nn->removed.size(), 0);
  nn->n_removed = 0;
  if (nn->max_abs_coord < 1.0) nn->max_abs_coord = 1.0;  // the unit hypercube
  return RKH_OK;
}

rkh_status rkh_nn_set_events(rkh_nn* nn, void* ev_start, void* ev_stop) {
  if (!nn) return RKH_ERR_BAD_ARG;
  nn->ev0 = static_cast<hipEvent_t>(ev_start);
  nn->ev1 = static_cast<hipEvent_t>(ev_stop);
  return RKH_OK;
}

const char* rkh_nn_kernel_name(void) { return nn_last_kernel_name(); }

}  // extern "C"
