// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// A static vantage-point tree for exact 1-NN under the Euclidean metric: the CPU yardstick for the NN sweep that is
// *not* a linear search.  It is NOT a restatement of the reference's dynamic vantage-point tree (DVP-tree,
// ctrl/path_planning/metric_space_search.hpp:172-, dvp_tree_detail.hpp:200-1421: arity 2/4, incremental insertion,
// random vantage-point chooser drawing from the global RNG) -- that is 1400 lines of BGL-based container code -- but it is
// the same search principle (prune a subtree when |d(q, vp) - mu| exceeds the best distance so far), so its timings on
// the box's CPU stand in for "what a tree-based CPU NN achieves" next to BASELINE.md's published DVP-tree numbers.
// Ties: the search returns the lowest index among equal distances, like min_dist_linear_search.
#ifndef REAK_ORACLE_VP_TREE_HPP
#define REAK_ORACLE_VP_TREE_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle {

class VpTree {
 public:
  VpTree(const double* pts, std::size_t n, int D) : pts_(pts), D_(D) {
    idx_.resize(n);
    for (std::size_t i = 0; i < n; ++i) idx_[i] = uint32_t(i);
    nodes_.reserve(n);
    root_ = build(0, n);
  }
  // exact nearest neighbour (index, distance)
  std::pair<uint32_t, double> nearest(const double* q) const {
    uint32_t best_i = 0xFFFFFFFFu;
    double best_d = std::numeric_limits<double>::infinity();
    search(root_, q, best_i, best_d);
    return std::make_pair(best_i, best_d);
  }

 private:
  struct Node {
    uint32_t vp;      // point index of the vantage point
    double mu;        // median distance of the rest to it
    int32_t inner, outer;
  };
  double dist(const double* a, const double* b) const {
    double r = 0.0;
    for (int i = 0; i < D_; ++i) {
      const double d = a[i] - b[i];
      r += d * d;
    }
    return std::sqrt(r);
  }
  int32_t build(std::size_t lo, std::size_t hi) {
    if (lo >= hi) return -1;
    const int32_t id = int32_t(nodes_.size());
    nodes_.push_back(Node{idx_[lo], 0.0, -1, -1});
    if (hi - lo == 1) return id;
    const double* vp = pts_ + std::size_t(idx_[lo]) * D_;
    const std::size_t mid = lo + 1 + (hi - lo - 1) / 2;
    std::nth_element(idx_.begin() + lo + 1, idx_.begin() + mid, idx_.begin() + hi, [&](uint32_t a, uint32_t b) {
      return dist(vp, pts_ + std::size_t(a) * D_) < dist(vp, pts_ + std::size_t(b) * D_);
    });
    const double mu = dist(vp, pts_ + std::size_t(idx_[mid]) * D_);
    const int32_t inner = build(lo + 1, mid), outer = build(mid, hi);
    nodes_[id].mu = mu;
    nodes_[id].inner = inner;
    nodes_[id].outer = outer;
    return id;
  }
  void search(int32_t id, const double* q, uint32_t& best_i, double& best_d) const {
    if (id < 0) return;
    const Node& nd = nodes_[id];
    const double d = dist(q, pts_ + std::size_t(nd.vp) * D_);
    if (d < best_d || (d == best_d && nd.vp < best_i)) {
      best_d = d;
      best_i = nd.vp;
    }
    // inner holds points with distance <= mu (up to ties at the median), outer those with distance >= mu
    if (d < nd.mu) {
      if (d - best_d <= nd.mu) search(nd.inner, q, best_i, best_d);
      if (d + best_d >= nd.mu) search(nd.outer, q, best_i, best_d);
    } else {
      if (d + best_d >= nd.mu) search(nd.outer, q, best_i, best_d);
      if (d - best_d <= nd.mu) search(nd.inner, q, best_i, best_d);
    }
  }
  const double* pts_;
  int D_;
  std::vector<uint32_t> idx_;
  std::vector<Node> nodes_;
  int32_t root_ = -1;
};

}  // namespace oracle
#endif
