// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// CPU restatement of ReaK's Dynamic Vantage-Point tree -- the reference's default NN structure
// (DVP_BF2_TREE_KNN, ctrl/path_planning/rrt_path_planner.hpp:113) and the subject of its only published benchmark
// (BASELINE.md section 1): ctrl/path_planning/dvp_tree_detail.hpp
//   random_best_vp_chooser                      :80-128   (draws from the GLOBAL rng: a side effect on the sample stream)
//   rearrange_with_chosen_vp / construct_node   :338-405  (breadth-first, std::nth_element partitions of equal size)
//   nearest_search_result_set                   :409-432  (bounded max-heap, std::push_heap / pop_heap)
//   find_nearest_impl                           :492-589  (stack search, children visited from the query's partition outwards)
//   get_leaf / update_mu_upwards / is_leaf_node / is_node_full :596-715
//   construction from a vertex range            :764-805
//   insert                                      :971-1031 (find the leaf, walk up to the first non-full ancestor, rebuild it)
// It serves as the honest "best CPU" baseline next to the GPU sweep (bench.py, tests/diag_knn.py); the product path
// never uses it.
//
// Parity pin status: the reference holds no expected outputs for the tree (test_vp_tree.cpp prints timings).  What can
// be checked, is: every query must return exactly the linear search's neighbours (tests/test_oracle_kat.py).
// Unpinned details restated from their published definitions: Boost's tree storage is replaced by a node array whose
// out_edges order is the insertion order of the children; BGL-Extra's remove_branch (the order in which insert()
// collects the vertices of the sub-tree it rebuilds) is taken as breadth-first, the order of the reference's own
// collect_vertices (:719-733) -- it changes the shape of rebuilt sub-trees, never a query result.
#ifndef REAK_ORACLE_DVP_TREE_HPP
#define REAK_ORACLE_DVP_TREE_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <queue>
#include <random>
#include <stack>
#include <unordered_map>
#include <utility>
#include <vector>

namespace oracle {

template <int Arity>
class DvpTree {
 public:
  typedef uint32_t key_type;  // index of the point in the caller's array (the graph's vertex descriptor)
  // points are rows of `pts` (row-major, D doubles each); rng = the global generator the vp chooser draws from
  DvpTree(const double* pts, int D, std::mt19937* rng, unsigned divider = 10)
      : pts_(pts), D_(D), rng_(rng), divider_(divider) {}

  // dvp_tree_impl(const Graph&, ...) :764-805: all vertices at once
  void build(std::size_t n) {
    nodes_.clear();
    root_ = -1;
    if (n == 0) return;
    std::vector<key_type> v_bin(n);
    for (std::size_t i = 0; i < n; ++i) v_bin[i] = key_type(i);
    rearrange_with_chosen_vp(v_bin, 0, n);
    root_ = new_node(v_bin[0], -1);
    construct_node(root_, v_bin, 1, n);
  }
  std::size_t size() const { return live_; }

  // insert(vertex_property) :971-1031
  void insert(key_type key) {
    if (live_ == 0) {
      nodes_.clear();
      root_ = new_node(key, -1);
      return;
    }
    const double* u_pt = pt(key);
    int u_subroot = get_leaf(u_pt, root_);
    if (u_subroot != root_) {
      int u_leaf = nodes_[u_subroot].parent;
      if ((int(nodes_[u_leaf].child.size()) == Arity) && is_leaf_node(u_leaf)) {
        int actual_depth_limit = 1;
        int last_depth_limit = actual_depth_limit;
        while ((u_leaf != root_) && is_node_full(u_leaf, last_depth_limit)) {
          u_leaf = nodes_[u_leaf].parent;
          last_depth_limit = ++actual_depth_limit;
        }
        bool is_p_full = false;
        if (u_leaf == root_) is_p_full = is_node_full(u_leaf, last_depth_limit);
        if ((!is_p_full) && (last_depth_limit >= 0)) u_subroot = u_leaf;
      } else {
        u_subroot = u_leaf;
      }
    }
    update_mu_upwards(u_pt, u_subroot);
    std::vector<key_type> prop_list;
    prop_list.push_back(key);
    // remove every branch below u_subroot, collecting their vertices (breadth-first per branch)
    while (!nodes_[u_subroot].child.empty()) {
      const int c = nodes_[u_subroot].child.front();
      std::queue<int> tasks;
      tasks.push(c);
      while (!tasks.empty()) {
        const int cur = tasks.front();
        tasks.pop();
        prop_list.push_back(nodes_[cur].key);
        for (int ch : nodes_[cur].child) tasks.push(ch);
        release_node(cur);
      }
      nodes_[u_subroot].child.erase(nodes_[u_subroot].child.begin());
      nodes_[u_subroot].mu.erase(nodes_[u_subroot].mu.begin());
    }
    construct_node(u_subroot, prop_list, 0, prop_list.size());
  }

  // find_nearest(aPoint) :1234-1243: one neighbour
  std::pair<key_type, double> find_nearest(const double* q) const {
    ResultSet rs(1, std::numeric_limits<double>::infinity());
    if (root_ >= 0) find_nearest_impl(q, rs);
    if (rs.Neighbors.empty()) return std::make_pair(key_type(0xFFFFFFFFu), std::numeric_limits<double>::infinity());
    return std::make_pair(rs.Neighbors.front().second, rs.Neighbors.front().first);
  }
  // find_nearest(aPoint, out, K, R) :1278-1288: nearest first (sort_heap)
  std::size_t find_nearest(const double* q, std::size_t K, double R, key_type* out_idx, double* out_dist) const {
    ResultSet rs(K, R);
    if (root_ >= 0) find_nearest_impl(q, rs);
    std::sort_heap(rs.Neighbors.begin(), rs.Neighbors.end(), priority_compare);
    for (std::size_t i = 0; i < rs.Neighbors.size(); ++i) {
      out_idx[i] = rs.Neighbors[i].second;
      out_dist[i] = rs.Neighbors[i].first;
    }
    return rs.Neighbors.size();
  }
  uint64_t distance_evaluations() const { return n_dist_; }

 private:
  struct Node {
    key_type key = 0;
    int parent = -1;
    std::vector<int> child;   // out_edges order = insertion order
    std::vector<double> mu;   // edge property of the edge to child[i]
    bool alive = false;
  };
  static bool priority_compare(const std::pair<double, key_type>& x, const std::pair<double, key_type>& y) {
    return x.first < y.first;  // :272-276
  }
  struct ResultSet {  // nearest_search_result_set :409-432
    std::vector<std::pair<double, key_type> > Neighbors;
    std::size_t K;
    double Radius;
    ResultSet(std::size_t aK, double aRadius) : K(aK), Radius(aRadius) {}
    void register_vantage_point(double current_dist, key_type current_vp) {
      if (current_dist < Radius) {
        Neighbors.push_back(std::make_pair(current_dist, current_vp));
        std::push_heap(Neighbors.begin(), Neighbors.end(), priority_compare);
        if (Neighbors.size() > K) {
          std::pop_heap(Neighbors.begin(), Neighbors.end(), priority_compare);
          Neighbors.pop_back();
          Radius = Neighbors.front().first;
        }
      }
    }
  };

  const double* pt(key_type k) const { return pts_ + std::size_t(k) * D_; }
  double distance(const double* a, const double* b) const {  // euclidean_distance_metric (vect_distance_metrics.hpp:113-137)
    ++n_dist_;
    double s = 0.0;
    for (int i = 0; i < D_; ++i) {
      const double d = a[i] - b[i];
      s += d * d;
    }
    return std::sqrt(s);
  }
  int new_node(key_type key, int parent) {
    int id;
    if (!free_.empty()) {
      id = free_.back();
      free_.pop_back();
      nodes_[id] = Node();
    } else {
      id = int(nodes_.size());
      nodes_.push_back(Node());
    }
    nodes_[id].key = key;
    nodes_[id].parent = parent;
    nodes_[id].alive = true;
    ++live_;
    return id;
  }
  void release_node(int id) {
    nodes_[id].alive = false;
    nodes_[id].child.clear();
    nodes_[id].mu.clear();
    free_.push_back(id);
    --live_;
  }

  // random_best_vp_chooser::operator() :102-127 on v[first, last); returns the chosen position (last = none)
  std::size_t choose_vp(const std::vector<key_type>& v, std::size_t first, std::size_t last) const {
    std::size_t best_pt = last;
    double best_dev = -1;
    const std::size_t n = last - first;
    for (unsigned int i = 0; i < n / divider_ + 1; ++i) {
      const std::size_t current_pt = first + ((*rng_)() % n);
      double current_mean = 0.0, current_dev = 0.0;
      const double* current_vp = pt(v[current_pt]);
      for (unsigned int j = 0; first + j != last; ++j) {
        const double dist = distance(current_vp, pt(v[first + j]));
        current_mean = (current_mean * j + dist) / (j + 1);
        current_dev = (current_dev * j + dist * dist) / (j + 1);
      }
      double current_var = current_dev - current_mean * current_mean;
      if (current_var < 0) current_var = 0.0;
      current_dev = std::sqrt(current_var);
      if (current_dev > best_dev) {
        best_pt = current_pt;
        best_dev = current_dev;
      }
    }
    return best_pt;
  }
  // rearrange_with_chosen_vp :338-353: the chosen vantage point goes to the front of the interval
  void rearrange_with_chosen_vp(std::vector<key_type>& v, std::size_t first, std::size_t last) const {
    if (first == last) return;
    const std::size_t chosen = choose_vp(v, first, last);
    if (chosen == last) return;
    std::swap(v[chosen], v[first]);
  }
  // construct_node :360-405 (breadth-first task queue)
  void construct_node(int aNode, std::vector<key_type>& v, std::size_t aBegin, std::size_t aEnd) {
    struct Task {
      int node;
      std::size_t first, last;
    };
    std::unordered_map<key_type, double> dist_map;
    std::queue<Task> tasks;
    tasks.push(Task{aNode, aBegin, aEnd});
    while (!tasks.empty()) {
      Task cur = tasks.front();
      tasks.pop();
      {
        const double* chosen_vp_pt = pt(nodes_[cur.node].key);
        for (std::size_t it = cur.first; it != cur.last; ++it) dist_map[v[it]] = distance(chosen_vp_pt, pt(v[it]));
      }
      std::size_t total_count = cur.last - cur.first;
      std::size_t child_count[Arity];
      for (std::size_t i = Arity; i > 0; --i) {
        child_count[i - 1] = total_count / i;
        total_count -= child_count[i - 1];
      }
      for (std::size_t i = 0; (i < std::size_t(Arity)) && (child_count[i] > 0); ++i) {
        std::nth_element(v.begin() + cur.first, v.begin() + cur.first + (child_count[i] - 1), v.begin() + cur.last,
                         [&](key_type a, key_type b) { return dist_map[a] < dist_map[b]; });  // closer :284-293
        std::size_t temp = cur.first;
        cur.first += child_count[i];
        const double ep = dist_map[v[cur.first - 1]];
        rearrange_with_chosen_vp(v, temp, cur.first);
        dist_map.erase(v[temp]);
        const int new_vp_node = new_node(v[temp], cur.node);
        nodes_[cur.node].child.push_back(new_vp_node);
        nodes_[cur.node].mu.push_back(ep);
        ++temp;
        if (temp != cur.first) tasks.push(Task{new_vp_node, temp, cur.first});
      }
    }
  }

  // find_nearest_impl :492-589
  void find_nearest_impl(const double* aPoint, ResultSet& aResult) const {
    std::stack<std::pair<int, double> > tasks;
    tasks.push(std::make_pair(root_, 0.0));
    while (!tasks.empty()) {
      const std::pair<int, double> cur_node = tasks.top();
      tasks.pop();
      if (cur_node.second > aResult.Radius) continue;
      const Node& nd = nodes_[cur_node.first];
      const double current_dist = distance(aPoint, pt(nd.key));
      aResult.register_vantage_point(current_dist, nd.key);
      const int deg = int(nd.child.size());
      if (deg == 0) continue;
      int ei = 0;
      for (; ei != deg; ++ei)
        if (current_dist <= nd.mu[ei]) break;
      if (ei == deg) --ei;
      std::stack<std::pair<int, double> > temp_invtasks;
      temp_invtasks.push(std::make_pair(nd.child[ei], 0.0));
      int ei_left = ei;
      int ei_right = ei + 1;
      bool left_stopped = (ei_left == 0);
      bool right_stopped = (ei_right == deg);
      while (true) {
        if (left_stopped) {
          int ei_rightleft = ei_right - 1;
          double temp_dist = 0.0;
          while ((ei_right != deg) && ((temp_dist = nd.mu[ei_rightleft] - current_dist) < aResult.Radius)) {
            temp_invtasks.push(std::make_pair(nd.child[ei_right], temp_dist));
            ++ei_rightleft;
            ++ei_right;
          }
          break;
        } else if (right_stopped) {
          int ei_leftleft = ei_left;
          double temp_dist = 0.0;
          while ((ei_left != 0) && ((temp_dist = current_dist - nd.mu[--ei_leftleft]) < aResult.Radius)) {
            temp_invtasks.push(std::make_pair(nd.child[ei_leftleft], temp_dist));
            --ei_left;
          }
          break;
        } else {
          const int ei_leftleft = ei_left - 1;
          const double d1 = nd.mu[ei_leftleft];
          const int ei_rightleft = ei_right - 1;
          const double d2 = nd.mu[ei_rightleft];
          if (d1 + d2 > 2.0 * current_dist) {
            if (d1 + aResult.Radius - current_dist > 0) {
              temp_invtasks.push(std::make_pair(nd.child[ei_leftleft], current_dist - d1));
              ei_left = ei_leftleft;
              if (d2 - aResult.Radius - current_dist < 0) {
                temp_invtasks.push(std::make_pair(nd.child[ei_right], d2 - current_dist));
                ++ei_right;
              } else {
                right_stopped = true;
              }
            } else {
              break;
            }
          } else {
            if (d2 - aResult.Radius - current_dist < 0) {
              temp_invtasks.push(std::make_pair(nd.child[ei_right], d2 - current_dist));
              ++ei_right;
              if (d1 + aResult.Radius - current_dist > 0) {
                temp_invtasks.push(std::make_pair(nd.child[ei_leftleft], current_dist - d1));
                ei_left = ei_leftleft;
              } else {
                left_stopped = true;
              }
            } else {
              break;
            }
          }
        }
        left_stopped = (ei_left == 0);
        right_stopped = (ei_right == deg);
      }
      while (!temp_invtasks.empty()) {
        tasks.push(temp_invtasks.top());
        temp_invtasks.pop();
      }
    }
  }

  // get_leaf :596-610
  int get_leaf(const double* aPoint, int aNode) const {
    while (!nodes_[aNode].child.empty()) {
      const double current_dist = distance(aPoint, pt(nodes_[aNode].key));
      int result = aNode;
      for (std::size_t ei = 0; ei < nodes_[aNode].child.size(); ++ei) {
        result = nodes_[aNode].child[ei];
        if (current_dist <= nodes_[aNode].mu[ei]) break;
      }
      aNode = result;
    }
    return aNode;
  }
  // update_mu_upwards :657-666
  void update_mu_upwards(const double* aPoint, int aNode) {
    while (aNode != root_) {
      const int parent = nodes_[aNode].parent;
      const double dist = distance(aPoint, pt(nodes_[parent].key));
      std::vector<int>& ch = nodes_[parent].child;
      const std::size_t e = std::size_t(std::find(ch.begin(), ch.end(), aNode) - ch.begin());
      if (dist > nodes_[parent].mu[e]) nodes_[parent].mu[e] = dist;
      aNode = parent;
    }
  }
  // is_leaf_node :670-678
  bool is_leaf_node(int aNode) const {
    if (nodes_[aNode].child.empty()) return true;
    for (int c : nodes_[aNode].child)
      if (!nodes_[c].child.empty()) return false;
    return true;
  }
  // is_node_full :684-713
  bool is_node_full(int aNode, int& depth_limit) const {
    if (depth_limit < 0) return false;
    std::queue<std::pair<int, int> > tasks;
    tasks.push(std::make_pair(aNode, depth_limit));
    while (!tasks.empty()) {
      std::pair<int, int> cur_task = tasks.front();
      tasks.pop();
      if (cur_task.second < depth_limit) depth_limit = cur_task.second;
      const int deg = int(nodes_[cur_task.first].child.size());
      if ((deg == 0) && (cur_task.second == 0)) continue;
      --(cur_task.second);
      if (((deg != 0) && (cur_task.second < 0)) || (deg < Arity) || ((cur_task.second > 0) && is_leaf_node(cur_task.first))) {
        depth_limit = cur_task.second;
        return false;
      }
      for (int c : nodes_[cur_task.first].child) tasks.push(std::make_pair(c, cur_task.second));
    }
    return (depth_limit == 0);
  }

  const double* pts_;
  int D_;
  std::mt19937* rng_;
  unsigned divider_;
  std::vector<Node> nodes_;
  std::vector<int> free_;
  int root_ = -1;
  std::size_t live_ = 0;
  mutable uint64_t n_dist_ = 0;
};

}  // namespace oracle
#endif
