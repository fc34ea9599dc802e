// malis.cpp -- MALIS loss weights and affinity-graph connected components (host code;
// SURVEY.md 8f-4).  Replaces malis/_malis_lib.cpp:38-125 (malis_loss_weights_cpp) and
// :128-167 (connected_components_cpp) -- the reference's one native component, there
// built on boost::disjoint_sets (Boost is not in this image): here a plain union-find.
//
// MALIS (Turaga et al.): run Kruskal on the affinity graph in order of DESCENDING edge
// weight (maximum spanning tree); when an edge joins two components, every pair of
// voxels (one from each side) has this edge as its maximin edge.  Each component keeps
// a sparse histogram {ground-truth id -> voxel count}; the joining edge is credited with
// sum n1*n2 over pairs of histogram entries with EQUAL ids (pos pass: pairs that must be
// connected) or DIFFERENT ids (neg pass).  Voxels with id 0 are in no histogram.
// Sequential by nature: stays on the host (the training loop overlaps it with the next
// forward pass).  Ties between equal weights are broken by std::sort on the edge index
// array exactly as the reference does, so its known-answer vectors
// (tests/test_malis.py:36-77) are reproduced.
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <numeric>
#include <vector>
#include "../../include/e2hip.h"

namespace {

struct UnionFind {
  std::vector<int> parent, rank;
  explicit UnionFind(int n) : parent(n), rank(n, 0) { std::iota(parent.begin(), parent.end(), 0); }
  int find(int x) {
    int r = x;
    while (parent[r] != r) r = parent[r];
    while (parent[x] != r) { const int nx = parent[x]; parent[x] = r; x = nx; }   // full compression
    return r;
  }
  // link two ROOTS by rank; returns the new root
  int link(int a, int b) {
    if (rank[a] > rank[b]) { parent[b] = a; return a; }
    parent[a] = b;
    if (rank[a] == rank[b]) ++rank[b];
    return b;
  }
};

struct ByWeightDesc {
  const float* w;
  bool operator()(const int& a, const int& b) const { return w[a] > w[b]; }
};

bool valid_edge(int a, int b, int n) { return a >= 0 && a < n && b >= 0 && b < n; }

}  // namespace

/* counts[e] (uint64, caller-zeroed or not: overwritten) = number of voxel pairs whose
 * maximin edge is e and whose ground-truth ids are equal (pos != 0) / different (pos == 0).
 * seg[n_vert] ground-truth ids (0 = unlabelled); node1/node2[n_edge] vertex indices
 * (out-of-range = edge absent); edge_weight[n_edge]. */
extern "C" int e2_malis_loss_weights(int n_vert, const int32_t* seg, int n_edge,
                                     const int32_t* node1, const int32_t* node2,
                                     const float* edge_weight, int pos, uint64_t* counts) {
  if (n_vert < 0 || n_edge < 0 || (n_vert && !seg) ||
      (n_edge && (!node1 || !node2 || !edge_weight || !counts)))
    return 2;
  std::fill(counts, counts + n_edge, (uint64_t)0);
  std::vector<std::map<int, uint64_t> > overlap(n_vert);
  UnionFind uf(n_vert);
  for (int i = 0; i < n_vert; ++i)
    if (seg[i] != 0) overlap[i].insert(std::make_pair((int)seg[i], (uint64_t)1));
  std::vector<int> order;
  order.reserve(n_edge);
  for (int i = 0; i < n_edge; ++i)
    if (valid_edge(node1[i], node2[i], n_vert)) order.push_back(i);
  std::sort(order.begin(), order.end(), ByWeightDesc{edge_weight});
  for (size_t k = 0; k < order.size(); ++k) {
    const int e = order[k];
    int s1 = uf.find(node1[e]), s2 = uf.find(node2[e]);
    if (s1 == s2) continue;
    uint64_t add = 0;
    for (const auto& a : overlap[s1])
      for (const auto& b : overlap[s2])
        if (pos ? (a.first == b.first) : (a.first != b.first)) add += a.second * b.second;
    counts[e] += add;
    const int keep = uf.link(s1, s2);
    const int drop = (keep == s1) ? s2 : s1;
    std::map<int, uint64_t>& into = overlap[keep];
    for (const auto& b : overlap[drop]) into[b.first] += b.second;
    overlap[drop].clear();
  }
  return 0;
}

/* seg[v] = 1 + representative of v's component under the edges with |weight| > 1e-5,
 * components of <= size_thresh voxels set to 0 (the caller renumbers). */
extern "C" int e2_malis_connected_components(int n_vert, int n_edge, const int32_t* node1,
                                             const int32_t* node2, const float* edge_weight,
                                             int size_thresh, int32_t* seg) {
  if (n_vert < 0 || n_edge < 0 || (n_vert && !seg) ||
      (n_edge && (!node1 || !node2 || !edge_weight)))
    return 2;
  UnionFind uf(n_vert);
  for (int i = 0; i < n_edge; ++i)
    if (std::fabs(edge_weight[i]) > 1e-5f && valid_edge(node1[i], node2[i], n_vert)) {
      const int a = uf.find(node1[i]), b = uf.find(node2[i]);
      if (a != b) uf.link(a, b);
    }
  std::map<int, int> sizes;
  for (int i = 0; i < n_vert; ++i) { seg[i] = uf.find(i) + 1; ++sizes[seg[i]]; }
  for (int i = 0; i < n_vert; ++i)
    if (sizes[seg[i]] <= size_thresh) seg[i] = 0;
  return 0;
}
