// hso_octree.cpp — spatial distribution of the keyframe candidates: the last step of
// FeatureExtractor::detect (reference src/feature_detection.cpp:449-455 calling
// computeKeyPointsOctTree :833-1122, ExtractorNode::DivideNode include/hso/feature_detection.h:217-272).
//
// Host code behind the C-ABI (no GPU work: a few thousand candidates, a sequential refinement whose
// every step depends on the previous one).  The reference keeps a std::list of nodes that each own
// a std::vector of KeyPoints and copies the keys on every split.  Here the candidates are never
// copied: one index array is partitioned in place (stably, so a child sees its keys in the parent's
// order — the order the reference's push_back produces), a node is {rectangle, index range, links}
// in a pool, and the list is intrusive (prev / next indices), which keeps the reference's visiting
// order: children are linked in at the front, a sweep walks the nodes that existed when it began.
//
// One deliberate definition: the reference sorts (size, node pointer) pairs, so among nodes of
// equal size the split order follows heap addresses — not defined by the program.  Here the pool
// index (creation order) stands in for the address.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>
#include "../../include/hso_gpu.h"

namespace {

struct Node {
  int x0, y0, x1, y1;     // UL = (x0, y0), UR = (x1, y0), BL = (x0, y1), BR = (x1, y1)
  int kb, ke;             // keys: order[kb .. ke)
  int prev, next;
  bool no_more;
};

struct Tree {
  const hso_keypoint* keys;
  std::vector<int> order, scratch;
  std::vector<Node> pool;
  int head = -1, count = 0;

  int push_front(const Node& n)
  {
    const int id = (int)pool.size();
    pool.push_back(n);
    pool[id].prev = -1; pool[id].next = head;
    if (head >= 0) pool[head].prev = id;
    head = id; ++count;
    return id;
  }
  void push_back_initial(const Node& n, int& tail)
  {
    const int id = (int)pool.size();
    pool.push_back(n);
    pool[id].prev = tail; pool[id].next = -1;
    if (tail >= 0) pool[tail].next = id; else head = id;
    tail = id; ++count;
  }
  void erase(int id)
  {
    const int p = pool[id].prev, n = pool[id].next;
    if (p >= 0) pool[p].next = n; else head = n;
    if (n >= 0) pool[n].prev = p;
    --count;
  }
  // DivideNode: the four children in the reference's order n1 (upper left), n2 (upper right),
  // n3 (lower left), n4 (lower right); their key ranges are laid out in that order inside the parent's
  void divide(int id, Node ch[4])
  {
    const Node P = pool[id];
    const int halfX = (int)std::ceil((float)(P.x1 - P.x0) / 2);
    const int halfY = (int)std::ceil((float)(P.y1 - P.y0) / 2);
    const int mx = P.x0 + halfX, my = P.y0 + halfY;
    ch[0] = {P.x0, P.y0, mx, my, 0, 0, -1, -1, false};
    ch[1] = {mx, P.y0, P.x1, my, 0, 0, -1, -1, false};
    ch[2] = {P.x0, my, mx, P.y1, 0, 0, -1, -1, false};
    ch[3] = {mx, my, P.x1, P.y1, 0, 0, -1, -1, false};
    int cnt[4] = {0, 0, 0, 0};
    scratch.resize(P.ke - P.kb);
    for (int i = P.kb; i < P.ke; ++i) {
      const hso_keypoint& k = keys[order[i]];
      const int q = (k.x < (float)mx ? 0 : 1) + (k.y < (float)my ? 0 : 2);
      scratch[i - P.kb] = q; ++cnt[q];
    }
    int at[4] = {P.kb, P.kb + cnt[0], P.kb + cnt[0] + cnt[1], P.kb + cnt[0] + cnt[1] + cnt[2]};
    for (int q = 0; q < 4; ++q) { ch[q].kb = at[q]; ch[q].ke = at[q] + cnt[q]; ch[q].no_more = cnt[q] == 1; }
    std::vector<int> moved(P.ke - P.kb);
    for (int i = P.kb; i < P.ke; ++i) moved[at[scratch[i - P.kb]]++ - P.kb] = order[i];
    std::copy(moved.begin(), moved.end(), order.begin() + P.kb);
  }
};

}  // namespace

extern "C" int hso_gpu_select_octree(const hso_keypoint* keys, int n, int min_x, int max_x, int min_y, int max_y, int n_features,
                                        hso_keypoint* out, int cap)
{
  if (n < 0 || cap < 0 || (n > 0 && !keys) || (cap > 0 && !out) || max_x <= min_x || max_y <= min_y) return HSO_E_INVALID;
  const int nIni = (int)std::round((float)(max_x - min_x) / (max_y - min_y));
  if (nIni < 1) return HSO_E_INVALID;
  const float hX = (float)(max_x - min_x) / nIni;
  Tree T;
  T.keys = keys;
  // initial column nodes, keys in input order
  std::vector<int> col(n);
  std::vector<int> col_count(nIni, 0);
  for (int i = 0; i < n; ++i) {
    const int x = (int)keys[i].x;
    const int c = (int)(x / hX);
    if (c < 0 || c >= nIni) return HSO_E_INVALID;      // out-of-range write in the reference
    col[i] = c; ++col_count[c];
  }
  std::vector<int> col_at(nIni + 1, 0);
  for (int c = 0; c < nIni; ++c) col_at[c + 1] = col_at[c] + col_count[c];
  T.order.resize(n);
  {
    std::vector<int> fill(col_at.begin(), col_at.end() - 1);
    for (int i = 0; i < n; ++i) T.order[fill[col[i]]++] = i;
  }
  T.pool.reserve((size_t)4 * (n_features > 0 ? n_features : 1) + 64);
  int tail = -1;
  for (int c = 0; c < nIni; ++c) {
    if (col_count[c] == 0) continue;                   // empty initial nodes are erased (:872-873)
    Node nd{(int)(hX * (float)c), min_y, (int)(hX * (float)(c + 1)), max_y, col_at[c], col_at[c + 1], -1, -1, col_count[c] == 1};
    T.push_back_initial(nd, tail);
  }
  std::vector<std::pair<int, int>> expandable;          // (key count, node) of the multi-key nodes created last
  bool finish = false;
  while (!finish) {
    const int prev_size = T.count;
    int n_to_expand = 0;
    expandable.clear();
    for (int it = T.head; it >= 0;) {
      const int next = T.pool[it].next;
      if (!T.pool[it].no_more) {
        Node ch[4];
        T.divide(it, ch);
        for (int q = 0; q < 4; ++q) {
          const int sz = ch[q].ke - ch[q].kb;
          if (sz == 0) continue;
          const int id = T.push_front(ch[q]);
          if (sz > 1) { ++n_to_expand; expandable.emplace_back(sz, id); }
        }
        T.erase(it);
      }
      it = next;
    }
    if (T.count >= n_features || T.count == prev_size) {
      finish = true;
    } else if (T.count + n_to_expand * 3 > n_features) {
      while (!finish) {
        const int prev2 = T.count;
        std::vector<std::pair<int, int>> todo;
        todo.swap(expandable);
        std::sort(todo.begin(), todo.end());
        for (int j = (int)todo.size() - 1; j >= 0; --j) {
          Node ch[4];
          T.divide(todo[j].second, ch);
          for (int q = 0; q < 4; ++q) {
            const int sz = ch[q].ke - ch[q].kb;
            if (sz == 0) continue;
            const int id = T.push_front(ch[q]);
            if (sz > 1) expandable.emplace_back(sz, id);
          }
          T.erase(todo[j].second);
          if (T.count >= n_features) break;
        }
        if (T.count >= n_features || T.count == prev2) finish = true;
      }
    }
  }
  // one key per node (:1083-1119): lowest species, then highest response; a node that meets an
  // occupancy key (kOccur) before the end of its list yields nothing
  int n_out = 0;
  for (int it = T.head; it >= 0; it = T.pool[it].next) {
    const Node& nd = T.pool[it];
    const hso_keypoint* best = &keys[T.order[nd.kb]];
    if (best->species == HSO_KP_OCCUR) continue;
    float max_score = best->response;
    bool have_occur = false;
    for (int i = nd.kb + 1; i < nd.ke; ++i) {
      const hso_keypoint* k = &keys[T.order[i]];
      if (k->species == HSO_KP_OCCUR) { have_occur = true; break; }
      if (best->species > k->species) { best = k; max_score = k->response; }
      else if (best->species == k->species && k->response > max_score) { best = k; max_score = k->response; }
    }
    if (have_occur) continue;
    if (n_out < cap) out[n_out] = *best;
    ++n_out;
  }
  return n_out;
}
