// Host-side tensorizers: abstract graphs (flat arrays) -> the reference's holder tables, or directly the CSR /
// byte-feature form the kernels consume.  Plain C++ over host memory; graphs of a batch are independent given
// their offsets, so a selection is cut into contiguous ranges that are filled side by side.
//
// Numbering follows the reference exactly:
//   molecules (fast_jtnn/mpn.py:110-161): bond i of a molecule, begin atom x, end atom y, becomes directed bond
//     b = (x -> y) and b + 1 = (y -> x), ids from 1 over the batch; in_bonds[y] gets b, in_bonds[x] gets b + 1;
//     agraph[a] = in_bonds[a]; bgraph[b1 = (x -> y)][i] = in_bonds[x][i] unless that bond starts at y (its slot
//     stays 0: zeros are interleaved, not trailing).
//   candidates (fast_jtnn/jtmpn.py:113-183): the same with ids from n_mess and rows from 0, plus, after the two
//     bond entries of a bond whose atoms belong to different tree nodes, the tree messages between those nodes.
//   trees (fast_jtnn/jtnn_enc.py:44-92): messages numbered from 1, node by node, neighbour by neighbour;
//     node_graph[y] = messages arriving at y, mess_graph[(y -> z)] = those of them that do not come from z, both
//     in increasing message id, zeros trailing.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/jtvae_hip.h"

namespace {

// fn(t, lo, hi) over [0, n) cut into contiguous ranges of about equal weight (weight prefix: w[i] .. w[i+1]).
// Returns the first non-zero code any range returned.
int run_ranges(int64_t n, const std::vector<int64_t>& wprefix, int n_threads,
               const std::function<int(int, int64_t, int64_t)>& fn, std::vector<int64_t>* bounds_out = nullptr) {
  int T = n_threads < 1 ? 1 : n_threads;
  if ((int64_t)T > n) T = n > 0 ? (int)n : 1;
  std::vector<int64_t> bounds((size_t)T + 1, 0);
  const int64_t total = wprefix.empty() ? n : wprefix[(size_t)n];
  for (int t = 1; t < T; ++t) {
    const int64_t want = total * t / T;
    int64_t pos;
    if (wprefix.empty()) {
      pos = want;
    } else {
      pos = std::lower_bound(wprefix.begin(), wprefix.begin() + n + 1, want) - wprefix.begin();
    }
    bounds[(size_t)t] = std::min<int64_t>(std::max<int64_t>(pos, bounds[(size_t)t - 1]), n);
  }
  bounds[(size_t)T] = n;
  if (bounds_out) *bounds_out = bounds;
  std::vector<int> rc((size_t)T, 0);
  if (T == 1) return fn(0, 0, n);
  std::vector<std::thread> pool;
  for (int t = 1; t < T; ++t)
    pool.emplace_back([&, t] { rc[(size_t)t] = fn(t, bounds[(size_t)t], bounds[(size_t)t + 1]); });
  rc[0] = fn(0, bounds[0], bounds[1]);
  for (auto& th : pool) th.join();
  for (int c : rc)
    if (c) return c;
  return 0;
}

inline int64_t sel(const int64_t* ids, int64_t i) { return ids ? ids[i] : i; }

// outgoing messages of every node, for the (x, y) -> message id lookups of the candidate graphs
struct OutIndex {
  std::vector<int32_t> ptr, list;
  const int32_t* dst = nullptr;
  int build(const jtvae_mess_index_t* m) {
    dst = m->dst;
    ptr.assign((size_t)m->n_nodes + 1, 0);
    list.resize((size_t)(m->n_mess > 0 ? m->n_mess : 0));
    for (int64_t k = 1; k < m->n_mess; ++k) {
      if (m->src[k] < 0 || m->src[k] >= m->n_nodes || m->dst[k] < 0 || m->dst[k] >= m->n_nodes) return JTVAE_E_RANGE;
      ++ptr[(size_t)m->src[k] + 1];
    }
    for (int64_t x = 0; x < m->n_nodes; ++x) ptr[(size_t)x + 1] += ptr[(size_t)x];
    std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t k = 1; k < m->n_mess; ++k) list[(size_t)cur[(size_t)m->src[k]]++] = (int32_t)k;
    return 0;
  }
  // id of message (x -> y), 0 if there is none; a pair that occurs twice resolves to the later id, as a
  // dictionary filled in id order does
  inline int32_t find(int64_t x, int64_t y) const {
    int32_t hit = 0;
    for (int32_t e = ptr[(size_t)x]; e < ptr[(size_t)x + 1]; ++e)
      if (dst[list[(size_t)e]] == y) hit = list[(size_t)e];
    return hit;
  }
};

int check_graph_set(const jtvae_graph_set_t* s, const int64_t* ids, int64_t n, const jtvae_mess_index_t* mess) {
  if (!s) return JTVAE_E_NULL;
  if (n < 0 || s->n_graphs < 0 || s->atom_fdim < 0 || s->bond_fdim < 0) return JTVAE_E_SHAPE;
  if (n > 0 && (!s->atom_ptr || !s->bond_ptr)) return JTVAE_E_NULL;
  if (mess && (!s->atom_nid || mess->n_mess < 1 || mess->n_nodes < 0 || !mess->node_base ||
               (mess->n_mess > 1 && (!mess->src || !mess->dst))))
    return mess->n_mess < 1 ? JTVAE_E_SHAPE : JTVAE_E_NULL;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    if (g < 0 || g >= s->n_graphs) return JTVAE_E_RANGE;
    if (s->atom_ptr[g + 1] < s->atom_ptr[g] || s->bond_ptr[g + 1] < s->bond_ptr[g]) return JTVAE_E_SHAPE;
  }
  return 0;
}

}  // namespace

extern "C" {

int jtvae_tensorize_graph_counts(const jtvae_graph_set_t* s, const int64_t* ids, int64_t n,
                                 const jtvae_mess_index_t* mess, int64_t* counts) {
  if (!counts) return JTVAE_E_NULL;
  int rc = check_graph_set(s, ids, n, mess);
  if (rc) return rc;
  int64_t atoms = 0, bonds = 0, a_cap = 0, b_cap = 0;
  std::vector<int32_t> deg, extra;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    const int64_t na = s->atom_ptr[g + 1] - s->atom_ptr[g], nb = s->bond_ptr[g + 1] - s->bond_ptr[g];
    if (nb > 0 && !s->bonds) return JTVAE_E_NULL;
    deg.assign((size_t)na, 0);
    extra.assign((size_t)na, 0);
    const int32_t* bd = s->bonds + 2 * s->bond_ptr[g];
    const int32_t* nid = mess ? s->atom_nid + s->atom_ptr[g] : nullptr;
    for (int64_t k = 0; k < nb; ++k) {
      const int32_t x = bd[2 * k], y = bd[2 * k + 1];
      if (x < 0 || x >= na || y < 0 || y >= na) return JTVAE_E_RANGE;
      ++deg[(size_t)x];
      ++deg[(size_t)y];
      if (nid && nid[x] > 0 && nid[y] > 0 && nid[x] != nid[y]) {  // at most one tree message each way
        ++extra[(size_t)x];
        ++extra[(size_t)y];
      }
    }
    atoms += na;
    bonds += nb;
    for (int64_t a = 0; a < na; ++a) {
      a_cap += deg[(size_t)a] + extra[(size_t)a];
      // a bond leaving x reads everything that arrives at x except (at least) its own reverse
      b_cap += (int64_t)deg[(size_t)a] * (deg[(size_t)a] + extra[(size_t)a] - 1);
    }
  }
  counts[0] = atoms;
  counts[1] = bonds;
  counts[2] = a_cap;
  counts[3] = b_cap;
  return 0;
}

int jtvae_tensorize_graphs(const jtvae_graph_set_t* s, const int64_t* ids, int64_t n, int32_t max_nb,
                           const jtvae_mess_index_t* mess, jtvae_graph_tables_t* tb, jtvae_graph_csr_t* cs,
                           int32_t n_threads) {
  int rc = check_graph_set(s, ids, n, mess);
  if (rc) return rc;
  if (max_nb < 1 || max_nb > 255) return JTVAE_E_SHAPE;
  const int32_t fa = s->atom_fdim, fb = s->bond_fdim, fw = fa + fb;
  const int64_t pad = mess ? 0 : 1;                   // molecules: row 0 of the bond tables is the padding row
  const int64_t first_id = mess ? mess->n_mess : 1;   // id of the batch's first directed bond
  const int64_t n_tree = mess ? mess->n_mess : 0;
  // offsets of every selected graph
  std::vector<int64_t> a0((size_t)n + 1, 0), b0((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    a0[(size_t)i + 1] = a0[(size_t)i] + (s->atom_ptr[g + 1] - s->atom_ptr[g]);
    b0[(size_t)i + 1] = b0[(size_t)i] + (s->bond_ptr[g + 1] - s->bond_ptr[g]);
  }
  const int64_t NA = a0[(size_t)n], NBu = b0[(size_t)n], rows = 2 * NBu + pad;
  if (first_id + 2 * NBu > INT32_MAX || NA > INT32_MAX) return JTVAE_E_RANGE;
  if (NA > 0 && (!s->atom_feats && fa > 0)) return JTVAE_E_NULL;
  if (NBu > 0 && (!s->bonds || (!s->bond_feats && fb > 0))) return JTVAE_E_NULL;
  if (cs && (!cs->a_rowptr || !cs->b_rowptr || (cs->a_cap > 0 && !cs->a_col) || (cs->b_cap > 0 && !cs->b_col)))
    return JTVAE_E_NULL;
  OutIndex out;
  if (mess) {
    rc = out.build(mess);
    if (rc) return rc;
  }
  if (pad) {  // the padding row
    if (tb && tb->fbonds) std::memset(tb->fbonds, 0, sizeof(float) * (size_t)fw);
    if (tb && tb->bgraph) std::memset(tb->bgraph, 0, sizeof(int64_t) * (size_t)max_nb);
    if (cs && cs->fbonds) std::memset(cs->fbonds, 0, (size_t)fw);
  }
  if (cs) {
    cs->a_rowptr[0] = 0;
    cs->b_rowptr[0] = 0;
    if (pad) cs->b_rowptr[1] = 0;
  }
  const int T = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  std::vector<std::vector<int32_t>> a_cols((size_t)T), b_cols((size_t)T);
  std::vector<int64_t> tree_a((size_t)T, 0), tree_b((size_t)T, 0), bounds;
  // weight of a graph ~ its bonds (+1 so that bond-less graphs still spread)
  std::vector<int64_t> w((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) w[(size_t)i + 1] = b0[(size_t)i + 1] + a0[(size_t)i + 1];

  auto work = [&](int t, int64_t lo, int64_t hi) -> int {
    std::vector<int32_t> lists;  // [na, max_nb]
    std::vector<uint8_t> cnt;
    std::vector<int32_t>& acol = a_cols[(size_t)t];
    std::vector<int32_t>& bcol = b_cols[(size_t)t];
    if (cs) {
      acol.reserve((size_t)(2 * (b0[(size_t)hi] - b0[(size_t)lo]) + 16));
      bcol.reserve((size_t)(4 * (b0[(size_t)hi] - b0[(size_t)lo]) + 16));
    }
    int64_t ta = 0, tbn = 0;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t g = sel(ids, i);
      const int64_t na = s->atom_ptr[g + 1] - s->atom_ptr[g], nb = s->bond_ptr[g + 1] - s->bond_ptr[g];
      const int64_t A0 = a0[(size_t)i], K0 = 2 * b0[(size_t)i];  // first atom, first directed bond (0-based)
      const uint8_t* af = s->atom_feats + (size_t)s->atom_ptr[g] * fa;
      const uint8_t* bf = s->bond_feats + (size_t)s->bond_ptr[g] * fb;
      const int32_t* bd = s->bonds + 2 * s->bond_ptr[g];
      const int32_t* nid = mess ? s->atom_nid + s->atom_ptr[g] : nullptr;
      const int64_t nbase = mess ? mess->node_base[i] : 0;
      lists.assign((size_t)na * max_nb, 0);
      cnt.assign((size_t)na, 0);
      auto push = [&](int32_t a, int64_t v) -> bool {
        if (cnt[(size_t)a] >= max_nb) return false;  // the reference's agraph[a, i] = b raises IndexError here
        lists[(size_t)a * max_nb + cnt[(size_t)a]++] = (int32_t)v;
        return true;
      };
      for (int64_t k = 0; k < nb; ++k) {
        const int32_t x = bd[2 * k], y = bd[2 * k + 1];
        if (x < 0 || x >= na || y < 0 || y >= na) return JTVAE_E_RANGE;
        const int64_t b = first_id + K0 + 2 * k;
        if (!push(y, b) || !push(x, b + 1)) return JTVAE_E_RANGE;
        if (nid) {
          const int32_t xn = nid[x], yn = nid[y];
          if (xn > 0 && yn > 0) {
            const int64_t xb = nbase + xn - 1, yb = nbase + yn - 1;
            if (xb >= mess->n_nodes || yb >= mess->n_nodes) return JTVAE_E_RANGE;
            if (xb != yb) {
              const int32_t m_xy = out.find(xb, yb), m_yx = out.find(yb, xb);
              if (m_xy && !push(y, m_xy)) return JTVAE_E_RANGE;
              if (m_yx && !push(x, m_yx)) return JTVAE_E_RANGE;
            }
          }
        }
      }
      // feature rows
      if (tb && tb->fatoms) {
        float* d = tb->fatoms + (size_t)A0 * fa;
        for (int64_t e = 0; e < na * fa; ++e) d[e] = (float)af[e];
      }
      if (cs && cs->fatoms) std::memcpy(cs->fatoms + (size_t)A0 * fa, af, (size_t)(na * fa));
      if (tb && tb->fbonds) {
        for (int64_t k = 0; k < nb; ++k)
          for (int side = 0; side < 2; ++side) {
            float* d = tb->fbonds + (size_t)(pad + K0 + 2 * k + side) * fw;
            const uint8_t* a = af + (size_t)bd[2 * k + side] * fa;
            for (int32_t c = 0; c < fa; ++c) d[c] = (float)a[c];
            for (int32_t c = 0; c < fb; ++c) d[fa + c] = (float)bf[(size_t)k * fb + c];
          }
      }
      if (cs && cs->fbonds) {
        for (int64_t k = 0; k < nb; ++k)
          for (int side = 0; side < 2; ++side) {
            uint8_t* d = cs->fbonds + (size_t)(pad + K0 + 2 * k + side) * fw;
            std::memcpy(d, af + (size_t)bd[2 * k + side] * fa, (size_t)fa);
            std::memcpy(d + fa, bf + (size_t)k * fb, (size_t)fb);
          }
      }
      // atom tables
      if (tb && tb->scope) {
        tb->scope[2 * i] = A0;
        tb->scope[2 * i + 1] = na;
      }
      if (cs && cs->seg) {
        cs->seg[2 * i] = (int32_t)A0;
        cs->seg[2 * i + 1] = (int32_t)na;
      }
      if (cs && cs->seg_of)
        for (int64_t a = 0; a < na; ++a) cs->seg_of[A0 + a] = (int32_t)i;
      if (tb && tb->agraph) {
        int64_t* d = tb->agraph + (size_t)A0 * max_nb;
        for (int64_t e = 0; e < na * max_nb; ++e) d[e] = lists[(size_t)e];
      }
      if (cs) {
        for (int64_t a = 0; a < na; ++a) {
          const int c = cnt[(size_t)a];
          cs->a_rowptr[A0 + a + 1] = c;  // counts for now; summed after the ranges are done
          for (int e = 0; e < c; ++e) {
            const int32_t v = lists[(size_t)a * max_nb + e];
            acol.push_back(v);
            ta += v < n_tree;
          }
        }
      }
      // bond tables: row (x -> y) keeps the slots of in_bonds[x] that are tree messages or do not start at y
      for (int64_t k = 0; k < 2 * nb; ++k) {
        const int32_t x = bd[k], y = bd[k ^ 1];  // k even: (begin -> end); odd: (end -> begin)
        const int64_t row = pad + K0 + k;
        const int c = cnt[(size_t)x];
        const int32_t* lx = &lists[(size_t)x * max_nb];
        int64_t* trow = (tb && tb->bgraph) ? tb->bgraph + (size_t)row * max_nb : nullptr;
        int kept = 0;
        for (int e = 0; e < c; ++e) {
          const int32_t v = lx[e];
          bool keep = true;
          if (v >= first_id) {
            const int64_t kk = v - first_id - K0;  // directed bond of this graph
            keep = bd[kk] != y;
          }
          if (trow) trow[e] = keep ? v : 0;
          if (keep && cs) {
            bcol.push_back(v);
            tbn += v < n_tree;
            ++kept;
          }
        }
        if (trow)
          for (int e = c; e < max_nb; ++e) trow[e] = 0;
        if (cs) cs->b_rowptr[row + 1] = kept;
      }
    }
    tree_a[(size_t)t] = ta;
    tree_b[(size_t)t] = tbn;
    return 0;
  };
  rc = run_ranges(n, w, T, work, &bounds);
  if (rc) return rc;
  if (cs) {
    for (int64_t a = 0; a < NA; ++a) cs->a_rowptr[a + 1] += cs->a_rowptr[a];
    for (int64_t r = 0; r < rows; ++r) cs->b_rowptr[r + 1] += cs->b_rowptr[r];
    if (cs->a_rowptr[NA] > cs->a_cap || cs->b_rowptr[rows] > cs->b_cap) return JTVAE_E_WORKSPACE;
    const int nr = (int)bounds.size() - 1;
    auto place = [&](int t) {
      const int64_t lo = bounds[(size_t)t];
      if (!a_cols[(size_t)t].empty())
        std::memcpy(cs->a_col + cs->a_rowptr[a0[(size_t)lo]], a_cols[(size_t)t].data(),
                    sizeof(int32_t) * a_cols[(size_t)t].size());
      if (!b_cols[(size_t)t].empty())
        std::memcpy(cs->b_col + cs->b_rowptr[pad + 2 * b0[(size_t)lo]], b_cols[(size_t)t].data(),
                    sizeof(int32_t) * b_cols[(size_t)t].size());
    };
    for (int t = 0; t < nr; ++t) place(t);
    cs->nnz[0] = cs->a_rowptr[NA];
    cs->nnz[1] = cs->b_rowptr[rows];
    cs->nnz[2] = cs->nnz[3] = 0;
    for (int t = 0; t < T; ++t) {
      cs->nnz[2] += tree_a[(size_t)t];
      cs->nnz[3] += tree_b[(size_t)t];
    }
    const int64_t n_src = n_tree + rows;
    if (cs->at_rowptr) {
      rc = jtvae_prep_transpose_csr(cs->a_rowptr, cs->a_col, NA, n_src, cs->at_rowptr, cs->at_col);
      if (rc) return rc;
    }
    if (cs->bt_rowptr) {
      rc = jtvae_prep_transpose_csr(cs->b_rowptr, cs->b_col, rows, n_src, cs->bt_rowptr, cs->bt_col);
      if (rc) return rc;
    }
  }
  return 0;
}

// ---- junction trees ----------------------------------------------------------------------------------

static int check_tree_set(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n) {
  if (!s) return JTVAE_E_NULL;
  if (n < 0 || s->n_trees < 0) return JTVAE_E_SHAPE;
  if (n > 0 && (!s->node_ptr || !s->nbr_ptr || !s->wid)) return JTVAE_E_NULL;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    if (g < 0 || g >= s->n_trees) return JTVAE_E_RANGE;
    if (s->node_ptr[g + 1] < s->node_ptr[g]) return JTVAE_E_SHAPE;
  }
  return 0;
}

namespace {

// Messages of one tree in local numbering: message e (0-based, node by node, neighbour by neighbour) runs
// x -> y; in_ptr / in_list: messages arriving at every node in increasing id.
struct LocalTree {
  std::vector<int32_t> mx, my, in_ptr, in_list, cur;
  int build(const jtvae_tree_set_t* s, int64_t g) {
    const int64_t n0 = s->node_ptr[g], k = s->node_ptr[g + 1] - n0;
    const int64_t e0 = s->nbr_ptr[n0], ne = s->nbr_ptr[n0 + k] - e0;
    if (ne < 0) return JTVAE_E_SHAPE;
    if (ne > 0 && !s->nbr) return JTVAE_E_NULL;
    mx.resize((size_t)ne);
    my.resize((size_t)ne);
    in_ptr.assign((size_t)k + 1, 0);
    for (int64_t x = 0; x < k; ++x) {
      if (s->nbr_ptr[n0 + x + 1] < s->nbr_ptr[n0 + x]) return JTVAE_E_SHAPE;
      for (int64_t e = s->nbr_ptr[n0 + x]; e < s->nbr_ptr[n0 + x + 1]; ++e) {
        const int32_t y = s->nbr[e];
        if (y < 0 || y >= k) return JTVAE_E_RANGE;
        mx[(size_t)(e - e0)] = (int32_t)x;
        my[(size_t)(e - e0)] = y;
        ++in_ptr[(size_t)y + 1];
      }
    }
    for (int64_t y = 0; y < k; ++y) in_ptr[(size_t)y + 1] += in_ptr[(size_t)y];
    cur.assign(in_ptr.begin(), in_ptr.end() - 1);
    in_list.resize((size_t)ne);
    for (int64_t e = 0; e < ne; ++e) in_list[(size_t)cur[(size_t)my[(size_t)e]]++] = (int32_t)e;
    return 0;
  }
};

}  // namespace

int jtvae_tensorize_tree_counts(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n, int32_t want_widths,
                                int32_t n_threads, int64_t* counts) {
  if (!counts) return JTVAE_E_NULL;
  int rc = check_tree_set(s, ids, n);
  if (rc) return rc;
  const int T = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  struct Part { int64_t nodes = 0, mess = 0, kn = 1, km = 1, m_cap = 0; };
  std::vector<Part> parts((size_t)T);
  auto work = [&](int t, int64_t lo, int64_t hi) -> int {
    Part& p = parts[(size_t)t];
    LocalTree lt;
    std::vector<int32_t> in_cnt;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t g = sel(ids, i);
      const int64_t n0 = s->node_ptr[g], k = s->node_ptr[g + 1] - n0;
      if (!want_widths) {
        // row (y -> z) holds at most the messages arriving at y: sum over nodes of in(y) * out(y) bounds m
        const int64_t e0 = s->nbr_ptr[n0], ne = s->nbr_ptr[n0 + k] - e0;
        if (ne < 0) return JTVAE_E_SHAPE;
        if (ne > 0 && !s->nbr) return JTVAE_E_NULL;
        in_cnt.assign((size_t)k, 0);
        for (int64_t e = 0; e < ne; ++e) {
          const int32_t y = s->nbr[e0 + e];
          if (y < 0 || y >= k) return JTVAE_E_RANGE;
          ++in_cnt[(size_t)y];
        }
        for (int64_t y = 0; y < k; ++y) {
          const int64_t out = s->nbr_ptr[n0 + y + 1] - s->nbr_ptr[n0 + y];
          // with the reverse edge present (a tree stored with both directions) each row drops one entry
          p.m_cap += (int64_t)in_cnt[(size_t)y] * out;
        }
        p.nodes += k;
        p.mess += ne;
        continue;
      }
      int r = lt.build(s, g);
      if (r) return r;
      const int64_t ne = (int64_t)lt.mx.size();
      p.nodes += k;
      p.mess += ne;
      for (int64_t y = 0; y < k; ++y) p.kn = std::max<int64_t>(p.kn, lt.in_ptr[(size_t)y + 1] - lt.in_ptr[(size_t)y]);
      for (int64_t e = 0; e < ne; ++e) {  // (y -> z) reads the messages arriving at y that do not come from z
        const int32_t y = lt.mx[(size_t)e], z = lt.my[(size_t)e];
        int64_t len = 0;
        for (int32_t q = lt.in_ptr[(size_t)y]; q < lt.in_ptr[(size_t)y + 1]; ++q) len += lt.mx[(size_t)lt.in_list[(size_t)q]] != z;
        p.km = std::max(p.km, len);
        p.m_cap += len;
      }
    }
    return 0;
  };
  rc = run_ranges(n, std::vector<int64_t>(), T, work);
  if (rc) return rc;
  Part a;
  for (const Part& p : parts) {
    a.nodes += p.nodes;
    a.mess += p.mess;
    a.kn = std::max(a.kn, p.kn);
    a.km = std::max(a.km, p.km);
    a.m_cap += p.m_cap;
  }
  if (a.mess + 1 > INT32_MAX || a.nodes > INT32_MAX) return JTVAE_E_RANGE;
  counts[0] = a.nodes;
  counts[1] = a.mess + 1;
  counts[2] = want_widths ? a.kn : 0;
  counts[3] = want_widths ? a.km : 0;
  counts[4] = a.m_cap;
  return 0;
}

int jtvae_tensorize_trees(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n, jtvae_tree_tables_t* tb,
                          jtvae_tree_csr_t* cs, int32_t n_threads) {
  int rc = check_tree_set(s, ids, n);
  if (rc) return rc;
  if (tb && (tb->Kn < 1 || tb->Km < 1)) return JTVAE_E_SHAPE;
  if (cs && (!cs->n_rowptr || !cs->m_rowptr || !cs->src || !cs->dst || !cs->wid || (cs->m_cap > 0 && !cs->m_col)))
    return JTVAE_E_NULL;
  std::vector<int64_t> n0((size_t)n + 1, 0), m0((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    const int64_t a = s->node_ptr[g], b = s->node_ptr[g + 1];
    n0[(size_t)i + 1] = n0[(size_t)i] + (b - a);
    m0[(size_t)i + 1] = m0[(size_t)i] + (s->nbr_ptr[b] - s->nbr_ptr[a]);
  }
  const int64_t NN = n0[(size_t)n], NM = 1 + m0[(size_t)n];
  if (NM > INT32_MAX || NN > INT32_MAX) return JTVAE_E_RANGE;
  if (tb) {
    if (tb->fmess) tb->fmess[0] = 0;
    if (tb->mess_graph) std::memset(tb->mess_graph, 0, sizeof(int64_t) * (size_t)tb->Km);
  }
  if (cs) {
    cs->src[0] = cs->dst[0] = 0;
    cs->n_rowptr[0] = 0;
    cs->m_rowptr[0] = cs->m_rowptr[1] = 0;
    if (cs->sf) cs->sf[0] = 1;
  }
  const int T = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  std::vector<std::vector<int32_t>> m_cols((size_t)T);
  std::vector<int32_t> max_sf((size_t)T, 0);
  std::vector<int64_t> bounds;
  std::vector<int64_t> w((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) w[(size_t)i + 1] = m0[(size_t)i + 1] + n0[(size_t)i + 1];

  auto work = [&](int t, int64_t lo, int64_t hi) -> int {
    LocalTree lt;
    std::vector<int32_t>& mcol = m_cols[(size_t)t];
    std::vector<int32_t> row_ptr, row_col, pending, ready, sfl;
    int32_t msf = 0;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t g = sel(ids, i);
      int r = lt.build(s, g);
      if (r) return r;
      const int64_t N0 = n0[(size_t)i], M0 = 1 + m0[(size_t)i];  // first node, first message id
      const int64_t k = s->node_ptr[g + 1] - s->node_ptr[g], ne = (int64_t)lt.mx.size();
      const int32_t* wid = s->wid + s->node_ptr[g];
      if (tb && tb->scope) {
        tb->scope[2 * i] = N0;
        tb->scope[2 * i + 1] = k;
      }
      if (cs && cs->roots) cs->roots[i] = (int32_t)N0;
      for (int64_t x = 0; x < k; ++x) {
        if (tb && tb->fnode) tb->fnode[N0 + x] = wid[x];
        if (cs) cs->wid[N0 + x] = wid[x];
      }
      for (int64_t e = 0; e < ne; ++e) {
        if (tb && tb->fmess) tb->fmess[M0 + e] = N0 + lt.mx[(size_t)e];
        if (cs) {
          cs->src[M0 + e] = (int32_t)(N0 + lt.mx[(size_t)e]);
          cs->dst[M0 + e] = (int32_t)(N0 + lt.my[(size_t)e]);
        }
      }
      for (int64_t y = 0; y < k; ++y) {
        const int32_t a = lt.in_ptr[(size_t)y], b = lt.in_ptr[(size_t)y + 1];
        if (tb && tb->node_graph) {
          if (b - a > tb->Kn) return JTVAE_E_SHAPE;
          int64_t* d = tb->node_graph + (size_t)(N0 + y) * tb->Kn;
          for (int32_t q = a; q < b; ++q) d[q - a] = M0 + lt.in_list[(size_t)q];
          for (int32_t q = b - a; q < tb->Kn; ++q) d[q] = 0;
        }
        if (cs) {
          cs->n_rowptr[N0 + y + 1] = (int32_t)(M0 - 1 + b);  // node lists are back to back in message order
          for (int32_t q = a; q < b; ++q) cs->n_col[M0 - 1 + q] = (int32_t)(M0 + lt.in_list[(size_t)q]);
        }
      }
      // mess_graph rows, local CSR first (the settling levels need it too)
      row_ptr.assign((size_t)ne + 1, 0);
      row_col.clear();
      for (int64_t e = 0; e < ne; ++e) {
        const int32_t y = lt.mx[(size_t)e], z = lt.my[(size_t)e];
        for (int32_t q = lt.in_ptr[(size_t)y]; q < lt.in_ptr[(size_t)y + 1]; ++q) {
          const int32_t m1 = lt.in_list[(size_t)q];
          if (lt.mx[(size_t)m1] != z) row_col.push_back(m1);
        }
        row_ptr[(size_t)e + 1] = (int32_t)row_col.size();
      }
      for (int64_t e = 0; e < ne; ++e) {
        const int32_t a = row_ptr[(size_t)e], b = row_ptr[(size_t)e + 1];
        if (tb && tb->mess_graph) {
          if (b - a > tb->Km) return JTVAE_E_SHAPE;
          int64_t* d = tb->mess_graph + (size_t)(M0 + e) * tb->Km;
          for (int32_t q = a; q < b; ++q) d[q - a] = M0 + row_col[(size_t)q];
          for (int32_t q = b - a; q < tb->Km; ++q) d[q] = 0;
        }
        if (cs) {
          cs->m_rowptr[M0 + e + 1] = b - a;  // counts for now
          for (int32_t q = a; q < b; ++q) mcol.push_back((int32_t)(M0 + row_col[(size_t)q]));
        }
      }
      if (cs && cs->sf) {
        // sf = 1 without inputs, else 1 + max over the inputs; messages on or behind a cycle never become
        // ready and keep 1 << 30
        pending.resize((size_t)ne);
        sfl.assign((size_t)ne, 1 << 30);
        ready.clear();
        for (int64_t e = 0; e < ne; ++e) {
          pending[(size_t)e] = row_ptr[(size_t)e + 1] - row_ptr[(size_t)e];
          if (!pending[(size_t)e]) {
            sfl[(size_t)e] = 1;
            ready.push_back((int32_t)e);
          }
        }
        for (size_t head = 0; head < ready.size(); ++head) {
          const int32_t j = ready[head];  // (x -> y) is read by the messages leaving y, except the one back to x
          const int32_t x = lt.mx[(size_t)j], y = lt.my[(size_t)j];
          const int64_t o0 = s->nbr_ptr[s->node_ptr[g] + y] - s->nbr_ptr[s->node_ptr[g]];
          const int64_t o1 = s->nbr_ptr[s->node_ptr[g] + y + 1] - s->nbr_ptr[s->node_ptr[g]];
          for (int64_t m = o0; m < o1; ++m) {
            if (lt.my[(size_t)m] == x) continue;
            if (--pending[(size_t)m] == 0) {
              int32_t mx_ = 0;
              for (int32_t q = row_ptr[(size_t)m]; q < row_ptr[(size_t)m + 1]; ++q)
                mx_ = std::max(mx_, sfl[(size_t)row_col[(size_t)q]]);
              sfl[(size_t)m] = mx_ + 1;
              ready.push_back((int32_t)m);
            }
          }
        }
        for (int64_t e = 0; e < ne; ++e) {
          cs->sf[M0 + e] = sfl[(size_t)e];
          msf = std::max(msf, sfl[(size_t)e]);
        }
      }
    }
    max_sf[(size_t)t] = msf;
    return 0;
  };
  rc = run_ranges(n, w, T, work, &bounds);
  if (rc) return rc;
  if (cs) {
    for (int64_t m = 1; m < NM; ++m) cs->m_rowptr[m + 1] += cs->m_rowptr[m];
    if (cs->m_rowptr[NM] > cs->m_cap) return JTVAE_E_WORKSPACE;
    for (int t = 0; t + 1 < (int)bounds.size(); ++t)
      if (!m_cols[(size_t)t].empty())
        std::memcpy(cs->m_col + cs->m_rowptr[1 + m0[(size_t)bounds[(size_t)t]]], m_cols[(size_t)t].data(),
                    sizeof(int32_t) * m_cols[(size_t)t].size());
    cs->nnz = cs->m_rowptr[NM];
    cs->max_sf = 0;
    for (int t = 0; t < T; ++t) cs->max_sf = std::max(cs->max_sf, max_sf[(size_t)t]);
  }
  return 0;
}

// ---- candidate selection -----------------------------------------------------------------------------

static int walk_candidates(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n, int64_t* cand_ids,
                           int32_t* batch_idx, int64_t* node_base, int32_t* seg_len, int32_t* label_pos,
                           int64_t* counts) {
  int rc = check_tree_set(s, ids, n);
  if (rc) return rc;
  if (n > 0 && !s->cand_ptr) return JTVAE_E_NULL;
  int64_t nc = 0, ns = 0, base = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = sel(ids, i);
    for (int64_t x = s->node_ptr[g]; x < s->node_ptr[g + 1]; ++x) {
      const int64_t c0 = s->cand_ptr[x], c1 = s->cand_ptr[x + 1], k = c1 - c0;
      if (k < 0) return JTVAE_E_SHAPE;
      const bool leaf = s->nbr_ptr[x + 1] - s->nbr_ptr[x] == 1;  // mol_tree.py: is_leaf = one neighbour
      if (leaf || k == 1) continue;                              // datautils.py:155
      if (k > 1) {                                               // jtnn_vae.py:119-123
        if (seg_len) {
          if (!s->label || s->label[x] < 0 || s->label[x] >= k) return JTVAE_E_RANGE;
          seg_len[ns] = (int32_t)k;
          label_pos[ns] = (int32_t)(nc + s->label[x]);
        }
        ++ns;
      }
      for (int64_t c = c0; c < c1; ++c, ++nc)
        if (cand_ids) {
          cand_ids[nc] = c;
          batch_idx[nc] = (int32_t)i;
          node_base[nc] = base;
        }
    }
    base += s->node_ptr[g + 1] - s->node_ptr[g];
  }
  if (nc > INT32_MAX) return JTVAE_E_RANGE;
  if (counts) {
    counts[0] = nc;
    counts[1] = ns;
  }
  return 0;
}

int jtvae_select_candidates_counts(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n, int64_t* counts) {
  if (!counts) return JTVAE_E_NULL;
  return walk_candidates(s, ids, n, nullptr, nullptr, nullptr, nullptr, nullptr, counts);
}

int jtvae_select_candidates(const jtvae_tree_set_t* s, const int64_t* ids, int64_t n, int64_t* cand_ids,
                            int32_t* batch_idx, int64_t* node_base, int32_t* seg_len, int32_t* label_pos) {
  if (!cand_ids || !batch_idx || !node_base || !seg_len || !label_pos) {
    int64_t c[2];
    int rc = walk_candidates(s, ids, n, nullptr, nullptr, nullptr, nullptr, nullptr, c);
    if (rc) return rc;
    if ((c[0] > 0 && (!cand_ids || !batch_idx || !node_base)) || (c[1] > 0 && (!seg_len || !label_pos)))
      return JTVAE_E_NULL;
    if (c[0] == 0 && c[1] == 0) return 0;
  }
  return walk_candidates(s, ids, n, cand_ids, batch_idx, node_base, seg_len, label_pos, nullptr);
}

}  // extern "C"
