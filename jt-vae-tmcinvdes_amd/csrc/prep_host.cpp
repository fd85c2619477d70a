// Host-side index preparation: the reference's padded int64 neighbour tables -> int32 CSR, its
// transpose, and the iteration from which each tree message stops changing.  Plain C++ over host
// memory (no GPU work): single counting passes where a numpy formulation needs sorts and scatter
// reductions.  Orders are the ones the kernels rely on: CSR entries keep the slot order of the
// table; a transposed list holds its readers in increasing (row, slot) order.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/jtvae_hip.h"

extern "C" {

int jtvae_prep_padded_to_csr(const int64_t* table, int64_t rows, int32_t K, int64_t n_src, int32_t* rowptr,
                             int32_t* col, int64_t col_capacity, int64_t* nnz_out) {
  if (rows < 0 || K < 0 || n_src < 0 || col_capacity < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !nnz_out || (rows > 0 && K > 0 && (!table || !col))) return JTVAE_E_NULL;
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t* t = table + r * K;
    for (int32_t k = 0; k < K; ++k) {
      const int64_t v = t[k];
      if (v == 0) continue;  // 0 = "no neighbour" (fast_jtnn/mpn.py:148-159, jtnn_enc.py:78-90)
      if (v < 0 || v >= n_src) return JTVAE_E_RANGE;
      if (nnz >= col_capacity) return JTVAE_E_WORKSPACE;
      col[nnz++] = (int32_t)v;
    }
    if (nnz > INT32_MAX) return JTVAE_E_RANGE;
    rowptr[r + 1] = (int32_t)nnz;
  }
  *nnz_out = nnz;
  return 0;
}

int jtvae_prep_transpose_csr(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int64_t n_src,
                             int32_t* trowptr, int32_t* tcol) {
  if (n_rows < 0 || n_src < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !trowptr) return JTVAE_E_NULL;
  const int64_t nnz = rowptr[n_rows];
  if (nnz > 0 && (!col || !tcol)) return JTVAE_E_NULL;
  std::memset(trowptr, 0, sizeof(int32_t) * (size_t)(n_src + 1));
  for (int64_t e = 0; e < nnz; ++e) {
    const int32_t j = col[e];
    if (j < 0 || j >= n_src) return JTVAE_E_RANGE;
    ++trowptr[j + 1];
  }
  for (int64_t j = 0; j < n_src; ++j) trowptr[j + 1] += trowptr[j];
  std::vector<int32_t> cursor(trowptr, trowptr + n_src);
  for (int64_t r = 0; r < n_rows; ++r)
    for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) tcol[cursor[col[e]]++] = (int32_t)r;
  return 0;
}

// sf[m] = 1 when row m reads nothing, else 1 + max over the rows it reads, capped at max(T, 1):
// the iteration from which message m keeps its value (fast_jtnn/jtnn_enc.py:107-128 recomputes all
// rows in all T iterations).  One pass in dependency order (rows become ready when all rows they read
// are done); rows on or behind a cycle never become ready and sit at the cap, which is where T rounds
// of the recurrence leave them.
int jtvae_prep_stable_from(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t T, int32_t* sf) {
  if (n < 0 || T < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !sf) return JTVAE_E_NULL;
  const int32_t cap = T > 1 ? T : 1;
  const int64_t nnz = n > 0 ? rowptr[n] : 0;
  if (nnz > 0 && !col) return JTVAE_E_NULL;
  // readers of every row (transposed lists), to release them when a row is done
  std::vector<int32_t> tptr((size_t)n + 1, 0), tcol((size_t)nnz), pending((size_t)n), ready;
  for (int64_t e = 0; e < nnz; ++e) {
    if (col[e] < 0 || col[e] >= n) return JTVAE_E_RANGE;
    ++tptr[(size_t)col[e] + 1];
  }
  for (int64_t j = 0; j < n; ++j) tptr[(size_t)j + 1] += tptr[(size_t)j];
  {
    std::vector<int32_t> cur(tptr.begin(), tptr.end() - 1);
    for (int64_t m = 0; m < n; ++m)
      for (int32_t e = rowptr[m]; e < rowptr[m + 1]; ++e) tcol[(size_t)cur[(size_t)col[e]]++] = (int32_t)m;
  }
  ready.reserve((size_t)n);
  for (int64_t m = 0; m < n; ++m) {
    pending[(size_t)m] = rowptr[m + 1] - rowptr[m];
    sf[m] = cap;  // stays for rows that never become ready
    if (pending[(size_t)m] == 0) {
      sf[m] = 1;
      ready.push_back((int32_t)m);
    }
  }
  for (size_t head = 0; head < ready.size(); ++head) {
    const int32_t j = ready[head];
    for (int32_t e = tptr[(size_t)j]; e < tptr[(size_t)j + 1]; ++e) {
      const int32_t m = tcol[(size_t)e];
      if (--pending[(size_t)m] == 0) {
        int32_t mx = 0;
        for (int32_t k = rowptr[m]; k < rowptr[m + 1]; ++k) mx = sf[col[k]] > mx ? sf[col[k]] : mx;
        sf[m] = mx + 1 < cap ? mx + 1 : cap;
        ready.push_back(m);
      }
    }
  }
  return 0;
}

// Atom and bond feature rows are one-hot blocks (fast_jtnn/mpn.py:18-31): every value is a small
// whole number, so the rows can cross the host-to-device link as bytes, a quarter of the float size,
// and be widened again on the device without changing a bit.  One pass converts and checks; *exact
// comes back 0 as soon as a value is not a whole number in [0, 255] (out is then incomplete and the
// caller ships the floats).
static bool features_to_u8_range(const float* x, int64_t b0, int64_t b1, uint8_t* out) {
  constexpr int64_t BLOCK = 4096;  // mismatches are looked for per block, so the inner loop stays branch-free
  for (int64_t b = b0; b < b1; b += BLOCK) {
    const int64_t e = b + BLOCK < b1 ? b + BLOCK : b1;
    int bad = 0;
    for (int64_t i = b; i < e; ++i) {
      const float v = x[i];
      const uint8_t u = (uint8_t)(int32_t)(v >= 0.f && v <= 255.f ? v : 0.f);
      const float w = (float)u;
      uint32_t bv, bw;
      std::memcpy(&bv, &v, 4);
      std::memcpy(&bw, &w, 4);
      out[i] = u;
      bad |= bv != bw;  // NaN, -0, fractions, negatives and values above 255 all land here
    }
    if (bad) return false;
  }
  return true;
}

int jtvae_prep_features_to_u8(const float* x, int64_t n, uint8_t* out, int32_t* exact) {
  if (n < 0) return JTVAE_E_SHAPE;
  if (!exact || (n > 0 && (!x || !out))) return JTVAE_E_NULL;
  // The pass streams 5 bytes per value and one core cannot saturate the memory system: large arrays (the bond
  // rows of a few thousand molecules are ~10^7 values) are cut into up to 4 ranges converted side by side.
  constexpr int64_t PER_THREAD = 1 << 21;
  const int n_threads = (int)(n / PER_THREAD < 1 ? 1 : (n / PER_THREAD > 4 ? 4 : n / PER_THREAD));
  if (n_threads == 1) {
    *exact = features_to_u8_range(x, 0, n, out) ? 1 : 0;
    return 0;
  }
  std::vector<std::thread> pool;
  std::vector<char> ok((size_t)n_threads, 0);
  const int64_t step = ((n + n_threads - 1) / n_threads + 4095) & ~(int64_t)4095;
  for (int t = 1; t < n_threads; ++t) {
    const int64_t b0 = t * step, b1 = b0 + step < n ? b0 + step : n;
    pool.emplace_back([=, &ok] { ok[(size_t)t] = features_to_u8_range(x, b0 < n ? b0 : n, b1, out); });
  }
  ok[0] = features_to_u8_range(x, 0, step < n ? step : n, out);
  bool all = ok[0];
  for (int t = 1; t < n_threads; ++t) {
    pool[(size_t)t - 1].join();
    all = all && ok[(size_t)t];
  }
  *exact = all ? 1 : 0;
  return 0;
}

// Stable grouping of items by key: order[] lists the items of key 0 in input order, then key 1 ...;
// keyptr[k] .. keyptr[k+1] delimit the group of key k.
int jtvae_prep_group_by_key(const int32_t* keys, int64_t n, int64_t n_keys, int32_t* keyptr, int32_t* order) {
  if (n < 0 || n_keys < 0) return JTVAE_E_SHAPE;
  if (!keyptr || (n > 0 && (!keys || !order))) return JTVAE_E_NULL;
  std::memset(keyptr, 0, sizeof(int32_t) * (size_t)(n_keys + 1));
  for (int64_t i = 0; i < n; ++i) {
    if (keys[i] < 0 || keys[i] >= n_keys) return JTVAE_E_RANGE;
    ++keyptr[keys[i] + 1];
  }
  for (int64_t k = 0; k < n_keys; ++k) keyptr[k + 1] += keyptr[k];
  std::vector<int32_t> cursor(keyptr, keyptr + n_keys);
  for (int64_t i = 0; i < n; ++i) order[cursor[keys[i]]++] = (int32_t)i;
  return 0;
}

// Index form of the teacher-forced decoder traces of a batch (fast_jtnn/jtnn_dec.py:59-167, dfs :299-305),
// from the junction-tree tables alone.  Directed tree edge m runs src[m] -> dst[m] (ids 1 .. n_mess-1 as
// JTNNEncoder.tensorize numbers them: node by node, neighbours in order; 0 is the padding message).  Every
// tree is walked depth first from its root; step t of tree i creates the hidden row of one edge (x -> y)
// from the rows that reached x before it, minus the one that came from y.  Rows of the step tables are
// ordered step by step, trees in batch order inside a step - the order in which the reference concatenates
// them - and the final stop decision at every root comes last.
int jtvae_prep_decoder_plan(const int32_t* src, const int32_t* dst, int64_t n_mess, const int32_t* wid,
                            int64_t n_nodes, const int32_t* roots, int64_t n_trees, int32_t* dep_rp,
                            int32_t* dep_col, int64_t dep_cap, int32_t* stop_rp, int32_t* stop_col,
                            int64_t stop_cap, int32_t* stop_word, int32_t* stop_ctx, float* stop_target,
                            int32_t* pred_mess, int32_t* pred_ctx, int32_t* pred_target, int64_t* counts) {
  if (n_mess < 1 || n_nodes < 0 || n_trees < 0 || dep_cap < 0 || stop_cap < 0) return JTVAE_E_SHAPE;
  if (!src || !dst || !wid || !roots || !dep_rp || !stop_rp || !stop_word || !stop_ctx || !stop_target ||
      !pred_mess || !pred_ctx || !pred_target || !counts)
    return JTVAE_E_NULL;
  // outgoing edges of every node, in id order (= neighbour order)
  std::vector<int32_t> out_ptr((size_t)n_nodes + 1, 0), out_edge((size_t)n_mess);
  for (int64_t m = 1; m < n_mess; ++m) {
    if (src[m] < 0 || src[m] >= n_nodes || dst[m] < 0 || dst[m] >= n_nodes) return JTVAE_E_RANGE;
    ++out_ptr[(size_t)src[m] + 1];
  }
  for (int64_t x = 0; x < n_nodes; ++x) out_ptr[(size_t)x + 1] += out_ptr[(size_t)x];
  {
    std::vector<int32_t> cur(out_ptr.begin(), out_ptr.end() - 1);
    for (int64_t m = 1; m < n_mess; ++m) out_edge[(size_t)cur[(size_t)src[m]]++] = (int32_t)m;
  }
  struct Step { int32_t t, tree, m, x, y, d; };
  std::vector<Step> steps;
  steps.reserve((size_t)n_mess);
  std::vector<std::vector<int32_t>> reached((size_t)n_nodes);  // edges that have arrived at a node, in order
  std::vector<int32_t> dep_off((size_t)n_mess, 0), dep_len((size_t)n_mess, 0), dep_tmp, stop_off, stop_len, stop_tmp;
  std::vector<char> visited((size_t)n_nodes, 0);
  struct Frame { int32_t x, parent, next; };
  std::vector<Frame> stack;
  int32_t max_t = 0;
  auto emit = [&](int32_t tree, int32_t t, int32_t m, int32_t x, int32_t y, int32_t d) {
    const std::vector<int32_t>& in = reached[(size_t)x];
    stop_off.push_back((int32_t)stop_tmp.size());
    stop_len.push_back((int32_t)in.size());
    dep_off[(size_t)m] = (int32_t)dep_tmp.size();
    for (int32_t e : in) {
      stop_tmp.push_back(e);
      if (src[e] != y) dep_tmp.push_back(e);
    }
    dep_len[(size_t)m] = (int32_t)dep_tmp.size() - dep_off[(size_t)m];
    steps.push_back(Step{t, tree, m, x, y, d});
    reached[(size_t)y].push_back(m);
  };
  for (int64_t i = 0; i < n_trees; ++i) {
    const int32_t root = roots[i];
    if (root < 0 || root >= n_nodes) return JTVAE_E_RANGE;
    int32_t t = 0;
    stack.clear();
    stack.push_back(Frame{root, -1, out_ptr[(size_t)root]});
    visited[(size_t)root] = 1;
    while (!stack.empty()) {
      Frame& f = stack.back();
      bool went_down = false;
      while (f.next < out_ptr[(size_t)f.x + 1]) {
        const int32_t m = out_edge[(size_t)f.next++];
        const int32_t y = dst[m];
        if (y == f.parent) continue;
        if (visited[(size_t)y]) return JTVAE_E_RANGE;  // not a tree
        visited[(size_t)y] = 1;
        emit((int32_t)i, t++, m, f.x, y, 1);
        const int32_t x = f.x;  // f dangles after the push
        stack.push_back(Frame{y, x, out_ptr[(size_t)y]});
        went_down = true;
        break;
      }
      if (went_down) continue;
      const Frame done = stack.back();
      stack.pop_back();
      if (!stack.empty()) {  // back to the parent: the edge (done.x -> parent)
        int32_t back = -1;
        for (int32_t e = out_ptr[(size_t)done.x]; e < out_ptr[(size_t)done.x + 1]; ++e)
          if (dst[out_edge[(size_t)e]] == done.parent) { back = out_edge[(size_t)e]; break; }
        if (back < 0) return JTVAE_E_RANGE;  // the reverse edge is missing
        emit((int32_t)i, t++, back, done.x, done.parent, 0);
      }
    }
    if (t > max_t) max_t = t;
  }
  const int64_t n_steps = (int64_t)steps.size();
  // dependencies by edge id
  int64_t nnz = 0;
  dep_rp[0] = 0;
  for (int64_t m = 0; m < n_mess; ++m) {
    if (nnz + dep_len[(size_t)m] > dep_cap) return JTVAE_E_WORKSPACE;
    for (int32_t k = 0; k < dep_len[(size_t)m]; ++k) dep_col[nnz++] = dep_tmp[(size_t)dep_off[(size_t)m] + k];
    dep_rp[m + 1] = (int32_t)nnz;
  }
  // step rows ordered by (t, tree): counting sort by t, stable in tree order
  std::vector<int64_t> t_ptr((size_t)max_t + 2, 0);
  for (const Step& s : steps) ++t_ptr[(size_t)s.t + 1];
  for (int32_t t = 0; t <= max_t; ++t) t_ptr[(size_t)t + 1] += t_ptr[(size_t)t];
  std::vector<int64_t> order((size_t)n_steps);
  for (int64_t k = 0; k < n_steps; ++k) order[(size_t)t_ptr[(size_t)steps[(size_t)k].t]++] = k;
  int64_t snz = 0, n_down = 0;
  stop_rp[0] = 0;
  for (int64_t i = 0; i < n_trees; ++i) {  // word prediction of the roots comes first
    pred_ctx[i] = (int32_t)i;
    pred_target[i] = wid[roots[i]];
  }
  for (int64_t r = 0; r < n_steps; ++r) {
    const int64_t k = order[(size_t)r];
    const Step& s = steps[(size_t)k];
    if (snz + stop_len[(size_t)k] > stop_cap) return JTVAE_E_WORKSPACE;
    for (int32_t j = 0; j < stop_len[(size_t)k]; ++j) stop_col[snz++] = stop_tmp[(size_t)stop_off[(size_t)k] + j];
    stop_rp[r + 1] = (int32_t)snz;
    stop_word[r] = wid[s.x];
    stop_ctx[r] = s.tree;
    stop_target[r] = (float)s.d;
    if (s.d == 1) {
      pred_mess[n_down] = s.m;
      pred_ctx[n_trees + n_down] = s.tree;
      pred_target[n_trees + n_down] = wid[s.y];
      ++n_down;
    }
  }
  for (int64_t i = 0; i < n_trees; ++i) {  // the last stop at every root
    const std::vector<int32_t>& in = reached[(size_t)roots[i]];
    if (snz + (int64_t)in.size() > stop_cap) return JTVAE_E_WORKSPACE;
    for (int32_t e : in) stop_col[snz++] = e;
    const int64_t r = n_steps + i;
    stop_rp[r + 1] = (int32_t)snz;
    stop_word[r] = wid[roots[i]];
    stop_ctx[r] = (int32_t)i;
    stop_target[r] = 0.f;
  }
  counts[0] = n_steps;
  counts[1] = n_down;
  counts[2] = nnz;
  counts[3] = snz;
  return 0;
}

// Backward plan of the compute-once tree-GRU schedule (each message visited once, fast_jtnn/jtnn_enc.py:107-128
// differentiated): which messages carry gradient, their reverse dependency level and the visiting order.
//   rowptr / col: the messages each message reads; seeds: messages with an outside gradient (n_seeds < 0: all).
//   act[m] = 1 when m is a seed or is read by a message with act = 1.
//   level of an active message = 0 when no active message reads it, else 1 + the largest level among its active
//   readers: its gradient is complete once every lower level has been visited.
//   order: the active messages by (level, rows that read something first, message id); lev_off[L] .. lev_off[L+1]
//   delimit level L.  Per level: n_dep = rows that read something, deg_sum = entries they read, tdeg_sum = active
//   readers of its rows.  counts = {active messages, levels}.
int jtvae_prep_backward_once(const int32_t* rowptr, const int32_t* col, int64_t n, const int32_t* seeds,
                             int64_t n_seeds, uint8_t* act, int32_t* order, int32_t* lev_off, int32_t* n_dep,
                             int64_t* deg_sum, int64_t* tdeg_sum, int64_t* counts) {
  if (n < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !act || !order || !lev_off || !n_dep || !deg_sum || !tdeg_sum || !counts) return JTVAE_E_NULL;
  const int64_t nnz = n > 0 ? rowptr[n] : 0;
  if (nnz > 0 && !col) return JTVAE_E_NULL;
  if (n_seeds > 0 && !seeds) return JTVAE_E_NULL;
  for (int64_t e = 0; e < nnz; ++e)
    if (col[e] < 0 || col[e] >= n) return JTVAE_E_RANGE;
  std::vector<int32_t> stack;
  if (n_seeds < 0) {
    std::memset(act, 1, (size_t)n);
  } else {
    std::memset(act, 0, (size_t)n);
    for (int64_t i = 0; i < n_seeds; ++i) {
      const int32_t m = seeds[i];
      if (m < 0 || m >= n) return JTVAE_E_RANGE;
      if (!act[m]) { act[m] = 1; stack.push_back(m); }
    }
    while (!stack.empty()) {
      const int32_t r = stack.back();
      stack.pop_back();
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e)
        if (!act[col[e]]) { act[col[e]] = 1; stack.push_back(col[e]); }
    }
  }
  // active readers still to be visited, per message; a message is released when that count reaches zero
  std::vector<int32_t> pending((size_t)n, 0), level((size_t)n, 0), ready;
  for (int64_t r = 0; r < n; ++r)
    if (act[r])
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) ++pending[(size_t)col[e]];
  std::vector<int32_t> tdeg(pending);  // active readers of every message
  ready.reserve((size_t)n);
  int64_t n_act = 0;
  for (int64_t m = 0; m < n; ++m)
    if (act[m]) {
      ++n_act;
      if (pending[(size_t)m] == 0) ready.push_back((int32_t)m);
    }
  int32_t max_level = -1;
  for (size_t head = 0; head < ready.size(); ++head) {
    const int32_t r = ready[head];
    if (level[(size_t)r] > max_level) max_level = level[(size_t)r];
    for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int32_t m = col[e];
      if (level[(size_t)m] < level[(size_t)r] + 1) level[(size_t)m] = level[(size_t)r] + 1;
      if (--pending[(size_t)m] == 0) ready.push_back(m);
    }
  }
  if ((int64_t)ready.size() != n_act) return JTVAE_E_RANGE;  // a dependency cycle: no compute-once order exists
  const int64_t n_lev = (int64_t)max_level + 1;
  // counting sort by (level, reads nothing), stable in message order
  std::vector<int64_t> start((size_t)(2 * n_lev + 1), 0);
  for (int64_t m = 0; m < n; ++m)
    if (act[m]) ++start[(size_t)(2 * level[(size_t)m] + (rowptr[m + 1] == rowptr[m] ? 1 : 0)) + 1];
  for (int64_t k = 0; k < 2 * n_lev; ++k) start[(size_t)k + 1] += start[(size_t)k];
  for (int64_t L = 0; L < n_lev; ++L) {
    lev_off[L] = (int32_t)start[(size_t)(2 * L)];
    n_dep[L] = (int32_t)(start[(size_t)(2 * L + 1)] - start[(size_t)(2 * L)]);
    deg_sum[L] = 0;
    tdeg_sum[L] = 0;
  }
  lev_off[n_lev] = (int32_t)n_act;
  for (int64_t m = 0; m < n; ++m)
    if (act[m]) {
      const int32_t L = level[(size_t)m];
      order[start[(size_t)(2 * L + (rowptr[m + 1] == rowptr[m] ? 1 : 0))]++] = (int32_t)m;
      deg_sum[L] += rowptr[m + 1] - rowptr[m];
      tdeg_sum[L] += tdeg[(size_t)m];
    }
  counts[0] = n_act;
  counts[1] = n_lev;
  return 0;
}

// Forward plan of the compute-once schedule: messages grouped by the iteration in which they settle.
int jtvae_prep_forward_once(const int32_t* rowptr, const int32_t* col, int64_t n, const int32_t* sf_true, int32_t T,
                            int32_t* sf, int32_t* order, int32_t* keyptr, int32_t* u_rows, int32_t* u_ptr,
                            int64_t* nnz_ptr) {
  if (n < 0 || T < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !keyptr || !u_ptr || !nnz_ptr || (n > 0 && (!sf_true || !sf || !order || !u_rows))) return JTVAE_E_NULL;
  const int32_t cap = T > 1 ? T : 1;
  const int64_t nnz = n > 0 ? rowptr[n] : 0;
  if (nnz > 0 && !col) return JTVAE_E_NULL;
  std::vector<uint8_t> read((size_t)n, 0);
  for (int64_t e = 0; e < nnz; ++e) {
    if (col[e] < 0 || col[e] >= n) return JTVAE_E_RANGE;
    read[(size_t)col[e]] = 1;
  }
  const int64_t nk = (int64_t)cap + 1;  // keys 0 .. cap
  std::memset(keyptr, 0, sizeof(int32_t) * (size_t)(nk + 1));
  for (int64_t m = 0; m < n; ++m) {
    if (sf_true[m] < 0) return JTVAE_E_RANGE;
    sf[m] = sf_true[m] < cap ? sf_true[m] : cap;
    ++keyptr[sf[m] + 1];
  }
  for (int64_t k = 0; k < nk; ++k) keyptr[k + 1] += keyptr[k];
  {
    std::vector<int32_t> cur(keyptr, keyptr + nk);
    for (int64_t m = 0; m < n; ++m) order[cur[(size_t)sf[m]]++] = (int32_t)m;
  }
  int64_t nu = 0, acc = 0, k = 0;
  for (int64_t i = 0; i <= n; ++i) {
    while (k <= nk && keyptr[k] == i) {  // group k starts here (empty groups share a start)
      u_ptr[k] = (int32_t)nu;
      nnz_ptr[k] = acc;
      ++k;
    }
    if (i == n) break;
    const int32_t m = order[i];
    if (read[(size_t)m]) u_rows[nu++] = m;
    acc += rowptr[m + 1] - rowptr[m];
  }
  return 0;
}

}  // extern "C"
