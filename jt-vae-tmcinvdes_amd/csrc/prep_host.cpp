// Host-side index preparation: the reference's padded int64 neighbour tables -> int32 CSR, its
// transpose, and the iteration from which each tree message stops changing.  Plain C++ over host
// memory (no GPU work): single counting passes where a numpy formulation needs sorts and scatter
// reductions.  Orders are the ones the kernels rely on: CSR entries keep the slot order of the
// table; a transposed list holds its readers in increasing (row, slot) order.
#include <cstdint>
#include <cstring>
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
// rows in all T iterations).  T rounds of relaxation also settle graphs with cycles at the cap.
int jtvae_prep_stable_from(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t T, int32_t* sf) {
  if (n < 0 || T < 0) return JTVAE_E_SHAPE;
  if (!rowptr || !sf) return JTVAE_E_NULL;
  const int32_t cap = T > 1 ? T : 1;
  for (int64_t m = 0; m < n; ++m) sf[m] = 1;
  std::vector<int32_t> next((size_t)n);
  for (int32_t round = 0; round <= cap; ++round) {
    bool changed = false;
    for (int64_t m = 0; m < n; ++m) {
      int32_t v = 1;
      if (rowptr[m + 1] > rowptr[m]) {
        int32_t mx = 0;
        for (int32_t e = rowptr[m]; e < rowptr[m + 1]; ++e) {
          const int32_t j = col[e];
          if (j < 0 || j >= n) return JTVAE_E_RANGE;
          mx = sf[j] > mx ? sf[j] : mx;
        }
        v = mx + 1 < cap ? mx + 1 : cap;
      }
      next[(size_t)m] = v;
      changed |= v != sf[m];
    }
    if (!changed) break;
    std::memcpy(sf, next.data(), sizeof(int32_t) * (size_t)n);
  }
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

}  // extern "C"
