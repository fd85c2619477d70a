// GraphGRU row kernels (fast_jtnn/jtnn_enc.py:107-128), forward and backward.
//
// Algebra used (exact, only the summation order differs from the reference):
//   U_r(h[j]) + b == (U_r h + b)[j]      -> the reset-gate product is formed once per
//                                           message (a dense GEMM) and gathered here,
//                                           not once per (message, neighbour) pair;
//   W_r x, W_z[:, :H] x + b_z, W_h[:, :H] x + b_h do not change over the iterations and
//   x = E[word of the source node], so they are small per-word tables (Pr, Pz, Ph)
//   looked up through wid[m]; they stay L2-resident.
//
// Active rows.  The update is synchronous, so h_{t+1}(m) is a function of h_t of the
// messages m reads.  A message that reads nothing is constant from iteration 1 on; a
// message whose inputs are constant from iteration k on is constant from k+1 on
// ("stable_from", sf).  On a tree every message becomes constant after (height of its
// subtree) iterations, so iteration t only has to update the rows with sf > t; the
// others would be recomputed to the bit-identical value.  The kernels therefore work on
// a list of active row ids.  Per-iteration state lives in slabs [T][n_mess][ld] of which
// only active rows are ever written; the value of row j "at iteration t" is found in
// slab min(t, sf[j]).
// One thread = one float4 of one active row.
#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

namespace {

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_iter0_kernel(const float* __restrict__ Pz,
                                                                const float* __restrict__ Ph,
                                                                const int* __restrict__ wid,
                                                                float* __restrict__ z, float* __restrict__ p,
                                                                float* __restrict__ h1, long total, int h4,
                                                                int H, int ld, int pad_row) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  int w = wid[q.row];
  float4 zz = sigmoid4(*rowp(Pz, w, ld, q.c4));
  float4 a = *rowp(Ph, w, ld, q.c4);
  float4 pp = make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
  float4 hh = zz * pp;
  if (q.row == pad_row) hh = f4zero();  // the padding message stays zero (jtnn_enc.py:108-109,126)
  if (z) *rowp(z, q.row, ld, q.c4) = zero_pad(zz, q.c4, H);
  if (p) *rowp(p, q.row, ld, q.c4) = zero_pad(pp, q.c4, H);
  *rowp(h1, q.row, ld, q.c4) = zero_pad(hh, q.c4, H);
}

template <bool TRAIN>
__global__ __launch_bounds__(JTVAE_BLOCK) void gru_gate_gather_kernel(
    const float* __restrict__ h, const float* __restrict__ Uh, size_t slab_stride, int t_iter,
    const int* __restrict__ sf, const float* __restrict__ Pr, const int* __restrict__ wid,
    const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ rows,
    float* __restrict__ s, float* __restrict__ g, float* __restrict__ t, long total, int h4, int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  const int m = rows ? rows[q.row] : q.row;
  int beg = rowptr[m], end = rowptr[m + 1];
  float4 wr = *rowp(Pr, wid[m], ld, q.c4);
  float4 ss = f4zero(), gg = f4zero(), tt = f4zero();
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  for (int e = beg; e < end; e += 2) {
    // two neighbours (four row loads) in flight
    int j0 = col[e];
    int j1 = (e + 1 < end) ? col[e + 1] : -1;
    size_t o0 = sf ? (size_t)imin(t_iter, sf[j0]) * slab_stride : 0;
    float4 h0 = *rowp(h + o0, j0, ld, q.c4);
    float4 u0 = *rowp(Uh + o0, j0, ld, q.c4);
    float4 h1v = f4zero(), u1 = f4zero();
    if (j1 >= 0) {
      size_t o1 = sf ? (size_t)imin(t_iter, sf[j1]) * slab_stride : 0;
      h1v = *rowp(h + o1, j1, ld, q.c4);
      u1 = *rowp(Uh + o1, j1, ld, q.c4);
    }
    float4 r0 = sigmoid4(wr + u0);
    ss = ss + h0;
    fma4(gg, r0, h0);
    if (TRAIN) fma4(tt, r0 * (one - r0), h0);
    if (j1 >= 0) {
      float4 r1 = sigmoid4(wr + u1);
      ss = ss + h1v;
      fma4(gg, r1, h1v);
      if (TRAIN) fma4(tt, r1 * (one - r1), h1v);
    }
  }
  *rowp(s, m, ld, q.c4) = zero_pad(ss, q.c4, H);
  *rowp(g, m, ld, q.c4) = zero_pad(gg, q.c4, H);
  if (TRAIN) *rowp(t, m, ld, q.c4) = zero_pad(tt, q.c4, H);
}

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_bwd_pointwise_kernel(
    const float* __restrict__ dh, const float* __restrict__ z, const float* __restrict__ p,
    const float* __restrict__ s, size_t slab_stride, int t_iter, const int* __restrict__ sf,
    const int* __restrict__ rows, float* __restrict__ dzpre, float* __restrict__ dppre,
    float* __restrict__ ds, long total, int h4, int H, int ld, int pad_row) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  const int m = rows ? rows[q.row] : q.row;
  // the state row m had at iteration t_iter is the one of iteration min(t_iter, sf[m]-1);
  // iteration 0 has s == 0 and keeps no s slab
  const int tau = sf ? imin(t_iter, sf[m] - 1) : t_iter;
  const size_t off = (size_t)tau * slab_stride;
  float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 d = *rowp(dh, m, ld, q.c4);
  if (m == pad_row) d = f4zero();  // backward of "* mask"
  float4 zz = *rowp(z + off, m, ld, q.c4);
  float4 pp = *rowp(p + off, m, ld, q.c4);
  float4 ss = tau > 0 ? *rowp(s + off, m, ld, q.c4) : f4zero();
  float4 dz = d * (pp - ss) * zz * (one - zz);
  float4 dp = d * zz * (one - pp * pp);
  float4 dsv = d * (one - zz);
  // gradients of the pre-activations are stored compactly (row i of this launch), ds by message id
  *rowp(dzpre, q.row, ld, q.c4) = zero_pad(dz, q.c4, H);
  *rowp(dppre, q.row, ld, q.c4) = zero_pad(dp, q.c4, H);
  *rowp(ds, m, ld, q.c4) = zero_pad(dsv, q.c4, H);
}

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_bwd_gather_kernel(
    const float* __restrict__ h, const float* __restrict__ Uh, size_t slab_stride, int t_iter,
    const int* __restrict__ sf, const unsigned long long* __restrict__ active, const float* __restrict__ Pr,
    const int* __restrict__ wid, const float* __restrict__ dg, const float* __restrict__ ds,
    const int* __restrict__ trowptr, const int* __restrict__ tcol, const int* __restrict__ rows,
    const float* __restrict__ dh_ext, float* __restrict__ dUh, float* __restrict__ dhdir, long total, int h4, int H,
    int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  const int j = rows ? rows[q.row] : q.row;
  int beg = trowptr[j], end = trowptr[j + 1];
  float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  const size_t off = sf ? (size_t)imin(t_iter, sf[j]) * slab_stride : 0;
  float4 hj = *rowp(h + off, j, ld, q.c4);
  float4 uj = *rowp(Uh + off, j, ld, q.c4);
  float4 acc_u = f4zero(), acc_h = f4zero();
  const unsigned long long bit = 1ull << t_iter;
  for (int e = beg; e < end; ++e) {
    int m = tcol[e];
    // a reader that receives no gradient at this iteration holds stale dg / ds rows
    if (active && !(active[m] & bit)) continue;
    float4 wr = *rowp(Pr, wid[m], ld, q.c4);
    float4 dgm = *rowp(dg, m, ld, q.c4);
    float4 dsm = *rowp(ds, m, ld, q.c4);
    float4 r = sigmoid4(wr + uj);
    acc_h = acc_h + dsm;
    fma4(acc_h, dgm, r);
    fma4(acc_u, dgm, r * (one - r));
  }
  if (dh_ext) acc_h = acc_h + *rowp(dh_ext, j, ld, q.c4);  // gradient handed in from outside the recurrence
  *rowp(dUh, q.row, ld, q.c4) = zero_pad(acc_u * hj, q.c4, H);  // compact: row i of this launch
  *rowp(dhdir, j, ld, q.c4) = zero_pad(acc_h, q.c4, H);
}

// out[i] = base[sel[i] * slab_stride + i * ld ...]: the final value of every message
__global__ __launch_bounds__(JTVAE_BLOCK) void select_slab_rows_kernel(float* __restrict__ out,
                                                                       const float* __restrict__ base,
                                                                       size_t slab_stride,
                                                                       const int* __restrict__ sel, long total,
                                                                       int h4, int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  float4 v = *rowp(base + (size_t)sel[q.row] * slab_stride, q.row, ld, q.c4);
  *rowp(out, q.row, ld, q.c4) = zero_pad(v, q.c4, H);
}

inline int check_rows(const void* p, int ld) {
  if (!p) return JTVAE_E_NULL;
  if (ld <= 0 || (ld & 3)) return JTVAE_E_ALIGN;
  if (reinterpret_cast<uintptr_t>(p) & 15) return JTVAE_E_ALIGN;
  return 0;
}
inline unsigned grid_for(long total) { return (unsigned)((total + JTVAE_BLOCK - 1) / JTVAE_BLOCK); }
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

#define JT_CHECK(expr)   \
  do {                   \
    int _rc = (expr);    \
    if (_rc) return _rc; \
  } while (0)

extern "C" {

int jtvae_gru_iter0(const float* Pz, const float* Ph, const int32_t* wid, float* z, float* p, float* h1,
                    int32_t n_mess, int32_t H, int32_t ld, int32_t pad_row, void* stream) {
  if (n_mess < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_mess == 0) return 0;
  if (!wid) return JTVAE_E_NULL;
  JT_CHECK(check_rows(Pz, ld));
  JT_CHECK(check_rows(Ph, ld));
  if (z) JT_CHECK(check_rows(z, ld));
  if (p) JT_CHECK(check_rows(p, ld));
  JT_CHECK(check_rows(h1, ld));
  int h4 = ld / 4;
  long total = (long)n_mess * h4;
  hipLaunchKernelGGL(gru_iter0_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, Pz, Ph,
                     wid, z, p, h1, total, h4, H, ld, pad_row);
  return launch_status();
}

int jtvae_gru_gate_gather(const float* h, const float* Uh, size_t slab_stride, int32_t t_iter,
                          const int32_t* sf, const float* Pr, const int32_t* wid, const int32_t* rowptr,
                          const int32_t* col, const int32_t* rows, int32_t n_rows, float* s, float* g, float* t,
                          int32_t H, int32_t ld, void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld || t_iter < 0 || (slab_stride & 3)) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!wid || !rowptr || !col) return JTVAE_E_NULL;
  JT_CHECK(check_rows(h, ld));
  JT_CHECK(check_rows(Uh, ld));
  JT_CHECK(check_rows(Pr, ld));
  JT_CHECK(check_rows(s, ld));
  JT_CHECK(check_rows(g, ld));
  if (t) JT_CHECK(check_rows(t, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipStream_t st = (hipStream_t)stream;
  if (t)
    hipLaunchKernelGGL(gru_gate_gather_kernel<true>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, st, h, Uh,
                       slab_stride, t_iter, sf, Pr, wid, rowptr, col, rows, s, g, t, total, h4, H, ld);
  else
    hipLaunchKernelGGL(gru_gate_gather_kernel<false>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, st, h, Uh,
                       slab_stride, t_iter, sf, Pr, wid, rowptr, col, rows, s, g, t, total, h4, H, ld);
  return launch_status();
}

int jtvae_gru_bwd_pointwise(const float* dh, const float* z, const float* p, const float* s, size_t slab_stride,
                            int32_t t_iter, const int32_t* sf, const int32_t* rows, int32_t n_rows, float* dzpre,
                            float* dppre, float* ds, int32_t H, int32_t ld, int32_t pad_row, void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld || t_iter < 0 || (slab_stride & 3)) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  JT_CHECK(check_rows(dh, ld));
  JT_CHECK(check_rows(z, ld));
  JT_CHECK(check_rows(p, ld));
  JT_CHECK(check_rows(s, ld));
  JT_CHECK(check_rows(dzpre, ld));
  JT_CHECK(check_rows(dppre, ld));
  JT_CHECK(check_rows(ds, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipLaunchKernelGGL(gru_bwd_pointwise_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream,
                     dh, z, p, s, slab_stride, t_iter, sf, rows, dzpre, dppre, ds, total, h4, H, ld, pad_row);
  return launch_status();
}

int jtvae_gru_bwd_gather(const float* h, const float* Uh, size_t slab_stride, int32_t t_iter, const int32_t* sf,
                         const uint64_t* active, const float* Pr, const int32_t* wid, const float* dg,
                         const float* ds, const int32_t* trowptr, const int32_t* tcol, const int32_t* rows,
                         int32_t n_rows, const float* dh_ext, float* dUh, float* dhdir, int32_t H, int32_t ld,
                         void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld || t_iter < 0 || t_iter > 63 || (slab_stride & 3)) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!wid || !trowptr || !tcol) return JTVAE_E_NULL;
  JT_CHECK(check_rows(h, ld));
  JT_CHECK(check_rows(Uh, ld));
  JT_CHECK(check_rows(Pr, ld));
  JT_CHECK(check_rows(dg, ld));
  JT_CHECK(check_rows(ds, ld));
  if (dh_ext) JT_CHECK(check_rows(dh_ext, ld));
  JT_CHECK(check_rows(dUh, ld));
  JT_CHECK(check_rows(dhdir, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipLaunchKernelGGL(gru_bwd_gather_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, h,
                     Uh, slab_stride, t_iter, sf, (const unsigned long long*)active, Pr, wid, dg, ds, trowptr, tcol,
                     rows, dh_ext, dUh, dhdir, total, h4, H, ld);
  return launch_status();
}

int jtvae_select_slab_rows(float* out, const float* base, size_t slab_stride, const int32_t* sel, int32_t n_rows,
                           int32_t H, int32_t ld, void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld || (slab_stride & 3)) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!sel) return JTVAE_E_NULL;
  JT_CHECK(check_rows(out, ld));
  JT_CHECK(check_rows(base, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipLaunchKernelGGL(select_slab_rows_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream,
                     out, base, slab_stride, sel, total, h4, H, ld);
  return launch_status();
}

}  // extern "C"
