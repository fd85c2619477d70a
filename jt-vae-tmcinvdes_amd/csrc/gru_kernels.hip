// GraphGRU row kernels (fast_jtnn/jtnn_enc.py:107-128), forward and backward.
//
// Algebra used (exact, only the summation order differs from the reference):
//   U_r(h[j]) + b == (U_r h + b)[j]      -> the reset-gate product is formed once per
//                                           message (a dense GEMM) and gathered here,
//                                           not once per (message, neighbour) pair;
//   W_r x, W_z[:, :H] x + b_z, W_h[:, :H] x + b_h do not change over the iterations and
//   x = E[word of the source node], so they are small per-word tables (Pr, Pz, Ph)
//   looked up through wid[m]; they stay L2-resident.
// One thread = one float4 of one message row.
#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

namespace {

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_iter0_kernel(const float* __restrict__ Pz,
                                                                const float* __restrict__ Ph,
                                                                const int* __restrict__ wid,
                                                                float* __restrict__ z, float* __restrict__ p,
                                                                float* __restrict__ h1, long total, int h4,
                                                                int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  int w = wid[q.row];
  float4 zz = sigmoid4(*rowp(Pz, w, ld, q.c4));
  float4 a = *rowp(Ph, w, ld, q.c4);
  float4 pp = make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
  float4 hh = zz * pp;
  if (q.row == 0) hh = f4zero();  // the padding message stays zero (jtnn_enc.py:108-109,126)
  *rowp(z, q.row, ld, q.c4) = zero_pad(zz, q.c4, H);
  *rowp(p, q.row, ld, q.c4) = zero_pad(pp, q.c4, H);
  *rowp(h1, q.row, ld, q.c4) = zero_pad(hh, q.c4, H);
}

template <bool TRAIN>
__global__ __launch_bounds__(JTVAE_BLOCK) void gru_gate_gather_kernel(
    const float* __restrict__ h, const float* __restrict__ Uh, const float* __restrict__ Pr,
    const int* __restrict__ wid, const int* __restrict__ rowptr, const int* __restrict__ col,
    float* __restrict__ s, float* __restrict__ g, float* __restrict__ t, long total, int h4, int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  int beg = rowptr[q.row], end = rowptr[q.row + 1];
  float4 wr = *rowp(Pr, wid[q.row], ld, q.c4);
  float4 ss = f4zero(), gg = f4zero(), tt = f4zero();
  for (int e = beg; e < end; e += 2) {
    // two neighbours (four row loads) in flight
    int j0 = col[e];
    int j1 = (e + 1 < end) ? col[e + 1] : -1;
    float4 h0 = *rowp(h, j0, ld, q.c4);
    float4 u0 = *rowp(Uh, j0, ld, q.c4);
    float4 h1v = f4zero(), u1 = f4zero();
    if (j1 >= 0) {
      h1v = *rowp(h, j1, ld, q.c4);
      u1 = *rowp(Uh, j1, ld, q.c4);
    }
    float4 r0 = sigmoid4(wr + u0);
    ss = ss + h0;
    fma4(gg, r0, h0);
    if (TRAIN) {
      float4 d0 = r0 * (make_float4(1.f, 1.f, 1.f, 1.f) - r0);
      fma4(tt, d0, h0);
    }
    if (j1 >= 0) {
      float4 r1 = sigmoid4(wr + u1);
      ss = ss + h1v;
      fma4(gg, r1, h1v);
      if (TRAIN) {
        float4 d1 = r1 * (make_float4(1.f, 1.f, 1.f, 1.f) - r1);
        fma4(tt, d1, h1v);
      }
    }
  }
  *rowp(s, q.row, ld, q.c4) = zero_pad(ss, q.c4, H);
  *rowp(g, q.row, ld, q.c4) = zero_pad(gg, q.c4, H);
  if (TRAIN) *rowp(t, q.row, ld, q.c4) = zero_pad(tt, q.c4, H);
}

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_bwd_pointwise_kernel(
    const float* __restrict__ dh, const float* __restrict__ z, const float* __restrict__ p,
    const float* __restrict__ s, float* __restrict__ dzpre, float* __restrict__ dppre,
    float* __restrict__ ds, long total, int h4, int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 d = *rowp(dh, q.row, ld, q.c4);
  if (q.row == 0) d = f4zero();  // backward of "* mask"
  float4 zz = *rowp(z, q.row, ld, q.c4);
  float4 pp = *rowp(p, q.row, ld, q.c4);
  float4 ss = *rowp(s, q.row, ld, q.c4);
  float4 dz = d * (pp - ss) * zz * (one - zz);
  float4 dp = d * zz * (one - pp * pp);
  float4 dsv = d * (one - zz);
  *rowp(dzpre, q.row, ld, q.c4) = zero_pad(dz, q.c4, H);
  *rowp(dppre, q.row, ld, q.c4) = zero_pad(dp, q.c4, H);
  *rowp(ds, q.row, ld, q.c4) = zero_pad(dsv, q.c4, H);
}

__global__ __launch_bounds__(JTVAE_BLOCK) void gru_bwd_gather_kernel(
    const float* __restrict__ h, const float* __restrict__ Uh, const float* __restrict__ Pr,
    const int* __restrict__ wid, const float* __restrict__ dg, const float* __restrict__ ds,
    const int* __restrict__ trowptr, const int* __restrict__ tcol, float* __restrict__ dUh,
    float* __restrict__ dhdir, long total, int h4, int H, int ld) {
  RowPos q = row_pos(total, h4);
  if (!q.ok) return;
  int beg = trowptr[q.row], end = trowptr[q.row + 1];
  float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 hj = *rowp(h, q.row, ld, q.c4);
  float4 uj = *rowp(Uh, q.row, ld, q.c4);
  float4 acc_u = f4zero(), acc_h = f4zero();
  for (int e = beg; e < end; ++e) {
    int m = tcol[e];
    float4 wr = *rowp(Pr, wid[m], ld, q.c4);
    float4 dgm = *rowp(dg, m, ld, q.c4);
    float4 dsm = *rowp(ds, m, ld, q.c4);
    float4 r = sigmoid4(wr + uj);
    acc_h = acc_h + dsm;
    fma4(acc_h, dgm, r);
    fma4(acc_u, dgm, r * (one - r));
  }
  *rowp(dUh, q.row, ld, q.c4) = zero_pad(acc_u * hj, q.c4, H);
  *rowp(dhdir, q.row, ld, q.c4) = zero_pad(acc_h, q.c4, H);
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
                    int32_t n_mess, int32_t H, int32_t ld, void* stream) {
  if (n_mess < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_mess == 0) return 0;
  if (!wid) return JTVAE_E_NULL;
  JT_CHECK(check_rows(Pz, ld));
  JT_CHECK(check_rows(Ph, ld));
  JT_CHECK(check_rows(z, ld));
  JT_CHECK(check_rows(p, ld));
  JT_CHECK(check_rows(h1, ld));
  int h4 = ld / 4;
  long total = (long)n_mess * h4;
  hipLaunchKernelGGL(gru_iter0_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, Pz, Ph,
                     wid, z, p, h1, total, h4, H, ld);
  return launch_status();
}

int jtvae_gru_gate_gather(const float* h, const float* Uh, const float* Pr, const int32_t* wid,
                          const int32_t* rowptr, const int32_t* col, float* s, float* g, float* t,
                          int32_t n_mess, int32_t H, int32_t ld, void* stream) {
  if (n_mess < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_mess == 0) return 0;
  if (!wid || !rowptr || !col) return JTVAE_E_NULL;
  JT_CHECK(check_rows(h, ld));
  JT_CHECK(check_rows(Uh, ld));
  JT_CHECK(check_rows(Pr, ld));
  JT_CHECK(check_rows(s, ld));
  JT_CHECK(check_rows(g, ld));
  if (t) JT_CHECK(check_rows(t, ld));
  int h4 = ld / 4;
  long total = (long)n_mess * h4;
  hipStream_t st = (hipStream_t)stream;
  if (t)
    hipLaunchKernelGGL(gru_gate_gather_kernel<true>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, st, h, Uh, Pr,
                       wid, rowptr, col, s, g, t, total, h4, H, ld);
  else
    hipLaunchKernelGGL(gru_gate_gather_kernel<false>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, st, h, Uh, Pr,
                       wid, rowptr, col, s, g, t, total, h4, H, ld);
  return launch_status();
}

int jtvae_gru_bwd_pointwise(const float* dh, const float* z, const float* p, const float* s, float* dzpre,
                            float* dppre, float* ds, int32_t n_mess, int32_t H, int32_t ld, void* stream) {
  if (n_mess < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_mess == 0) return 0;
  JT_CHECK(check_rows(dh, ld));
  JT_CHECK(check_rows(z, ld));
  JT_CHECK(check_rows(p, ld));
  JT_CHECK(check_rows(s, ld));
  JT_CHECK(check_rows(dzpre, ld));
  JT_CHECK(check_rows(dppre, ld));
  JT_CHECK(check_rows(ds, ld));
  int h4 = ld / 4;
  long total = (long)n_mess * h4;
  hipLaunchKernelGGL(gru_bwd_pointwise_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream,
                     dh, z, p, s, dzpre, dppre, ds, total, h4, H, ld);
  return launch_status();
}

int jtvae_gru_bwd_gather(const float* h, const float* Uh, const float* Pr, const int32_t* wid, const float* dg,
                         const float* ds, const int32_t* trowptr, const int32_t* tcol, float* dUh, float* dhdir,
                         int32_t n_mess, int32_t H, int32_t ld, void* stream) {
  if (n_mess < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_mess == 0) return 0;
  if (!wid || !trowptr || !tcol) return JTVAE_E_NULL;
  JT_CHECK(check_rows(h, ld));
  JT_CHECK(check_rows(Uh, ld));
  JT_CHECK(check_rows(Pr, ld));
  JT_CHECK(check_rows(dg, ld));
  JT_CHECK(check_rows(ds, ld));
  JT_CHECK(check_rows(dUh, ld));
  JT_CHECK(check_rows(dhdir, ld));
  int h4 = ld / 4;
  long total = (long)n_mess * h4;
  hipLaunchKernelGGL(gru_bwd_gather_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, h,
                     Uh, Pr, wid, dg, ds, trowptr, tcol, dUh, dhdir, total, h4, H, ld);
  return launch_status();
}

}  // extern "C"
