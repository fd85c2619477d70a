// HBM-bound row kernels of the encoder path: CSR neighbour gather-sum (forward and,
// on the transposed graph, backward), per-molecule mean readout, copies.
//
// One thread = one float4 of one output row; a row's neighbour rows are read with
// 16-byte loads by consecutive lanes (coalesced 1.8 KB row reads), summed in
// registers and written once.  Nothing of size [rows, K, H] ever exists.
#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

namespace {

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f, v.w > 0.f ? v.w : 0.f);
}

template <bool TWO_SRC>
__global__ __launch_bounds__(JTVAE_BLOCK) void csr_gather_sum_kernel(
    float* __restrict__ out, const float* __restrict__ src0, int n_src0, const float* __restrict__ src1,
    const int* __restrict__ rowptr, const int* __restrict__ col, long total, int h4, int H, int ld,
    const float* __restrict__ relu_mask, int accumulate, float* __restrict__ acc2, int relu_src) {
  RowPos p = row_pos(total, h4);
  if (!p.ok) return;
  int beg = rowptr[p.row], end = rowptr[p.row + 1];
  float4 acc = f4zero();
  // relu_src: the rows of the LAST source (src1 if given, else src0) hold pre-activations; their ReLU is
  // taken as they are read, so the activated copy never has to be written
  const bool relu0 = relu_src && !TWO_SRC;
  // four independent row loads in flight per thread
  for (int e = beg; e < end; e += 4) {
    int j[4];
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) j[u] = (e + u < end) ? col[e + u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = f4zero();
      if (j[u] >= 0) {
        if (TWO_SRC && j[u] >= n_src0) {
          v[u] = *rowp(src1, j[u] - n_src0, ld, p.c4);
          if (relu_src) v[u] = relu4(v[u]);
        } else {
          v[u] = *rowp(src0, j[u], ld, p.c4);
          if (relu0) v[u] = relu4(v[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = acc + v[u];
  }
  if (relu_mask) {
    float4 m = *rowp(relu_mask, p.row, ld, p.c4);
    acc.x = m.x > 0.f ? acc.x : 0.f;
    acc.y = m.y > 0.f ? acc.y : 0.f;
    acc.z = m.z > 0.f ? acc.z : 0.f;
    acc.w = m.w > 0.f ? acc.w : 0.f;
  }
  float4* o = rowp(out, p.row, ld, p.c4);
  if (accumulate) acc = acc + *o;
  acc = zero_pad(acc, p.c4, H);
  *o = acc;
  if (acc2) {
    float4* a = rowp(acc2, p.row, ld, p.c4);
    *a = zero_pad(*a + acc, p.c4, H);
  }
}

__global__ __launch_bounds__(JTVAE_BLOCK) void segment_mean_kernel(float* __restrict__ out,
                                                                   const float* __restrict__ rows,
                                                                   const int* __restrict__ seg, long total,
                                                                   int h4, int H, int ld) {
  RowPos p = row_pos(total, h4);
  if (!p.ok) return;
  int st = seg[2 * p.row], le = seg[2 * p.row + 1];
  float4 acc = f4zero();
  int a = 0;
  for (; a + 4 <= le; a += 4) {
    float4 v0 = *rowp(rows, st + a + 0, ld, p.c4);
    float4 v1 = *rowp(rows, st + a + 1, ld, p.c4);
    float4 v2 = *rowp(rows, st + a + 2, ld, p.c4);
    float4 v3 = *rowp(rows, st + a + 3, ld, p.c4);
    acc = acc + v0;
    acc = acc + v1;
    acc = acc + v2;
    acc = acc + v3;
  }
  for (; a < le; ++a) acc = acc + *rowp(rows, st + a, ld, p.c4);
  // true division, as the reference's sum/le and mean(): le == 0 gives nan there too
  float inv = 1.0f / (float)le;
  *rowp(out, p.row, ld, p.c4) = zero_pad(acc * inv, p.c4, H);
}

__global__ __launch_bounds__(JTVAE_BLOCK) void segment_mean_bwd_kernel(
    float* __restrict__ drows, const float* __restrict__ dout, const float* __restrict__ act,
    const int* __restrict__ seg, const int* __restrict__ seg_of, long total, int h4, int H, int ld) {
  RowPos p = row_pos(total, h4);
  if (!p.ok) return;
  int b = seg_of[p.row];
  if (b < 0) {  // row outside every segment: the readout never saw it
    *rowp(drows, p.row, ld, p.c4) = f4zero();
    return;
  }
  float inv = 1.0f / (float)seg[2 * b + 1];
  float4 g = *rowp(dout, b, ld, p.c4) * inv;
  float4 a = *rowp(act, p.row, ld, p.c4);
  g.x = a.x > 0.f ? g.x : 0.f;
  g.y = a.y > 0.f ? g.y : 0.f;
  g.z = a.z > 0.f ? g.z : 0.f;
  g.w = a.w > 0.f ? g.w : 0.f;
  *rowp(drows, p.row, ld, p.c4) = zero_pad(g, p.c4, H);
}

// any-alignment row copy, one thread per destination float (pads written as zero)
__global__ __launch_bounds__(JTVAE_BLOCK) void copy_rows_kernel(const float* __restrict__ src, int ld_src,
                                                                float* __restrict__ dst, int ld_dst,
                                                                long total, int cols) {
  long gid = (long)blockIdx.x * JTVAE_BLOCK + threadIdx.x;
  if (gid >= total) return;
  long r = gid / ld_dst;
  int c = (int)(gid - r * ld_dst);
  dst[gid] = c < cols ? src[r * ld_src + c] : 0.f;
}

// byte features widened to float rows, one thread per destination float (pads written as zero)
__global__ __launch_bounds__(JTVAE_BLOCK) void widen_u8_rows_kernel(const uint8_t* __restrict__ src,
                                                                    float* __restrict__ dst, int ld_dst,
                                                                    long total, int cols) {
  long gid = (long)blockIdx.x * JTVAE_BLOCK + threadIdx.x;
  if (gid >= total) return;
  long r = gid / ld_dst;
  int c = (int)(gid - r * ld_dst);
  dst[gid] = c < cols ? (float)src[r * cols + c] : 0.f;
}

// Several small matrices (the Linear weights of one module) packed or transposed into 16-byte-row
// buffers by one launch: block b serves one 32 x 32 tile of one destination.  Padding columns of
// every destination row are written as zero.
struct PackArgs {
  jtvae_pack_item_t it[JTVAE_PACK_MAX];
  int tile_end[JTVAE_PACK_MAX];  // running total of tiles after item i
  int n;
};
__global__ __launch_bounds__(256) void pack_batch_kernel(PackArgs a) {
  __shared__ float tile[32][33];
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.tile_end[i]) ++i;
  const jtvae_pack_item_t& it = a.it[i];
  const int local = blockIdx.x - (i ? a.tile_end[i - 1] : 0);
  const int drows = it.transpose ? it.cols : it.rows;   // destination rows
  const int dvalid = it.transpose ? it.rows : it.cols;  // destination columns that hold data
  const int tiles_c = (it.ld_dst + 31) / 32;
  const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (it.transpose) {
    // destination tile [r0.., c0..] = source tile [c0.., r0..] turned through LDS
    for (int k = ty; k < 32; k += 8) {
      const int sr = c0 + k, sc = r0 + tx;
      tile[k][tx] = (sr < it.rows && sc < it.cols) ? it.src[(size_t)sr * it.ld_src + sc] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      const int dr = r0 + k, dc = c0 + tx;
      if (dr < drows && dc < it.ld_dst) it.dst[(size_t)dr * it.ld_dst + dc] = dc < dvalid ? tile[tx][k] : 0.f;
    }
  } else {
    for (int k = ty; k < 32; k += 8) {
      const int dr = r0 + k, dc = c0 + tx;
      if (dr < drows && dc < it.ld_dst)
        it.dst[(size_t)dr * it.ld_dst + dc] = dc < dvalid ? it.src[(size_t)dr * it.ld_src + dc] : 0.f;
    }
  }
}

// Candidate-assembly loss (fast_jtnn/jtnn_vae.py:116-135): per tree node one cross entropy over the
// scores of its candidates.  One thread per segment (a node has a handful of candidates): loss, whether
// the label scores highest, and d(loss)/d(score) = softmax - onehot for the backward pass.
__global__ __launch_bounds__(JTVAE_BLOCK) void segment_ce_kernel(const float* __restrict__ scores,
                                                                 const int* __restrict__ seg_ptr,
                                                                 const int* __restrict__ label_pos, int n_seg,
                                                                 float* __restrict__ loss_seg,
                                                                 int* __restrict__ hit_seg,
                                                                 float* __restrict__ dscores) {
  int sgm = blockIdx.x * JTVAE_BLOCK + threadIdx.x;
  if (sgm >= n_seg) return;
  const int b = seg_ptr[sgm], e = seg_ptr[sgm + 1], lab = label_pos[sgm];
  float mx = -INFINITY;
  for (int i = b; i < e; ++i) mx = fmaxf(mx, scores[i]);
  float den = 0.f;
  for (int i = b; i < e; ++i) den += expf(scores[i] - mx);
  const float sl = scores[lab];
  loss_seg[sgm] = logf(den) + mx - sl;
  hit_seg[sgm] = sl >= mx ? 1 : 0;
  if (dscores) {
    const float inv = 1.f / den;
    for (int i = b; i < e; ++i) dscores[i] = expf(scores[i] - mx) * inv - (i == lab ? 1.f : 0.f);
  }
}

// score[c] = <q[owner[c]], cand[c]>, one wave per candidate
__global__ __launch_bounds__(JTVAE_BLOCK) void row_dot_kernel(float* __restrict__ score,
                                                              const float* __restrict__ q,
                                                              const int* __restrict__ owner,
                                                              const float* __restrict__ cand, int n_cand,
                                                              int h4, int H, int ld) {
  int wave = (int)(((long)blockIdx.x * JTVAE_BLOCK + threadIdx.x) >> 6);
  int lane = threadIdx.x & 63;
  if (wave >= n_cand) return;
  int o = owner[wave];
  float acc = 0.f;
  for (int c = lane; c < h4; c += 64) {
    float4 a = zero_pad(*rowp(q, o, ld, c), c, H);  // padding columns never enter the dot
    float4 b = zero_pad(*rowp(cand, wave, ld, c), c, H);
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) score[wave] = acc;
}

// Backward of row_dot.  Rows [0, n_cand): dcand[c] = dscore[c] * q[owner[c]].  Rows [n_cand, n_cand + n_owner):
// dq[b] = sum over the candidates c of owner b (own_ptr[b] .. own_ptr[b+1], consecutive) of dscore[c] * cand[c],
// in candidate order.  One thread = one float4 of one output row.
__global__ __launch_bounds__(JTVAE_BLOCK) void row_dot_bwd_kernel(const float* __restrict__ dscore,
                                                                  const float* __restrict__ q,
                                                                  const int* __restrict__ owner,
                                                                  const float* __restrict__ cand,
                                                                  const int* __restrict__ own_ptr, int n_cand,
                                                                  float* __restrict__ dq, float* __restrict__ dcand,
                                                                  long total, int h4, int H, int ld) {
  RowPos p = row_pos(total, h4);
  if (!p.ok) return;
  if (p.row < n_cand) {
    const float g = dscore[p.row];
    *rowp(dcand, p.row, ld, p.c4) = zero_pad(*rowp(q, owner[p.row], ld, p.c4) * g, p.c4, H);
    return;
  }
  const int b = p.row - n_cand;
  float4 acc = f4zero();
  for (int c = own_ptr[b]; c < own_ptr[b + 1]; ++c) {
    const float g = dscore[c];
    const float4 v = *rowp(cand, c, ld, p.c4);
    acc.x = fmaf(g, v.x, acc.x);
    acc.y = fmaf(g, v.y, acc.y);
    acc.z = fmaf(g, v.z, acc.z);
    acc.w = fmaf(g, v.w, acc.w);
  }
  *rowp(dq, b, ld, p.c4) = zero_pad(acc, p.c4, H);
}

inline int check_rows(const void* p, int ld) {
  if (!p) return JTVAE_E_NULL;
  if (ld <= 0 || (ld & 3)) return JTVAE_E_ALIGN;
  if (reinterpret_cast<uintptr_t>(p) & 15) return JTVAE_E_ALIGN;
  return 0;
}
inline int check_rows_opt(const void* p, int ld) { return p ? check_rows(p, ld) : 0; }

inline unsigned grid_for(long total) { return (unsigned)((total + JTVAE_BLOCK - 1) / JTVAE_BLOCK); }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

#define JT_CHECK(expr)        \
  do {                        \
    int _rc = (expr);         \
    if (_rc) return _rc;      \
  } while (0)

extern "C" {

int jtvae_csr_gather_sum(float* out, const float* src0, int32_t n_src0, const float* src1,
                         const int32_t* rowptr, const int32_t* col, int32_t n_rows, int32_t H, int32_t ld,
                         const float* relu_mask, int32_t accumulate, float* acc2, int32_t relu_src, void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld || n_src0 < 0) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!rowptr || !col) return JTVAE_E_NULL;
  JT_CHECK(check_rows(out, ld));
  JT_CHECK(check_rows(src0, ld));
  JT_CHECK(check_rows_opt(src1, ld));
  JT_CHECK(check_rows_opt(relu_mask, ld));
  JT_CHECK(check_rows_opt(acc2, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipStream_t s = (hipStream_t)stream;
  if (src1)
    hipLaunchKernelGGL(csr_gather_sum_kernel<true>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, s, out, src0,
                       n_src0, src1, rowptr, col, total, h4, H, ld, relu_mask, accumulate, acc2, relu_src);
  else
    hipLaunchKernelGGL(csr_gather_sum_kernel<false>, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, s, out, src0,
                       n_src0, src1, rowptr, col, total, h4, H, ld, relu_mask, accumulate, acc2, relu_src);
  return launch_status();
}

int jtvae_segment_mean(float* out, const float* rows, const int32_t* seg, int32_t n_seg, int32_t H, int32_t ld,
                       void* stream) {
  if (n_seg < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_seg == 0) return 0;
  if (!seg) return JTVAE_E_NULL;
  JT_CHECK(check_rows(out, ld));
  JT_CHECK(check_rows(rows, ld));
  int h4 = ld / 4;
  long total = (long)n_seg * h4;
  hipLaunchKernelGGL(segment_mean_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, out,
                     rows, seg, total, h4, H, ld);
  return launch_status();
}

int jtvae_segment_mean_bwd(float* drows, const float* dout, const float* act, const int32_t* seg,
                           const int32_t* seg_of, int32_t n_rows, int32_t H, int32_t ld, void* stream) {
  if (n_rows < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!seg || !seg_of) return JTVAE_E_NULL;
  JT_CHECK(check_rows(drows, ld));
  JT_CHECK(check_rows(dout, ld));
  JT_CHECK(check_rows(act, ld));
  int h4 = ld / 4;
  long total = (long)n_rows * h4;
  hipLaunchKernelGGL(segment_mean_bwd_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream,
                     drows, dout, act, seg, seg_of, total, h4, H, ld);
  return launch_status();
}

int jtvae_copy_rows(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols,
                    void* stream) {
  if (rows < 0 || cols <= 0 || cols > ld_src || cols > ld_dst) return JTVAE_E_SHAPE;
  if (rows == 0) return 0;
  if (!src || !dst) return JTVAE_E_NULL;
  long total = (long)rows * ld_dst;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, src,
                     ld_src, dst, ld_dst, total, cols);
  return launch_status();
}

int jtvae_widen_u8_rows(const uint8_t* src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols,
                        void* stream) {
  if (rows < 0 || cols <= 0 || cols > ld_dst) return JTVAE_E_SHAPE;
  if (rows == 0) return 0;
  if (!src || !dst) return JTVAE_E_NULL;
  long total = (long)rows * ld_dst;
  hipLaunchKernelGGL(widen_u8_rows_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, src,
                     dst, ld_dst, total, cols);
  return launch_status();
}

int jtvae_pack_batch(const jtvae_pack_item_t* items, int32_t n_items, void* stream) {
  if (n_items < 0 || n_items > JTVAE_PACK_MAX) return JTVAE_E_SHAPE;
  if (n_items == 0) return 0;
  if (!items) return JTVAE_E_NULL;
  PackArgs a;
  int total = 0;
  for (int i = 0; i < n_items; ++i) {
    const jtvae_pack_item_t& it = items[i];
    if (it.rows <= 0 || it.cols <= 0 || it.cols > it.ld_src) return JTVAE_E_SHAPE;
    if ((it.transpose ? it.rows : it.cols) > it.ld_dst) return JTVAE_E_SHAPE;
    if (!it.src || !it.dst) return JTVAE_E_NULL;
    const int drows = it.transpose ? it.cols : it.rows;
    total += ((drows + 31) / 32) * ((it.ld_dst + 31) / 32);
    a.it[i] = it;
    a.tile_end[i] = total;
  }
  a.n = n_items;
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int jtvae_row_dot(float* score, const float* q, const int32_t* owner, const float* cand, int32_t n_cand,
                  int32_t H, int32_t ld, void* stream) {
  if (n_cand < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_cand == 0) return 0;
  if (!score || !owner) return JTVAE_E_NULL;
  JT_CHECK(check_rows(q, ld));
  JT_CHECK(check_rows(cand, ld));
  long total = (long)n_cand * 64;
  hipLaunchKernelGGL(row_dot_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, score, q,
                     owner, cand, n_cand, ld / 4, H, ld);
  return launch_status();
}

int jtvae_row_dot_bwd(const float* dscore, const float* q, const int32_t* owner, const float* cand,
                      const int32_t* own_ptr, int32_t n_cand, int32_t n_owner, float* dq, float* dcand, int32_t H,
                      int32_t ld, void* stream) {
  if (n_cand < 0 || n_owner < 0 || H <= 0 || H > ld) return JTVAE_E_SHAPE;
  if (n_cand + n_owner == 0) return 0;
  if (!dscore || !owner || !own_ptr) return JTVAE_E_NULL;
  JT_CHECK(check_rows(q, ld));
  JT_CHECK(check_rows(cand, ld));
  JT_CHECK(check_rows(dq, ld));
  JT_CHECK(check_rows(dcand, ld));
  long total = ((long)n_cand + n_owner) * (ld / 4);
  hipLaunchKernelGGL(row_dot_bwd_kernel, dim3(grid_for(total)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, dscore, q,
                     owner, cand, own_ptr, n_cand, dq, dcand, total, ld / 4, H, ld);
  return launch_status();
}

int jtvae_segment_ce(const float* scores, const int32_t* seg_ptr, const int32_t* label_pos, int32_t n_seg,
                     float* loss_seg, int32_t* hit_seg, float* dscores, void* stream) {
  if (n_seg < 0) return JTVAE_E_SHAPE;
  if (n_seg == 0) return 0;
  if (!scores || !seg_ptr || !label_pos || !loss_seg || !hit_seg) return JTVAE_E_NULL;
  hipLaunchKernelGGL(segment_ce_kernel, dim3(grid_for(n_seg)), dim3(JTVAE_BLOCK), 0, (hipStream_t)stream, scores,
                     seg_ptr, label_pos, n_seg, loss_seg, hit_seg, dscores);
  return launch_status();
}

}  // extern "C"
