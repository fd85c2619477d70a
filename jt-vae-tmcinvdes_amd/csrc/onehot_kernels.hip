// Linear layers whose input rows are one-hot blocks (fast_jtnn/mpn.py:87,99; fast_jtnn/jtmpn.py:89,103: W_i on
// the bond feature rows, the atom-feature columns of W_o).  A feature row of 39..51 entries holds ~8 ones, so
//   out[r] = sum over the set entries j of W^T[j]                       (forward)
//   dW^T[j] += dOut[r] for the set entries j of row r                   (weight gradient)
// are row kernels bound by the 1.8 KB row they write / read, not products worth the matrix cores (a dense K = 51
// product spends 64-wide MFMA tiles on 8 useful terms).  The rows arrive as bytes (whole numbers 0..255, which is
// what the reference's feature extraction yields); an entry other than 0 / 1 multiplies its table row.
//
// The whole transposed weight ([F, ld] floats, 92 KB for F = 51) sits in LDS, one block per CU; blocks walk
// contiguous row ranges in chunks whose feature bytes are staged through LDS.  No atomics: the gradient is
// accumulated per block in LDS (a thread owns two columns and visits the rows in order) and the per-block
// tables are summed in block order by a second launch.
#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

namespace {

constexpr int OH_CHUNK = 128;           // rows whose feature bytes are staged at a time
constexpr size_t OH_LDS_MAX = 150 << 10;

__device__ __forceinline__ void stage_features(const uint8_t* __restrict__ feats, long row0, int n, int F, int Fw,
                                               uint32_t* fs) {
  // consecutive threads read consecutive bytes; the padding bytes of a staged row stay zero
  uint8_t* fb = reinterpret_cast<uint8_t*>(fs);
  const uint8_t* src = feats + (size_t)row0 * F;
  for (int idx = threadIdx.x; idx < n * F; idx += 256) {
    const int r = idx / F, j = idx - r * F;
    fb[r * Fw * 4 + j] = src[idx];
  }
}

__global__ __launch_bounds__(256) void onehot_linear_kernel(const uint8_t* __restrict__ feats, int n_rows, int F,
                                                            const float* __restrict__ Wt, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int ld, int H, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oh_smem[];
  float* Ws = reinterpret_cast<float*>(oh_smem);                       // [F][ldw]
  const int Fw = (F + 3) >> 2;
  uint32_t* fs = reinterpret_cast<uint32_t*>(Ws + (size_t)F * ldw);    // [OH_CHUNK][Fw]
  const int tid = threadIdx.x;
  const int w4 = ldw >> 2, h4 = ld >> 2;
  for (int i = tid; i < F * w4; i += 256) reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(Wt)[i];
  for (int i = tid; i < OH_CHUNK * Fw; i += 256) fs[i] = 0u;
  __syncthreads();
  const long r_beg = (long)blockIdx.x * rows_per_block;
  const long r_end = r_beg + rows_per_block < n_rows ? r_beg + rows_per_block : n_rows;
  const int slots = 256 / h4 > 0 ? 256 / h4 : 1;  // rows handled side by side (2 for ld = 452)
  const int slot = tid / h4, c4 = tid - slot * h4;
  float4 b4 = f4zero();
  if (bias && slot < slots) {
    const int c = 4 * c4;
    b4.x = c + 0 < H ? bias[c + 0] : 0.f;
    b4.y = c + 1 < H ? bias[c + 1] : 0.f;
    b4.z = c + 2 < H ? bias[c + 2] : 0.f;
    b4.w = c + 3 < H ? bias[c + 3] : 0.f;
  }
  for (long row0 = r_beg; row0 < r_end; row0 += OH_CHUNK) {
    const int n = (int)(r_end - row0 < OH_CHUNK ? r_end - row0 : OH_CHUNK);
    stage_features(feats, row0, n, F, Fw, fs);
    __syncthreads();
    if (slot < slots) {
      for (int lr = slot; lr < n; lr += slots) {
        float4 acc = f4zero();
        const uint32_t* fr = fs + lr * Fw;
        for (int w = 0; w < Fw; ++w) {
          const uint32_t word = fr[w];
          if (!word) continue;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t v = (word >> (8 * b)) & 255u;
            if (v) {
              const float4 t = reinterpret_cast<const float4*>(Ws)[(4 * w + b) * w4 + c4];
              const float f = (float)v;
              acc.x = fmaf(f, t.x, acc.x);
              acc.y = fmaf(f, t.y, acc.y);
              acc.z = fmaf(f, t.z, acc.z);
              acc.w = fmaf(f, t.w, acc.w);
            }
          }
        }
        *rowp(out, (int)(row0 + lr), ld, c4) = zero_pad(acc + b4, c4, H);
      }
    }
    __syncthreads();
  }
}

// per-block partial of dW^T: T[j][c] = sum over the block's rows r (in order) of feats[r][j] * dOut[r][c]
__global__ __launch_bounds__(256) void onehot_linear_bwd_kernel(const uint8_t* __restrict__ feats, int n_rows, int F,
                                                                const float* __restrict__ dOut, int ld, int H,
                                                                float* __restrict__ partial, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oh_smem[];
  float* Ts = reinterpret_cast<float*>(oh_smem);                      // [F][ld]
  const int Fw = (F + 3) >> 2;
  uint32_t* fs = reinterpret_cast<uint32_t*>(Ts + (size_t)F * ld);    // [OH_CHUNK][Fw]
  const int tid = threadIdx.x;
  const int h2 = ld >> 1;
  for (int i = tid; i < F * (ld >> 2); i += 256) reinterpret_cast<float4*>(Ts)[i] = f4zero();
  for (int i = tid; i < OH_CHUNK * Fw; i += 256) fs[i] = 0u;
  __syncthreads();
  const long r_beg = (long)blockIdx.x * rows_per_block;
  const long r_end = r_beg + rows_per_block < n_rows ? r_beg + rows_per_block : n_rows;
  // a thread owns columns [2 * c2, 2 * c2 + 2) of every table row; with more than 512 columns it owns several pairs
  for (long row0 = r_beg; row0 < r_end; row0 += OH_CHUNK) {
    const int n = (int)(r_end - row0 < OH_CHUNK ? r_end - row0 : OH_CHUNK);
    stage_features(feats, row0, n, F, Fw, fs);
    __syncthreads();
    for (int c2 = tid; c2 < h2; c2 += 256) {
      float2* tcol = reinterpret_cast<float2*>(Ts) + c2;
      const bool in0 = 2 * c2 < H, in1 = 2 * c2 + 1 < H;
      for (int lr0 = 0; lr0 < n; lr0 += 4) {
        float2 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four rows' loads in flight
          const int lr = lr0 + u < n ? lr0 + u : n - 1;
          d[u] = reinterpret_cast<const float2*>(dOut + (size_t)(row0 + lr) * ld)[c2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (lr0 + u >= n) break;
          const float dx = in0 ? d[u].x : 0.f, dy = in1 ? d[u].y : 0.f;
          const uint32_t* fr = fs + (lr0 + u) * Fw;
          for (int w = 0; w < Fw; ++w) {
            const uint32_t word = fr[w];
            if (!word) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const uint32_t v = (word >> (8 * b)) & 255u;
              if (v) {
                // LDS adds that return nothing: they queue behind each other in issue order, so a table entry
                // (touched by this thread only) sees its rows' contributions in row order, without a
                // read-modify-write round trip per term
                float* t = reinterpret_cast<float*>(tcol + (size_t)(4 * w + b) * h2);
                const float f = (float)v;
                (void)__hip_atomic_fetch_add(t, f * dx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(t + 1, f * dy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  float4* P = reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * F * ld);
  for (int i = tid; i < F * (ld >> 2); i += 256) P[i] = reinterpret_cast<const float4*>(Ts)[i];
}

// dW[n][j] (+)= sum over blocks (in block order) of partial[b][j][n]
__global__ __launch_bounds__(256) void onehot_reduce_kernel(const float* __restrict__ partial, int n_blocks, int F,
                                                            int ld, int H, float* __restrict__ dW, int lddw,
                                                            int accumulate) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)F * H) return;
  const int j = (int)(gid / H), n = (int)(gid - (long)j * H);
  const float* p = partial + (size_t)j * ld + n;
  const size_t stride = (size_t)F * ld;
  float acc = 0.f;
  int b = 0;
  for (; b + 8 <= n_blocks; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; b < n_blocks; ++b) acc += p[(size_t)b * stride];
  float* o = dW + (size_t)n * lddw + j;
  *o = accumulate ? *o + acc : acc;
}

inline size_t lds_bytes(int F, int ldt) { return (size_t)F * ldt * 4 + (size_t)OH_CHUNK * ((F + 3) / 4) * 4; }

inline int plan_blocks(int n_rows, int* rows_per_block) {
  int blocks = (n_rows + 255) / 256;  // at least 256 rows per block: the table load has to pay off
  if (blocks > 256) blocks = 256;     // one block per CU holds the table
  if (blocks < 1) blocks = 1;
  int rpb = (n_rows + blocks - 1) / blocks;
  rpb = (rpb + OH_CHUNK - 1) / OH_CHUNK * OH_CHUNK;
  *rows_per_block = rpb;
  return (n_rows + rpb - 1) / rpb;
}

inline int allow_lds(const void* kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int jtvae_onehot_linear(const uint8_t* feats, int32_t n_rows, int32_t F, const float* Wt, int32_t ldw,
                        const float* bias, float* out, int32_t ld, int32_t H, void* stream) {
  if (n_rows < 0 || F <= 0 || H <= 0 || ld < H || ldw < H) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!feats || !Wt || !out) return JTVAE_E_NULL;
  if ((ld & 3) || (ldw & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(Wt) & 15))
    return JTVAE_E_ALIGN;
  const size_t lds = lds_bytes(F, ldw);
  if (lds > OH_LDS_MAX || ld / 4 > 256) return JTVAE_E_WORKSPACE;  // table does not fit: use jtvae_gemm_nt
  int rc = allow_lds(reinterpret_cast<const void*>(onehot_linear_kernel), lds);
  if (rc) return rc;
  int rpb;
  const int blocks = plan_blocks(n_rows, &rpb);
  hipLaunchKernelGGL(onehot_linear_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, feats, n_rows, F, Wt, ldw,
                     bias, out, ld, H, rpb);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

size_t jtvae_onehot_linear_bwd_workspace_bytes(int32_t n_rows, int32_t F, int32_t ld) {
  if (n_rows <= 0 || F <= 0 || ld <= 0) return 0;
  int rpb;
  return (size_t)plan_blocks(n_rows, &rpb) * F * ld * sizeof(float);
}

int jtvae_onehot_linear_bwd(const uint8_t* feats, int32_t n_rows, int32_t F, const float* dOut, int32_t ld, int32_t H,
                            float* dW, int32_t lddw, int32_t accumulate, void* workspace, size_t workspace_bytes,
                            void* stream) {
  if (n_rows < 0 || F <= 0 || H <= 0 || ld < H || lddw < F) return JTVAE_E_SHAPE;
  if (!dW) return JTVAE_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)F * H;
  if (n_rows == 0) {
    if (!accumulate)
      hipLaunchKernelGGL(onehot_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, nullptr, 0, F, ld,
                         H, dW, lddw, 0);
    return 0;
  }
  if (!feats || !dOut || !workspace) return JTVAE_E_NULL;
  if ((ld & 3) || (reinterpret_cast<uintptr_t>(dOut) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return JTVAE_E_ALIGN;
  const size_t lds = lds_bytes(F, ld);
  if (lds > OH_LDS_MAX) return JTVAE_E_WORKSPACE;
  int rpb;
  const int blocks = plan_blocks(n_rows, &rpb);
  if (workspace_bytes < (size_t)blocks * F * ld * sizeof(float)) return JTVAE_E_WORKSPACE;
  int rc = allow_lds(reinterpret_cast<const void*>(onehot_linear_bwd_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(onehot_linear_bwd_kernel, dim3(blocks), dim3(256), lds, s, feats, n_rows, F, dOut, ld, H,
                     (float*)workspace, rpb);
  hipLaunchKernelGGL(onehot_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     (const float*)workspace, blocks, F, ld, H, dW, lddw, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
