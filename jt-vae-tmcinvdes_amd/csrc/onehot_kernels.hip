// Linear layers whose input rows are one-hot blocks (fast_jtnn/mpn.py:87,99; fast_jtnn/jtmpn.py:89,103: W_i on
// the bond feature rows, the atom-feature columns of W_o).  A feature row of 39..51 entries holds ~8 ones, so
//   out[r] = sum over the set entries j of W^T[j]                       (forward)
//   dW^T[j] += dOut[r] for the set entries j of row r                   (weight gradient)
// are row kernels bound by the 1.8 KB row they write / read, not products worth the matrix cores (a dense K = 51
// product spends 64-wide MFMA tiles on 8 useful terms).  The rows arrive as bytes (whole numbers 0..255, which is
// what the reference's feature extraction yields); an entry other than 0 / 1 multiplies its table row.
//
// Forward: the whole transposed weight ([F, ld] floats, 92 KB for F = 51) sits in LDS, one 1024-thread block per
// CU; blocks walk contiguous row ranges in chunks whose feature bytes are staged through LDS.  Gradient: thread
// groups own disjoint features and keep their accumulators in registers (see the kernel); the per-block results
// are summed in block order by a second launch.  No atomics anywhere: results are bitwise reproducible.
#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

namespace {

constexpr int OH_THREADS = 1024;        // 16 waves on the one block a CU holds: the latency of the LDS reads hides
constexpr int OH_CHUNK = 128;           // forward: rows whose feature bytes are staged at a time
constexpr int OH_WORDS = 16;            // feature rows of up to 64 entries
constexpr size_t OH_LDS_MAX = 150 << 10;

__device__ __forceinline__ void stage_features(const uint8_t* __restrict__ feats, long row0, int n, int F, int Fw,
                                               uint32_t* fs) {
  // consecutive threads read consecutive bytes; the padding bytes of a staged row stay zero
  uint8_t* fb = reinterpret_cast<uint8_t*>(fs);
  const uint8_t* src = feats + (size_t)row0 * F;
  for (int idx = threadIdx.x; idx < n * F; idx += OH_THREADS) {
    const int r = idx / F, j = idx - r * F;
    fb[r * Fw * 4 + j] = src[idx];
  }
}

__global__ __launch_bounds__(OH_THREADS) void onehot_linear_kernel(const uint8_t* __restrict__ feats, int n_rows, int F,
                                                                   const float* __restrict__ Wt, int ldw,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ out, int ld, int H,
                                                                   int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oh_smem[];
  float* Ws = reinterpret_cast<float*>(oh_smem);                       // [F][ldw]
  const int Fw = (F + 3) >> 2;
  uint32_t* fs = reinterpret_cast<uint32_t*>(Ws + (size_t)F * ldw);    // [OH_CHUNK][Fw]
  const int tid = threadIdx.x;
  const int w4 = ldw >> 2, h4 = ld >> 2;
  for (int i = tid; i < F * w4; i += OH_THREADS)
    reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(Wt)[i];
  for (int i = tid; i < OH_CHUNK * Fw; i += OH_THREADS) fs[i] = 0u;
  __syncthreads();
  const long r_beg = (long)blockIdx.x * rows_per_block;
  const long r_end = r_beg + rows_per_block < n_rows ? r_beg + rows_per_block : n_rows;
  const int slots = OH_THREADS / h4;  // rows handled side by side (9 for ld = 452)
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
        const uint32_t* fr = fs + lr * Fw;
        uint32_t word[OH_WORDS];
#pragma unroll
        for (int w = 0; w < OH_WORDS; ++w) word[w] = w < Fw ? fr[w] : 0u;  // all reads issued before the first use
        float4 acc = f4zero();
#pragma unroll
        for (int w = 0; w < OH_WORDS; ++w) {
          if (!word[w]) continue;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t v = (word[w] >> (8 * b)) & 255u;
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

// Weight gradient.  The block's threads form `slots` groups of ld / 4 threads; group s owns the features
// [s * FS, s * FS + FS) and a thread of it four columns: FS float4 accumulators in registers, nothing shared.
// Every group walks ALL rows of the block's range (a chunk of rows is staged in LDS once and read by each group),
// adding feats[r][j] * dOut[r] in row order: fixed order, no read-modify-write anywhere.
constexpr int OH_BWD_ROWS = 32;  // rows of dOut staged at a time
constexpr int OH_FS_MAX = 8;     // features per group

__global__ __launch_bounds__(OH_THREADS) void onehot_linear_bwd_kernel(const uint8_t* __restrict__ feats, int n_rows,
                                                                       int F, const float* __restrict__ dOut, int ld,
                                                                       int H, float* __restrict__ partial,
                                                                       int rows_per_block, int FS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char oh_smem[];
  float4* Ds = reinterpret_cast<float4*>(oh_smem);                               // [OH_BWD_ROWS][ld / 4]
  const int h4 = ld >> 2;
  uint8_t* fb = reinterpret_cast<uint8_t*>(Ds + (size_t)OH_BWD_ROWS * h4);       // [OH_BWD_ROWS][Fp] bytes
  const int slots = OH_THREADS / h4;
  const int Fp = slots * FS;  // staged feature rows are padded to the features the groups cover (zeros beyond F)
  const int tid = threadIdx.x;
  const int slot = tid / h4, c4 = tid - slot * h4;
  const int j0 = slot * FS;
  const long r_beg = (long)blockIdx.x * rows_per_block;
  const long r_end = r_beg + rows_per_block < n_rows ? r_beg + rows_per_block : n_rows;
  // two levels of accumulation (per staged chunk, then over the chunks): tens of thousands of rows added into one
  // float one after the other would lose digits that the slab sums of the dense weight gradient keep
  float4 acc[OH_FS_MAX], tot[OH_FS_MAX];
#pragma unroll
  for (int k = 0; k < OH_FS_MAX; ++k) acc[k] = tot[k] = f4zero();
  for (int i = tid; i < OH_BWD_ROWS * Fp; i += OH_THREADS) fb[i] = 0;
  __syncthreads();
  for (long row0 = r_beg; row0 < r_end; row0 += OH_BWD_ROWS) {
    const int n = (int)(r_end - row0 < OH_BWD_ROWS ? r_end - row0 : OH_BWD_ROWS);
    for (int i = tid; i < n * h4; i += OH_THREADS) {
      const int r = i / h4, c = i - r * h4;
      Ds[i] = zero_pad(*rowp(dOut, (int)(row0 + r), ld, c), c, H);
    }
    for (int i = tid; i < n * F; i += OH_THREADS) {
      const int r = i / F, j = i - r * F;
      fb[r * Fp + j] = feats[(size_t)row0 * F + i];
    }
    __syncthreads();
    if (slot < slots) {
      for (int r = 0; r < n; ++r) {
        const float4 d = Ds[r * h4 + c4];
        const uint8_t* fr = fb + r * Fp + j0;
#pragma unroll
        for (int k = 0; k < OH_FS_MAX; ++k) {
          if (k < FS) {
            const float f = (float)fr[k];
            acc[k].x = fmaf(f, d.x, acc[k].x);
            acc[k].y = fmaf(f, d.y, acc[k].y);
            acc[k].z = fmaf(f, d.z, acc[k].z);
            acc[k].w = fmaf(f, d.w, acc[k].w);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < OH_FS_MAX; ++k) {
        tot[k] = tot[k] + acc[k];
        acc[k] = f4zero();
      }
    }
    __syncthreads();
  }
  if (slot < slots) {
    float* P = partial + (size_t)blockIdx.x * F * ld;
#pragma unroll
    for (int k = 0; k < OH_FS_MAX; ++k)
      if (k < FS && j0 + k < F) *rowp(P, j0 + k, ld, c4) = tot[k];
  }
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
  int blocks = (n_rows + 511) / 512;  // at least 512 rows per block: the table load has to pay off
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
  if (lds > OH_LDS_MAX || ld / 4 > OH_THREADS || F > 4 * OH_WORDS) return JTVAE_E_WORKSPACE;  // use jtvae_gemm_nt
  int rc = allow_lds(reinterpret_cast<const void*>(onehot_linear_kernel), lds);
  if (rc) return rc;
  int rpb;
  const int blocks = plan_blocks(n_rows, &rpb);
  hipLaunchKernelGGL(onehot_linear_kernel, dim3(blocks), dim3(OH_THREADS), lds, (hipStream_t)stream, feats, n_rows, F, Wt, ldw,
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
  const int slots = OH_THREADS / (ld / 4);
  if (slots < 1) return JTVAE_E_WORKSPACE;
  const int FS = (F + slots - 1) / slots;
  const size_t lds = (size_t)OH_BWD_ROWS * ld * 4 + (size_t)OH_BWD_ROWS * slots * FS;
  if (FS > OH_FS_MAX || lds > OH_LDS_MAX) return JTVAE_E_WORKSPACE;  // use jtvae_gemm_tn
  int rpb;
  const int blocks = plan_blocks(n_rows, &rpb);
  if (workspace_bytes < (size_t)blocks * F * ld * sizeof(float)) return JTVAE_E_WORKSPACE;
  int rc = allow_lds(reinterpret_cast<const void*>(onehot_linear_bwd_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(onehot_linear_bwd_kernel, dim3(blocks), dim3(OH_THREADS), lds, s, feats, n_rows, F, dOut, ld, H,
                     (float*)workspace, rpb, FS);
  hipLaunchKernelGGL(onehot_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     (const float*)workspace, blocks, F, ld, H, dW, lddw, accumulate);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
