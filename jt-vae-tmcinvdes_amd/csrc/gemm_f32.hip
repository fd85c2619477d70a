// Dense contractions of the encoder path on the CDNA4 matrix cores.
//
// v_mfma_f32_32x32x2_f32 is an exact-fp32 fma chain (no reduced-precision fast
// path exists on gfx950), which is what the 1e-4 parity bound over 20 recurrent
// GRU iterations needs; it serves every product up to ~10 000 rows.  The products
// that fill the chip run in the split form instead (split3 below): fp32 operands
// cut into three bf16 parts, six bf16 matrix instructions per fp32 one, fp32-class
// accuracy at 1.4-1.5x the fp32 instruction's throughput.  Two kernels:
//
//   gemm_nt : C[M,N] = epilogue(A[M,K] . W[N,K]^T)        forward Linear, and dgrad
//                                                         with a transposed weight copy
//   gemm_tn : dW[N,K] = X[M,N]^T . Y[M,K]                 weight gradients, reduction over
//                                                         the (huge) row dimension, split
//                                                         into slabs summed in fixed order
//
// Both take optional row lists: the tree GRU only updates the messages that can still
// change in an iteration, so A rows are gathered and C rows scattered by message id, and the
// weight gradients contract over a list of (gradient row, saved-state row) pairs.
//
// N = K = 450 on the hot calls, M = 4e4..9e5.  Tiles are cut for that: the NT block
// is 128 rows x 96 columns (5 column blocks cover 480 >= 450: 6 % padding waste,
// against 14 % for 128/256-wide tiles); one wave owns 32 rows x 96 columns = three
// 32x32 accumulators, which keeps the kernel near 128 VGPRs so that four blocks
// (four waves per SIMD) are resident per CU: the matrix pipe of a SIMD then always has
// a wave with MFMAs to issue while the others sit in a barrier or stage the next tile.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "jtvae_common.h"
#include "../../include/jtvae_hip.h"

using namespace jtvae;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Optional in-kernel timeline of the tiled NT kernel (builds with -DJTVAE_GEMM_TRACE only): lane 0
// of every wave stamps the shader clock at the phase boundaries of its K loop into a buffer the
// tracing harness reads back.  Product builds compile the marks away.
#ifdef JTVAE_GEMM_TRACE
extern "C" __device__ unsigned long long* jtvae_trace_buf = nullptr;  // [blocks*4][JTVAE_TRACE_SLOTS]
#define JTVAE_TRACE_SLOTS 96
#define TRACE_MARK(slot)                                                                              \
  do {                                                                                                \
    if ((threadIdx.x & 63) == 0 && jtvae_trace_buf && (slot) < JTVAE_TRACE_SLOTS)                      \
      jtvae_trace_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * JTVAE_TRACE_SLOTS + (slot)] =   \
          __builtin_readcyclecounter();                                                               \
  } while (0)
#define TRACE_HWID(slot)                                                                              \
  do {                                                                                                \
    if ((threadIdx.x & 63) == 0 && jtvae_trace_buf)                                                    \
      jtvae_trace_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * JTVAE_TRACE_SLOTS + (slot)] =   \
          ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |                     \
          (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                                         \
  } while (0)
#else
#define TRACE_MARK(slot) do {} while (0)
#define TRACE_HWID(slot) do {} while (0)
#endif

// Slab partials of jtvae_gemm_tn_static: a fixed array in device memory, one copy per GPU, held for
// the life of the process.  64 MiB holds the partials of every N, K <= 450 product for any M (at most
// 64 slabs of 450 x 452 floats).  In host code `tn_scratch` is only a handle for the symbol: the
// address a kernel may use comes from hipGetSymbolAddress, once per device.
constexpr size_t TN_SCRATCH_BYTES = (size_t)64 << 20;
constexpr int TN_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char tn_scratch[TN_SCRATCH_BYTES];
static void* tn_scratch_ptr[TN_MAX_DEVICES];
static std::mutex tn_scratch_mutex;


namespace {

constexpr int NT_BM = 128;
constexpr int NT_BK = 32;
constexpr int NT_LDS = 36;  // floats per LDS row: 32 + 4 -> ds_read_b128 of 16 rows hits 16 distinct 16-B slots

// blocks that share an XCD (ids equal mod 8) get consecutive tiles, so the column
// blocks of one row block read their A rows through one L2
__device__ __forceinline__ int xcd_remap(int id, int n) {
  int q = n >> 3, r = n & 7;
  int x = id & 7;
  int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (id >> 3);
}

// ---------------------------------------------------------------------------------------
// Split form of the large products.  The fp32 matrix instruction runs at 1/16 of the bf16 one, so a product
// that fills the chip is computed on the bf16 pipe instead, at fp32 accuracy: every operand is written as the sum
// of three bf16 numbers, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 = 24
// significant bits, the residual is below 2^-25 |x|), and the six cross products x_i * y_j with i + j <= 4 are
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are below 2^-23 |x y|; measured against
// fp64 the result is as close as the fp32 fma chain (1.1e-7 vs 1.7e-7 of sum |x y| at K = 450).  Six bf16
// instructions of K = 16 replace eight fp32 instructions of K = 2 that take twice as long each.
__device__ __forceinline__ unsigned pack_bf16(float x, float y) {
  f32x2 v = {x, y};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ void split3(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pack_bf16(x, y);
  float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  p2 = pack_bf16(rx, ry);
  rx -= __uint_as_float(p2 << 16);
  ry -= __uint_as_float(p2 & 0xffff0000u);
  p3 = pack_bf16(rx, ry);
}
// LDS image of a split tile: three planes of [rows][32 k] bf16, 64-byte rows whose four 16-byte chunks are
// rotated by (row >> 2): the ds_read_b128 of a fragment (16-lane groups of rows that differ in row >> 2) and the
// ds_write_b64 of the staging (two adjacent rows per 16 lanes) are both free of bank conflicts without padding.
constexpr int SPLIT_ROW = 64;
__device__ __forceinline__ int split_off(int row, int chunk) { return row * SPLIT_ROW + 16 * ((chunk + (row >> 2)) & 3); }

struct NtArgs {
  const float* A;
  const float* W;
  float* C;
  int lda, ldw, ldc;
  int M, N, K;
  int n_tiles_n;
  int n_blocks;
  int evec;  // epilogue operands are float4-addressable
  jtvae_epilogue_t e;
  // EPI_GRU_BWD only: s rows (by message id, like e.aux0 = z and e.aux1 = p) and the compact outputs
  const float* bw_s;
  float* bw_dz;
  float* bw_dp;
};

// Internal epilogue mode of jtvae_gru_bwd_dh: the product is the gradient dh of the level's rows, and what the
// pointwise backward of the blend / sigmoid / tanh makes of it is written in the same pass (dz, dp compactly by
// launch row; ds by message id into e.out2).
constexpr int EPI_GRU_BWD = 3;

// Tile loads are split in two so that they stay in flight behind the MFMAs of the current
// tile: load4_raw issues an unconditional load from a clamped (always valid) address, mask4
// zeroes what lies outside the matrix once the value is consumed.  A bounds check wrapped around
// the load itself makes the compiler wait for each load separately, right where it is issued.
template <bool VEC>
__device__ __forceinline__ float4 load4_raw(const float* rowbase, int k, int K) {
  const int kc = k < K ? k : 0;
  const float* p = rowbase + kc;
  if (VEC) return *reinterpret_cast<const float4*>(p);  // k % 4 == 0 and ld % 4 == 0: inside the row
  const int last = K - 1 - kc;
  float4 v;
  v.x = p[0];
  v.y = p[min(1, last)];
  v.z = p[min(2, last)];
  v.w = p[min(3, last)];
  return v;
}
__device__ __forceinline__ float4 mask4(float4 v, bool row_ok, int k, int K) {
  v.x = (row_ok && k + 0 < K) ? v.x : 0.f;
  v.y = (row_ok && k + 1 < K) ? v.y : 0.f;
  v.z = (row_ok && k + 2 < K) ? v.z : 0.f;
  v.w = (row_ok && k + 3 < K) ? v.w : 0.f;
  return v;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case JTVAE_ACT_RELU: return v > 0.f ? v : 0.f;
    case JTVAE_ACT_SIGMOID: return sigmoidf_fast(v);
    case JTVAE_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Per-output-row state of the epilogue, shared by both NT kernels: three integers per row, so
// that a lane can hold several rows without crowding the accumulators out of the register file;
// the row pointers are rebuilt from them where they are used.
struct EpiRow {
  int m;      // id of the output row
  int add_m;  // row of the add operand
  int tau;    // MUL_ACC: slab of the multiplier row, -1 = no multiplier (nothing to accumulate)
              // GRU_BWD: slab of the row's saved state (0: the row has no s)
  int i;      // row of the launch (compact outputs are indexed by it)
};

template <int MODE>
__device__ __forceinline__ EpiRow epi_row(const NtArgs& a, int i) {
  const jtvae_epilogue_t& e = a.e;
  EpiRow R;
  R.m = e.c_rows ? e.c_rows[i] : i;
  R.add_m = (e.add && e.add_idx) ? e.add_idx[R.m] : R.m;
  R.tau = 0;
  R.i = i;
  if (MODE == JTVAE_EPI_MUL_ACC && e.aux_sf) {
    R.tau = min(e.aux_t, e.aux_sf[R.m] - 1);
    if (R.tau <= 0) R.tau = -1;  // slab 0 is all zero
  }
  if (MODE == EPI_GRU_BWD) R.tau = e.aux_sf ? min(e.aux_t, e.aux_sf[R.m] - 1) : e.aux_t;
  return R;
}
__device__ __forceinline__ float* epi_c_row(const NtArgs& a, const EpiRow& R) { return a.C + (size_t)R.m * a.ldc; }
__device__ __forceinline__ const float* epi_add_row(const NtArgs& a, const EpiRow& R) {
  return a.e.add ? a.e.add + (size_t)R.add_m * a.e.ld_add : nullptr;
}
__device__ __forceinline__ size_t epi_aux_off(const NtArgs& a, const EpiRow& R) { return (size_t)R.m * a.e.ld_aux; }
__device__ __forceinline__ const float* epi_mul_row(const NtArgs& a, const EpiRow& R) {
  return R.tau < 0 ? nullptr : a.e.aux0 + (size_t)R.tau * a.e.aux_slab_stride + epi_aux_off(a, R);
}

template <int MODE>
__device__ __forceinline__ void epi_store(const NtArgs& a, const EpiRow& R, int n, float v) {
  const jtvae_epilogue_t& e = a.e;
  float* c_row = epi_c_row(a, R);
  const size_t aux_off = epi_aux_off(a, R);
  if (n >= a.N) {
    // padding columns of the outputs are defined as zero
    if (n < a.ldc) c_row[n] = 0.f;
    if (MODE == JTVAE_EPI_PLAIN && e.pre_out && n < e.ld_pre) e.pre_out[(size_t)R.m * e.ld_pre + n] = 0.f;
    if (MODE == JTVAE_EPI_GRU_BLEND && n < e.ld_aux) e.out2[aux_off + n] = 0.f;
    if (MODE == EPI_GRU_BWD && n < e.ld_aux) {
      a.bw_dz[(size_t)R.i * e.ld_aux + n] = 0.f;
      a.bw_dp[(size_t)R.i * e.ld_aux + n] = 0.f;
      e.out2[aux_off + n] = 0.f;
    }
    return;
  }
  if (e.bias) v += e.bias[n];
  if (e.add) v += epi_add_row(a, R)[n];
  if (MODE == JTVAE_EPI_PLAIN) {
    if (e.pre_out) e.pre_out[(size_t)R.m * e.ld_pre + n] = v;
    c_row[n] = act_apply(v, e.act);
  } else if (MODE == JTVAE_EPI_GRU_BLEND) {
    const float p = tanhf(v);
    const float z = e.aux0[aux_off + n];
    const float s = e.aux1[aux_off + n];
    c_row[n] = p;
    e.out2[aux_off + n] = (R.m == e.zero_row) ? 0.f : fmaf(z, p - s, s);  // (1-z)s + zp
  } else if (MODE == JTVAE_EPI_MUL_ACC) {
    c_row[n] = v;
    const float* mul_row = epi_mul_row(a, R);
    if (mul_row) e.out2[aux_off + n] = fmaf(v, mul_row[n], e.out2[aux_off + n]);
  } else {  // EPI_GRU_BWD
    c_row[n] = v;
    const size_t st = (size_t)R.tau * e.aux_slab_stride + aux_off + n;
    const float d = R.m == e.zero_row ? 0.f : v;
    const float zz = e.aux0[st], pp = e.aux1[st];
    const float ss = R.tau > 0 ? a.bw_s[st] : 0.f;
    a.bw_dz[(size_t)R.i * e.ld_aux + n] = d * (pp - ss) * zz * (1.f - zz);
    a.bw_dp[(size_t)R.i * e.ld_aux + n] = d * zz * (1.f - pp * pp);
    e.out2[aux_off + n] = d * (1.f - zz);
  }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// zero the components at or past column count nv
__device__ __forceinline__ float4 keep_cols(float4 v, int nv) {
  v.x = nv > 0 ? v.x : 0.f;
  v.y = nv > 1 ? v.y : 0.f;
  v.z = nv > 2 ? v.z : 0.f;
  v.w = nv > 3 ? v.w : 0.f;
  return v;
}

// Vector form of the epilogue: J output rows x four consecutive columns [n, n+4) per lane.
// Every operand row is read and written as one float4 and all reads are issued before the
// first write, so the J rows' memory round trips overlap.  Needs 16-byte aligned operands
// (n % 4 == 0, leading dimensions % 4 == 0); the arithmetic and its order are those of epi_store.
template <int MODE, int J>
__device__ __forceinline__ void epi_store_vec(const NtArgs& a, const EpiRow (&R)[J], const bool (&ok)[J], int n,
                                              const float4 (&v)[J]) {
  const jtvae_epilogue_t& e = a.e;
  const int nv = a.N - n;  // columns of this float4 inside the matrix; <= 0: padding only
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e.bias) {
    b4.x = nv > 0 ? e.bias[n] : 0.f;
    b4.y = nv > 1 ? e.bias[n + 1] : 0.f;
    b4.z = nv > 2 ? e.bias[n + 2] : 0.f;
    b4.w = nv > 3 ? e.bias[n + 3] : 0.f;
  }
  const bool c_in = n < a.ldc;
  const bool aux_in = n < e.ld_aux;
  float4 add4[J], x0[J], x1[J], x2[MODE == EPI_GRU_BWD ? J : 1];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    add4[j] = x0[j] = x1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == EPI_GRU_BWD) x2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok[j]) continue;
    if (e.add && n < e.ld_add) add4[j] = ld4(epi_add_row(a, R[j]) + n);
    if (MODE == JTVAE_EPI_GRU_BLEND && aux_in) {
      x0[j] = ld4(e.aux0 + epi_aux_off(a, R[j]) + n);
      x1[j] = ld4(e.aux1 + epi_aux_off(a, R[j]) + n);
    }
    if (MODE == JTVAE_EPI_MUL_ACC && aux_in && R[j].tau >= 0) {
      x0[j] = ld4(epi_mul_row(a, R[j]) + n);
      x1[j] = ld4(e.out2 + epi_aux_off(a, R[j]) + n);
    }
    if (MODE == EPI_GRU_BWD && aux_in) {
      const size_t st = (size_t)R[j].tau * e.aux_slab_stride + epi_aux_off(a, R[j]) + n;
      x0[j] = ld4(e.aux0 + st);
      x1[j] = ld4(e.aux1 + st);
      if (R[j].tau > 0) x2[j] = ld4(a.bw_s + st);  // iteration 0 has s == 0 and keeps no s
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (!ok[j]) continue;
    float4 x = v[j];
    x.x = (x.x + b4.x) + add4[j].x;
    x.y = (x.y + b4.y) + add4[j].y;
    x.z = (x.z + b4.z) + add4[j].z;
    x.w = (x.w + b4.w) + add4[j].w;
    x = keep_cols(x, nv);
    if (MODE == JTVAE_EPI_PLAIN) {
      if (e.pre_out && n < e.ld_pre) st4(e.pre_out + (size_t)R[j].m * e.ld_pre + n, x);
      float4 y;
      y.x = act_apply(x.x, e.act);
      y.y = act_apply(x.y, e.act);
      y.z = act_apply(x.z, e.act);
      y.w = act_apply(x.w, e.act);
      if (c_in) st4(epi_c_row(a, R[j]) + n, keep_cols(y, nv));
    } else if (MODE == JTVAE_EPI_GRU_BLEND) {
      float4 pp;
      pp.x = tanhf(x.x);
      pp.y = tanhf(x.y);
      pp.z = tanhf(x.z);
      pp.w = tanhf(x.w);
      if (c_in) st4(epi_c_row(a, R[j]) + n, pp);  // tanh(0) = 0 keeps the padding columns zero
      if (aux_in) {
        float4 h;
        h.x = fmaf(x0[j].x, pp.x - x1[j].x, x1[j].x);  // (1-z)s + zp
        h.y = fmaf(x0[j].y, pp.y - x1[j].y, x1[j].y);
        h.z = fmaf(x0[j].z, pp.z - x1[j].z, x1[j].z);
        h.w = fmaf(x0[j].w, pp.w - x1[j].w, x1[j].w);
        if (R[j].m == e.zero_row) h = make_float4(0.f, 0.f, 0.f, 0.f);
        st4(e.out2 + epi_aux_off(a, R[j]) + n, keep_cols(h, nv));
      }
    } else if (MODE == JTVAE_EPI_MUL_ACC) {
      if (c_in) st4(epi_c_row(a, R[j]) + n, x);
      if (aux_in && R[j].tau >= 0) {
        float4 o;
        o.x = fmaf(x.x, x0[j].x, x1[j].x);
        o.y = fmaf(x.y, x0[j].y, x1[j].y);
        o.z = fmaf(x.z, x0[j].z, x1[j].z);
        o.w = fmaf(x.w, x0[j].w, x1[j].w);
        st4(e.out2 + epi_aux_off(a, R[j]) + n, o);
      }
    } else {  // EPI_GRU_BWD: the arithmetic of the stand-alone pointwise kernel on the value just formed
      if (c_in) st4(epi_c_row(a, R[j]) + n, x);
      if (aux_in) {
        using namespace jtvae;
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 d = R[j].m == e.zero_row ? f4zero() : x;  // backward of "* mask"
        const float4 zz = x0[j], pp = x1[j], ss = x2[j];
        const float4 dz = d * (pp - ss) * zz * (one - zz);
        const float4 dp = d * zz * (one - pp * pp);
        const float4 dsv = d * (one - zz);
        st4(a.bw_dz + (size_t)R[j].i * e.ld_aux + n, keep_cols(dz, nv));
        st4(a.bw_dp + (size_t)R[j].i * e.ld_aux + n, keep_cols(dp, nv));
        st4(e.out2 + epi_aux_off(a, R[j]) + n, keep_cols(dsv, nv));
      }
    }
  }
}

constexpr int NT_EPI_LD = 40;  // LDS row stride of a wave's 32 x 32 staging tile: the two half-waves
                               // (rows r and r + 4) land in disjoint bank halves

// NACC = 32-column accumulators per wave: 3 (block tile 128 x 96) for launches that fill the chip, 1
// (128 x 32) for mid-size launches, where three times as many, shorter-lived blocks cut the latency
// chain of a block that has a CU to itself.
template <int NACC, bool SPLIT>
constexpr int nt_smem_floats() {
  return SPLIT ? 3 * (NT_BM + 32 * NACC) * SPLIT_ROW / 4 : (NT_BM + 32 * NACC) * NT_LDS;
}

template <bool VEC, int MODE, bool EVEC, int NACC, bool SPLIT = false>
__device__ __forceinline__ void gemm_nt_body(const NtArgs& a, const int block_id, float* smem) {
  constexpr int BN = 32 * NACC;
  static_assert(4 * 32 * NT_EPI_LD <= nt_smem_floats<NACC, SPLIT>(), "epilogue staging must fit the operand tiles");
  float* As = smem;
  float* Bs = smem + NT_BM * NT_LDS;
  unsigned char* planes = reinterpret_cast<unsigned char*>(smem);  // SPLIT: plane p at p * PLANE
  constexpr int PLANE = (NT_BM + BN) * SPLIT_ROW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tile = xcd_remap(block_id, a.n_blocks);
  const int m0 = (tile / a.n_tiles_n) * NT_BM;
  const int n0 = (tile % a.n_tiles_n) * BN;
  const jtvae_epilogue_t& e = a.e;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int srow = tid >> 3;  // staging: 8 threads (8 x float4 = 32 floats) per tile row
  const int sc4 = tid & 7;
  float4 ra[NT_BM / 32], rb[NACC];

  // row base pointers of the staged rows do not change along K: resolve row lists once
  const float* a_row[NT_BM / 32];
  const float* b_row[NACC];
#pragma unroll
  for (int i = 0; i < NT_BM / 32; ++i) {
    int m = min(m0 + srow + 32 * i, a.M - 1);
    if (e.a_rows) m = e.a_rows[m];
    a_row[i] = a.A + (size_t)m * a.lda;
  }
#pragma unroll
  for (int i = 0; i < NACC; ++i) b_row[i] = a.W + (size_t)min(n0 + srow + 32 * i, a.N - 1) * a.ldw;

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NT_BM / 32; ++i) ra[i] = load4_raw<VEC>(a_row[i], k0 + 4 * sc4, a.K);
#pragma unroll
    for (int i = 0; i < NACC; ++i) rb[i] = load4_raw<VEC>(b_row[i], k0 + 4 * sc4, a.K);
  };
  // SPLIT: the four values of a staged float4 become three 8-byte groups, one per plane
  auto put_split = [&](int row, float4 v) {
    unsigned p1a, p2a, p3a, p1b, p2b, p3b;
    split3(v.x, v.y, p1a, p2a, p3a);
    split3(v.z, v.w, p1b, p2b, p3b);
    const int off = split_off(row, sc4 >> 1) + 8 * (sc4 & 1);
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(p1a, p1b);
    *reinterpret_cast<uint2*>(planes + PLANE + off) = make_uint2(p2a, p2b);
    *reinterpret_cast<uint2*>(planes + 2 * PLANE + off) = make_uint2(p3a, p3b);
  };
  auto store_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NT_BM / 32; ++i) {
      const float4 v = mask4(ra[i], m0 + srow + 32 * i < a.M, k0 + 4 * sc4, a.K);
      if (SPLIT)
        put_split(srow + 32 * i, v);
      else
        *reinterpret_cast<float4*>(&As[(srow + 32 * i) * NT_LDS + 4 * sc4]) = v;
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const float4 v = mask4(rb[i], n0 + srow + 32 * i < a.N, k0 + 4 * sc4, a.K);
      if (SPLIT)
        put_split(NT_BM + srow + 32 * i, v);
      else
        *reinterpret_cast<float4*>(&Bs[(srow + 32 * i) * NT_LDS + 4 * sc4]) = v;
    }
  };

  const int nk = (a.K + NT_BK - 1) / NT_BK;
  TRACE_HWID(0);
  TRACE_MARK(1);
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  TRACE_MARK(2);

  // operand fragment addresses: lane l supplies row (l & 31); its half (l >> 5) picks which
  // four of the eight k values of a step group it feeds.  The same k permutation is used
  // for A and B, so each MFMA sums matching products.
  const float* a_frag = &As[(wave * 32 + (lane & 31)) * NT_LDS + 4 * (lane >> 5)];
  const float* b_frag = &Bs[(lane & 31) * NT_LDS + 4 * (lane >> 5)];

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tiles((kt + 1) * NT_BK);  // in flight during the MFMAs below
    if (SPLIT) {
      // lane (r, h) supplies row r and, in k step s, the eight k values of chunk 2s + h; the six cross products
      // go into the accumulator smallest first
      const int r = lane & 31, h = lane >> 5;
      const int arow = wave * 32 + r;
      const int ks_end = min(NT_BK / 16, (a.K - kt * NT_BK + 15) >> 4);
#pragma unroll
      for (int ks = 0; ks < NT_BK / 16; ++ks) {
        if (ks >= ks_end) break;
        const int ao = split_off(arow, 2 * ks + h);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(planes + ao);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(planes + PLANE + ao);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(planes + 2 * PLANE + ao);
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
          const int bo = split_off(NT_BM + 32 * t + r, 2 * ks + h);
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(planes + bo);
          const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(planes + PLANE + bo);
          const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(planes + 2 * PLANE + bo);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
        }
      }
    }
    // the last tile of K = 450 holds two valid columns: skip its empty 8-column groups
    const int kk_end = SPLIT ? 0 : min(NT_BK / 8, (a.K - kt * NT_BK + 7) >> 3);
#pragma unroll
    for (int kk = 0; kk < NT_BK / 8; ++kk) {
      if (kk >= kk_end) break;
      float4 av = *reinterpret_cast<const float4*>(a_frag + 8 * kk);
      float4 bv[NACC];
#pragma unroll
      for (int t = 0; t < NACC; ++t) bv[t] = *reinterpret_cast<const float4*>(b_frag + t * 32 * NT_LDS + 8 * kk);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
    }
    TRACE_MARK(3 + 4 * kt);  // MFMAs of this tile issued
    __syncthreads();
    TRACE_MARK(4 + 4 * kt);  // every wave of the block is past its MFMAs
    if (more) {
      store_tiles((kt + 1) * NT_BK);
      TRACE_MARK(5 + 4 * kt);  // next tile arrived and written to LDS
      __syncthreads();
    }
    TRACE_MARK(6 + 4 * kt);  // next tile staged by every wave
  }

  // epilogue: accumulator register r of lane l is C[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31]
  if (EVEC) {
    // Turn each 32 x 32 accumulator through the wave's own LDS patch (the operand tiles are dead
    // after the loop's last barrier) so that a lane ends up with float4s along the rows: the
    // epilogue operands and results then move as 16-byte accesses, 128 contiguous bytes per
    // eight lanes.  A wave's LDS instructions execute in order: no barrier is needed.
    float* stage = smem + wave * (32 * NT_EPI_LD);
    EpiRow R[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = m0 + wave * 32 + (lane >> 3) + 8 * j;
      ok[j] = i < a.M;
      R[j] = epi_row<MODE>(a, ok[j] ? i : 0);
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * NT_EPI_LD + (lane & 31)] = acc[t][r];
      __builtin_amdgcn_wave_barrier();
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ld4(&stage[((lane >> 3) + 8 * j) * NT_EPI_LD + 4 * (lane & 7)]);
      __builtin_amdgcn_wave_barrier();
      TRACE_MARK(70 + 2 * t);  // accumulator turned through LDS
      // two rows at a time: their operand reads overlap, and the register footprint stays under
      // the 128-VGPR budget of four resident blocks
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const EpiRow R2[2] = {R[j], R[j + 1]};
        const bool ok2[2] = {ok[j], ok[j + 1]};
        const float4 v2[2] = {v[j], v[j + 1]};
        epi_store_vec<MODE, 2>(a, R2, ok2, n0 + 32 * t + 4 * (lane & 7), v2);
      }
      TRACE_MARK(71 + 2 * t);  // its results issued to memory
    }
  } else {
    const int col_in_tile = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (i >= a.M) continue;
      const EpiRow R = epi_row<MODE>(a, i);
#pragma unroll
      for (int t = 0; t < NACC; ++t) epi_store<MODE>(a, R, n0 + 32 * t + col_in_tile, acc[t][r]);
    }
  }
  TRACE_MARK(80);  // results written
}

template <bool VEC, int MODE, bool EVEC, int NACC>
__global__ __launch_bounds__(256, 4) void gemm_nt_kernel(NtArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[(NT_BM + 32 * NACC) * NT_LDS];
  gemm_nt_body<VEC, MODE, EVEC, NACC>(a, blockIdx.x, smem);
}

// the split form: 42 KB of LDS and ~140 registers, three blocks per CU
template <int MODE, int NACC>
__global__ __launch_bounds__(256, 3) void gemm_nt_split_kernel(NtArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[nt_smem_floats<NACC, true>()];
  gemm_nt_body<true, MODE, true, NACC, true>(a, blockIdx.x, smem);
}

// Two independent products in one launch (blocks [0, a.n_blocks) serve the first): mid-size launches
// leave most CUs idle, so two of them that do not depend on each other share the chip.
template <int MODE_A, int MODE_B, int NACC>
__global__ __launch_bounds__(256, 4) void gemm_nt_pair_kernel(NtArgs a, NtArgs b) {
  __shared__ __attribute__((aligned(16))) float smem[(NT_BM + 32 * NACC) * NT_LDS];
  if ((int)blockIdx.x < a.n_blocks)
    gemm_nt_body<true, MODE_A, true, NACC>(a, blockIdx.x, smem);
  else
    gemm_nt_body<true, MODE_B, true, NACC>(b, blockIdx.x - a.n_blocks, smem);
}

// ---------------------------------------------------------------------------------------
// Small-M variant.  The tail iterations of the tree GRU touch a few to a few hundred rows; with
// the tiled kernel such a launch is one or two blocks walking 15 dependent K tiles (load ->
// LDS -> MFMA -> barrier), ~45 us of pure latency.  Here one block owns a 32 x 32 output tile,
// pulls the whole K extent of its A and W rows into LDS with every load in flight at once,
// splits K over its four waves and adds the four partial tiles through LDS: one memory round
// trip instead of fifteen.  K <= SM_KMAX, aligned operands.
constexpr int SM_KMAX = 456;
constexpr int SM_LD = 460;                 // LDS row stride (floats): 115 float4, odd -> conflict-free b128 reads
constexpr int SM_C4 = SM_KMAX / 4;         // float4 per staged row
constexpr int SM_PER_THREAD = (32 * SM_C4 + 255) / 256;

template <int MODE>
__device__ __forceinline__ void gemm_nt_small_body(const NtArgs& a, const int block_id, float* sm) {
  float* As = sm;
  float* Ws = sm + 32 * SM_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = (block_id / a.n_tiles_n) * 32;
  const int n0 = (block_id % a.n_tiles_n) * 32;
  const jtvae_epilogue_t& e = a.e;
  const int c4n = (a.K + 3) >> 2;  // float4 per row that hold data

  float4 va[SM_PER_THREAD], vw[SM_PER_THREAD];
#pragma unroll
  for (int u = 0; u < SM_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    const int row = min(idx / SM_C4, 31), c4 = idx % SM_C4;
    int m = min(m0 + row, a.M - 1);
    if (e.a_rows) m = e.a_rows[m];
    const int n = min(n0 + row, a.N - 1);
    const int k = c4 < c4n ? 4 * c4 : 0;
    va[u] = *reinterpret_cast<const float4*>(a.A + (size_t)m * a.lda + k);
    vw[u] = *reinterpret_cast<const float4*>(a.W + (size_t)n * a.ldw + k);
  }
#pragma unroll
  for (int u = 0; u < SM_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    if (idx < 32 * SM_C4) {
      const int row = idx / SM_C4, c4 = idx % SM_C4;
      const bool in_k = c4 < c4n;
      *reinterpret_cast<float4*>(&As[row * SM_LD + 4 * c4]) = mask4(va[u], in_k && m0 + row < a.M, 4 * c4, a.K);
      *reinterpret_cast<float4*>(&Ws[row * SM_LD + 4 * c4]) = mask4(vw[u], in_k && n0 + row < a.N, 4 * c4, a.K);
    }
  }
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* a_frag = &As[(lane & 31) * SM_LD + 4 * (lane >> 5)];
  const float* w_frag = &Ws[(lane & 31) * SM_LD + 4 * (lane >> 5)];
  const int groups = (a.K + 7) >> 3;
  for (int g = wave; g < groups; g += 4) {
    const float4 av = *reinterpret_cast<const float4*>(a_frag + 8 * g);
    const float4 bv = *reinterpret_cast<const float4*>(w_frag + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
  __syncthreads();  // operands no longer needed: reuse their LDS for the four partial tiles
  float* red = sm + wave * (32 * 33);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[r];
  __syncthreads();
  const int row = tid >> 3, nq = (tid & 7) * 4;
  if (m0 + row < a.M) {
    const EpiRow R = epi_row<MODE>(a, m0 + row);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = row * 33 + nq + j;
      // fixed order: wave 0 + wave 1 + wave 2 + wave 3
      v[j] = ((sm[o] + sm[o + 32 * 33]) + sm[o + 2 * 32 * 33]) + sm[o + 3 * 32 * 33];
    }
    if (a.evec) {
      const EpiRow R1[1] = {R};
      const bool ok1[1] = {true};
      const float4 v1[1] = {make_float4(v[0], v[1], v[2], v[3])};
      epi_store_vec<MODE, 1>(a, R1, ok1, n0 + nq, v1);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) epi_store<MODE>(a, R, n0 + nq + j, v[j]);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_nt_small_kernel(NtArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * 32 * SM_LD];  // 115 KiB: one block per CU
  gemm_nt_small_body<MODE>(a, blockIdx.x, sm);
}

template <int MODE_A, int MODE_B>
__global__ __launch_bounds__(256, 1) void gemm_nt_small_pair_kernel(NtArgs a, NtArgs b) {
  __shared__ __attribute__((aligned(16))) float sm[2 * 32 * SM_LD];
  if ((int)blockIdx.x < a.n_blocks)
    gemm_nt_small_body<MODE_A>(a, blockIdx.x, sm);
  else
    gemm_nt_small_body<MODE_B>(b, blockIdx.x - a.n_blocks, sm);
}

// ---------------------------------------------------------------------------------------
// GRU update of one level in ONE launch (fast_jtnn/jtnn_enc.py:115,124-126; fast_jtnn/nnutils.py:59-70):
//   z = sigmoid(S . Wz^T + Pz[wid]),  p = tanh(G . Wh^T + Ph[wid]),  h' = (1 - z) s + z p
// Two products with different operands feed one epilogue, so a block keeps two accumulators for its
// output tile and runs the K loop twice (same LDS tiles, same summation order as gemm_nt); z never
// makes a round trip through memory before the blend.  Tiled form: 128 x 32 tiles (levels of a few
// hundred to ~10 000 rows); small form: one 32 x 32 tile per block with the whole K extent in LDS.
struct GruUpArgs {
  const float* S;
  const float* G;
  const float* Wz;
  const float* Wh;
  const float* Pz;
  const float* Ph;
  const int* wid;
  const int* rows;  // nullable: message id of launch row i
  float* Z;
  float* P;
  float* Hout;
  int ld, ldw, ldp;
  int M, N, K, zero_row;
  int n_tiles_n, n_blocks;
};

// K loop of the 128 x 32 tile, operands float4-addressable: acc += A[rows, :] . W[n0.., :]^T
__device__ __forceinline__ void nt_kloop_narrow(const float* A, int lda, const float* W, int ldw, const int* a_rows,
                                                int M, int N, int K, int m0, int n0, float* As, float* Bs,
                                                f32x16& acc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int srow = tid >> 3, sc4 = tid & 7;
  float4 ra[NT_BM / 32], rb;
  const float* a_row[NT_BM / 32];
#pragma unroll
  for (int i = 0; i < NT_BM / 32; ++i) {
    int m = min(m0 + srow + 32 * i, M - 1);
    if (a_rows) m = a_rows[m];
    a_row[i] = A + (size_t)m * lda;
  }
  const float* b_row = W + (size_t)min(n0 + srow, N - 1) * ldw;
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NT_BM / 32; ++i) ra[i] = load4_raw<true>(a_row[i], k0 + 4 * sc4, K);
    rb = load4_raw<true>(b_row, k0 + 4 * sc4, K);
  };
  auto store_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NT_BM / 32; ++i)
      *reinterpret_cast<float4*>(&As[(srow + 32 * i) * NT_LDS + 4 * sc4]) =
          mask4(ra[i], m0 + srow + 32 * i < M, k0 + 4 * sc4, K);
    *reinterpret_cast<float4*>(&Bs[srow * NT_LDS + 4 * sc4]) = mask4(rb, n0 + srow < N, k0 + 4 * sc4, K);
  };
  const int nk = (K + NT_BK - 1) / NT_BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const float* a_frag = &As[(wave * 32 + (lane & 31)) * NT_LDS + 4 * (lane >> 5)];
  const float* b_frag = &Bs[(lane & 31) * NT_LDS + 4 * (lane >> 5)];
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tiles((kt + 1) * NT_BK);
    const int kk_end = min(NT_BK / 8, (K - kt * NT_BK + 7) >> 3);
#pragma unroll
    for (int kk = 0; kk < NT_BK / 8; ++kk) {
      if (kk >= kk_end) break;
      const float4 av = *reinterpret_cast<const float4*>(a_frag + 8 * kk);
      const float4 bv = *reinterpret_cast<const float4*>(b_frag + 8 * kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_tiles((kt + 1) * NT_BK);
      __syncthreads();
    }
  }
}

// z, p, h' of one output row at columns [n, n + 4) from the two pre-activation sums
__device__ __forceinline__ void gru_up_store(const GruUpArgs& a, int m, int n, float4 vz, float4 vp) {
  if (n >= a.ld) return;
  const int nv = a.N - n;
  const int w = a.wid[m];
  const float4 pz = ld4(a.Pz + (size_t)w * a.ldp + n);
  const float4 ph = ld4(a.Ph + (size_t)w * a.ldp + n);
  const float4 sv = ld4(a.S + (size_t)m * a.ld + n);
  float4 z, pp, h;
  z.x = sigmoidf_fast(vz.x + pz.x);
  z.y = sigmoidf_fast(vz.y + pz.y);
  z.z = sigmoidf_fast(vz.z + pz.z);
  z.w = sigmoidf_fast(vz.w + pz.w);
  pp.x = tanhf(vp.x + ph.x);
  pp.y = tanhf(vp.y + ph.y);
  pp.z = tanhf(vp.z + ph.z);
  pp.w = tanhf(vp.w + ph.w);
  h.x = fmaf(z.x, pp.x - sv.x, sv.x);  // (1 - z) s + z p
  h.y = fmaf(z.y, pp.y - sv.y, sv.y);
  h.z = fmaf(z.z, pp.z - sv.z, sv.z);
  h.w = fmaf(z.w, pp.w - sv.w, sv.w);
  if (m == a.zero_row) h = make_float4(0.f, 0.f, 0.f, 0.f);
  st4(a.Z + (size_t)m * a.ld + n, keep_cols(z, nv));
  st4(a.P + (size_t)m * a.ld + n, keep_cols(pp, nv));
  st4(a.Hout + (size_t)m * a.ld + n, keep_cols(h, nv));
}

__global__ __launch_bounds__(256, 4) void gru_update_kernel(GruUpArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[(NT_BM + 32) * NT_LDS];
  float* As = smem;
  float* Bs = smem + NT_BM * NT_LDS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = xcd_remap(blockIdx.x, a.n_blocks);
  const int m0 = (tile / a.n_tiles_n) * NT_BM;
  const int n0 = (tile % a.n_tiles_n) * 32;
  f32x16 acc_z, acc_p;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_z[r] = acc_p[r] = 0.f;
  nt_kloop_narrow(a.S, a.ld, a.Wz, a.ldw, a.rows, a.M, a.N, a.K, m0, n0, As, Bs, acc_z);
  nt_kloop_narrow(a.G, a.ld, a.Wh, a.ldw, a.rows, a.M, a.N, a.K, m0, n0, As, Bs, acc_p);
  // both accumulators through the wave's own LDS patch (the operand tiles are dead after the last barrier
  // of the loop), so that a lane holds float4s along rows
  float* stage = smem + wave * (32 * NT_EPI_LD);
  float4 vz[4], vp[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * NT_EPI_LD + (lane & 31)] = acc_z[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) vz[j] = ld4(&stage[((lane >> 3) + 8 * j) * NT_EPI_LD + 4 * (lane & 7)]);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * NT_EPI_LD + (lane & 31)] = acc_p[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) vp[j] = ld4(&stage[((lane >> 3) + 8 * j) * NT_EPI_LD + 4 * (lane & 7)]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = m0 + wave * 32 + (lane >> 3) + 8 * j;
    if (i >= a.M) continue;
    const int m = a.rows ? a.rows[i] : i;
    gru_up_store(a, m, n0 + 4 * (lane & 7), vz[j], vp[j]);
  }
}

// one product of the small form: the 32 x 32 tile of A[rows] . W^T, K split over the four waves, partial
// tiles added through LDS in fixed order; returns this thread's four consecutive columns of row (tid >> 3)
__device__ __forceinline__ float4 nt_small_product(const float* A, int lda, const float* W, int ldw, const int* a_rows,
                                                   int M, int N, int K, int m0, int n0, float* sm) {
  float* As = sm;
  float* Ws = sm + 32 * SM_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4n = (K + 3) >> 2;
  float4 va[SM_PER_THREAD], vw[SM_PER_THREAD];
#pragma unroll
  for (int u = 0; u < SM_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    const int row = min(idx / SM_C4, 31), c4 = idx % SM_C4;
    int m = min(m0 + row, M - 1);
    if (a_rows) m = a_rows[m];
    const int n = min(n0 + row, N - 1);
    const int k = c4 < c4n ? 4 * c4 : 0;
    va[u] = *reinterpret_cast<const float4*>(A + (size_t)m * lda + k);
    vw[u] = *reinterpret_cast<const float4*>(W + (size_t)n * ldw + k);
  }
#pragma unroll
  for (int u = 0; u < SM_PER_THREAD; ++u) {
    const int idx = tid + 256 * u;
    if (idx < 32 * SM_C4) {
      const int row = idx / SM_C4, c4 = idx % SM_C4;
      const bool in_k = c4 < c4n;
      *reinterpret_cast<float4*>(&As[row * SM_LD + 4 * c4]) = mask4(va[u], in_k && m0 + row < M, 4 * c4, K);
      *reinterpret_cast<float4*>(&Ws[row * SM_LD + 4 * c4]) = mask4(vw[u], in_k && n0 + row < N, 4 * c4, K);
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* a_frag = &As[(lane & 31) * SM_LD + 4 * (lane >> 5)];
  const float* w_frag = &Ws[(lane & 31) * SM_LD + 4 * (lane >> 5)];
  const int groups = (K + 7) >> 3;
  for (int g = wave; g < groups; g += 4) {
    const float4 av = *reinterpret_cast<const float4*>(a_frag + 8 * g);
    const float4 bv = *reinterpret_cast<const float4*>(w_frag + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
  }
  __syncthreads();
  float* red = sm + wave * (32 * 33);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[r];
  __syncthreads();
  const int o = (tid >> 3) * 33 + (tid & 7) * 4;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = ((sm[o + j] + sm[o + j + 32 * 33]) + sm[o + j + 2 * 32 * 33]) + sm[o + j + 3 * 32 * 33];
  __syncthreads();  // the LDS is free for the next product
  return make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256, 1) void gru_update_small_kernel(GruUpArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * 32 * SM_LD];
  const int m0 = (blockIdx.x / a.n_tiles_n) * 32;
  const int n0 = (blockIdx.x % a.n_tiles_n) * 32;
  const float4 vz = nt_small_product(a.S, a.ld, a.Wz, a.ldw, a.rows, a.M, a.N, a.K, m0, n0, sm);
  const float4 vp = nt_small_product(a.G, a.ld, a.Wh, a.ldw, a.rows, a.M, a.N, a.K, m0, n0, sm);
  const int i = m0 + (threadIdx.x >> 3);
  if (i < a.M) gru_up_store(a, a.rows ? a.rows[i] : i, n0 + 4 * (threadIdx.x & 7), vz, vp);
}

// ---------------------------------------------------------------------------------------
// gemm_tn: partial[s][n][k] = sum_{m in slab s} X[xr(m)][n] * Y[yr(m)][k]
// Block tile 128 (n) x 96 (k) (64 / 32 for narrow feature operands); wave w owns n rows [32w, 32w+32) and one
// 32x32 accumulator per 32 k columns.
// Operand tiles sit in LDS row-major as loaded (m-major): for one MFMA step lane l needs
// X[m = 2*step + (l>>5)][n = l&31], i.e. 32 consecutive floats per half-wave: conflict-free
// ds_read_b32 with no transpose anywhere.
constexpr int TN_BN = 128;
constexpr int TN_BM = 32;

struct TnArgs {
  const float* X;
  const float* Y;
  const int* x_rows;  // nullable: row of X for reduction index m
  const int* y_rows;  // nullable: row of Y for reduction index m
  float* P;           // [splits][N][ldp]
  int ldx, ldy, ldp;
  int M, N, K;
  int tiles_n, tiles_k, rows_per_split;
  int ones_col;  // 1: Y carries a virtual column K of ones, so P[:, K] = column sums of X (bias gradient)
};

// mask4 for the Y tile: the virtual ones column sits at k == K
__device__ __forceinline__ float4 mask4_ones(float4 v, bool row_ok, int k, int K, int ones_col) {
  v = mask4(v, row_ok, k, K);
  if (ones_col && row_ok) {
    if (k + 0 == K) v.x = 1.f;
    if (k + 1 == K) v.y = 1.f;
    if (k + 2 == K) v.z = 1.f;
    if (k + 3 == K) v.w = 1.f;
  }
  return v;
}

// NACC = 32-column accumulators per wave (k tile 32 * NACC): 3..5 for the wide products, whichever pads K least,
// 2 / 1 when the reduced-over operand is a narrow feature matrix (K + bias column <= 64 / 32): no MFMA is spent
// on padding columns.
// SPLIT (the split form, see split3): the tiles are kept as three bf16 planes in the same m-major layout, and since
// the bf16 instruction wants eight consecutive m per lane - a column of the image - the fragments come through the
// transposing LDS read (ds_read_b64_tr_b16: 16 lanes fetch a 4 m x 16 column block and each receives one column).
// Plane rows are pitched so that the four rows of a block fall into different quarters of the 64 banks.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int TN_XPITCH = TN_BN * 2 + 64;  // bytes per m row of an X plane (256 + 64)
template <int NACC>
constexpr int tn_ypitch() { return (64 * NACC) % 256 == 0 ? 64 * NACC + 64 : 64 * NACC; }
template <int NACC, bool SPLIT>
constexpr int tn_smem_bytes() {
  return SPLIT ? 3 * TN_BM * (TN_XPITCH + tn_ypitch<NACC>()) : 4 * TN_BM * (TN_BN + 32 * NACC);
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <bool VEC, int NACC, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? (NACC >= 4 ? 2 : 3) : (NACC >= 4 ? 3 : 2)) void gemm_tn_kernel(TnArgs a) {
  constexpr int BKK = 32 * NACC;  // k tile
  constexpr int Y4 = 8 * NACC;    // float4 per staged Y row
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[tn_smem_bytes<NACC, SPLIT>()];
  float* Xs = reinterpret_cast<float*>(smem_raw);
  float* Ys = Xs + TN_BM * TN_BN;
  constexpr int YPITCH = tn_ypitch<NACC>();
  constexpr int XPLANE = TN_BM * TN_XPITCH, YPLANE = TN_BM * YPITCH;
  unsigned char* Xp = smem_raw;               // SPLIT: X planes, then Y planes
  unsigned char* Yp = smem_raw + 3 * XPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = a.tiles_n * a.tiles_k;
  // the tiles of one slab re-read the slab's X and Y rows: keep them on one XCD so that its L2 serves the re-reads
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid / tiles;
  const int tile = bid % tiles;
  const int n0 = (tile / a.tiles_k) * TN_BN;
  const int k0 = (tile % a.tiles_k) * BKK;
  const int m_beg = split * a.rows_per_split;
  const int m_end = min(a.M, m_beg + a.rows_per_split);

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 rx[4], ry[NACC];
  int ix[4], iy[NACC];  // source rows of the NEXT tile to load (row lists resolved one tile ahead)
  auto load_rows = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = min(mb + ((tid + 256 * i) >> 5), m_end - 1);
      ix[i] = a.x_rows ? a.x_rows[m] : m;
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      int m = min(mb + (tid + 256 * i) / Y4, m_end - 1);
      iy[i] = a.y_rows ? a.y_rows[m] : m;
    }
  };
  auto load_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rx[i] = load4_raw<VEC>(a.X + (size_t)ix[i] * a.ldx, n0 + 4 * ((tid + 256 * i) & 31), a.N);
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      ry[i] = load4_raw<VEC>(a.Y + (size_t)iy[i] * a.ldy, k0 + 4 * ((tid + 256 * i) % Y4), a.K);
  };
  auto put_split = [&](unsigned char* at, int plane, float4 v) {
    unsigned p1a, p2a, p3a, p1b, p2b, p3b;
    split3(v.x, v.y, p1a, p2a, p3a);
    split3(v.z, v.w, p1b, p2b, p3b);
    *reinterpret_cast<uint2*>(at) = make_uint2(p1a, p1b);
    *reinterpret_cast<uint2*>(at + plane) = make_uint2(p2a, p2b);
    *reinterpret_cast<uint2*>(at + 2 * plane) = make_uint2(p3a, p3b);
  };
  auto store_tiles = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int idx = tid + 256 * i;
      const float4 v = mask4(rx[i], mb + (idx >> 5) < m_end, n0 + 4 * (idx & 31), a.N);
      if (SPLIT)
        put_split(Xp + (idx >> 5) * TN_XPITCH + 8 * (idx & 31), XPLANE, v);
      else
        *reinterpret_cast<float4*>(&Xs[(idx >> 5) * TN_BN + 4 * (idx & 31)]) = v;
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      int idx = tid + 256 * i;
      const float4 v = mask4_ones(ry[i], mb + idx / Y4 < m_end, k0 + 4 * (idx % Y4), a.K, a.ones_col);
      if (SPLIT)
        put_split(Yp + (idx / Y4) * YPITCH + 8 * (idx % Y4), YPLANE, v);
      else
        *reinterpret_cast<float4*>(&Ys[(idx / Y4) * BKK + 4 * (idx % Y4)]) = v;
    }
  };

  if (m_beg < m_end) {
    load_rows(m_beg);
    load_tiles();
    store_tiles(m_beg);
    if (m_beg + TN_BM < m_end) load_rows(m_beg + TN_BM);
    __syncthreads();
    const float* xf = &Xs[(lane >> 5) * TN_BN + wave * 32 + (lane & 31)];
    const float* yf = &Ys[(lane >> 5) * BKK + (lane & 31)];
    for (int mb = m_beg; mb < m_end; mb += TN_BM) {
      const bool more = mb + TN_BM < m_end;
      if (more) {
        load_tiles();                                            // tile mb + TN_BM, rows known already
        if (mb + 2 * TN_BM < m_end) load_rows(mb + 2 * TN_BM);   // row ids of the tile after it
      }
      if (SPLIT) {
        // 16-lane group g: columns 16 (g & 1) .. + 16 of the wave's 32, m rows 8 (g >> 1) .. + 8 of the k step, in two
        // blocks of four; lane 4q + p of a group supplies the address of row q, columns 4p .. 4p + 3
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int mrow = 8 * (g >> 1) + q, col = 16 * (g & 1) + 4 * pp;
#pragma unroll
        for (int ks = 0; ks < TN_BM / 16; ++ks) {
          const unsigned char* xa = Xp + (16 * ks + mrow) * TN_XPITCH + (wave * 32 + col) * 2;
          const bf16x8 a1 = tr_frag(xa, xa + 4 * TN_XPITCH);
          const bf16x8 a2 = tr_frag(xa + XPLANE, xa + XPLANE + 4 * TN_XPITCH);
          const bf16x8 a3 = tr_frag(xa + 2 * XPLANE, xa + 2 * XPLANE + 4 * TN_XPITCH);
#pragma unroll
          for (int t = 0; t < NACC; ++t) {
            const unsigned char* ya = Yp + (16 * ks + mrow) * YPITCH + (32 * t + col) * 2;
            const bf16x8 b1 = tr_frag(ya, ya + 4 * YPITCH);
            const bf16x8 b2 = tr_frag(ya + YPLANE, ya + YPLANE + 4 * YPITCH);
            const bf16x8 b3 = tr_frag(ya + 2 * YPLANE, ya + 2 * YPLANE + 4 * YPITCH);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int st = 0; st < (SPLIT ? 0 : TN_BM / 2); ++st) {
        float xv = xf[2 * st * TN_BN];
        float yv[NACC];
#pragma unroll
        for (int t = 0; t < NACC; ++t) yv[t] = yf[2 * st * BKK + 32 * t];
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, yv[t], acc[t], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        store_tiles(mb + TN_BM);
        __syncthreads();
      }
    }
  }

  float* P = a.P + (size_t)split * a.N * a.ldp;
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    const int k = k0 + 32 * t + (lane & 31);
    if (k >= a.K + a.ones_col) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < a.N) P[(size_t)n * a.ldp + k] = acc[t][r];
    }
  }
}

// sums the slab partials in slab order; column K (present when dbias is given) goes to dbias.  A thread owns
// four consecutive columns of one row and keeps eight slabs' loads in flight; the additions stay in slab order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ P, int splits, int N, int K,
                                                          int ldp, float* __restrict__ dW, int lddw,
                                                          float* __restrict__ dbias, int accumulate) {
  const int Kx = K + (dbias ? 1 : 0);
  const int quads = (Kx + 3) >> 2;
  long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)N * quads) return;
  const int n = (int)(gid / quads), k = 4 * (int)(gid - (long)n * quads);
  const float* p = P + (size_t)n * ldp + k;  // ldp % 4 == 0 and the slabs are 16-byte aligned
  const size_t stride = (size_t)N * ldp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld4(p + (size_t)(s + j) * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
    }
  }
  for (; s < splits; ++s) {
    float4 v = ld4(p + (size_t)s * stride);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (k + j >= Kx) break;
    float* o = k + j < K ? dW + (size_t)n * lddw + k + j : dbias + n;
    *o = accumulate ? *o + r[j] : r[j];
  }
}

inline bool aligned16(const void* p, int ld) { return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((ld & 3) == 0); }
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

struct TnPlan {
  int tiles_n, tiles_k, splits, rows_per_split, ldp, nacc;
};
// Slab length above which the product runs in several rounds of the resident grid instead of one round of longer
// slabs (0 = never): the tiles of a slab share its rows through one L2, and over thousands of K steps they drift
// apart until every tile fetches the rows from HBM for itself; 8192-row slabs keep them in step (measured on
// M = 5.3 M rows: 98 -> 102 TFLOP/s, and the shorter fp32 chains lose fewer digits).
constexpr int TN_MAX_SLAB_ROWS = 8192;

inline TnPlan tn_plan(int M, int N, int K, int ones_col, int max_slab = TN_MAX_SLAB_ROWS, bool split = false) {
  TnPlan p;
  const int kk = K + ones_col;
  if (kk <= 64) {
    p.nacc = kk <= 32 ? 1 : 2;
  } else {  // widest k tile among those that pad the k extent least: more MFMAs per byte staged
    int best = 0;
    p.nacc = 3;
    for (int n = 5; n >= 3; --n) {
      int padded = (kk + 32 * n - 1) / (32 * n) * (32 * n);
      if (!best || padded < best) { best = padded; p.nacc = n; }
    }
    // a product too short to fill the chip with wide tiles runs on more, shorter-lived blocks instead
    const int wide_tiles = ((N + TN_BN - 1) / TN_BN) * (best / (32 * p.nacc));
    if (p.nacc > 3 && (M + 255) / 256 < 768 / wide_tiles) p.nacc = 3;
  }
  p.tiles_n = (N + TN_BN - 1) / TN_BN;
  p.tiles_k = (kk + 32 * p.nacc - 1) / (32 * p.nacc);
  int tiles = p.tiles_n * p.tiles_k;
  int want = (M + 255) / 256;               // at least 256 rows per slab
  // every block does the same work, so the grid is sized to what the chip holds at once (256 CUs x the blocks
  // a CU keeps resident) and never a few blocks more: those would run alone behind a full round
  int resident = 256 * (split ? (p.nacc <= 3 ? 3 : 2) : (p.nacc <= 3 ? 4 : 3));
  int cap = resident / tiles;
  int s = want < cap ? want : cap;
  if (s < 1) s = 1;
  if (max_slab > 0 && cap >= 1 && (M + s - 1) / s > max_slab) {
    int rounds = ((M + max_slab - 1) / max_slab + cap - 1) / cap;
    s = rounds * cap;  // whole rounds: a few blocks more than the chip holds would run alone behind a full round
  }
  int rows = (M + s - 1) / s;
  rows = (rows + TN_BM - 1) / TN_BM * TN_BM;
  if (rows < TN_BM) rows = TN_BM;
  p.rows_per_split = rows;
  p.splits = M > 0 ? (M + rows - 1) / rows : 1;
  p.ldp = (K + 1 + 3) & ~3;  // room for the ones column either way
  return p;
}

constexpr int SM_MAX_ROWS_DEFAULT = 512;  // above this the tiled kernel fills the chip anyway
// JTVAE_NT_SMALL_MAX_M overrides it (tests lower it to send small products through the tiled forms)
thread_local int small_rows_override = -1;  // jtvae_set_small_tile_rows
int small_max_rows() {
  static const int v = [] {
    const char* e = getenv("JTVAE_NT_SMALL_MAX_M");
    return e ? atoi(e) : SM_MAX_ROWS_DEFAULT;
  }();
  return small_rows_override >= 0 ? small_rows_override : v;
}
#define SM_MAX_ROWS small_max_rows()

void launch_nt_small(NtArgs a, hipStream_t s) {
  a.n_tiles_n = (a.N + 31) / 32;
  const int blocks = ((a.M + 31) / 32) * a.n_tiles_n;
  a.n_blocks = blocks;
  switch (a.e.mode) {
    case JTVAE_EPI_GRU_BLEND:
      hipLaunchKernelGGL((gemm_nt_small_kernel<JTVAE_EPI_GRU_BLEND>), dim3(blocks), dim3(256), 0, s, a);
      break;
    case JTVAE_EPI_MUL_ACC:
      hipLaunchKernelGGL((gemm_nt_small_kernel<JTVAE_EPI_MUL_ACC>), dim3(blocks), dim3(256), 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((gemm_nt_small_kernel<JTVAE_EPI_PLAIN>), dim3(blocks), dim3(256), 0, s, a);
  }
}

template <bool VEC, bool EVEC, int NACC>
void launch_nt(NtArgs a, hipStream_t s) {
  a.n_tiles_n = (a.N + 32 * NACC - 1) / (32 * NACC);
  a.n_blocks = ((a.M + NT_BM - 1) / NT_BM) * a.n_tiles_n;
  switch (a.e.mode) {
    case JTVAE_EPI_GRU_BLEND:
      hipLaunchKernelGGL((gemm_nt_kernel<VEC, JTVAE_EPI_GRU_BLEND, EVEC, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
      break;
    case JTVAE_EPI_MUL_ACC:
      hipLaunchKernelGGL((gemm_nt_kernel<VEC, JTVAE_EPI_MUL_ACC, EVEC, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((gemm_nt_kernel<VEC, JTVAE_EPI_PLAIN, EVEC, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
  }
}

template <int NACC>
void launch_nt_split(NtArgs a, hipStream_t s) {
  a.n_tiles_n = (a.N + 32 * NACC - 1) / (32 * NACC);
  a.n_blocks = ((a.M + NT_BM - 1) / NT_BM) * a.n_tiles_n;
  switch (a.e.mode) {
    case JTVAE_EPI_GRU_BLEND:
      hipLaunchKernelGGL((gemm_nt_split_kernel<JTVAE_EPI_GRU_BLEND, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
      break;
    case JTVAE_EPI_MUL_ACC:
      hipLaunchKernelGGL((gemm_nt_split_kernel<JTVAE_EPI_MUL_ACC, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((gemm_nt_split_kernel<JTVAE_EPI_PLAIN, NACC>), dim3(a.n_blocks), dim3(256), 0, s, a);
  }
}

// Products of more than narrow_max_rows() rows with an inner extent of at least this run in the split form
// (JTVAE_SPLIT_BF16=0 keeps every product on the fp32 matrix instruction).
constexpr int SPLIT_MIN_K = 64;
constexpr int TN_SPLIT_MIN_ROWS = 4096;
std::atomic<int> split_mode{-1};  // -1: not decided yet (environment), 0 / 1: off / on
bool split_enabled() {
  int v = split_mode.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("JTVAE_SPLIT_BF16");
    v = (e && e[0] == '0') ? 0 : 1;
    split_mode.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

// Row count up to which the narrow (128 x 32) tiling is used; JTVAE_NT_NARROW_MAX_M overrides it for tuning.
int narrow_max_rows() {
  static const int v = [] {
    const char* e = getenv("JTVAE_NT_NARROW_MAX_M");
    return e ? atoi(e) : 10000;
  }();
  return v;
}

// every row buffer the epilogue touches can be accessed as float4 at columns n % 4 == 0
bool epilogue_aligned(const NtArgs& a) {
  const jtvae_epilogue_t& e = a.e;
  if (!aligned16(a.C, a.ldc)) return false;
  if (e.add && !aligned16(e.add, e.ld_add)) return false;
  if (e.pre_out && !aligned16(e.pre_out, e.ld_pre)) return false;
  if (e.mode != JTVAE_EPI_PLAIN) {
    if (!aligned16(e.aux0, e.ld_aux) || !aligned16(e.out2, e.ld_aux)) return false;
    if (e.mode == JTVAE_EPI_GRU_BLEND && !aligned16(e.aux1, e.ld_aux)) return false;
    if (e.aux_sf && (e.aux_slab_stride & 3)) return false;
  }
  return true;
}

// argument checks and kernel arguments of one NT product; vec = operands are float4-addressable
int nt_setup(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int32_t M, int32_t N,
             int32_t K, const jtvae_epilogue_t* epi, NtArgs& a, bool& vec) {
  vec = false;
  if (M < 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldc < N) return JTVAE_E_SHAPE;
  if (M == 0) return 0;
  if (!A || !W || !C) return JTVAE_E_NULL;
  a.A = A; a.W = W; a.C = C;
  a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  a.M = M; a.N = N; a.K = K;
  a.n_tiles_n = 0; a.n_blocks = 0;
  a.bw_s = nullptr; a.bw_dz = nullptr; a.bw_dp = nullptr;
  if (epi) {
    a.e = *epi;
  } else {
    a.e = jtvae_epilogue_t();
    a.e.zero_row = -1;
  }
  const jtvae_epilogue_t& e = a.e;
  if (e.mode < 0 || e.mode > JTVAE_EPI_MUL_ACC || e.act < 0 || e.act > JTVAE_ACT_TANH) return JTVAE_E_RANGE;
  if (e.add && e.ld_add < N) return JTVAE_E_SHAPE;
  if (e.pre_out && e.ld_pre < N) return JTVAE_E_SHAPE;
  if (e.mode == JTVAE_EPI_GRU_BLEND && (!e.aux0 || !e.aux1 || !e.out2 || e.ld_aux < N)) return JTVAE_E_NULL;
  if (e.mode == JTVAE_EPI_MUL_ACC && (!e.aux0 || !e.out2 || e.ld_aux < N)) return JTVAE_E_NULL;
  if (e.aux_sf && e.aux_t < 0) return JTVAE_E_SHAPE;
  vec = aligned16(A, lda) && aligned16(W, ldw);
  a.evec = epilogue_aligned(a) ? 1 : 0;
  return 0;
}

// NT-family kernels launched by this host thread so far: a call that cannot take its fused form runs as two
// launches, and whoever prices a call per launch (a profiler's view) can ask how many it made
thread_local unsigned long long nt_launches = 0;

// picks the kernel for one product by its row count and operand alignment
void launch_nt_any(const NtArgs& a, bool vec, hipStream_t s) {
  ++nt_launches;
  if (vec && a.M <= SM_MAX_ROWS && a.K <= SM_KMAX)
    launch_nt_small(a, s);
  else if (vec && a.evec && a.M <= narrow_max_rows())
    launch_nt<true, true, 1>(a, s);
  else if (vec && a.evec && a.K >= SPLIT_MIN_K && split_enabled())
    launch_nt_split<3>(a, s);
  else if (vec && a.evec)
    launch_nt<true, true, 3>(a, s);
  else if (vec)
    launch_nt<true, false, 3>(a, s);
  else
    launch_nt<false, false, 3>(a, s);
}

int gemm_tn_run(const float* X, int32_t ldx, const int32_t* x_rows, const float* Y, int32_t ldy,
                const int32_t* y_rows, float* dW, int32_t lddw, float* dbias, int32_t M, int32_t N, int32_t K,
                int32_t accumulate, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (M < 0 || N <= 0 || K <= 0 || ldx < N || ldy < K || lddw < K) return JTVAE_E_SHAPE;
  if (!X || !Y || !dW || !workspace) return JTVAE_E_NULL;
  const int ones_col = dbias ? 1 : 0;
  const bool vec = aligned16(X, ldx) && aligned16(Y, ldy);
  // wide products over many rows take the split form (the narrow feature-row products are bound by the rows they
  // move, short ones by latency)
  const bool split = vec && split_enabled() && K + ones_col > 64 && M >= TN_SPLIT_MIN_ROWS;
  TnPlan p = tn_plan(M, N, K, ones_col, TN_MAX_SLAB_ROWS, split);
  if (workspace_bytes < (size_t)p.splits * N * p.ldp * sizeof(float))
    p = tn_plan(M, N, K, ones_col, 0, split);  // a smaller workspace (the static array) takes one round of longer slabs
  if (workspace_bytes < (size_t)p.splits * N * p.ldp * sizeof(float)) return JTVAE_E_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return JTVAE_E_ALIGN;
  TnArgs a;
  a.X = X; a.Y = Y; a.x_rows = x_rows; a.y_rows = y_rows; a.P = (float*)workspace;
  a.ldx = ldx; a.ldy = ldy; a.ldp = p.ldp;
  a.M = M; a.N = N; a.K = K;
  a.tiles_n = p.tiles_n; a.tiles_k = p.tiles_k; a.rows_per_split = p.rows_per_split;
  a.ones_col = ones_col;
  int blocks = p.tiles_n * p.tiles_k * p.splits;
#define JTVAE_TN_LAUNCH(NACC)                                                                   \
  do {                                                                                          \
    if (split && NACC >= 3)                                                                     \
      hipLaunchKernelGGL((gemm_tn_kernel<true, (NACC >= 3 ? NACC : 3), true>), dim3(blocks), dim3(256), 0, s, a); \
    else if (vec)                                                                               \
      hipLaunchKernelGGL((gemm_tn_kernel<true, NACC>), dim3(blocks), dim3(256), 0, s, a);       \
    else                                                                                        \
      hipLaunchKernelGGL((gemm_tn_kernel<false, NACC>), dim3(blocks), dim3(256), 0, s, a);      \
  } while (0)
  switch (p.nacc) {
    case 1: JTVAE_TN_LAUNCH(1); break;
    case 2: JTVAE_TN_LAUNCH(2); break;
    case 3: JTVAE_TN_LAUNCH(3); break;
    case 4: JTVAE_TN_LAUNCH(4); break;
    default: JTVAE_TN_LAUNCH(5); break;
  }
#undef JTVAE_TN_LAUNCH
  long total = (long)N * ((K + ones_col + 3) / 4);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.P, p.splits, N,
                     K, p.ldp, dW, lddw, dbias, accumulate);
  return launch_status();
}

// The static array of the device that owns `stream` (the current device for the null stream).  hipGetSymbolAddress
// answers for the CURRENT device, so the first call for a device makes it current for the duration of the lookup.
int tn_scratch_for_stream(hipStream_t stream, void** out) {
  int cur = -1, dev = -1;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return (int)e;
  dev = cur;
  if (stream) {
    hipDevice_t d;
    e = hipStreamGetDevice(stream, &d);
    if (e != hipSuccess) return (int)e;
    dev = (int)d;
  }
  if (dev < 0 || dev >= TN_MAX_DEVICES) return JTVAE_E_RANGE;
  std::lock_guard<std::mutex> lock(tn_scratch_mutex);
  if (!tn_scratch_ptr[dev]) {
    void* p = nullptr;
    if (dev != cur && (e = hipSetDevice(dev)) != hipSuccess) return (int)e;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(tn_scratch));
    if (dev != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) return (int)e;
    if (!p) return JTVAE_E_NULL;
    tn_scratch_ptr[dev] = p;
  }
  *out = tn_scratch_ptr[dev];
  return 0;
}

}  // namespace

extern "C" {

int jtvae_gemm_nt(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int32_t M,
                  int32_t N, int32_t K, const jtvae_epilogue_t* epi, void* stream) {
  NtArgs a;
  bool vec;
  const int rc = nt_setup(A, lda, W, ldw, C, ldc, M, N, K, epi, a, vec);
  if (rc != 0 || M == 0) return rc;
  launch_nt_any(a, vec, (hipStream_t)stream);
  return launch_status();
}

int jtvae_gemm_nt_pair(const jtvae_gemm_nt_problem_t* p0, const jtvae_gemm_nt_problem_t* p1, void* stream) {
  if (!p0 || !p1) return JTVAE_E_NULL;
  NtArgs a, b;
  bool vec_a, vec_b;
  int rc = nt_setup(p0->A, p0->lda, p0->W, p0->ldw, p0->C, p0->ldc, p0->M, p0->N, p0->K, p0->epi, a, vec_a);
  if (rc != 0) return rc;
  rc = nt_setup(p1->A, p1->lda, p1->W, p1->ldw, p1->C, p1->ldc, p1->M, p1->N, p1->K, p1->epi, b, vec_b);
  if (rc != 0) return rc;
  if (p0->M == 0 && p1->M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // one launch for the combination the tree GRU's backward pass issues (plain + multiply-accumulate
  // epilogue) when both products are small or mid-size; anything else runs as two launches
  const bool combo = a.e.mode == JTVAE_EPI_PLAIN && b.e.mode == JTVAE_EPI_MUL_ACC && vec_a && vec_b &&
                     p0->M > 0 && p1->M > 0;
  const int big = p0->M > p1->M ? p0->M : p1->M;
  if (combo && big <= SM_MAX_ROWS && a.K <= SM_KMAX && b.K <= SM_KMAX) {
    a.n_tiles_n = (a.N + 31) / 32;
    a.n_blocks = ((a.M + 31) / 32) * a.n_tiles_n;
    b.n_tiles_n = (b.N + 31) / 32;
    b.n_blocks = ((b.M + 31) / 32) * b.n_tiles_n;
    hipLaunchKernelGGL((gemm_nt_small_pair_kernel<JTVAE_EPI_PLAIN, JTVAE_EPI_MUL_ACC>), dim3(a.n_blocks + b.n_blocks),
                       dim3(256), 0, s, a, b);
    ++nt_launches;
  } else if (combo && a.evec && b.evec && big <= narrow_max_rows()) {
    a.n_tiles_n = (a.N + 31) / 32;
    a.n_blocks = ((a.M + NT_BM - 1) / NT_BM) * a.n_tiles_n;
    b.n_tiles_n = (b.N + 31) / 32;
    b.n_blocks = ((b.M + NT_BM - 1) / NT_BM) * b.n_tiles_n;
    hipLaunchKernelGGL((gemm_nt_pair_kernel<JTVAE_EPI_PLAIN, JTVAE_EPI_MUL_ACC, 1>), dim3(a.n_blocks + b.n_blocks),
                       dim3(256), 0, s, a, b);
    ++nt_launches;
  } else {
    if (p0->M > 0) launch_nt_any(a, vec_a, s);
    if (p1->M > 0) launch_nt_any(b, vec_b, s);
  }
  return launch_status();
}

int jtvae_gru_update(const float* S, const float* G, int32_t ld, const float* Wz, const float* Wh, int32_t ldw,
                     const float* Pz, const float* Ph, int32_t ldp, const int32_t* wid, const int32_t* rows,
                     int32_t n_rows, float* Z, float* P, float* Hout, int32_t H, int32_t zero_row, void* stream) {
  if (n_rows < 0 || H <= 0 || ld < H || ldw < H || ldp < H) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!S || !G || !Wz || !Wh || !Pz || !Ph || !wid || !Z || !P || !Hout) return JTVAE_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = aligned16(S, ld) && aligned16(G, ld) && aligned16(Wz, ldw) && aligned16(Wh, ldw) &&
                   aligned16(Pz, ldp) && aligned16(Ph, ldp) && aligned16(Z, ld) && aligned16(P, ld) &&
                   aligned16(Hout, ld);
  if (!vec || n_rows > narrow_max_rows() || (n_rows <= SM_MAX_ROWS && H > SM_KMAX)) {
    // the general route: two products, the second with the blend epilogue
    jtvae_epilogue_t e = jtvae_epilogue_t();
    e.zero_row = -1;
    e.add = Pz; e.ld_add = ldp; e.add_idx = wid; e.act = JTVAE_ACT_SIGMOID; e.a_rows = rows; e.c_rows = rows;
    int rc = jtvae_gemm_nt(S, ld, Wz, ldw, Z, ld, n_rows, H, H, &e, stream);
    if (rc) return rc;
    e.add = Ph; e.act = JTVAE_ACT_NONE; e.mode = JTVAE_EPI_GRU_BLEND;
    e.aux0 = Z; e.aux1 = S; e.out2 = Hout; e.ld_aux = ld; e.zero_row = zero_row;
    return jtvae_gemm_nt(G, ld, Wh, ldw, P, ld, n_rows, H, H, &e, stream);
  }
  GruUpArgs a;
  a.S = S; a.G = G; a.Wz = Wz; a.Wh = Wh; a.Pz = Pz; a.Ph = Ph; a.wid = wid; a.rows = rows;
  a.Z = Z; a.P = P; a.Hout = Hout;
  a.ld = ld; a.ldw = ldw; a.ldp = ldp;
  a.M = n_rows; a.N = H; a.K = H; a.zero_row = zero_row;
  a.n_tiles_n = (H + 31) / 32;
  ++nt_launches;
  if (n_rows <= SM_MAX_ROWS) {
    a.n_blocks = ((n_rows + 31) / 32) * a.n_tiles_n;
    hipLaunchKernelGGL(gru_update_small_kernel, dim3(a.n_blocks), dim3(256), 0, s, a);
  } else {
    a.n_blocks = ((n_rows + NT_BM - 1) / NT_BM) * a.n_tiles_n;
    hipLaunchKernelGGL(gru_update_kernel, dim3(a.n_blocks), dim3(256), 0, s, a);
  }
  return launch_status();
}

int jtvae_gru_bwd_dh(const float* du, const float* UrT, int32_t ldw, const float* dhdir, float* dh,
                     const int32_t* rows, int32_t n_rows, const float* z, const float* p, const float* s,
                     size_t slab_stride, int32_t t_iter, const int32_t* sf, float* dzpre, float* dppre, float* ds,
                     int32_t H, int32_t ld, int32_t pad_row, void* stream) {
  if (n_rows < 0 || H <= 0 || ld < H || ldw < H || t_iter < 0 || (slab_stride & 3)) return JTVAE_E_SHAPE;
  if (n_rows == 0) return 0;
  if (!du || !UrT || !dhdir || !dh || !z || !p || !s || !dzpre || !dppre || !ds) return JTVAE_E_NULL;
  NtArgs a;
  bool vec;
  jtvae_epilogue_t e = jtvae_epilogue_t();
  e.zero_row = -1;
  e.add = dhdir;
  e.ld_add = ld;
  e.c_rows = rows;
  int rc = nt_setup(du, ld, UrT, ldw, dh, ld, n_rows, H, H, &e, a, vec);
  if (rc) return rc;
  const bool fused = vec && a.evec && n_rows <= narrow_max_rows() && aligned16(z, ld) && aligned16(p, ld) &&
                     aligned16(s, ld) && aligned16(dzpre, ld) && aligned16(dppre, ld) && aligned16(ds, ld);
  if (!fused) {  // the general route: the product, then the pointwise kernel on what it wrote
    launch_nt_any(a, vec, (hipStream_t)stream);
    rc = launch_status();
    if (rc) return rc;
    return jtvae_gru_bwd_pointwise(dh, z, p, s, slab_stride, t_iter, sf, rows, n_rows, dzpre, dppre, ds, H, ld,
                                   pad_row, stream);
  }
  a.e.mode = EPI_GRU_BWD;
  a.e.aux0 = z; a.e.aux1 = p; a.e.out2 = ds; a.e.ld_aux = ld;
  a.e.aux_sf = sf; a.e.aux_t = t_iter; a.e.aux_slab_stride = slab_stride;
  a.e.zero_row = pad_row;
  a.bw_s = s; a.bw_dz = dzpre; a.bw_dp = dppre;
  ++nt_launches;
  hipStream_t st = (hipStream_t)stream;
  if (n_rows <= SM_MAX_ROWS && H <= SM_KMAX) {
    a.n_tiles_n = (H + 31) / 32;
    a.n_blocks = ((n_rows + 31) / 32) * a.n_tiles_n;
    hipLaunchKernelGGL((gemm_nt_small_kernel<EPI_GRU_BWD>), dim3(a.n_blocks), dim3(256), 0, st, a);
  } else {
    a.n_tiles_n = (H + 31) / 32;
    a.n_blocks = ((n_rows + NT_BM - 1) / NT_BM) * a.n_tiles_n;
    hipLaunchKernelGGL((gemm_nt_kernel<true, EPI_GRU_BWD, true, 1>), dim3(a.n_blocks), dim3(256), 0, st, a);
  }
  return launch_status();
}

size_t jtvae_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  // enough for both forms of the call (with and without the bias column)
  size_t s = 0;
  for (int split = 0; split < 2; ++split)
    for (int ones = 0; ones < 2; ++ones) {
      const size_t n = (size_t)tn_plan(M, N, K, ones, TN_MAX_SLAB_ROWS, split != 0).splits;
      s = n > s ? n : s;
    }
  return s * N * tn_plan(M, N, K, 0).ldp * sizeof(float);
}

int jtvae_gemm_tn(const float* X, int32_t ldx, const int32_t* x_rows, const float* Y, int32_t ldy,
                  const int32_t* y_rows, float* dW, int32_t lddw, float* dbias, int32_t M, int32_t N, int32_t K,
                  int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_tn_run(X, ldx, x_rows, Y, ldy, y_rows, dW, lddw, dbias, M, N, K, accumulate, workspace,
                     workspace_bytes, (hipStream_t)stream);
}

size_t jtvae_gemm_tn_static_max_bytes(void) { return TN_SCRATCH_BYTES; }

int jtvae_gemm_tn_static(const float* X, int32_t ldx, const int32_t* x_rows, const float* Y, int32_t ldy,
                         const int32_t* y_rows, float* dW, int32_t lddw, float* dbias, int32_t M, int32_t N,
                         int32_t K, int32_t accumulate, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return JTVAE_E_SHAPE;
  // same slab plan as jtvae_gemm_tn whenever that plan's partials fit the array (bit-identical results then);
  // a product of more than ~0.7 M rows keeps one round of longer slabs here
  {
    TnPlan p0 = tn_plan(M, N, K, 0, 0), p1 = tn_plan(M, N, K, 1, 0);
    const size_t need = (size_t)(p0.splits > p1.splits ? p0.splits : p1.splits) * N * p0.ldp * sizeof(float);
    if (need > TN_SCRATCH_BYTES) return JTVAE_E_WORKSPACE;
  }
  void* ws = nullptr;
  const int rc = tn_scratch_for_stream((hipStream_t)stream, &ws);
  if (rc != 0) return rc;
  // every partial the slab sum reads is written by this call's product kernel first, so the
  // contents the previous call left in the array never need clearing
  return gemm_tn_run(X, ldx, x_rows, Y, ldy, y_rows, dW, lddw, dbias, M, N, K, accumulate, ws, TN_SCRATCH_BYTES,
                     (hipStream_t)stream);
}

int jtvae_abi_version(void) { return JTVAE_ABI_VERSION; }

int jtvae_set_small_tile_rows(int rows) {
  const int before = small_rows_override;
  small_rows_override = rows < 0 ? -1 : rows;
  return before;
}

int jtvae_set_split_products(int on) {
  const int before = split_enabled() ? 1 : 0;
  if (on >= 0) split_mode.store(on ? 1 : 0, std::memory_order_relaxed);
  return before;
}

uint64_t jtvae_nt_launch_count(void) { return nt_launches; }

const char* jtvae_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case JTVAE_E_NULL: return "required pointer is NULL";
    case JTVAE_E_SHAPE: return "negative or inconsistent sizes";
    case JTVAE_E_ALIGN: return "row buffer not 16-byte aligned (base address or leading dimension)";
    case JTVAE_E_RANGE: return "mode or enum out of range";
    case JTVAE_E_WORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

}  // extern "C"
