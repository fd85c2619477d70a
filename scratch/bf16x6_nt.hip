// Feasibility probe: C[M,N] = A[M,K] . W[N,K]^T with fp32 operands split into three bf16 parts each
// (a = a1 + a2 + a3, 24 significant bits in all) and the six largest cross products a_i * b_j (i + j <= 4)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: fp32-class accuracy at 16/6 of the fp32 MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x6_nt scratch/bf16x6_nt.hip && /tmp/bf16x6_nt [M]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;

// blocks that share an XCD (ids equal mod 8) get consecutive tiles
__device__ __forceinline__ int xcd_remap(int id, int n) {
  int q = n >> 3, r = n & 7;
  int x = id & 7;
  int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (id >> 3);
}

__device__ __forceinline__ unsigned pk(float x, float y) {
  f32x2 v = {x, y};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}
// (x, y) -> three packed bf16 pairs with x = x1 + x2 + x3 (+ 2^-26 |x|)
__device__ __forceinline__ void split3(float x, float y, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pk(x, y);
#ifdef NO_SPLIT
  p2 = p1; p3 = p1; return;
#endif
  float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
  p2 = pk(rx, ry);
  rx -= __uint_as_float(p2 << 16);
  ry -= __uint_as_float(p2 & 0xffff0000u);
  p3 = pk(rx, ry);
}

template <int NACC>
__global__ __launch_bounds__(256, 3) void nt_bf16x6(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                    int ldw, float* __restrict__ C, int ldc, int M, int N, int K,
                                                    int n_tiles_n, unsigned long long* trace, const uint4* __restrict__ Wp, int wp_rows, int wp_ld8) {
  constexpr int BN = 32 * NACC;
#ifdef TRACE
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tt = __builtin_readcyclecounter(), tn;
  const unsigned long long wall0 = wall_clock64(), cyc0 = tt;
#define PH(i) do { tn = __builtin_readcyclecounter(); tph[i] += tn - tt; tt = tn; } while (0)
#else
#define PH(i) do {} while (0)
#endif
  constexpr int ROWS = BM + BN;
#ifndef LDS_PAD
#define LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * ROWS * 64 + LDS_PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / n_tiles_n) * BM, n0 = (tile % n_tiles_n) * BN;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int srow = tid >> 3, sc4 = tid & 7;
  float4 ra[BM / 32], rb[NACC];
  const float* a_row[BM / 32];
  const float* b_row[NACC];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) a_row[i] = A + (size_t)min(m0 + srow + 32 * i, M - 1) * lda;
#pragma unroll
  for (int i = 0; i < NACC; ++i) b_row[i] = W + (size_t)min(n0 + srow + 32 * i, N - 1) * ldw;

  float4 ra2[BM / 32], rb2[NACC];
  auto load_into = [&](float4 (&xa)[BM / 32], float4 (&xb)[NACC], int k0) {
    const int k = k0 + 4 * sc4;
    const int kc = k < K ? k : 0;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) xa[i] = *reinterpret_cast<const float4*>(a_row[i] + kc);
#pragma unroll
    for (int i = 0; i < NACC; ++i) xb[i] = *reinterpret_cast<const float4*>(b_row[i] + kc);
  };
  auto put = [&](int row, float4 v, bool row_ok, int k) {
    v.x = (row_ok && k + 0 < K) ? v.x : 0.f;
    v.y = (row_ok && k + 1 < K) ? v.y : 0.f;
    v.z = (row_ok && k + 2 < K) ? v.z : 0.f;
    v.w = (row_ok && k + 3 < K) ? v.w : 0.f;
    unsigned p1a, p2a, p3a, p1b, p2b, p3b;
    split3(v.x, v.y, p1a, p2a, p3a);
    split3(v.z, v.w, p1b, p2b, p3b);
    const int off = row * 64 + 16 * (((sc4 >> 1) + (row >> 2)) & 3) + 8 * (sc4 & 1);
    *reinterpret_cast<uint2*>(smem + off) = make_uint2(p1a, p1b);
    *reinterpret_cast<uint2*>(smem + ROWS * 64 + off) = make_uint2(p2a, p2b);
    *reinterpret_cast<uint2*>(smem + 2 * ROWS * 64 + off) = make_uint2(p3a, p3b);
  };
  auto store_from = [&](const float4 (&xa)[BM / 32], const float4 (&xb)[NACC], int k0) {
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) put(srow + 32 * i, xa[i], m0 + srow + 32 * i < M, k0 + 4 * sc4);
#pragma unroll
    for (int i = 0; i < NACC; ++i) put(BM + srow + 32 * i, xb[i], n0 + srow + 32 * i < N, k0 + 4 * sc4);
  };
#ifdef PRESPLIT
  // W arrives split: three planes [wp_rows][wp_ld8 chunks of 8 bf16], zero padded in both directions; the A side
  // is read from clamped addresses and never masked (what lies past K meets zeros of W)
  constexpr int WCH = 3 * BN * 4;  // 16-byte chunks of a W K-tile
  constexpr int WPT = (WCH + 255) / 256;
  u32x4 rw[WPT];
  auto load_w = [&](int k0) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 256 * i, WCH - 1);  // unconditional: a guarded load is waited for where it is issued
      const int plane = c / (BN * 4), rc = c % (BN * 4), row = rc >> 2, ch = rc & 3;
      rw[i] = reinterpret_cast<const u32x4*>(Wp)[((size_t)plane * wp_rows + n0 + row) * wp_ld8 + (k0 >> 3) + ch];
    }
  };
  auto st_w = [&](int c, u32x4 v) {
    const int plane = c / (BN * 4), rc = c % (BN * 4), row = BM + (rc >> 2), ch = rc & 3;
    *reinterpret_cast<u32x4*>(smem + plane * ROWS * 64 + row * 64 + 16 * ((ch + (row >> 2)) & 3)) = v;
  };
  auto store_w = [&]() {
    // threads past the last chunk hold a copy of the last one and store it again: no branch, no guarded access
#pragma unroll
    for (int i = 0; i < WPT; ++i) st_w(min(tid + 256 * i, WCH - 1), rw[i]);
  };
  auto put_a = [&](int row, float4 v) {
    unsigned p1a, p2a, p3a, p1b, p2b, p3b;
    split3(v.x, v.y, p1a, p2a, p3a);
    split3(v.z, v.w, p1b, p2b, p3b);
    const int off = row * 64 + 16 * (((sc4 >> 1) + (row >> 2)) & 3) + 8 * (sc4 & 1);
    *reinterpret_cast<uint2*>(smem + off) = make_uint2(p1a, p1b);
    *reinterpret_cast<uint2*>(smem + ROWS * 64 + off) = make_uint2(p2a, p2b);
    *reinterpret_cast<uint2*>(smem + 2 * ROWS * 64 + off) = make_uint2(p3a, p3b);
  };
  auto load_tiles = [&](int k0) {
    const int k = k0 + 4 * sc4;
    const int kc = k < K ? k : 0;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) ra[i] = *reinterpret_cast<const float4*>(a_row[i] + kc);
    load_w(k0);
  };
  auto store_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) put_a(srow + 32 * i, ra[i]);
    store_w();
  };
#else
  auto load_tiles = [&](int k0) { load_into(ra, rb, k0); };
  auto store_tiles = [&](int k0) { store_from(ra, rb, k0); };
#endif

  const int nk = (K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  const int arow = wave * 32 + r;
  PH(0);
#ifdef INTERLEAVE
  // no matrix instruction depends on the one issued just before it: the six terms go round the accumulators
  auto mfma_tile = [&](int kt) {
    const int ks_end = min(2, (K - kt * BK + 15) >> 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s >= ks_end) break;
      const int slot = 2 * s + h;
      const int aoff = arow * 64 + 16 * ((slot + (arow >> 2)) & 3);
      bf16x8 a[3], b[NACC][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(smem + p * ROWS * 64 + aoff);
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const int brow = BM + 32 * t + r;
        const int boff = brow * 64 + 16 * ((slot + (brow >> 2)) & 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const bf16x8*>(smem + p * ROWS * 64 + boff);
      }
      constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int t = 0; t < NACC; ++t)
        {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[IA[j]], b[t][IB[j]], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
#elif defined(LOOKAHEAD)
  // fragments one group of six matrix instructions ahead: A of both k steps up front, B of group g + 1 requested
  // before the instructions of group g are issued
  auto mfma_tile = [&](int kt) {
    const int ks_end = min(2, (K - kt * BK + 15) >> 4);
    bf16x8 af[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int slot = 2 * s + h;
      const int aoff = arow * 64 + 16 * ((slot + (arow >> 2)) & 3);
#pragma unroll
      for (int p = 0; p < 3; ++p) af[s][p] = *reinterpret_cast<const bf16x8*>(smem + p * ROWS * 64 + aoff);
    }
    auto ldb = [&](int g, bf16x8 (&b)[3]) {
      const int s = g / NACC, t = g % NACC;
      const int brow = BM + 32 * t + r;
      const int boff = brow * 64 + 16 * ((2 * s + h + (brow >> 2)) & 3);
#pragma unroll
      for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(smem + p * ROWS * 64 + boff);
    };
    bf16x8 bq[2][3];
    ldb(0, bq[0]);
    const int groups = ks_end * NACC;
#pragma unroll
    for (int g = 0; g < 2 * NACC; ++g) {
      if (g >= groups) break;
      if (g + 1 < 2 * NACC) ldb(g + 1, bq[(g + 1) & 1]);
      const int s = g / NACC, t = g % NACC;
      const bf16x8 (&b)[3] = bq[g & 1];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][2], b[0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], b[2], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][1], b[1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][1], b[0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], b[1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0], b[0], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#else
  auto mfma_tile = [&](int kt) {
    const int ks_end = min(2, (K - kt * BK + 15) >> 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s >= ks_end) break;
      const int slot = 2 * s + h;
#ifdef V1_NO_LDSREAD
      const int aoff = 0;
#else
      const int aoff = arow * 64 + 16 * ((slot + (arow >> 2)) & 3);
#endif
      bf16x8 a1 = *reinterpret_cast<const bf16x8*>(smem + aoff);
      bf16x8 a2 = *reinterpret_cast<const bf16x8*>(smem + ROWS * 64 + aoff);
      bf16x8 a3 = *reinterpret_cast<const bf16x8*>(smem + 2 * ROWS * 64 + aoff);
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const int brow = BM + 32 * t + r;
#ifdef V1_NO_LDSREAD
        const int boff = 0;
#else
        const int boff = brow * 64 + 16 * ((slot + (brow >> 2)) & 3);
#endif
        bf16x8 b1 = *reinterpret_cast<const bf16x8*>(smem + boff);
        bf16x8 b2 = *reinterpret_cast<const bf16x8*>(smem + ROWS * 64 + boff);
        bf16x8 b3 = *reinterpret_cast<const bf16x8*>(smem + 2 * ROWS * 64 + boff);
#if !defined(ONE_MFMA) && !defined(THREE_MFMA)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
#endif
#if !defined(ONE_MFMA)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
#endif
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      }
    }
  };
#endif
#ifdef DEEP
  // tile kt + 1 waits in one register set while tile kt + 2 is in flight into the other
  if (nk > 1) load_into(ra, rb, BK);
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) load_into(ra2, rb2, (kt + 2) * BK);
    mfma_tile(kt);
    PH(1);
#ifndef V1_NO_BARRIER
    __syncthreads();
#endif
    PH(2);
    if (kt + 1 < nk) {
      store_from(ra, rb, (kt + 1) * BK);
      PH(3);
#ifndef V1_NO_BARRIER
      __syncthreads();
#endif
      PH(4);
      if (kt + 3 < nk) load_into(ra, rb, (kt + 3) * BK);
      mfma_tile(kt + 1);
      PH(1);
      __syncthreads();
      PH(2);
      if (kt + 2 < nk) {
        store_from(ra2, rb2, (kt + 2) * BK);
        PH(3);
        __syncthreads();
        PH(4);
      }
    }
  }
#else
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tiles((kt + 1) * BK);
    mfma_tile(kt);
    PH(1);
#ifndef V1_NO_BARRIER
    __syncthreads();
#endif
    PH(2);
    if (more) {
      store_tiles((kt + 1) * BK);
      PH(3);
#ifndef V1_NO_BARRIER
      __syncthreads();
#endif
      PH(4);
    }
  }
#endif
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = m0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (i >= M) continue;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int n = n0 + 32 * t + r;
#ifdef NO_STORE
      if (n < N && acc[t][q] == 123.456f) C[(size_t)i * ldc + n] = acc[t][q];
#else
      if (n < N) C[(size_t)i * ldc + n] = acc[t][q];
#endif
    }
  }
#ifdef TRACE
  __builtin_amdgcn_s_waitcnt(0);
  PH(5);
  if (lane == 0 && blockIdx.x == 1000 && wave == 0)
    printf("block 1000: %llu shader cycles in %llu ticks of the 100 MHz wall clock\n", tt - cyc0, wall_clock64() - wall0);
  if (lane == 0)
    for (int i = 0; i < 8; ++i) trace[((size_t)blockIdx.x * 4 + wave) * 8 + i] = tph[i];
#endif
}


// v2: the A operand never touches LDS.  Lane (r, h) of a wave owns row r of the wave's 32 rows and, per K tile of
// 32, the 16 consecutive k values [16h, 16h + 16) of it (k step s takes the 8 starting at 16h + 8s; W's image uses
// the same permutation), loads them as 64 contiguous bytes and splits them in registers.  W arrives pre-split and
// goes through a double-buffered LDS image: one barrier per K tile.
template <int NACC>
__global__ __launch_bounds__(256, NACC > 3 ? 2 : 3) void nt_v2(const float* __restrict__ A, int lda, float* __restrict__ C, int ldc,
                                                int M, int N, int K, int n_tiles_n, const u32x4* __restrict__ Wp,
                                                int wp_rows, int wp_ld8) {
  constexpr int BN = 32 * NACC;
  constexpr int WCH = 3 * BN * 4;  // 16-byte chunks of a W K-tile
  constexpr int WPT = (WCH + 255) / 256;
  constexpr int WBUF = 3 * BN * 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WBUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / n_tiles_n) * BM, n0 = (tile % n_tiles_n) * BN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

#ifdef SAME_A
  const float* a_row = A + (size_t)min((m0 & 1023) + wave * 32 + r, M - 1) * lda + 16 * h;
#else
  const float* a_row = A + (size_t)min(m0 + wave * 32 + r, M - 1) * lda + 16 * h;
#endif
  float4 ra[4];
  u32x4 rw[WPT];
  const int nk = (K + BK - 1) / BK;
  auto load_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kt * BK + 16 * h + 4 * i;
      ra[i] = *reinterpret_cast<const float4*>(a_row + (k + 3 < lda ? kt * BK + 4 * i : -16 * h));
    }
  };
  auto load_w = [&](int kt) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 256 * i, WCH - 1);
      const int plane = c / (BN * 4), rc = c % (BN * 4), row = rc >> 2, ch = rc & 3;
      rw[i] = Wp[((size_t)plane * wp_rows + n0 + row) * wp_ld8 + kt * 4 + ch];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 256 * i, WCH - 1);
      const int plane = c / (BN * 4), rc = c % (BN * 4), row = rc >> 2, ch = rc & 3;
      *reinterpret_cast<u32x4*>(smem + buf * WBUF + (plane * BN + row) * 64 + 16 * ((ch + (row >> 2)) & 3)) = rw[i];
    }
  };
  load_a(0);
  load_w(0);
  store_w(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    // this tile's A values -> three fragments per k step
    float4 v[4] = {ra[0], ra[1], ra[2], ra[3]};
    if (kt == nk - 1 && (K & 31)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = kt * BK + 16 * h + 4 * i;
        v[i].x = k + 0 < K ? v[i].x : 0.f;
        v[i].y = k + 1 < K ? v[i].y : 0.f;
        v[i].z = k + 2 < K ? v[i].z : 0.f;
        v[i].w = k + 3 < K ? v[i].w : 0.f;
      }
    }
    if (more) {
      load_a(kt + 1);
      load_w(kt + 1);
    }
    const unsigned char* wb = smem + (kt & 1) * WBUF;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned p[3][4];
      split3(v[2 * s].x, v[2 * s].y, p[0][0], p[1][0], p[2][0]);
      split3(v[2 * s].z, v[2 * s].w, p[0][1], p[1][1], p[2][1]);
      split3(v[2 * s + 1].x, v[2 * s + 1].y, p[0][2], p[1][2], p[2][2]);
      split3(v[2 * s + 1].z, v[2 * s + 1].w, p[0][3], p[1][3], p[2][3]);
      u32x4 q1 = {p[0][0], p[0][1], p[0][2], p[0][3]};
      u32x4 q2 = {p[1][0], p[1][1], p[1][2], p[1][3]};
      u32x4 q3 = {p[2][0], p[2][1], p[2][2], p[2][3]};
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, q1), a2 = __builtin_bit_cast(bf16x8, q2),
                   a3 = __builtin_bit_cast(bf16x8, q3);
      const int slot = 2 * h + s;  // chunk of W's row that holds k = 16h + 8s .. + 8
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const int brow = 32 * t + r;
#ifdef NO_LDSREAD
        const int boff = 0;
#else
        const int boff = brow * 64 + 16 * ((slot + (brow >> 2)) & 3);
#endif
        bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wb + boff);
        bf16x8 b2 = *reinterpret_cast<const bf16x8*>(wb + BN * 64 + boff);
        bf16x8 b3 = *reinterpret_cast<const bf16x8*>(wb + 2 * BN * 64 + boff);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      }
    }
    if (more) store_w((kt + 1) & 1);
#ifndef NO_BARRIER
    __syncthreads();
#endif
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = m0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (i >= M) continue;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int n = n0 + 32 * t + r;
#ifdef NO_STORE
      if (n < N && acc[t][q] == 123.456f) C[(size_t)i * ldc + n] = acc[t][q];
#else
      if (n < N) C[(size_t)i * ldc + n] = acc[t][q];
#endif
    }
  }
}


// v5: one block owns 128 rows x ALL columns (15 column tiles), eight waves = four row groups x two column halves
// (tiles 0-7 and 8-14), two waves per SIMD.  Each A element is loaded by two waves (not five blocks), straight into
// registers in the matrix instruction's own layout (lane (r, h): row r, k = 8h .. 8h + 8 of a K step of 16) and split
// there; W arrives pre-split and streams through a two-stage LDS ring of [3 planes][480 rows][16 k] bf16 (46 KB a
// stage), whose 32-byte rows make every fragment read and every staging write one contiguous kilobyte per wave.
// One barrier per K step of 16; per step a wave issues 48 (42) matrix instructions against ~25 vector and 9 LDS ones.
constexpr int V5_ROWS = 480, V5_PLANE = V5_ROWS * 32, V5_STAGE = 3 * V5_PLANE;

// the matrix work of one K step for a wave with NT column tiles: the three W fragments of tile t + 1 are requested
// before the six instructions of tile t are issued (sched_group_barrier pins that order), so the LDS latency of a
// tile hides behind the 192 cycles of the tile before it
template <int NT>
__device__ __forceinline__ void v5_step(f32x16 (&acc)[8], const bf16x8 a1, const bf16x8 a2, const bf16x8 a3,
                                        const unsigned char* wb) {
  bf16x8 b[2][3];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) b[0][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * V5_PLANE);
#ifndef V5_NO_SCHED
  __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // the first tile's fragments
#endif
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = t & 1;
    if (t + 1 < NT) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        b[c ^ 1][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * V5_PLANE + (t + 1) * 1024);
    }
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b[c][0], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[c][2], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[c][1], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[c][0], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[c][1], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[c][0], acc[t], 0, 0, 0);
#ifndef V5_NO_SCHED
    if (t + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // 3 LDS reads (next tile)
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                  // 6 matrix instructions (this tile)
#endif
  }
}

template <int CHALF>
__device__ __forceinline__ void v5_body(const float* __restrict__ A, int lda, float* __restrict__ C, int ldc, int M,
                                        int N, int K, const u32x4* __restrict__ Wp, int wp_rows, int wp_ld8,
                                        unsigned char* smem) {
  constexpr int T0 = CHALF ? 8 : 0, NT = CHALF ? 7 : 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rg = wave & 3;
  const int m0 = blockIdx.x * BM;
  const int r = lane & 31, h = lane >> 5;
  constexpr int WCH = 3 * V5_ROWS * 2;  // 16-byte chunks of a stage
  constexpr int WPT = (WCH + 511) / 512;

  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  const float* a_row = A + (size_t)min(m0 + rg * 32 + r, M - 1) * lda + 8 * h;
  float4 ra[2];
  u32x4 rw[WPT];
  const int nk = (K + 15) / 16;
  auto load_a = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = 16 * s + 8 * h + 4 * i;
      ra[i] = *reinterpret_cast<const float4*>(a_row + (k + 3 < lda ? 16 * s + 4 * i : -8 * h));
    }
  };
  auto load_w = [&](int s) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 512 * i, WCH - 1);
      const int plane = c / (V5_ROWS * 2), rc = c % (V5_ROWS * 2), row = rc >> 1, ch = rc & 1;
      rw[i] = Wp[((size_t)plane * wp_rows + row) * wp_ld8 + 2 * s + ch];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 512 * i, WCH - 1);
      *reinterpret_cast<u32x4*>(smem + buf * V5_STAGE + 16 * c) = rw[i];  // chunk c = (plane, row, half) in order
    }
  };
  load_a(0);
  load_w(0);
  store_w(0);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const bool more = s + 1 < nk;
    float4 v[2] = {ra[0], ra[1]};
    if (s == nk - 1 && (K & 15)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = 16 * s + 8 * h + 4 * i;
        v[i].x = k + 0 < K ? v[i].x : 0.f;
        v[i].y = k + 1 < K ? v[i].y : 0.f;
        v[i].z = k + 2 < K ? v[i].z : 0.f;
        v[i].w = k + 3 < K ? v[i].w : 0.f;
      }
    }
    if (more) {
      load_a(s + 1);
      load_w(s + 1);
    }
    unsigned p[3][4];
    split3(v[0].x, v[0].y, p[0][0], p[1][0], p[2][0]);
    split3(v[0].z, v[0].w, p[0][1], p[1][1], p[2][1]);
    split3(v[1].x, v[1].y, p[0][2], p[1][2], p[2][2]);
    split3(v[1].z, v[1].w, p[0][3], p[1][3], p[2][3]);
    const u32x4 q1 = {p[0][0], p[0][1], p[0][2], p[0][3]};
    const u32x4 q2 = {p[1][0], p[1][1], p[1][2], p[1][3]};
    const u32x4 q3 = {p[2][0], p[2][1], p[2][2], p[2][3]};
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, q1), a2 = __builtin_bit_cast(bf16x8, q2),
                 a3 = __builtin_bit_cast(bf16x8, q3);
#ifdef V5_NO_LDSREAD
    const unsigned char* wb = smem + (s & 1) * V5_STAGE + 16 * h - (size_t)lane * 0;
    v5_step<NT>(acc, a1, a2, a3, smem + 16 * lane - 1024 * 0);
#else
    const unsigned char* wb = smem + (s & 1) * V5_STAGE + (32 * T0 + r) * 32 + 16 * h;
    v5_step<NT>(acc, a1, a2, a3, wb);
#endif
    if (more) store_w((s + 1) & 1);
#ifndef V5_NO_BARRIER
    __syncthreads();
#endif
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = m0 + rg * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (i >= M) continue;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 32 * (T0 + t) + r;
      if (n < N) C[(size_t)i * ldc + n] = acc[t][q];
    }
  }
}

__global__ __launch_bounds__(512, 1) void nt_v5(const float* __restrict__ A, int lda, float* __restrict__ C, int ldc,
                                                int M, int N, int K, const u32x4* __restrict__ Wp, int wp_rows,
                                                int wp_ld8) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * V5_STAGE];
  if (threadIdx.x >> 8)
    v5_body<1>(A, lda, C, ldc, M, N, K, Wp, wp_rows, wp_ld8, smem);
  else
    v5_body<0>(A, lda, C, ldc, M, N, K, Wp, wp_rows, wp_ld8, smem);
}

// v3: K steps of 16 through a two-stage LDS ring.  While the matrix instructions of step s run from stage s & 1, the
// same wave splits the (already arrived) global values of step s + 1 and writes them into the other stage, in program
// order between the groups of matrix instructions, and the loads of step s + 2 are in flight: one barrier per step,
// and no phase in which a block does nothing but stage.
template <int NACC>
__global__ __launch_bounds__(256, 3) void nt_v3(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                int ldw, float* __restrict__ C, int ldc, int M, int N, int K,
                                                int n_tiles_n) {
  constexpr int BN = 32 * NACC;
  constexpr int ROWS = BM + BN;
  constexpr int PLANE = ROWS * 32;  // bytes: 16 bf16 per row
  constexpr int STAGE = 3 * PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / n_tiles_n) * BM, n0 = (tile % n_tiles_n) * BN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  // staging: four threads per row (4 k each), rows tid >> 2 and + 64 of the A tile; W rows (tid >> 2) and, for the
  // first half of the block, 64 + (tid >> 2) (the second half repeats row 95's last store: same bytes, no branch)
  const int srow = tid >> 2, kc = tid & 3;
  constexpr int NW = (BN * 4 + 255) / 256;
  const float* a_ptr[2];
  const float* w_ptr[NW];
  int a_off[2], w_off[NW];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = srow + 64 * i;
    a_ptr[i] = A + (size_t)min(m0 + row, M - 1) * lda + 4 * kc;
    a_off[i] = row * 32 + 16 * ((kc >> 1) ^ ((row >> 3) & 1)) + 8 * (kc & 1);
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int row = min(srow + 64 * i, BN - 1);
    w_ptr[i] = W + (size_t)min(n0 + row, N - 1) * ldw + 4 * kc;
    w_off[i] = (BM + row) * 32 + 16 * ((kc >> 1) ^ (((BM + row) >> 3) & 1)) + 8 * (kc & 1);
  }
  const int ns = (K + 15) >> 4;
  float4 ga[2][2], gw[2][NW];  // [register set][item]
  auto gload = [&](int set, int s) {
    // the last step reaches past K: its loads come from column 0 instead and are zeroed when consumed
    const int k0 = (16 * s + 4 * kc + 3 < lda && 16 * s + 4 * kc + 3 < ldw) ? 16 * s : -4 * kc;
#pragma unroll
    for (int i = 0; i < 2; ++i) ga[set][i] = *reinterpret_cast<const float4*>(a_ptr[i] + k0);
#pragma unroll
    for (int i = 0; i < NW; ++i) gw[set][i] = *reinterpret_cast<const float4*>(w_ptr[i] + k0);
  };
  auto put = [&](unsigned char* stage, int off, float4 v, int s) {
    if (16 * s + 16 > K) {  // uniform: only the last step
      const int k = 16 * s + 4 * kc;
      v.x = k + 0 < K ? v.x : 0.f;
      v.y = k + 1 < K ? v.y : 0.f;
      v.z = k + 2 < K ? v.z : 0.f;
      v.w = k + 3 < K ? v.w : 0.f;
    }
    unsigned p1a, p2a, p3a, p1b, p2b, p3b;
    split3(v.x, v.y, p1a, p2a, p3a);
    split3(v.z, v.w, p1b, p2b, p3b);
    *reinterpret_cast<uint2*>(stage + off) = make_uint2(p1a, p1b);
    *reinterpret_cast<uint2*>(stage + PLANE + off) = make_uint2(p2a, p2b);
    *reinterpret_cast<uint2*>(stage + 2 * PLANE + off) = make_uint2(p3a, p3b);
  };
  const int a_frag = (wave * 32 + r) * 32 + 16 * (h ^ (((wave * 32 + r) >> 3) & 1));
  int b_frag[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) b_frag[t] = (BM + 32 * t + r) * 32 + 16 * (h ^ (((BM + 32 * t + r) >> 3) & 1));

  auto step = [&](int s, int set_next) {
    // set_next holds the values of step s + 1 (if any); the other set is free for the loads of step s + 2
    const unsigned char* cur = smem + (s & 1) * STAGE;
    unsigned char* nxt = smem + ((s + 1) & 1) * STAGE;
    const bool has_next = s + 1 < ns;
    if (s + 2 < ns) gload(set_next ^ 1, s + 2);
    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(cur + a_frag);
    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(cur + PLANE + a_frag);
    const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(cur + 2 * PLANE + a_frag);
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(cur + b_frag[t]);
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(cur + PLANE + b_frag[t]);
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(cur + 2 * PLANE + b_frag[t]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      // a share of the next step's staging behind each group of six
      if (has_next) {
        if (t == 0) {
          put(nxt, a_off[0], ga[set_next][0], s + 1);
          put(nxt, a_off[1], ga[set_next][1], s + 1);
        } else if (t - 1 < NW) {
          put(nxt, w_off[t - 1], gw[set_next][t - 1], s + 1);
        }
      }
#ifdef V3_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();
  };

  gload(0, 0);
  if (ns > 1) gload(1, 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) put(smem, a_off[i], ga[0][i], 0);
#pragma unroll
  for (int i = 0; i < NW; ++i) put(smem, w_off[i], gw[0][i], 0);
  __syncthreads();
  for (int s = 0; s < ns; s += 2) {
    step(s, 1);
    if (s + 1 < ns) step(s + 1, 0);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = m0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (i >= M) continue;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int n = n0 + 32 * t + r;
      if (n < N) C[(size_t)i * ldc + n] = acc[t][q];
    }
  }
}

// v4: A goes through LDS as fp32 (one 16-byte store per staged float4, no vector work in the staging phase) and is
// split in registers after the fragment read - every wave reads only its own 32 rows, so each element is split
// once per block as before; W arrives pre-split (three bf16 planes in global memory, zero padded) and is copied
// to LDS as it is.  34 KB of LDS per block.
template <int NACC>
__global__ __launch_bounds__(256, 3) void nt_v4(const float* __restrict__ A, int lda, float* __restrict__ C, int ldc,
                                                int M, int N, int K, int n_tiles_n, const u32x4* __restrict__ Wp,
                                                int wp_rows, int wp_ld8) {
  constexpr int BN = 32 * NACC;
  constexpr int ABYTES = BM * 128;           // fp32 A tile, 128-byte rows, 16-byte slots XORed with (row >> 1) & 7
  constexpr int WPLANE = BN * 64;
  constexpr int WCH = 3 * BN * 4, WPT = (WCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ABYTES + 3 * WPLANE];
  unsigned char* wl = smem + ABYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / n_tiles_n) * BM, n0 = (tile % n_tiles_n) * BN;
  const int r = lane & 31, h = lane >> 5;

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  const int srow = tid >> 3, sc4 = tid & 7;
  float4 ra[BM / 32];
  u32x4 rw[WPT];
  const float* a_row[BM / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) a_row[i] = A + (size_t)min(m0 + srow + 32 * i, M - 1) * lda;
  const int nk = (K + BK - 1) / BK;
  auto load_tiles = [&](int kt) {
    const int k = kt * BK + 4 * sc4;
    const int kc = k + 3 < lda ? k : 0;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) ra[i] = *reinterpret_cast<const float4*>(a_row[i] + kc);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 256 * i, WCH - 1);
      const int plane = c / (BN * 4), rc = c % (BN * 4), row = rc >> 2, ch = rc & 3;
      rw[i] = Wp[((size_t)plane * wp_rows + n0 + row) * wp_ld8 + kt * 4 + ch];
    }
  };
  auto store_tiles = [&](int kt) {
    const bool tail = kt * BK + BK > K;
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
      float4 v = ra[i];
      if (tail) {  // what lies past K may be anything (also NaN): W's zeros there do not protect against it
        const int k = kt * BK + 4 * sc4;
        v.x = k + 0 < K ? v.x : 0.f;
        v.y = k + 1 < K ? v.y : 0.f;
        v.z = k + 2 < K ? v.z : 0.f;
        v.w = k + 3 < K ? v.w : 0.f;
      }
      const int row = srow + 32 * i;
      *reinterpret_cast<float4*>(smem + row * 128 + 16 * (sc4 ^ ((row >> 1) & 7))) = v;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = min(tid + 256 * i, WCH - 1);
      const int plane = c / (BN * 4), rc = c % (BN * 4), row = rc >> 2, ch = rc & 3;
      *reinterpret_cast<u32x4*>(wl + plane * WPLANE + row * 64 + 16 * ((ch + (row >> 2)) & 3)) = rw[i];
    }
  };
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int arow = wave * 32 + r;
  const int asw = (arow >> 1) & 7;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tiles(kt + 1);
    const int ks_end = min(2, (K - kt * BK + 15) >> 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s >= ks_end) break;
      // lane (r, h): k = 16 s + 8 h .. + 8 of its row: slots 4 s + 2 h and + 1
      const int sl = 4 * s + 2 * h;
      const float4 x0 = *reinterpret_cast<const float4*>(smem + arow * 128 + 16 * (sl ^ asw));
      const float4 x1 = *reinterpret_cast<const float4*>(smem + arow * 128 + 16 * ((sl + 1) ^ asw));
      unsigned p[3][4];
      split3(x0.x, x0.y, p[0][0], p[1][0], p[2][0]);
      split3(x0.z, x0.w, p[0][1], p[1][1], p[2][1]);
      split3(x1.x, x1.y, p[0][2], p[1][2], p[2][2]);
      split3(x1.z, x1.w, p[0][3], p[1][3], p[2][3]);
      const u32x4 q1 = {p[0][0], p[0][1], p[0][2], p[0][3]};
      const u32x4 q2 = {p[1][0], p[1][1], p[1][2], p[1][3]};
      const u32x4 q3 = {p[2][0], p[2][1], p[2][2], p[2][3]};
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, q1), a2 = __builtin_bit_cast(bf16x8, q2),
                   a3 = __builtin_bit_cast(bf16x8, q3);
      const int slot = 2 * s + h;
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const int brow = 32 * t + r;
        const int boff = brow * 64 + 16 * ((slot + (brow >> 2)) & 3);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wl + boff);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(wl + WPLANE + boff);
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(wl + 2 * WPLANE + boff);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_tiles(kt + 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = m0 + wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (i >= M) continue;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const int n = n0 + 32 * t + r;
      if (n < N) C[(size_t)i * ldc + n] = acc[t][q];
    }
  }
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 85181, N = 450, K = 450, ld = 452;
  std::vector<float> hA((size_t)M * ld, 0.f), hW((size_t)N * ld, 0.f), hC((size_t)M * ld);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    unsigned hi = s >> 8;  // 24 random bits
    s = s * 1664525u + 1013904223u;
    return (float)(((double)hi + (double)(s >> 8) / 16777216.0) / 8388608.0 - 1.0);  // full 24-bit significands
  };
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < K; ++k) hA[(size_t)i * ld + k] = getenv("ZERO_A") ? 0.f : rnd() * (1.f + (i % 7));
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) hW[(size_t)n * ld + k] = getenv("ZERO_W") ? 0.f : rnd() * 0.05f;
  float *dA, *dW, *dC;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dC, hC.size() * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0, hC.size() * 4));
#ifndef NACC_
#define NACC_ 3
#endif
  constexpr int NACC = NACC_;
  const int ntn = (N + 32 * NACC - 1) / (32 * NACC), ntm = (M + BM - 1) / BM;
  const int wp_rows = 480, wp_ld8 = 480 / 8;
  std::vector<unsigned short> hWp((size_t)3 * wp_rows * 480, 0);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      float x = hW[(size_t)n * ld + k];
      for (int pl = 0; pl < 3; ++pl) {
        unsigned u;
        memcpy(&u, &x, 4);
        unsigned rnd_ = ((u >> 16) & 1) + 0x7fffu;
        unsigned short hb = (unsigned short)((u + rnd_) >> 16);
        hWp[((size_t)pl * wp_rows + n) * 480 + k] = hb;
        unsigned back = (unsigned)hb << 16;
        float fb;
        memcpy(&fb, &back, 4);
        x -= fb;
      }
    }
  uint4* dWp;
  CK(hipMalloc(&dWp, hWp.size() * 2));
  CK(hipMemcpy(dWp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
  unsigned long long* dT;
  CK(hipMalloc(&dT, (size_t)ntm * ntn * 4 * 8 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const bool use_v2 = getenv("V2") != nullptr, use_v3 = getenv("V3") != nullptr;
  const bool use_v4 = getenv("V4") != nullptr, use_v5 = getenv("V5") != nullptr;
  auto launch = [&]() {
    if (use_v5)
      hipLaunchKernelGGL(nt_v5, dim3(ntm), dim3(512), 0, 0, dA, ld, dC, ld, M, N, K, (const u32x4*)dWp, wp_rows, wp_ld8);
    else if (use_v4)
      hipLaunchKernelGGL(nt_v4<NACC>, dim3(ntm * ntn), dim3(256), 0, 0, dA, ld, dC, ld, M, N, K, ntn,
                         (const u32x4*)dWp, wp_rows, wp_ld8);
    else if (use_v3)
      hipLaunchKernelGGL(nt_v3<NACC>, dim3(ntm * ntn), dim3(256), 0, 0, dA, ld, dW, ld, dC, ld, M, N, K, ntn);
    else if (use_v2)
      hipLaunchKernelGGL(nt_v2<NACC>, dim3(ntm * ntn), dim3(256), 0, 0, dA, ld, dC, ld, M, N, K, ntn,
                         (const u32x4*)dWp, wp_rows, wp_ld8);
    else
      hipLaunchKernelGGL(nt_bf16x6<NACC>, dim3(ntm * ntn), dim3(256), 0, 0, dA, ld, dW, ld, dC, ld, M, N, K, ntn, dT, dWp, wp_rows, wp_ld8);
  };
  for (int it = 0; it < (getenv("WARM") ? atoi(getenv("WARM")) : 3); ++it)
    launch();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int it = 0; it < reps; ++it) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s (fp32-equivalent)\n", M, N, K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) / 1e12);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
#ifdef TRACE
  {
    std::vector<unsigned long long> hT((size_t)ntm * ntn * 4 * 8);
    CK(hipMemcpy(hT.data(), dT, hT.size() * 8, hipMemcpyDeviceToHost));
    double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t w = 0; w < (size_t)ntm * ntn * 4; ++w)
      for (int i = 0; i < 8; ++i) ph[i] += hT[w * 8 + i];
    const char* nm[8] = {"prologue", "frag reads + mfma issue", "barrier 1", "split + lds store issue", "barrier 2", "epilogue", "wait for global values", "wait for lds stores"};
    for (int i = 0; i < 8; ++i) printf("  %-34s %9.0f cycles per wave (whole block life)\n", nm[i], ph[i] / (ntm * ntn * 4.0));
  }
#endif
  // accuracy against fp64 on a sample of rows, beside a plain fp32 fma chain
  double worst = 0, worst32 = 0;
  for (int i = 0; i < M; i += 997) {
    for (int n = 0; n < N; n += 7) {
      double ref = 0, mag = 0;
      float c32 = 0.f;
      for (int k = 0; k < K; ++k) {
        double p = (double)hA[(size_t)i * ld + k] * hW[(size_t)n * ld + k];
        ref += p;
        mag += fabs(p);
        c32 = fmaf(hA[(size_t)i * ld + k], hW[(size_t)n * ld + k], c32);
      }
      worst = fmax(worst, fabs(hC[(size_t)i * ld + n] - ref) / mag);
      worst32 = fmax(worst32, fabs(c32 - ref) / mag);
    }
  }
  printf("max |err| / sum|a b|: split-bf16 %.3e   fp32 fma chain %.3e\n", worst, worst32);
  return 0;
}
