// GEMM lab v2 (not part of the product): tile-shape / prefetch-depth exploration for the NT kernel.
// Build with -DNW=4|8 (waves; BM = 32*NW) -DBN=96|160 -DDEPTH=1|2 -DMINW=..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef AWRAP
#define AWRAP 1000000000
#endif
constexpr int BM = 32 * NW, BK = 32, LDS_S = 36, NT = BN / 32, THREADS = 64 * NW;
constexpr int NB_LOADS = (BN * 8 + THREADS - 1) / THREADS;
__device__ __forceinline__ int xcd_remap(int id, int n) { int q = n >> 3, r = n & 7, x = id & 7; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3); }
__global__ __launch_bounds__(THREADS, MINW) void lab(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                     int M, int N, int K, int ld, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_S];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int srow = tid >> 3, sc4 = tid & 7;
  constexpr int RSTEP = THREADS / 8;  // rows covered per staging pass
  const float* arow[4]; const float* brow[NB_LOADS];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = A + (size_t)(min(m0 + srow + RSTEP * i, M - 1) % AWRAP) * ld + 4 * sc4;
#pragma unroll
  for (int i = 0; i < NB_LOADS; ++i) brow[i] = W + (size_t)min(n0 + min(srow + RSTEP * i, BN - 1), N - 1) * ld + 4 * sc4;
  float4 ra[DEPTH][4], rb[DEPTH][NB_LOADS];
  float4 sa[4], sb[NB_LOADS];
  for (int i = 0; i < 4; ++i) sa[i] = make_float4(0.1f * i, 0.2f, 0.3f, 0.4f);
  for (int i = 0; i < NB_LOADS; ++i) sb[i] = make_float4(0.1f, 0.2f * i, 0.3f, 0.4f);
#define LOAD(S, k0) { _Pragma("unroll") for (int i = 0; i < 4; ++i) ra[S][i] = *(const float4*)(arow[i] + ((k0) + 4 * sc4 < K ? (k0) : -4 * sc4)); \
                      _Pragma("unroll") for (int i = 0; i < NB_LOADS; ++i) rb[S][i] = *(const float4*)(brow[i] + ((k0) + 4 * sc4 < K ? (k0) : -4 * sc4)); }
#ifdef NOSTOREDEP
#define STORE(S) { _Pragma("unroll") for (int i = 0; i < 4; ++i) *(float4*)&As[(srow + RSTEP * i) * LDS_S + 4 * sc4] = sa[i]; \
                   _Pragma("unroll") for (int i = 0; i < NB_LOADS; ++i) if (srow + RSTEP * i < BN) *(float4*)&Bs[(srow + RSTEP * i) * LDS_S + 4 * sc4] = sb[i]; }
#else
#define STORE(S) { _Pragma("unroll") for (int i = 0; i < 4; ++i) *(float4*)&As[(srow + RSTEP * i) * LDS_S + 4 * sc4] = ra[S][i]; \
                   _Pragma("unroll") for (int i = 0; i < NB_LOADS; ++i) if (srow + RSTEP * i < BN) *(float4*)&Bs[(srow + RSTEP * i) * LDS_S + 4 * sc4] = rb[S][i]; }
#endif
  const float* a_frag = &As[(wave * 32 + (lane & 31)) * LDS_S + 4 * (lane >> 5)];
  const float* b_frag = &Bs[(lane & 31) * LDS_S + 4 * (lane >> 5)];
#define COMPUTE() { _Pragma("unroll") for (int kk = 0; kk < BK / 8; ++kk) { \
      float4 av = *(const float4*)(a_frag + 8 * kk); float4 bv[NT]; \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) bv[t] = *(const float4*)(b_frag + t * 32 * LDS_S + 8 * kk); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0); \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0); } }
  const int nk = (K + BK - 1) / BK;
#if DEPTH == 1
  LOAD(0, 0) STORE(0) __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) LOAD(0, (kt + 1) * BK)
    COMPUTE()
    __syncthreads();
    if (kt + 1 < nk) { STORE(0) __syncthreads(); }
  }
#else
  // two tiles in flight in registers: set 0 holds odd tiles, set 1 even tiles (static indexing: loop unrolled by 2)
  LOAD(0, 0) STORE(0)
  if (1 < nk) LOAD(0, BK)
  if (2 < nk) LOAD(1, 2 * BK)
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    COMPUTE()                                   // tile kt
    __syncthreads();
    if (kt + 1 < nk) { STORE(0) }               // tile kt+1
    if (kt + 3 < nk) LOAD(0, (kt + 3) * BK)
    __syncthreads();
    if (kt + 1 < nk) {
      COMPUTE()                                 // tile kt+1
      __syncthreads();
      if (kt + 2 < nk) { STORE(1) }             // tile kt+2
      if (kt + 4 < nk) LOAD(1, (kt + 4) * BK)
      __syncthreads();
    }
  }
#endif
#ifdef NOSTOREDEP
  for (int d = 0; d < DEPTH; ++d) { for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ra[d][i])); for (int i = 0; i < NB_LOADS; ++i) asm volatile("" ::"v"(rb[d][i])); }
#endif
#ifdef NOEPI
  for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(acc[t]));
#else
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= M) continue;
    float* crow = C + (size_t)m * ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) { const int n = n0 + 32 * t + (lane & 31); if (n < N) crow[n] = acc[t][r]; }
  }
#endif
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 42943, N = 450, K = 450, ld = argc > 2 ? atoi(argv[2]) : 452;
  float *A, *W, *C;
  if (hipMalloc(&A, (size_t)M * ld * 4) != hipSuccess || hipMalloc(&W, (size_t)N * ld * 4) != hipSuccess || hipMalloc(&C, (size_t)M * ld * 4) != hipSuccess) return 1;
  std::vector<float> h((size_t)M * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), (size_t)N * ld * 4, hipMemcpyHostToDevice);
  int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) lab<<<dim3(tiles_m * tiles_n), dim3(THREADS), 0, 0>>>(A, W, C, M, N, K, ld, tiles_n);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) lab<<<dim3(tiles_m * tiles_n), dim3(THREADS), 0, 0>>>(A, W, C, M, N, K, ld, tiles_n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float us = ms / reps * 1e3f;
  // checksum against a host reference of a few entries
  std::vector<float> c((size_t)ld); (void)hipMemcpy(c.data(), C + (size_t)(M - 1) * ld, ld * 4, hipMemcpyDeviceToHost);
  double ref = 0; for (int k = 0; k < K; ++k) ref += (double)h[(size_t)(M - 1) * ld + k] * h[(size_t)(N - 1) * ld + k];
  printf("NW=%d BN=%3d depth=%d minw=%d ld=%d M=%d blocks=%d : %7.1f us %6.1f TF  (check %.4f vs %.4f)\n", NW, BN, DEPTH, MINW, ld, M, tiles_m * tiles_n, us, 2.0 * M * N * K / 1e6 / us, c[N - 1], ref);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
