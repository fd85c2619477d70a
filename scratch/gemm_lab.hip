// Ablation lab for the fp32 MFMA NT GEMM (not part of the product). Build with -DBN=.. -DFLAGS=.. -DMINW=..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BK = 32, LDS_S = 36;
constexpr int NT = BN / 32;
// FLAGS bit0: skip epilogue stores; bit1: skip global loads in loop; bit2: skip LDS stores+barriers in loop; bit3: skip ds_reads
__global__ __launch_bounds__(256, MINW) void lab(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                  int M, int N, int K, int ld, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_S];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_S];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb_ = gridDim.x; const int q_ = nb_ >> 3, r_ = nb_ & 7, x_ = blockIdx.x & 7;
  const int tile = (x_ < r_ ? x_ * (q_ + 1) : r_ * (q_ + 1) + (x_ - r_) * q_) + (blockIdx.x >> 3);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int srow = tid >> 3, sc4 = tid & 7;
  float4 ra[4], rb[NT];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { int r = min(m0 + srow + 32 * i, M - 1); int k = k0 + 4 * sc4; k = k < K ? k : 0; ra[i] = *(const float4*)(A + (size_t)r * ld + k); }
#pragma unroll
    for (int i = 0; i < NT; ++i) { int r = min(n0 + srow + 32 * i, N - 1); int k = k0 + 4 * sc4; k = k < K ? k : 0; rb[i] = *(const float4*)(W + (size_t)r * ld + k); }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float4*)&As[(srow + 32 * i) * LDS_S + 4 * sc4] = ra[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) *(float4*)&Bs[(srow + 32 * i) * LDS_S + 4 * sc4] = rb[i];
  };
  const int nk = (K + BK - 1) / BK;
  load_tiles(0); store_tiles(); __syncthreads();
  const float* a_frag = &As[(wave * 32 + (lane & 31)) * LDS_S + 4 * (lane >> 5)];
  const float* b_frag = &Bs[(lane & 31) * LDS_S + 4 * (lane >> 5)];
  float4 av = *(const float4*)a_frag; float4 bv[NT];
  for (int t = 0; t < NT; ++t) bv[t] = *(const float4*)(b_frag + t * 32 * LDS_S);
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more && !(FLAGS & 2)) load_tiles((kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      if (!(FLAGS & 8)) {
        av = *(const float4*)(a_frag + 8 * kk);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = *(const float4*)(b_frag + t * 32 * LDS_S + 8 * kk);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
    }
    if (!(FLAGS & 4)) {
      __syncthreads();
      if (more) { store_tiles(); __syncthreads(); }
    }
  }
  if (FLAGS & 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(acc[t]));
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 32 * t + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M && n < N) C[(size_t)m * ld + n] = acc[t][r];
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 42943, N = 450, K = 450, ld = argc > 2 ? atoi(argv[2]) : 452;
  float *A, *W, *C;
  if (hipMalloc(&A, (size_t)M * ld * 4) != hipSuccess) return 1;
  if (hipMalloc(&W, (size_t)N * ld * 4) != hipSuccess) return 1;
  if (hipMalloc(&C, (size_t)M * ld * 4) != hipSuccess) return 1;
  std::vector<float> h((size_t)M * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(W, h.data(), (size_t)N * ld * 4, hipMemcpyHostToDevice);
  int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) lab<<<dim3(tiles_m * tiles_n), dim3(256), 0, 0>>>(A, W, C, M, N, K, ld, tiles_n);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) lab<<<dim3(tiles_m * tiles_n), dim3(256), 0, 0>>>(A, W, C, M, N, K, ld, tiles_n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float us = ms / reps * 1e3f;
  printf("ld=%d M=%d BN=%3d flags=%2d minw=%d blocks=%d : %7.1f us  %6.1f TF (alg 2MNK)\n", ld, M, BN, FLAGS, MINW, tiles_m * tiles_n, us, 2.0 * M * N * K / 1e6 / us);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
