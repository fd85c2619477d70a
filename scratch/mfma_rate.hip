// Bare matrix-pipe rate: every SIMD of the chip issues v_mfma_f32_32x32x16_bf16 (or the fp32 32x32x2 form) back to
// back on register operands holding random data; reports the sustained rate and the shader clock it ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__global__ __launch_bounds__(256) void rate(float* out, int iters, unsigned long long* clk) {
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  u32x4 qa[3], qb[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      // random bf16 pairs with exponents near 1
      qa[i][j] = (rnd() & 0x807f807fu) | 0x3f003f00u;
      qb[i][j] = (rnd() & 0x807f807fu) | 0x3f003f00u;
    }
  f32x16 acc[3];
  for (int t = 0; t < 3; ++t)
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (BF) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3 - i; ++j)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[i]),
                                                             __builtin_bit_cast(bf16x8, qb[j]), acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(qa[i % 3][i % 4]),
                                                        __uint_as_float(qb[i % 3][(i + 1) % 4]), acc[t], 0, 0, 0);
      }
    }
  }
  float x = 0;
  for (int t = 0; t < 3; ++t)
    for (int q = 0; q < 16; ++q) x += acc[t][q];
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if (blockIdx.x == 100 && threadIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - c0;
    clk[1] = wall_clock64() - w0;
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 768, iters = 4000;
  float* out;
  unsigned long long *clk, h[2];
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int bf = 1; bf >= 0; --bf) {
    for (int rep = 0; rep < 60; ++rep) {
      if (rep == 40) hipEventRecord(e0);
      if (bf) hipLaunchKernelGGL(rate<true>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
      else hipLaunchKernelGGL(rate<false>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    }
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    // one iteration of a wave: 3 tiles x (6 bf16 MFMAs of K = 16 | 8 fp32 MFMAs of K = 2) = 3 x 32 x 32 x 16 x 2 flop
    double flop = (double)blocks * 4 * iters * 3 * 32 * 32 * 16 * 2;
    printf("%s: %.3f ms per launch, %.1f TFLOP/s fp32-equivalent (%s), shader clock %.2f GHz\n",
           bf ? "split-bf16 (6 x 32x32x16 bf16)" : "fp32 (8 x 32x32x2 f32)", ms, flop / (ms * 1e-3) / 1e12,
           bf ? "bf16 rate / 6" : "fp32 rate", h[0] / (h[1] * 10.0) / 1e3 * 1e3 / 1e3);
  }
  return 0;
}
