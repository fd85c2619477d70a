// Do vector ALU instructions overlap with matrix instructions on a gfx950 SIMD?
//   intra:  one wave per SIMD issues groups of {1 MFMA (32x32x16 bf16) + V independent v_fma_f32}
//   inter:  two waves per SIMD, one issuing only MFMAs, the other only v_fma_f32 (V per MFMA of its neighbour)
// Reports time per launch for V = 0..12 and, for the two-wave form, each wave kind alone.
// hipcc --offload-arch=gfx950 -O3 scratch/coissue.hip -o scratch/bin/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GROUP = 18;  // MFMAs per loop iteration (three accumulators x six products)

template <int V>
__device__ __forceinline__ void valu_ops(float (&f)[8], float m, float c) {
#pragma unroll
  for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v & 7]) : "v"(m), "v"(c));
}

// MODE 0: waves 0-3 do MFMA + V VALU per MFMA, waves 4-7 leave.  MODE 1: waves 0-3 MFMA only, waves 4-7 VALU only (V per MFMA).
// MODE 2: waves 0-3 MFMA only, waves 4-7 leave.  MODE 3: waves 4-7 VALU only, waves 0-3 leave.
// MODE 4: all eight waves MFMA only (two matrix waves per SIMD).  MODE 5: waves 0-3 MFMA only, waves 4-7 issue V
// ds_read_b128 per MFMA of their neighbour.  MODE 6: the LDS-reading waves alone.  MODE 7: waves 0-3 issue MFMA + V
// ds_read_b128 per MFMA themselves (results awaited once per group of 18).
// The inner loop of a product kernel and nothing else: per tile three ds_read_b128 (requested one tile ahead) feed six
// dependent MFMAs; PER_SIMD waves per SIMD; READS = false keeps the MFMAs and drops the reads.
template <int PER_SIMD, bool READS>
__global__ __launch_bounds__(512) void fed(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 480 * 32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 4 * PER_SIMD) return;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  for (int i = threadIdx.x; i < 3 * 480 * 32 / 4; i += 256 * PER_SIMD)
    reinterpret_cast<unsigned*>(lds)[i] = (rnd() & 0x807f807fu) | 0x3f003f00u;
  __syncthreads();
  u32x4 qa[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) qa[i][j] = (rnd() & 0x807f807fu) | 0x3f003f00u;
  const bf16x8 a1 = __builtin_bit_cast(bf16x8, qa[0]), a2 = __builtin_bit_cast(bf16x8, qa[1]),
               a3 = __builtin_bit_cast(bf16x8, qa[2]);
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  const unsigned char* wb = lds + (wave >> 2) * 7 * 1024 + (lane & 31) * 32 + 16 * (lane >> 5);
  constexpr int PLANE = 480 * 32;
  u32x4 b[2][3];
  const unsigned addr = (unsigned)(size_t)wb;
#define RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  RD(b[0][0], 0);
  RD(b[0][1], PLANE);
  RD(b[0][2], 2 * PLANE);
  if (!READS) asm volatile("s_waitcnt lgkmcnt(0)");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = READS ? (t & 1) : 0;
      if (READS) {
        // next tile (tile 0 of the next pass after the last one)
        RD(b[c ^ 1][0], ((t + 1) & 7) * 1024);
        RD(b[c ^ 1][1], PLANE + ((t + 1) & 7) * 1024);
        RD(b[c ^ 1][2], 2 * PLANE + ((t + 1) & 7) * 1024);
        asm volatile("s_waitcnt lgkmcnt(3)");
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, b[c][0]), b2 = __builtin_bit_cast(bf16x8, b[c][1]),
                   b3 = __builtin_bit_cast(bf16x8, b[c][2]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
#undef RD
  float x = 0;
  for (int t = 0; t < 8; ++t)
    for (int q = 0; q < 16; ++q) x += acc[t][q];
  out[blockIdx.x * 512 + threadIdx.x] = x;
}

template <int PER_SIMD, bool READS>
float run_fed(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 30; ++rep) {
    if (rep == 20) hipEventRecord(e0);
    hipLaunchKernelGGL((fed<PER_SIMD, READS>), dim3(blocks), dim3(512), 0, 0, out, iters);
  }
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

template <int V, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
  __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 64 * 16 * 4];
  if ((MODE == 0 || MODE == 7 || MODE == 8) && wave >= 4) return;  // same matrix work per SIMD in every mode but 4
  const bool do_mfma = MODE == 0 || MODE == 4 || MODE == 7 || MODE == 8 || MODE == 9 || ((MODE == 1 || MODE == 2 || MODE == 5) && wave < 4);
  const bool do_valu = MODE == 0 || ((MODE == 1 || MODE == 3) && wave >= 4);
  const bool do_lds = (MODE == 5 || MODE == 6) && wave >= 4;
  if (!do_mfma && !do_valu && !do_lds) return;
  const unsigned lds_addr = (unsigned)(size_t)(lds) + threadIdx.x * 16;  // consecutive 16-byte chunks: conflict-free
  if (do_lds || MODE == 7) {
    u32x4 sink[4];
    if (do_lds) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUP; ++g)
#pragma unroll
          for (int v = 0; v < V; ++v)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[v & 3]) : "v"(lds_addr), "n"(8192 * ((v) & 3)));
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
      out[blockIdx.x * 512 + threadIdx.x] = __uint_as_float(sink[0][0] ^ sink[1][1] ^ sink[2][2] ^ sink[3][3]);
      return;
    }
  }
  u32x4 qa[3], qb[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      qa[i][j] = (rnd() & 0x807f807fu) | 0x3f003f00u;
      qb[i][j] = (rnd() & 0x807f807fu) | 0x3f003f00u;
    }
  f32x16 acc[3];
  for (int t = 0; t < 3; ++t)
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = (float)(rnd() & 1023) * 1e-3f;
  const float m = 0.999f, c = 1e-3f;
  if (MODE == 8 || MODE == 9) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        // six in a row on one accumulator, as the product kernels issue them
        acc[g / 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[g % 3]),
                                                             __builtin_bit_cast(bf16x8, qb[(g / 2) % 3]), acc[g / 6], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (MODE == 0 || MODE == 7 || (do_mfma && !do_valu)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        // round-robin over the accumulators: no MFMA waits for the one before it
        acc[g % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[(g / 3) % 3]),
                                                             __builtin_bit_cast(bf16x8, qb[(g / 9) % 3]), acc[g % 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
          valu_ops<V>(f, m, c);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 7) {
          u32x4 sink7;
#pragma unroll
          for (int v = 0; v < V; ++v)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink7) : "v"(lds_addr), "n"(8192 * ((v) & 3)));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)");
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < GROUP; ++g) valu_ops<V>(f, m, c);
    }
  }
  float x = 0;
  for (int t = 0; t < 3; ++t)
    for (int q = 0; q < 16; ++q) x += acc[t][q];
  for (int i = 0; i < 8; ++i) x += f[i];
  out[blockIdx.x * 512 + threadIdx.x] = x;
}

template <int V, int MODE>
float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 30; ++rep) {
    if (rep == 20) hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, MODE>), dim3(blocks), dim3(512), 0, 0, out, iters);
  }
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

template <int V>
void row(float* out, int blocks, int iters) {
  const float a = run<V, 0>(out, blocks, iters), b = run<V, 1>(out, blocks, iters), c = run<V, 2>(out, blocks, iters),
              d = run<V, 3>(out, blocks, iters);
  printf("V=%2d  same wave %.3f ms   two waves: both %.3f  mfma wave alone %.3f  valu wave alone %.3f\n",
         V, a, b, c, d);
  if (V == 0) printf("      two matrix waves per SIMD: %.3f ms\n", run<0, 4>(out, blocks, iters));
  if (V == 0)
    printf("      chains of six dependent MFMAs per accumulator: one wave per SIMD %.3f ms, two waves %.3f ms (twice the work)\n",
           run<0, 8>(out, blocks, iters), run<0, 9>(out, blocks, iters));
  if (V > 0 && V <= 4)
    printf("      LDS reads (ds_read_b128 per MFMA): other wave %.3f  reading wave alone %.3f  same wave %.3f\n",
           run<V, 5>(out, blocks, iters), run<V, 6>(out, blocks, iters), run<V, 7>(out, blocks, iters));
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = 2000;
  float* out;
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  // warm the clocks
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<0, 2>), dim3(blocks), dim3(512), 0, 0, out, iters);
  hipDeviceSynchronize();
  {
    const int it = 750;  // 750 x 48 MFMAs = the 36 000 of the rows below
    printf("LDS-fed loop (3 ds_read_b128 one tile ahead -> 6 dependent MFMAs, 48 MFMAs per pass):\n");
    printf("  one wave per SIMD : with reads %.3f ms   without %.3f ms\n", run_fed<1, true>(out, blocks, it),
           run_fed<1, false>(out, blocks, it));
    printf("  two waves per SIMD: with reads %.3f ms   without %.3f ms (twice the work)\n",
           run_fed<2, true>(out, blocks, it), run_fed<2, false>(out, blocks, it));
  }
  row<0>(out, blocks, iters);
  row<1>(out, blocks, iters);
  row<2>(out, blocks, iters);
  row<4>(out, blocks, iters);
  row<6>(out, blocks, iters);
  row<8>(out, blocks, iters);
  row<12>(out, blocks, iters);
  return 0;
}
