// Timeline harness for the tiled NT GEMM: builds the product source with its trace marks on,
// runs one GEMM and dumps the per-wave stamps.   usage: gemm_trace M out.bin
// build with -DJTVAE_GEMM_TRACE for the timeline, without it for plain timing
#include "../jt-vae-tmcinvdes_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 85181;
  const char* out = argc > 2 ? argv[2] : "trace.bin";
  const int N = argc > 3 ? atoi(argv[3]) : 450, K = N, ld = (N + 3) & ~3;
  std::vector<float> hA((size_t)M * ld), hW((size_t)N * ld);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 500.f - 1.f;
  for (size_t i = 0; i < hW.size(); ++i) hW[i] = ((i * 40503u) % 1000) / 500.f - 1.f;
  float *A, *W, *C;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&C, (size_t)M * ld * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  int blocks = ((M + 127) / 128) * ((N + 95) / 96);
#ifndef JTVAE_GEMM_TRACE
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) if (jtvae_gemm_nt(A, ld, W, ld, C, ld, M, N, K, nullptr, nullptr)) return 2;
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) if (jtvae_gemm_nt(A, ld, W, ld, C, ld, M, N, K, nullptr, nullptr)) return 2;
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M=%d blocks=%d  %.1f us  %.1f TF\n", M, blocks, ms * 50, 2.0 * M * N * K / (ms * 50e-6) / 1e12);
    return 0;
  }
#else
  // the library may pick a tiling with more blocks than the 128 x 96 one counted above (128 x 32 for mid-size
  // launches): size the stamp buffer for the finest tiling any kernel uses
  const int blocks_max = ((M + 31) / 32) * ((N + 31) / 32);
  size_t nst = (size_t)blocks_max * 4 * JTVAE_TRACE_SLOTS;
  unsigned long long* T;
  CK(hipMalloc(&T, nst * 8)); CK(hipMemset(T, 0, nst * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) if (jtvae_gemm_nt(A, ld, W, ld, C, ld, M, N, K, nullptr, nullptr)) return 2;
  CK(hipDeviceSynchronize());
  CK(hipMemcpyToSymbol(HIP_SYMBOL(jtvae_trace_buf), &T, sizeof(T)));
  CK(hipEventRecord(e0));
  if (jtvae_gemm_nt(A, ld, W, ld, C, ld, M, N, K, nullptr, nullptr)) return 2;
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("M=%d blocks=%d traced launch %.1f us\n", M, blocks, ms * 1e3);
  std::vector<unsigned long long> h(nst);
  CK(hipMemcpy(h.data(), T, nst * 8, hipMemcpyDeviceToHost));
  FILE* f = fopen(out, "wb"); fwrite(h.data(), 8, nst, f); fclose(f);
  return 0;
#endif
}
