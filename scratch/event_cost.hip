// Host cost of the cross-stream primitives: hipEventRecord + hipStreamWaitEvent between two busy streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(float* p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.1f; p[threadIdx.x] = x; }
int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  float* p; hipMalloc(&p, 4096);
  const int N = 2000;
  hipEvent_t ev[64];
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, p, 10);
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, p, 10);
      hipEventRecord(ev[i & 63], a);
      hipStreamWaitEvent(b, ev[i & 63], 0);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, p + 64, 10);
    }
    auto t3 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto t4 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      hipEventRecord(ev[i & 63], a);
    }
    auto t5 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      hipStreamWaitEvent(b, ev[i & 63], 0);
    }
    auto t6 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    printf("launch alone %.2f us   launch+record+wait+launch %.2f us   record %.2f us   wait %.2f us\n", us(t0, t1) / N,
           us(t2, t3) / N, us(t4, t5) / N, us(t5, t6) / N);
  }
  return 0;
}
