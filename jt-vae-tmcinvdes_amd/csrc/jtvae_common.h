// Shared device helpers for the row kernels.
//
// Row buffers are [rows, ld] fp32 with ld % 4 == 0 and a 16-byte aligned base; a
// thread owns one float4 of one row ("flat" mapping: consecutive lanes walk along
// a row and on into the next one, so every lane of a wave is busy although
// H = 450 is not a multiple of 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JTVAE_BLOCK 256

namespace jtvae {

struct RowPos {
  int row;
  int c4;   // float4 index inside the row
  bool ok;
};

__device__ __forceinline__ RowPos row_pos(long total, int h4) {
  long gid = (long)blockIdx.x * JTVAE_BLOCK + threadIdx.x;
  RowPos p;
  p.ok = gid < total;
  if (!p.ok) gid = 0;
  p.row = (int)(gid / h4);
  p.c4 = (int)(gid - (long)p.row * h4);
  return p;
}

__device__ __forceinline__ const float4* rowp(const float* base, int row, int ld, int c4) {
  return reinterpret_cast<const float4*>(base + (size_t)row * ld) + c4;
}
__device__ __forceinline__ float4* rowp(float* base, int row, int ld, int c4) {
  return reinterpret_cast<float4*>(base + (size_t)row * ld) + c4;
}

// columns >= H of a row are padding: keep them zero whatever was computed there
__device__ __forceinline__ float4 zero_pad(float4 v, int c4, int H) {
  int c = c4 * 4;
  if (c + 3 >= H) {
    if (c + 0 >= H) v.x = 0.f;
    if (c + 1 >= H) v.y = 0.f;
    if (c + 2 >= H) v.z = 0.f;
    v.w = 0.f;
  }
  return v;
}

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 operator-(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 operator*(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 operator*(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ void fma4(float4& acc, float4 a, float4 b) {
  acc.x = fmaf(a.x, b.x, acc.x);
  acc.y = fmaf(a.y, b.y, acc.y);
  acc.z = fmaf(a.z, b.z, acc.z);
  acc.w = fmaf(a.w, b.w, acc.w);
}

// sigmoid / tanh on the hardware exp2 + rcp units (1 ulp each; the parity
// tolerance of the path is 1e-4 relative)
__device__ __forceinline__ float sigmoidf_fast(float x) {
  float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * x);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float tanhf_fast(float x) {
  // tanh(x) = 2*sigmoid(2x) - 1; exact limits for |x| large (e -> 0 or inf)
  float e = __builtin_amdgcn_exp2f(-2.88539008177792681472f * x);
  return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + e), -1.0f);
}
__device__ __forceinline__ float4 sigmoid4(float4 v) {
  return make_float4(sigmoidf_fast(v.x), sigmoidf_fast(v.y), sigmoidf_fast(v.z), sigmoidf_fast(v.w));
}
__device__ __forceinline__ float4 tanh4(float4 v) {
  return make_float4(tanhf_fast(v.x), tanhf_fast(v.y), tanhf_fast(v.z), tanhf_fast(v.w));
}

}  // namespace jtvae
