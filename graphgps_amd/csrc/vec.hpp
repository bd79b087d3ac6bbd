// Small fixed-width float vectors for the HBM-bound row kernels: one lane owns VEC consecutive
// channels of a node/edge row, so a row of d floats is d/VEC consecutive lanes and every global
// access is a fully coalesced 4/8/16-byte-per-lane transaction.
#pragma once
#include <hip/hip_runtime.h>

template <int VEC>
struct Vec;

template <>
struct Vec<4> {
  float4 v;
  __device__ static Vec load(const float* p) { Vec r; r.v = *reinterpret_cast<const float4*>(p); return r; }
  __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ static Vec zero() { Vec r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
  __device__ float& operator[](int i) { return (&v.x)[i]; }
  __device__ float operator[](int i) const { return (&v.x)[i]; }
};
template <>
struct Vec<2> {
  float2 v;
  __device__ static Vec load(const float* p) { Vec r; r.v = *reinterpret_cast<const float2*>(p); return r; }
  __device__ void store(float* p) const { *reinterpret_cast<float2*>(p) = v; }
  __device__ static Vec zero() { Vec r; r.v = make_float2(0.f, 0.f); return r; }
  __device__ float& operator[](int i) { return (&v.x)[i]; }
  __device__ float operator[](int i) const { return (&v.x)[i]; }
};
template <>
struct Vec<1> {
  float v;
  __device__ static Vec load(const float* p) { Vec r; r.v = *p; return r; }
  __device__ void store(float* p) const { *p = v; }
  __device__ static Vec zero() { Vec r; r.v = 0.f; return r; }
  __device__ float& operator[](int) { return v; }
  __device__ float operator[](int) const { return v; }
};

__device__ __forceinline__ float sigmoidf_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
// The gate of the GatedGCN kernels, once per edge element: v_exp_f32 + v_rcp_f32 (each 1 ulp) instead of the
// correctly rounded expf and IEEE division (~20 VALU instructions) -- <= 3e-7 relative on sigma, which enters
// x~ through a ratio of sums of ~3 terms.  GPS_EXACT_SIGMOID at compile time restores the exact form.
__device__ __forceinline__ float sigmoidf_fast(float x) {
#ifdef GPS_EXACT_SIGMOID
  return sigmoidf_exact(x);
#else
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
#endif
}

// Dispatch on the widest vector that divides d AND keeps every row pointer aligned.
#define GPS_DISPATCH_VEC(d, ld_ok4, ld_ok2, ...)                              \
  do {                                                                        \
    if ((d) % 4 == 0 && (ld_ok4)) { constexpr int VEC = 4; __VA_ARGS__; }     \
    else if ((d) % 2 == 0 && (ld_ok2)) { constexpr int VEC = 2; __VA_ARGS__; } \
    else { constexpr int VEC = 1; __VA_ARGS__; }                              \
  } while (0)
