// Pure-MFMA ceiling on gfx950: v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32, W waves per SIMD,
// no memory traffic.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16 (NACC independent accumulator chains) and v_mfma_f32_16x16x32_bf16
template <int NACC>
__global__ void kb32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  bf16x8 av, bv;
  for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ void kb16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  bf16x8 av, bv;
  for (int q = 0; q < 8; ++q) { av[q] = (__bf16)a; bv[q] = (__bf16)b; }
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}

template <typename F>
double run(F launch, double flops) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flops * 5 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD
    const int block = 256, grid = 256 * wps;
    const double waves = (double)grid * 4;
    double t32 = run([&] { k32<4><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 4096.0);
    double t16 = run([&] { k16<4><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 2048.0);
    double t16b = run([&] { k16<2><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 2 * 2048.0);
    double b32 = run([&] { kb32<4><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 32768.0);
    double b32c = run([&] { kb32<1><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 1 * 32768.0);
    double b16 = run([&] { kb16<4><<<grid, block>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 32768.0);
    printf("waves/SIMD %d: bf16 32x32x16 (4 chains) %.0f TF  (1 chain) %.0f TF   bf16 16x16x32 (4 chains) %.0f TF\n",
           wps, b32, b32c, b16);
    printf("waves/SIMD %d: 32x32x2 (4 chains) %.1f TF   16x16x4 (4 chains) %.1f TF   16x16x4 (2 chains) %.1f TF\n",
           wps, t32, t16, t16b);
  }
  return 0;
}
