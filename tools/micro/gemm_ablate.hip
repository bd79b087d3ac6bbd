// Ablation harness for csrc/gemm_split.hip (bf16-split NT GEMM): times gps_gemm_nt on one shape with
// hipEvents, no torch.  Built in variants by tools/micro/gemm_ablate.sh (-DGPS_ABL_*).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gps_hip.h"

int main(int argc, char** argv) {
  const int64_t R = argc > 1 ? atoll(argv[1]) : 7569;
  const int K = argc > 2 ? atoi(argv[2]) : 384;
  const int M = argc > 3 ? atoi(argv[3]) : 2688;
  const char* tag = argc > 4 ? argv[4] : "";
  std::vector<float> ha((size_t)R * K), hb((size_t)M * K);
  for (size_t i = 0; i < ha.size(); ++i) ha[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 40503u) % 1999) / 1000.0f - 1.0f;
  float *A, *B, *C;
  hipMalloc(&A, ha.size() * 4); hipMalloc(&B, hb.size() * 4); hipMalloc(&C, (size_t)R * M * 4);
  hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 5; ++i)
    if (gps_gemm_nt(A, K, B, K, R, M, K, nullptr, nullptr, 0, C, M, s) != 0) { printf("launch failed: %s\n", gps_last_error()); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStreamSynchronize(s);
  const int iters = 30;
  hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) gps_gemm_nt(A, K, B, K, R, M, K, nullptr, nullptr, 0, C, M, s);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;
  printf("%-14s R=%lld K=%d M=%d: %8.1f us  %6.1f TF/s (fp32-equivalent)\n", tag, (long long)R, K, M, us,
         2.0 * R * K * M / us / 1e6);
  return 0;
}
