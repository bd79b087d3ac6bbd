// Random-walk structural encoding (RWSE): landing probabilities diag(P^k), k = kmin..kmax, P = D^-1 A,
// for every graph of a batch in one launch.
//
// Replaces graphgps/transform/posenc_stats.py:184-230 (get_rw_landing_probs): per graph, on the CPU, a
// dense n x n `to_dense_adj`, `matrix_power` and K-1 dense matmuls -- minutes to hours of preprocessing
// for the 3.7 M graphs of PCQM4Mv2.  Here: one workgroup per graph walks the
// by-source CSR of the batch's graph index, builds P in LDS (graphs of <= 128 nodes; larger ones use a
// global scratch), and repeats Pk <- Pk P with fp32 FMAs, emitting the diagonal after every step.
// Same arithmetic as the reference: out-degree normalisation with 1/0 -> 0, multi-edges add, P^kmin by
// repeated multiplication, the k^(space_dim/2) correction.
#include "gps_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kLdsNodes = 128;   // 2 x 128 x 128 fp32 = 128 KB of the CU's 160 KB

__global__ __launch_bounds__(kThreads) void k_rwse(const int32_t* __restrict__ rowptr_src,
                                                   const int32_t* __restrict__ dst_by_src,
                                                   const int32_t* __restrict__ ptr, int kmin, int kmax,
                                                   float space_dim, float* __restrict__ scratch,
                                                   const int64_t* __restrict__ scratch_off,
                                                   float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int g = blockIdx.x;
  const int n0 = ptr[g], n = ptr[g + 1] - n0;
  if (n <= 0) return;
  const int K = kmax - kmin + 1;
  const bool in_lds = n <= kLdsNodes;
  // three n x n matrices: P, Pk, and the product being formed
  float* P = in_lds ? lds : scratch + scratch_off[g];
  float* A = P + (size_t)n * n;
  float* Bm = in_lds ? nullptr : A + (size_t)n * n;
  const int t = threadIdx.x;
  for (int i = t; i < n * n; i += kThreads) P[i] = 0.0f;
  __syncthreads();
  // P = D^-1 A from the by-source CSR (each source row is owned by one thread: no atomics)
  for (int v = t; v < n; v += kThreads) {
    const int beg = rowptr_src[n0 + v], end = rowptr_src[n0 + v + 1];
    const float inv = end > beg ? 1.0f / (float)(end - beg) : 0.0f;   // deg^-1, inf -> 0
    for (int k = beg; k < end; ++k) P[(size_t)v * n + (dst_by_src[k] - n0)] += 1.0f;
    for (int j = 0; j < n; ++j) P[(size_t)v * n + j] *= inv;
  }
  __syncthreads();
  for (int i = t; i < n * n; i += kThreads) A[i] = P[i];
  __syncthreads();
  // with two LDS matrices the product overwrites A row by row through registers: thread -> (row, cols)
  // simpler and exact: form the product into a third matrix when in global scratch, or into
  // registers + barrier when in LDS
  auto multiply = [&]() {
    if (in_lds) {
      // each thread owns whole output elements; accumulate all of them first, then write.  The element loop is unrolled
      // over its compile-time maximum so that `acc` is 64 registers, not a private array in scratch memory (it was:
      // the only kernel of the library with a scratch segment).
      constexpr int kAcc = (kLdsNodes * kLdsNodes + kThreads - 1) / kThreads;
      float acc[kAcc];
#pragma unroll
      for (int cnt = 0; cnt < kAcc; ++cnt) {
        const int i = t + cnt * kThreads;
        float s = 0.0f;
        if (i < n * n) {
          const int r = i / n, c = i - r * n;
          for (int k = 0; k < n; ++k) s += A[r * n + k] * P[k * n + c];
        }
        acc[cnt] = s;
      }
      __syncthreads();
#pragma unroll
      for (int cnt = 0; cnt < kAcc; ++cnt) {
        const int i = t + cnt * kThreads;
        if (i < n * n) A[i] = acc[cnt];
      }
      __syncthreads();
    } else {
      for (int i = t; i < n * n; i += kThreads) {
        const int r = i / n, c = i - r * n;
        float s = 0.0f;
        for (int k = 0; k < n; ++k) s += A[(size_t)r * n + k] * P[(size_t)k * n + c];
        Bm[i] = s;
      }
      __syncthreads();
      for (int i = t; i < n * n; i += kThreads) A[i] = Bm[i];
      __syncthreads();
    }
  };
  for (int k = 1; k < kmin; ++k) multiply();                 // A = P^kmin  (matrix_power(min(ksteps)))
  for (int k = kmin; k <= kmax; ++k) {
    const float corr = powf((float)k, 0.5f * space_dim);      // k ** (space_dim / 2)
    for (int v = t; v < n; v += kThreads) out[(size_t)(n0 + v) * K + (k - kmin)] = A[(size_t)v * n + v] * corr;
    if (k < kmax) multiply();
  }
}

}  // namespace

extern "C" {

int gps_rwse_lds_nodes(void) { return kLdsNodes; }

int gps_rwse(const int32_t* rowptr_src, const int32_t* dst_by_src, const int32_t* ptr, int64_t B, int64_t N,
             int kmin, int kmax, float space_dim, float* scratch, const int64_t* scratch_off, float* out,
             gps_stream_t stream) {
  GPS_REQUIRE(B >= 0 && N >= 0 && kmin >= 1 && kmax >= kmin, "gps_rwse: bad arguments (kmin=%d kmax=%d)", kmin, kmax);
  if (B == 0 || N == 0) return GPS_OK;
  GPS_REQUIRE(rowptr_src && ptr && out && scratch_off, "gps_rwse: null buffer");
  const size_t lds_bytes = 2 * (size_t)kLdsNodes * kLdsNodes * sizeof(float);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rwse),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  GPS_REQUIRE(attr == hipSuccess, "gps_rwse: cannot reserve %zu bytes of LDS", lds_bytes);
  k_rwse<<<(unsigned)B, kThreads, lds_bytes, gps::as_stream(stream)>>>(rowptr_src, dst_by_src, ptr, kmin, kmax,
                                                                       space_dim, scratch, scratch_off, out);
  return gps::launch_status("gps_rwse");
}

}  // extern "C"
