// Fused BatchNorm epilogues for the [N,d] / [E,d] activation streams of a GPS layer.
//
// Replaces the chains the reference builds out of separate torch modules around every BatchNorm1d
//   x = x_in + dropout(relu(bn_node_x(x)))        graphgps/layer/gatedgcn_layer.py:72-83
//   h = norm(h_in + dropout(branch))               graphgps/layer/gps_layer.py:191-194,212-217,225-229
//   relu -> dropout inside the FFN                 graphgps/layer/gps_layer.py:253-257
// (ATen: collect_statistics + transform + relu + dropout + add forward, reduce + elemt + masked_scale +
// threshold backward; the [E,384] reductions alone ran ~60 us each on MI355X) with:
//   stats   : per-column (mean_b, M2_b) of ~512 row blocks (block-local shifted sums) combined by a
//             two-pass stage 2 (global mean, then sum of M2_b + n_b (mean_b - mean)^2) -> batch mean /
//             biased var, running-stat update.  Deterministic, Welford-grade accuracy
//             (tests/test_hip_ops.py::test_bn_stats_large_mean_is_accurate).
//   apply   : y = res + drop(relu((z - mean) * rstd * gamma + beta))     (each stage optional)
//   bwd     : column sums of g and g*zhat (g = dL/d(bn output), ReLU and dropout masks RECOMPUTED from
//             z and the counter hash -> nothing but z is saved), then
//             g_z = gamma * rstd * (g - mean(g) - zhat * mean(g*zhat))
// HBM-bound row kernels, same lane-owns-4-channels mapping as gatedgcn.hip.  Training-mode batch
// statistics follow torch.nn.BatchNorm1d: biased variance for normalisation, unbiased for the running
// estimate, eps inside the sqrt.
#include <algorithm>

#include "gps_common.hpp"
#include "vec.hpp"

namespace {

constexpr int TARGET_BLOCKS = 512;   // stage-1 partial blocks: ~2 per CU so the reduction fills the chip
constexpr int FCOLS = 16, FCHUNKS = 16;  // stage-2 block = 16 columns x 16 partial-list chunks

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Same counter hash as seg_attention.hip (row id, element index) -> keep decision.
__device__ __forceinline__ uint32_t row_hash(uint32_t rowid, uint64_t seed) {
  return mix32(rowid ^ (uint32_t)seed) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ bool keep_elem(uint32_t rh, uint32_t col, float p_drop) {
  const uint32_t r = mix32(rh + col * 0x9E3779B9U);
  return (float)(r >> 8) * (1.0f / 16777216.0f) >= p_drop;
}

// ---- statistics ---------------------------------------------------------------------------------
// ws layout: [nblocks][2][d] = (mean_b, M2_b) per block and column.
template <int VEC>
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ z, int64_t R, int d,
                                                    int rpb, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [RS][2][d]
  const int L = d / VEC;            // lanes per row
  const int RS = 256 / L;           // row sub-groups in the block
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * VEC;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  const int64_t row1 = min(R, row0 + rpb);
  const bool active = rsub < RS;
  Vec<VEC> k = Vec<VEC>::zero(), s1 = Vec<VEC>::zero(), s2 = Vec<VEC>::zero();
  if (active) {
    // shift = the block's own first row: inside a ~15-row block the shifted sums have no
    // E[x^2]-E[x]^2 cancellation to speak of, so (mean_b, M2_b) are accurate to ~1 ulp
    k = Vec<VEC>::load(z + row0 * d + c);
    for (int64_t r = row0 + rsub; r < row1; r += RS) {
      const Vec<VEC> v = Vec<VEC>::load(z + r * d + c);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float t = v[j] - k[j];
        s1[j] += t;
        s2[j] += t * t;
      }
    }
    if (rsub > 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        lds[(rsub * 2 + 0) * d + c + j] = s1[j];
        lds[(rsub * 2 + 1) * d + c + j] = s2[j];
      }
    }
  }
  __syncthreads();
  if (active && rsub == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = s1[j], b = s2[j];
      for (int q = 1; q < RS; ++q) {   // fixed order
        a += lds[(q * 2 + 0) * d + c + j];
        b += lds[(q * 2 + 1) * d + c + j];
      }
      const float n = (float)(row1 - row0);
      float* o = ws + (int64_t)blockIdx.x * 2 * d;   // [nblocks][2][d] = (mean_b, M2_b)
      o[c + j] = k[j] + a / n;
      o[d + c + j] = fmaxf(b - a * a / n, 0.0f);
    }
  }
}

// Stage 2: block = FCOLS columns x FCHUNKS chunks of the partial list; every thread keeps its chunk
// of (mean_b, M2_b) in registers.  Pass 1: global mean = sum n_b mean_b / n.  Pass 2: M2 = sum
// [M2_b + n_b (mean_b - mean)^2] -- a sum of non-negative terms, no cancellation, no sequential
// Chan chain, loads issued once and pipelined.  Fixed summation tree -> deterministic.
constexpr int FPER = (TARGET_BLOCKS + FCHUNKS - 1) / FCHUNKS;   // partials per thread (32)

__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ ws, int nblocks, int d,
                                                     int rpb, float count, float eps, float momentum,
                                                     float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out,
                                                     float* __restrict__ running_mean,
                                                     float* __restrict__ running_var) {
  __shared__ float sh[FCHUNKS][FCOLS];
  __shared__ float sh_mean[FCOLS];
  const int col = threadIdx.x % FCOLS, chunk = threadIdx.x / FCOLS;
  const int c = blockIdx.x * FCOLS + col;
  const int per = (nblocks + FCHUNKS - 1) / FCHUNKS;   // <= FPER
  const int b0 = chunk * per;
  float mb[FPER], qb[FPER], nbv[FPER];
  float a = 0.f;
#pragma unroll
  for (int j = 0; j < FPER; ++j) {
    const int b = b0 + j;
    const bool ok = c < d && j < per && b < nblocks;
    mb[j] = ok ? ws[(int64_t)b * 2 * d + c] : 0.f;
    qb[j] = ok ? ws[(int64_t)b * 2 * d + d + c] : 0.f;
    // rows in block b: rpb, except the last block
    nbv[j] = ok ? fminf((float)rpb, count - (float)b * (float)rpb) : 0.f;
  }
#pragma unroll
  for (int j = 0; j < FPER; ++j) a += nbv[j] * mb[j];
  sh[chunk][col] = a;
  __syncthreads();
  if (chunk == 0) {
    for (int q = 1; q < FCHUNKS; ++q) a += sh[q][col];
    sh_mean[col] = a / count;
  }
  __syncthreads();
  const float mean = sh_mean[col];
  float m2 = 0.f;
#pragma unroll
  for (int j = 0; j < FPER; ++j) {
    const float dl = mb[j] - mean;
    m2 += qb[j] + nbv[j] * dl * dl;
  }
  __syncthreads();
  sh[chunk][col] = m2;
  __syncthreads();
  if (chunk == 0 && c < d) {
    for (int q = 1; q < FCHUNKS; ++q) m2 += sh[q][col];
    mean_out[c] = mean;
    rstd_out[c] = 1.0f / sqrtf(m2 / count + eps);
    if (running_mean) {
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (m2 / fmaxf(count - 1.0f, 1.0f));
    }
  }
}

// ---- forward apply --------------------------------------------------------------------------------
template <int VEC, bool RELU, bool DROP, bool RES>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ z,
                                                  const float* __restrict__ mean,
                                                  const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma,
                                                  const float* __restrict__ beta,
                                                  const float* __restrict__ res, int64_t R, int d,
                                                  float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ y) {
  seed = gps::salted_seed(seed, salt);
  const int L = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = t / L;
  if (row >= R) return;
  const int c = (int)(t - row * L) * VEC;
  const Vec<VEC> v = Vec<VEC>::load(z + row * d + c);
  const Vec<VEC> mu = Vec<VEC>::load(mean + c), rs = Vec<VEC>::load(rstd + c);
  const Vec<VEC> ga = Vec<VEC>::load(gamma + c), be = Vec<VEC>::load(beta + c);
  Vec<VEC> o;
  const uint32_t rh = DROP ? row_hash((uint32_t)row, seed) : 0u;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float u = (v[j] - mu[j]) * rs[j] * ga[j] + be[j];
    if (RELU) u = fmaxf(u, 0.0f);
    if (DROP) u = keep_elem(rh, (uint32_t)(c + j), p_drop) ? u * inv_keep : 0.0f;
    o[j] = u;
  }
  if (RES) {
    const Vec<VEC> rr = Vec<VEC>::load(res + row * d + c);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = rr[j] + o[j];
  }
  o.store(y + row * d + c);
}

// ---- backward -------------------------------------------------------------------------------------
// g (grad wrt the BN output) = [relu mask] * [dropout mask / (1-p)] * g_y, masks recomputed.
template <int VEC, bool RELU, bool DROP>
__device__ __forceinline__ void bn_out_grad(const Vec<VEC>& v, const Vec<VEC>& gy, const Vec<VEC>& mu,
                                            const Vec<VEC>& rs, const Vec<VEC>& ga, const Vec<VEC>& be,
                                            uint32_t rh, int c, float p_drop, float inv_keep,
                                            Vec<VEC>& g, Vec<VEC>& zh) {
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    zh[j] = (v[j] - mu[j]) * rs[j];
    float gg = gy[j];
    if (DROP) gg = keep_elem(rh, (uint32_t)(c + j), p_drop) ? gg * inv_keep : 0.0f;
    if (RELU) gg = (zh[j] * ga[j] + be[j]) > 0.0f ? gg : 0.0f;
    g[j] = gg;
  }
}

// ws layout: [nblocks][2][d] = (sum g, sum g*zhat)
template <int VEC, bool RELU, bool DROP>
__global__ __launch_bounds__(256) void k_bn_bwd_partial(
    const float* __restrict__ z, const float* __restrict__ g_y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    int64_t R, int d, int rpb, float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ ws) {
  seed = gps::salted_seed(seed, salt);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = d / VEC;
  const int RS = 256 / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * VEC;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  const int64_t row1 = min(R, row0 + rpb);
  const bool active = rsub < RS;
  Vec<VEC> sg = Vec<VEC>::zero(), sgz = Vec<VEC>::zero();
  if (active) {
    const Vec<VEC> mu = Vec<VEC>::load(mean + c), rs = Vec<VEC>::load(rstd + c);
    const Vec<VEC> ga = Vec<VEC>::load(gamma + c), be = Vec<VEC>::load(beta + c);
    const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int64_t r = row0 + rsub; r < row1; r += RS) {
      const Vec<VEC> v = Vec<VEC>::load(z + r * d + c);
      const Vec<VEC> gy = Vec<VEC>::load(g_y + r * d + c);
      Vec<VEC> g, zh;
      bn_out_grad<VEC, RELU, DROP>(v, gy, mu, rs, ga, be, DROP ? row_hash((uint32_t)r, seed) : 0u, c,
                                   p_drop, inv_keep, g, zh);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        sg[j] += g[j];
        sgz[j] += g[j] * zh[j];
      }
    }
    if (rsub > 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        lds[(rsub * 2 + 0) * d + c + j] = sg[j];
        lds[(rsub * 2 + 1) * d + c + j] = sgz[j];
      }
    }
  }
  __syncthreads();
  if (active && rsub == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = sg[j], b = sgz[j];
      for (int q = 1; q < RS; ++q) {
        a += lds[(q * 2 + 0) * d + c + j];
        b += lds[(q * 2 + 1) * d + c + j];
      }
      ws[(int64_t)blockIdx.x * 2 * d + c + j] = a;
      ws[(int64_t)blockIdx.x * 2 * d + d + c + j] = b;
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ ws, int nblocks,
                                                         int d, float* __restrict__ g_beta,
                                                         float* __restrict__ g_gamma) {
  __shared__ float sh[FCHUNKS][FCOLS][2];
  const int col = threadIdx.x % FCOLS, chunk = threadIdx.x / FCOLS;
  const int c = blockIdx.x * FCOLS + col;
  const int per = (nblocks + FCHUNKS - 1) / FCHUNKS;
  const int b0 = chunk * per, b1 = min(nblocks, b0 + per);
  float a = 0.f, b = 0.f;
  if (c < d) {
#pragma unroll 8
    for (int k = b0; k < b1; ++k) {
      a += ws[(int64_t)k * 2 * d + c];
      b += ws[(int64_t)k * 2 * d + d + c];
    }
  }
  sh[chunk][col][0] = a; sh[chunk][col][1] = b;
  __syncthreads();
  if (chunk == 0 && c < d) {
    for (int q = 1; q < FCHUNKS; ++q) { a += sh[q][col][0]; b += sh[q][col][1]; }
    g_beta[c] = a;
    g_gamma[c] = b;
  }
}

template <int VEC, bool RELU, bool DROP>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(
    const float* __restrict__ z, const float* __restrict__ g_y, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gz, int64_t R, int d, float p_drop,
    uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ g_z) {
  seed = gps::salted_seed(seed, salt);
  const int L = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = t / L;
  if (row >= R) return;
  const int c = (int)(t - row * L) * VEC;
  const Vec<VEC> v = Vec<VEC>::load(z + row * d + c);
  const Vec<VEC> gy = Vec<VEC>::load(g_y + row * d + c);
  const Vec<VEC> mu = Vec<VEC>::load(mean + c), rs = Vec<VEC>::load(rstd + c);
  const Vec<VEC> ga = Vec<VEC>::load(gamma + c), be = Vec<VEC>::load(beta + c);
  const Vec<VEC> s1 = Vec<VEC>::load(sum_g + c), s2 = Vec<VEC>::load(sum_gz + c);
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
  const float inv_n = 1.0f / (float)R;
  Vec<VEC> g, zh, o;
  bn_out_grad<VEC, RELU, DROP>(v, gy, mu, rs, ga, be, DROP ? row_hash((uint32_t)row, seed) : 0u, c,
                               p_drop, inv_keep, g, zh);
#pragma unroll
  for (int j = 0; j < VEC; ++j)
    o[j] = ga[j] * rs[j] * (g[j] - s1[j] * inv_n - zh[j] * s2[j] * inv_n);
  o.store(g_z + row * d + c);
}

// ---- elementwise helpers ---------------------------------------------------------------------------
// out = a + drop(b)       (gps_layer.py:212-213, 225 + ff_dropout2)
// out = drop(relu(b))     (a == nullptr, RELU)   (gps_layer.py:256)
template <int VEC, bool RELU, bool DROP, bool ADD>
__global__ __launch_bounds__(256) void k_act_drop_add(const float* __restrict__ a,
                                                      const float* __restrict__ b, int64_t R, int d,
                                                      float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
                                                      float* __restrict__ out) {
  seed = gps::salted_seed(seed, salt);
  const int L = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = t / L;
  if (row >= R) return;
  const int c = (int)(t - row * L) * VEC;
  const Vec<VEC> v = Vec<VEC>::load(b + row * d + c);
  const uint32_t rh = DROP ? row_hash((uint32_t)row, seed) : 0u;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
  Vec<VEC> o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float u = v[j];
    if (RELU) u = fmaxf(u, 0.0f);
    if (DROP) u = keep_elem(rh, (uint32_t)(c + j), p_drop) ? u * inv_keep : 0.0f;
    o[j] = u;
  }
  if (ADD) {
    const Vec<VEC> aa = Vec<VEC>::load(a + row * d + c);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = aa[j] + o[j];
  }
  o.store(out + row * d + c);
}

// g_b = [b > 0] * [keep / (1-p)] * g      (pre = the forward's input b, only read when RELU)
template <int VEC, bool RELU, bool DROP>
__global__ __launch_bounds__(256) void k_act_drop_bwd(const float* __restrict__ g,
                                                      const float* __restrict__ pre, int64_t R, int d,
                                                      float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
                                                      float* __restrict__ g_b) {
  seed = gps::salted_seed(seed, salt);
  const int L = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t row = t / L;
  if (row >= R) return;
  const int c = (int)(t - row * L) * VEC;
  const Vec<VEC> gg = Vec<VEC>::load(g + row * d + c);
  const uint32_t rh = DROP ? row_hash((uint32_t)row, seed) : 0u;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;
  Vec<VEC> o;
  Vec<VEC> pv = Vec<VEC>::zero();
  if (RELU) pv = Vec<VEC>::load(pre + row * d + c);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float u = gg[j];
    if (DROP) u = keep_elem(rh, (uint32_t)(c + j), p_drop) ? u * inv_keep : 0.0f;
    if (RELU) u = pv[j] > 0.0f ? u : 0.0f;
    o[j] = u;
  }
  o.store(g_b + row * d + c);
}

// ---- plain column sum (bias gradients of the dense projections) ------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int64_t R, int d,
                                                        int rpb, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [RS][Lb*VEC]
  const int L = d / VEC;                       // lanes per full row
  const int Lb = L < 256 ? L : 256;            // lanes of this block's column slab (blockIdx.y)
  const int RS = 256 / Lb;
  const int rsub = threadIdx.x / Lb;
  const int lane = threadIdx.x - rsub * Lb;
  const int c = (blockIdx.y * Lb + lane) * VEC;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  const int64_t row1 = min(R, row0 + rpb);
  const bool active = rsub < RS && c < d;
  Vec<VEC> acc = Vec<VEC>::zero();
  if (active) {
    for (int64_t r = row0 + rsub; r < row1; r += RS) {
      const Vec<VEC> v = Vec<VEC>::load(x + r * d + c);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += v[j];
    }
    if (rsub > 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) lds[(rsub * Lb + lane) * VEC + j] = acc[j];
    }
  }
  __syncthreads();
  if (active && rsub == 0) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = acc[j];
      for (int q = 1; q < RS; ++q) a += lds[(q * Lb + lane) * VEC + j];
      ws[(int64_t)blockIdx.x * d + c + j] = a;
    }
  }
}

__global__ __launch_bounds__(256) void k_colsum_finalize(const float* __restrict__ ws, int nblocks,
                                                         int d, float* __restrict__ out) {
  __shared__ float sh[FCHUNKS][FCOLS];
  const int col = threadIdx.x % FCOLS, chunk = threadIdx.x / FCOLS;
  const int c = blockIdx.x * FCOLS + col;
  const int per = (nblocks + FCHUNKS - 1) / FCHUNKS;
  const int b0 = chunk * per, b1 = min(nblocks, b0 + per);
  float a = 0.f;
  if (c < d) {
#pragma unroll 8
    for (int k = b0; k < b1; ++k) a += ws[(int64_t)k * d + c];
  }
  sh[chunk][col] = a;
  __syncthreads();
  if (chunk == 0 && c < d) {
    for (int q = 1; q < FCHUNKS; ++q) a += sh[q][col];
    out[c] = a;
  }
}

inline bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
// rows per stage-1 block (a multiple of the block's row sub-groups is not required) and block count
inline int rows_per_block(int64_t R) { return (int)std::max<int64_t>(8, (R + TARGET_BLOCKS - 1) / TARGET_BLOCKS); }
inline int nblocks_for(int64_t R) { const int rpb = rows_per_block(R); return (int)((R + rpb - 1) / rpb); }

}  // namespace

#define GPS_BOOL3(A, B, C, ...)                                         \
  do {                                                                  \
    if (A) { constexpr bool kA = true;                                  \
      if (B) { constexpr bool kB = true;                                \
        if (C) { constexpr bool kC = true; __VA_ARGS__; } else { constexpr bool kC = false; __VA_ARGS__; } \
      } else { constexpr bool kB = false;                               \
        if (C) { constexpr bool kC = true; __VA_ARGS__; } else { constexpr bool kC = false; __VA_ARGS__; } \
      }                                                                 \
    } else { constexpr bool kA = false;                                 \
      if (B) { constexpr bool kB = true;                                \
        if (C) { constexpr bool kC = true; __VA_ARGS__; } else { constexpr bool kC = false; __VA_ARGS__; } \
      } else { constexpr bool kB = false;                               \
        if (C) { constexpr bool kC = true; __VA_ARGS__; } else { constexpr bool kC = false; __VA_ARGS__; } \
      }                                                                 \
    }                                                                   \
  } while (0)

extern "C" {

size_t gps_bn_workspace_floats(int64_t R, int d) { return (size_t)nblocks_for(R) * 3 * d; }

int gps_bn_stats(const float* z, int64_t R, int d, float eps, float momentum, float* mean, float* rstd,
                 float* running_mean, float* running_var, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(R >= 2 && d > 0 && d <= 4096, "gps_bn_stats: need R >= 2 rows (got %lld) and 0 < d <= 4096",
              (long long)R);
  GPS_REQUIRE(z && mean && rstd && ws && ((running_mean == nullptr) == (running_var == nullptr)),
              "gps_bn_stats: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const int nb = nblocks_for(R);
  GPS_DISPATCH_VEC(d, al(z, 16) && d / 4 <= 256, al(z, 8) && d / 2 <= 256, {
    GPS_REQUIRE(d / VEC <= 256, "gps_bn_stats: d=%d too wide for this vector width", d);
    const int RS = 256 / (d / VEC);
    k_bn_partial<VEC><<<nb, 256, sizeof(float) * 2 * RS * d, s>>>(z, R, d, rows_per_block(R), ws);
  });
  k_bn_finalize<<<gps::grid_for(d, FCOLS), 256, 0, s>>>(ws, nb, d, rows_per_block(R), (float)R, eps, momentum,
                                                        mean, rstd, running_mean, running_var);
  return gps::launch_status("gps_bn_stats");
}

int gps_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, const float* res, int64_t R, int d, int relu, float p_drop,
                 uint64_t seed, float* y, gps_stream_t stream) {
  GPS_REQUIRE(R >= 0 && d > 0 && p_drop >= 0.f && p_drop < 1.f, "gps_bn_apply: bad arguments");
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(z && mean && rstd && gamma && beta && y, "gps_bn_apply: null buffer");
  GPS_REQUIRE(R < INT32_MAX, "gps_bn_apply: row id exceeds 32 bits");
  hipStream_t s = gps::as_stream(stream);
  const bool a16 = al(z, 16) && al(y, 16) && al(res, 16) && al(mean, 16) && al(rstd, 16) && al(gamma, 16) && al(beta, 16);
  const bool a8 = al(z, 8) && al(y, 8) && al(res, 8) && al(mean, 8) && al(rstd, 8) && al(gamma, 8) && al(beta, 8);
  GPS_DISPATCH_VEC(d, a16, a8, {
    const unsigned grid = gps::grid_for(R * (int64_t)(d / VEC), 256);
    GPS_BOOL3(relu != 0, p_drop > 0.f, res != nullptr,
              (k_bn_apply<VEC, kA, kB, kC><<<grid, 256, 0, s>>>(z, mean, rstd, gamma, beta, res, R, d,
                                                                p_drop, seed, gps::dropout_salt(), y)));
  });
  return gps::launch_status("gps_bn_apply");
}

int gps_bn_bwd(const float* z, const float* g_y, const float* mean, const float* rstd,
               const float* gamma, const float* beta, int64_t R, int d, int relu, float p_drop,
               uint64_t seed, float* g_z, float* g_gamma, float* g_beta, float* ws,
               gps_stream_t stream) {
  GPS_REQUIRE(R >= 1 && d > 0 && d <= 4096 && p_drop >= 0.f && p_drop < 1.f, "gps_bn_bwd: bad arguments");
  GPS_REQUIRE(z && g_y && mean && rstd && gamma && beta && g_z && g_gamma && g_beta && ws,
              "gps_bn_bwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const int nb = nblocks_for(R);
  const bool a16 = al(z, 16) && al(g_y, 16) && al(g_z, 16) && al(mean, 16) && al(rstd, 16) && al(gamma, 16) &&
                   al(beta, 16) && al(g_gamma, 16) && al(g_beta, 16);
  const bool a8 = al(z, 8) && al(g_y, 8) && al(g_z, 8) && al(mean, 8) && al(rstd, 8) && al(gamma, 8) &&
                  al(beta, 8) && al(g_gamma, 8) && al(g_beta, 8);
  GPS_DISPATCH_VEC(d, a16 && d / 4 <= 256, a8 && d / 2 <= 256, {
    GPS_REQUIRE(d / VEC <= 256, "gps_bn_bwd: d=%d too wide for this vector width", d);
    const int RS = 256 / (d / VEC);
    const unsigned grid = gps::grid_for(R * (int64_t)(d / VEC), 256);
    GPS_BOOL3(relu != 0, p_drop > 0.f, false, {
      (void)kC;
      k_bn_bwd_partial<VEC, kA, kB><<<nb, 256, sizeof(float) * 2 * RS * d, s>>>(
          z, g_y, mean, rstd, gamma, beta, R, d, rows_per_block(R), p_drop, seed, gps::dropout_salt(), ws);
      k_bn_bwd_finalize<<<gps::grid_for(d, FCOLS), 256, 0, s>>>(ws, nb, d, g_beta, g_gamma);
      k_bn_bwd_apply<VEC, kA, kB><<<grid, 256, 0, s>>>(z, g_y, mean, rstd, gamma, beta, g_beta, g_gamma,
                                                       R, d, p_drop, seed, gps::dropout_salt(), g_z);
    });
  });
  return gps::launch_status("gps_bn_bwd");
}

int gps_colsum(const float* x, int64_t R, int d, float* out, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(R >= 1 && d > 0 && x && out && ws, "gps_colsum: bad arguments");
  hipStream_t s = gps::as_stream(stream);
  const int nb = nblocks_for(R);
  GPS_DISPATCH_VEC(d, al(x, 16), al(x, 8), {
    const int L = d / VEC, Lb = L < 256 ? L : 256;
    const dim3 grid(nb, (L + Lb - 1) / Lb);
    k_colsum_partial<VEC><<<grid, 256, sizeof(float) * (256 / Lb) * Lb * VEC, s>>>(x, R, d, rows_per_block(R), ws);
  });
  k_colsum_finalize<<<gps::grid_for(d, FCOLS), 256, 0, s>>>(ws, nb, d, out);
  return gps::launch_status("gps_colsum");
}

int gps_act_drop_add(const float* a, const float* b, int64_t R, int d, int relu, float p_drop,
                     uint64_t seed, float* out, gps_stream_t stream) {
  GPS_REQUIRE(R >= 0 && d > 0 && p_drop >= 0.f && p_drop < 1.f, "gps_act_drop_add: bad arguments");
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(b && out && R < INT32_MAX, "gps_act_drop_add: null buffer");
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, al(a, 16) && al(b, 16) && al(out, 16), al(a, 8) && al(b, 8) && al(out, 8), {
    const unsigned grid = gps::grid_for(R * (int64_t)(d / VEC), 256);
    GPS_BOOL3(relu != 0, p_drop > 0.f, a != nullptr,
              (k_act_drop_add<VEC, kA, kB, kC><<<grid, 256, 0, s>>>(a, b, R, d, p_drop, seed, gps::dropout_salt(), out)));
  });
  return gps::launch_status("gps_act_drop_add");
}

int gps_act_drop_bwd(const float* g, const float* pre, int64_t R, int d, int relu, float p_drop,
                     uint64_t seed, float* g_b, gps_stream_t stream) {
  GPS_REQUIRE(R >= 0 && d > 0 && p_drop >= 0.f && p_drop < 1.f, "gps_act_drop_bwd: bad arguments");
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(g && g_b && (!relu || pre) && R < INT32_MAX, "gps_act_drop_bwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, al(g, 16) && al(pre, 16) && al(g_b, 16), al(g, 8) && al(pre, 8) && al(g_b, 8), {
    const unsigned grid = gps::grid_for(R * (int64_t)(d / VEC), 256);
    GPS_BOOL3(relu != 0, p_drop > 0.f, false, {
      (void)kC;
      k_act_drop_bwd<VEC, kA, kB><<<grid, 256, 0, s>>>(g, pre, R, d, p_drop, seed, gps::dropout_salt(), g_b);
    });
  });
  return gps::launch_status("gps_act_drop_bwd");
}

}  // extern "C"
