// Task-list BatchNorm / residual / dropout kernels for one CustomGatedGCN+Transformer GPS block.
//
// bn_fused.hip gives every BatchNorm1d of the reference block its own three launches (partial
// statistics, finalize, apply) plus one launch per residual+dropout (gatedgcn_layer.py:72-83,
// gps_layer.py:191-194,212-217,225-229): 18 forward + 19 backward launches per layer of 5-10 us
// each -- latency-bound kernels that the PCQM4M-size step cannot hide.  Here the same arithmetic
// (same formulas, same counter-hash dropout, same two-pass statistics) is issued as LISTS of up to
// four independent row-stream tasks per launch, and neighbouring stages are merged:
//
//   forward   stats{x~, e^}                                           partial + finalize
//             {x1 = x + drop(relu(BN_x(x~))) [+ stats of x1],
//              e1 = e + drop(relu(BN_e(e^))),
//              za = x + drop(attn_out)       [+ stats of za]}         one launch
//             finalize{x1, za}
//             h = BN_l(x1) + BN_a(za)                                 one launch (dual apply)
//             z2 = h + drop(ffn_out) [+ stats] ; finalize ; BN_2      three launches
//   backward  BN_2: partial, finalize, apply -> (g_z2, g_f2 = dropmask(g_z2))
//             {BN_l, BN_a} share dL/dh: partial (reads dL/dh once), finalize,
//                 apply -> (g_x1, g_x1 + g_za, g_ao = dropmask(g_za))
//             {BN_x, BN_e}: partial, finalize, apply as two-task lists
//
// 10 + 9 launches instead of 18 + 19, ~25 % less HBM traffic on these stages.
// Row kernels keep the lane-owns-4-channels mapping (d % 4 == 0, d <= 1024).
#include <algorithm>

#include "gps_common.hpp"
#include "vec.hpp"

namespace {

constexpr int TARGET_BLOCKS = 512;       // stage-1 partial blocks per task (~2 per CU)
constexpr int FCOLS = 16, FCHUNKS = 16;  // stage-2 block = 16 columns x 16 partial-list chunks
constexpr int FPER = (TARGET_BLOCKS + FCHUNKS - 1) / FCHUNKS;
constexpr int kMaxTasks = 4;
typedef Vec<4> V4;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// identical to bn_fused.hip / seg_attention.hip: (row id, element index) -> keep decision
__device__ __forceinline__ uint32_t row_hash(uint32_t rowid, uint64_t seed) {
  return mix32(rowid ^ (uint32_t)seed) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ bool keep_elem(uint32_t rh, uint32_t col, float p_drop) {
  const uint32_t r = mix32(rh + col * 0x9E3779B9U);
  return (float)(r >> 8) * (1.0f / 16777216.0f) >= p_drop;
}

struct Bn {
  const float *mean, *rstd, *gamma, *beta;
};
struct Col {
  V4 mu, rs, ga, be;
};
__device__ __forceinline__ Col load_col(const Bn& b, int c) {
  Col o;
  o.mu = V4::load(b.mean + c); o.rs = V4::load(b.rstd + c);
  o.ga = V4::load(b.gamma + c); o.be = V4::load(b.beta + c);
  return o;
}

// ------------------------------------------------------------------------------------------------
// forward: produce rows (optionally store them, optionally accumulate their column statistics)
// ------------------------------------------------------------------------------------------------
enum { K_LOAD = 0, K_ADD_DROP = 1, K_BN_ACT = 2, K_BN_DUAL = 3 };

struct FwdTask {
  const float *a, *b, *res;
  Bn bn1, bn2;
  float* out;     // produced rows (nullptr: K_LOAD)
  float* ws;      // [nblk][2][d] (mean_b, M2_b) of the produced rows, or nullptr
  int64_t R;
  uint64_t seed;
  float p;
  int kind, relu, rpb, nblk, block_begin;
};
struct FwdGroup {
  FwdTask t[kMaxTasks];
  const uint64_t* salt;
  int n, d;
};

template <int KIND, bool RELU, bool DROP>
__device__ __forceinline__ V4 eval_row(const FwdTask& T, int64_t r, int c, int d, const Col& c1,
                                       const Col& c2, uint64_t seed, float inv_keep) {
  V4 o;
  if (KIND == K_LOAD) {
    o = V4::load(T.a + r * d + c);
  } else if (KIND == K_ADD_DROP) {             // a + drop(b)
    const V4 a = V4::load(T.a + r * d + c), b = V4::load(T.b + r * d + c);
    const uint32_t rh = DROP ? row_hash((uint32_t)r, seed) : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = b[j];
      if (DROP) u = keep_elem(rh, (uint32_t)(c + j), T.p) ? u * inv_keep : 0.0f;
      o[j] = a[j] + u;
    }
  } else if (KIND == K_BN_ACT) {               // res + drop(relu(BN(a)))
    const V4 v = V4::load(T.a + r * d + c);
    const uint32_t rh = DROP ? row_hash((uint32_t)r, seed) : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = (v[j] - c1.mu[j]) * c1.rs[j] * c1.ga[j] + c1.be[j];
      if (RELU) u = fmaxf(u, 0.0f);
      if (DROP) u = keep_elem(rh, (uint32_t)(c + j), T.p) ? u * inv_keep : 0.0f;
      o[j] = u;
    }
    if (T.res) {
      const V4 rr = V4::load(T.res + r * d + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = rr[j] + o[j];
    }
  } else {                                     // BN1(a) + BN2(b)
    const V4 v1 = V4::load(T.a + r * d + c), v2 = V4::load(T.b + r * d + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u1 = (v1[j] - c1.mu[j]) * c1.rs[j] * c1.ga[j] + c1.be[j];
      const float u2 = (v2[j] - c2.mu[j]) * c2.rs[j] * c2.ga[j] + c2.be[j];
      o[j] = u1 + u2;
    }
  }
  return o;
}

template <int KIND, bool RELU, bool DROP>
__device__ __forceinline__ void run_fwd(const FwdTask& T, int d, int local_block, uint64_t seed, float* lds) {
  const int L = d >> 2;
  const int RS = 256 / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * 4;
  const int64_t row0 = (int64_t)local_block * T.rpb;
  const int64_t row1 = min(T.R, row0 + T.rpb);
  const bool active = rsub < RS;
  const bool stats = T.ws != nullptr;
  const float inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
  V4 k = V4::zero(), s1 = V4::zero(), s2 = V4::zero();
  if (active) {
    Col c1, c2;
    if (KIND == K_BN_ACT || KIND == K_BN_DUAL) c1 = load_col(T.bn1, c);
    if (KIND == K_BN_DUAL) c2 = load_col(T.bn2, c);
    // shift = the block's own first produced row (see bn_fused.hip: block-local shifted sums)
    if (stats) k = eval_row<KIND, RELU, DROP>(T, row0, c, d, c1, c2, seed, inv_keep);
    for (int64_t r = row0 + rsub; r < row1; r += RS) {
      const V4 v = eval_row<KIND, RELU, DROP>(T, r, c, d, c1, c2, seed, inv_keep);
      if (T.out) v.store(T.out + r * d + c);
      if (stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float t = v[j] - k[j];
          s1[j] += t;
          s2[j] += t * t;
        }
      }
    }
    if (stats && rsub > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lds[(rsub * 2 + 0) * d + c + j] = s1[j];
        lds[(rsub * 2 + 1) * d + c + j] = s2[j];
      }
    }
  }
  if (!stats) return;            // block-uniform
  __syncthreads();
  if (active && rsub == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = s1[j], b = s2[j];
      for (int q = 1; q < RS; ++q) {   // fixed order
        a += lds[(q * 2 + 0) * d + c + j];
        b += lds[(q * 2 + 1) * d + c + j];
      }
      const float n = (float)(row1 - row0);
      float* o = T.ws + (int64_t)local_block * 2 * d;
      o[c + j] = k[j] + a / n;
      o[d + c + j] = fmaxf(b - a * a / n, 0.0f);
    }
  }
}

__global__ __launch_bounds__(256) void k_rows_fwd(const FwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [RS][2][d]
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const FwdTask& T = G.t[ti];
  const int lb = blockIdx.x - T.block_begin;
  if (lb >= T.nblk) return;
  const uint64_t seed = gps::salted_seed(T.seed, G.salt);
  const bool drop = T.p > 0.0f;
  switch (T.kind) {
    case K_LOAD: run_fwd<K_LOAD, false, false>(T, G.d, lb, seed, lds); break;
    case K_ADD_DROP:
      if (drop) run_fwd<K_ADD_DROP, false, true>(T, G.d, lb, seed, lds);
      else run_fwd<K_ADD_DROP, false, false>(T, G.d, lb, seed, lds);
      break;
    case K_BN_ACT:
      if (T.relu) {
        if (drop) run_fwd<K_BN_ACT, true, true>(T, G.d, lb, seed, lds);
        else run_fwd<K_BN_ACT, true, false>(T, G.d, lb, seed, lds);
      } else {
        if (drop) run_fwd<K_BN_ACT, false, true>(T, G.d, lb, seed, lds);
        else run_fwd<K_BN_ACT, false, false>(T, G.d, lb, seed, lds);
      }
      break;
    default: run_fwd<K_BN_DUAL, false, false>(T, G.d, lb, seed, lds); break;
  }
}

// Stage 2 of the statistics, for a list of tasks: identical arithmetic to bn_fused.hip:k_bn_finalize
// (pass 1 global mean, pass 2 sum of M2_b + n_b (mean_b - mean)^2; fixed tree).
struct FinTask {
  const float* ws;
  float *mean, *rstd, *running_mean, *running_var;
  float count, eps, momentum;
  int nblk, rpb, block_begin;
};
struct FinGroup {
  FinTask t[kMaxTasks];
  int n, d;
};

__global__ __launch_bounds__(256) void k_stats_finalize(const FinGroup G) {
  __shared__ float sh[FCHUNKS][FCOLS];
  __shared__ float sh_mean[FCOLS];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const FinTask& T = G.t[ti];
  const int d = G.d;
  const int col = threadIdx.x % FCOLS, chunk = threadIdx.x / FCOLS;
  const int c = (blockIdx.x - T.block_begin) * FCOLS + col;
  const int per = (T.nblk + FCHUNKS - 1) / FCHUNKS;   // <= FPER
  const int b0 = chunk * per;
  float mb[FPER], qb[FPER], nbv[FPER];
  float a = 0.f;
  // every partial is requested before the first one is used: the loads are unconditional from a clamped address and
  // the out-of-range ones are zeroed by a select afterwards (a load under a condition becomes its own exec-masked
  // block and the 64 loads of a thread stop being one burst: this launch is pure latency, 10 -> ~4 us)
  const int cc = min(c, d - 1);
#pragma unroll
  for (int j = 0; j < FPER; ++j) {
    const int bb = min(b0 + j, T.nblk - 1);
    mb[j] = T.ws[(int64_t)bb * 2 * d + cc];
    qb[j] = T.ws[(int64_t)bb * 2 * d + d + cc];
  }
#pragma unroll
  for (int j = 0; j < FPER; ++j) {
    const int b = b0 + j;
    const bool ok = c < d && j < per && b < T.nblk;
    mb[j] = ok ? mb[j] : 0.f;
    qb[j] = ok ? qb[j] : 0.f;
    nbv[j] = ok ? fminf((float)T.rpb, T.count - (float)b * (float)T.rpb) : 0.f;
  }
#pragma unroll
  for (int j = 0; j < FPER; ++j) a += nbv[j] * mb[j];
  sh[chunk][col] = a;
  __syncthreads();
  if (chunk == 0) {
    for (int q = 1; q < FCHUNKS; ++q) a += sh[q][col];
    sh_mean[col] = a / T.count;
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
    T.mean[c] = mean;
    T.rstd[c] = 1.0f / sqrtf(m2 / T.count + T.eps);
    if (T.running_mean) {
      T.running_mean[c] = (1.0f - T.momentum) * T.running_mean[c] + T.momentum * mean;
      T.running_var[c] =
          (1.0f - T.momentum) * T.running_var[c] + T.momentum * (m2 / fmaxf(T.count - 1.0f, 1.0f));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct BwdTask {
  const float *z, *g_y;
  Bn bn;                 // primary BN; relu / (p, seed) are the masks applied to ITS output
  const float* z2;       // dual: a second BN (no masks) fed by the same g_y, or nullptr
  Bn bn2;
  float* ws;             // [nblk][3][d]: sum g, sum g*zhat, sum g*zhat2
  float *g_beta, *g_gamma, *g_beta2, *g_gamma2;
  float* g_z;            // primary input gradient
  float* g_sum;          // dual: g_z + g_z2
  float* g_drop;         // dropmask(seed2, p2) of g_z (dual: of g_z2), or nullptr
  float p1x;             // dual only: dropout (p1x, seed1x) applied to the STORED g_z (g_sum stays unmasked)
  uint64_t seed1x;
  int64_t R;
  uint64_t seed, seed2;
  float p, p2;
  int relu, rpb, nblk, block_begin, fin_begin;
  int64_t thread_begin;  // apply kernel: first flat thread of this task
};
struct BwdGroup {
  BwdTask t[kMaxTasks];
  const uint64_t* salt;
  int n, d;
};

// g (grad wrt the BN output) = [relu mask] * [dropout mask / (1-p)] * g_y, masks recomputed from z
template <bool RELU, bool DROP>
__device__ __forceinline__ void out_grad(const V4& v, const V4& gy, const Col& cc, uint32_t rh, int c,
                                         float p, float inv_keep, V4& g, V4& zh) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    zh[j] = (v[j] - cc.mu[j]) * cc.rs[j];
    float gg = gy[j];
    if (DROP) gg = keep_elem(rh, (uint32_t)(c + j), p) ? gg * inv_keep : 0.0f;
    if (RELU) gg = (zh[j] * cc.ga[j] + cc.be[j]) > 0.0f ? gg : 0.0f;
    g[j] = gg;
  }
}

template <bool RELU, bool DROP, bool DUAL>
__device__ __forceinline__ void run_bwd_partial(const BwdTask& T, int d, int lb, uint64_t seed, float* lds) {
  const int L = d >> 2;
  const int RS = 256 / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * 4;
  const int64_t row0 = (int64_t)lb * T.rpb;
  const int64_t row1 = min(T.R, row0 + T.rpb);
  const bool active = rsub < RS;
  V4 sg = V4::zero(), sgz = V4::zero(), sgz2 = V4::zero();
  if (active) {
    const Col c1 = load_col(T.bn, c);
    Col c2;
    if (DUAL) c2 = load_col(T.bn2, c);
    const float inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
    for (int64_t r = row0 + rsub; r < row1; r += RS) {
      const V4 v = V4::load(T.z + r * d + c);
      const V4 gy = V4::load(T.g_y + r * d + c);
      V4 g, zh;
      out_grad<RELU, DROP>(v, gy, c1, DROP ? row_hash((uint32_t)r, seed) : 0u, c, T.p, inv_keep, g, zh);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sg[j] += g[j];
        sgz[j] += g[j] * zh[j];
      }
      if (DUAL) {
        const V4 v2 = V4::load(T.z2 + r * d + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) sgz2[j] += g[j] * ((v2[j] - c2.mu[j]) * c2.rs[j]);
      }
    }
    if (rsub > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lds[(rsub * 3 + 0) * d + c + j] = sg[j];
        lds[(rsub * 3 + 1) * d + c + j] = sgz[j];
        if (DUAL) lds[(rsub * 3 + 2) * d + c + j] = sgz2[j];
      }
    }
  }
  __syncthreads();
  if (active && rsub == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = sg[j], b = sgz[j], e = sgz2[j];
      for (int q = 1; q < RS; ++q) {
        a += lds[(q * 3 + 0) * d + c + j];
        b += lds[(q * 3 + 1) * d + c + j];
        if (DUAL) e += lds[(q * 3 + 2) * d + c + j];
      }
      float* o = T.ws + (int64_t)lb * 3 * d;
      o[c + j] = a;
      o[d + c + j] = b;
      if (DUAL) o[2 * d + c + j] = e;
    }
  }
}

__global__ __launch_bounds__(256) void k_bwd_partial(const BwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [RS][3][d]
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const BwdTask& T = G.t[ti];
  const int lb = blockIdx.x - T.block_begin;
  if (lb >= T.nblk) return;
  const uint64_t seed = gps::salted_seed(T.seed, G.salt);
  const bool drop = T.p > 0.0f;
  if (T.z2) { run_bwd_partial<false, false, true>(T, G.d, lb, seed, lds); return; }
  if (T.relu) {
    if (drop) run_bwd_partial<true, true, false>(T, G.d, lb, seed, lds);
    else run_bwd_partial<true, false, false>(T, G.d, lb, seed, lds);
  } else {
    if (drop) run_bwd_partial<false, true, false>(T, G.d, lb, seed, lds);
    else run_bwd_partial<false, false, false>(T, G.d, lb, seed, lds);
  }
}

__global__ __launch_bounds__(256) void k_bwd_finalize(const BwdGroup G) {
  __shared__ float sh[FCHUNKS][FCOLS][3];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].fin_begin) ti = i;
  const BwdTask& T = G.t[ti];
  const int d = G.d;
  const int col = threadIdx.x % FCOLS, chunk = threadIdx.x / FCOLS;
  const int c = (blockIdx.x - T.fin_begin) * FCOLS + col;
  const int per = (T.nblk + FCHUNKS - 1) / FCHUNKS;
  const int b0 = chunk * per, b1 = min(T.nblk, b0 + per);
  const bool dual = T.z2 != nullptr;
  float a = 0.f, b = 0.f, e = 0.f;
  {
    // one burst of loads (clamped addresses, masked afterwards), summed in ascending block order
    float va[FPER], vb[FPER], vc[FPER];
    const int cc = min(c, d - 1);
    const int64_t third = dual ? 2 * d : d;        // non-dual: re-read the second column block (unused)
#pragma unroll
    for (int j = 0; j < FPER; ++j) {
      const int kk = min(b0 + j, T.nblk - 1);
      va[j] = T.ws[(int64_t)kk * 3 * d + cc];
      vb[j] = T.ws[(int64_t)kk * 3 * d + d + cc];
      vc[j] = T.ws[(int64_t)kk * 3 * d + third + cc];
    }
#pragma unroll
    for (int j = 0; j < FPER; ++j) {
      const bool ok = c < d && b0 + j < b1;
      a += ok ? va[j] : 0.f;
      b += ok ? vb[j] : 0.f;
      e += (ok && dual) ? vc[j] : 0.f;
    }
  }
  sh[chunk][col][0] = a; sh[chunk][col][1] = b; sh[chunk][col][2] = e;
  __syncthreads();
  if (chunk == 0 && c < d) {
    for (int q = 1; q < FCHUNKS; ++q) { a += sh[q][col][0]; b += sh[q][col][1]; e += sh[q][col][2]; }
    T.g_beta[c] = a;
    T.g_gamma[c] = b;
    if (dual) { T.g_beta2[c] = a; T.g_gamma2[c] = e; }
  }
}

template <bool RELU, bool DROP, bool DUAL>
__device__ __forceinline__ void run_bwd_apply(const BwdTask& T, int d, int64_t row, int c, uint64_t seed,
                                              uint64_t seed2, uint64_t seed1x = 0) {
  const V4 v = V4::load(T.z + row * d + c);
  const V4 gy = V4::load(T.g_y + row * d + c);
  const Col c1 = load_col(T.bn, c);
  const V4 s1 = V4::load(T.g_beta + c), s2 = V4::load(T.g_gamma + c);
  const float inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
  const float inv_n = 1.0f / (float)T.R;
  V4 g, zh, o;
  out_grad<RELU, DROP>(v, gy, c1, DROP ? row_hash((uint32_t)row, seed) : 0u, c, T.p, inv_keep, g, zh);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = c1.ga[j] * c1.rs[j] * (g[j] - s1[j] * inv_n - zh[j] * s2[j] * inv_n);
  if (DUAL && T.p1x > 0.0f) {
    const uint32_t rh1 = row_hash((uint32_t)row, seed1x);
    const float ik1 = 1.0f / (1.0f - T.p1x);
    V4 om;
#pragma unroll
    for (int j = 0; j < 4; ++j) om[j] = keep_elem(rh1, (uint32_t)(c + j), T.p1x) ? o[j] * ik1 : 0.0f;
    om.store(T.g_z + row * d + c);
  } else {
    o.store(T.g_z + row * d + c);
  }
  V4 last = o;
  if (DUAL) {
    const V4 v2 = V4::load(T.z2 + row * d + c);
    const Col c2 = load_col(T.bn2, c);
    const V4 s3 = V4::load(T.g_gamma2 + c);
    V4 o2, sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float zh2 = (v2[j] - c2.mu[j]) * c2.rs[j];
      o2[j] = c2.ga[j] * c2.rs[j] * (g[j] - s1[j] * inv_n - zh2 * s3[j] * inv_n);
      sum[j] = o[j] + o2[j];
    }
    if (T.g_sum) sum.store(T.g_sum + row * d + c);
    last = o2;
  }
  if (T.g_drop) {
    V4 q = last;
    if (T.p2 > 0.0f) {
      const uint32_t rh2 = row_hash((uint32_t)row, seed2);
      const float ik2 = 1.0f / (1.0f - T.p2);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = keep_elem(rh2, (uint32_t)(c + j), T.p2) ? last[j] * ik2 : 0.0f;
    }
    q.store(T.g_drop + row * d + c);
  }
}

__global__ __launch_bounds__(256) void k_bwd_apply(const BwdGroup G) {
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && gt >= G.t[i].thread_begin) ti = i;
  const BwdTask& T = G.t[ti];
  const int d = G.d, L = d >> 2;
  const int64_t t = gt - T.thread_begin;
  const int64_t row = t / L;
  if (row >= T.R) return;
  const int c = (int)(t - row * L) * 4;
  const uint64_t seed = gps::salted_seed(T.seed, G.salt), seed2 = gps::salted_seed(T.seed2, G.salt);
  const bool drop = T.p > 0.0f;
  if (T.z2) {
    run_bwd_apply<false, false, true>(T, d, row, c, seed, seed2, gps::salted_seed(T.seed1x, G.salt));
    return;
  }
  if (T.relu) {
    if (drop) run_bwd_apply<true, true, false>(T, d, row, c, seed, seed2);
    else run_bwd_apply<true, false, false>(T, d, row, c, seed, seed2);
  } else {
    if (drop) run_bwd_apply<false, true, false>(T, d, row, c, seed, seed2);
    else run_bwd_apply<false, false, false>(T, d, row, c, seed, seed2);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }
inline int rows_per_block(int64_t R) { return (int)std::max<int64_t>(8, (R + TARGET_BLOCKS - 1) / TARGET_BLOCKS); }
inline int nblocks_for(int64_t R) { const int rpb = rows_per_block(R); return (int)((R + rpb - 1) / rpb); }
inline Bn bn_of(const gps_bn* b) { return Bn{b->mean, b->rstd, b->gamma, b->beta}; }

int check_common(const char* who, int64_t R, int d) {
  GPS_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024, "%s: d=%d must be a multiple of 4, <= 1024", who, d);
  GPS_REQUIRE(R >= 2 && R < INT32_MAX, "%s: need 2 <= rows < 2^31 (got %lld)", who, (long long)R);
  return GPS_OK;
}
int check_bn(const char* who, const gps_bn* b, bool need_stats_out) {
  GPS_REQUIRE(b && b->gamma && b->beta && b->mean && b->rstd, "%s: incomplete gps_bn", who);
  GPS_REQUIRE(al16(b->gamma) && al16(b->beta) && al16(b->mean) && al16(b->rstd), "%s: gps_bn buffers must be 16-byte aligned", who);
  GPS_REQUIRE((b->running_mean == nullptr) == (b->running_var == nullptr), "%s: running stats", who);
  (void)need_stats_out;
  return GPS_OK;
}

struct FwdPlan {
  FwdGroup g;
  FinGroup f;
  int blocks, fin_blocks;
};

void add_fwd(FwdPlan& P, int kind, const float* a, const float* b, const float* res, const gps_bn* bn1,
             const gps_bn* bn2, int relu, float p, uint64_t seed, float* out, int64_t R,
             const gps_bn* stats_for, float*& ws) {
  FwdTask& T = P.g.t[P.g.n++];
  T = FwdTask{};
  T.a = a; T.b = b; T.res = res;
  if (bn1) T.bn1 = bn_of(bn1);
  if (bn2) T.bn2 = bn_of(bn2);
  T.out = out; T.R = R; T.seed = seed; T.p = p; T.kind = kind; T.relu = relu;
  T.rpb = rows_per_block(R); T.nblk = nblocks_for(R);
  T.block_begin = P.blocks;
  P.blocks += T.nblk;
  if (stats_for) {
    T.ws = ws;
    FinTask& F = P.f.t[P.f.n++];
    F = FinTask{};
    F.ws = ws; F.mean = stats_for->mean; F.rstd = stats_for->rstd;
    F.running_mean = stats_for->running_mean; F.running_var = stats_for->running_var;
    F.count = (float)R; F.eps = stats_for->eps; F.momentum = stats_for->momentum;
    F.nblk = T.nblk; F.rpb = T.rpb; F.block_begin = P.fin_blocks;
    P.fin_blocks += (P.g.d + FCOLS - 1) / FCOLS;
    ws += (size_t)T.nblk * 2 * P.g.d;
  }
}

int launch_fwd(FwdPlan& P, hipStream_t s, const char* who) {
  const int d = P.g.d;
  P.f.d = d;
  P.g.salt = gps::dropout_salt();
  const int RS = 256 / (d / 4);
  if (P.blocks > 0) k_rows_fwd<<<(unsigned)P.blocks, 256, sizeof(float) * 2 * RS * d, s>>>(P.g);
  if (P.f.n > 0) k_stats_finalize<<<(unsigned)P.fin_blocks, 256, 0, s>>>(P.f);
  return gps::launch_status(who);
}

void add_bwd(BwdGroup& G, int& blocks, int& fin_blocks, int64_t& threads, const float* z, const float* g_y,
             const gps_bn* bn, int relu, float p, uint64_t seed, const float* z2, const gps_bn* bn2,
             float* g_beta, float* g_gamma, float* g_beta2, float* g_gamma2, float* g_z, float* g_sum,
             float* g_drop, float p2, uint64_t seed2, int64_t R, float*& ws) {
  BwdTask& T = G.t[G.n++];
  T = BwdTask{};
  T.z = z; T.g_y = g_y; T.bn = bn_of(bn); T.z2 = z2;
  if (bn2) T.bn2 = bn_of(bn2);
  T.ws = ws; T.g_beta = g_beta; T.g_gamma = g_gamma; T.g_beta2 = g_beta2; T.g_gamma2 = g_gamma2;
  T.g_z = g_z; T.g_sum = g_sum; T.g_drop = g_drop;
  T.R = R; T.seed = seed; T.seed2 = seed2; T.p = p; T.p2 = p2; T.relu = relu;
  T.rpb = rows_per_block(R); T.nblk = nblocks_for(R);
  T.block_begin = blocks; blocks += T.nblk;
  T.fin_begin = fin_blocks; fin_blocks += (G.d + FCOLS - 1) / FCOLS;
  T.thread_begin = threads;
  threads += (R * (int64_t)(G.d / 4) + 255) / 256 * 256;
  ws += (size_t)T.nblk * 3 * G.d;
}

int launch_bwd(BwdGroup& G, int blocks, int fin_blocks, int64_t threads, hipStream_t s, const char* who) {
  G.salt = gps::dropout_salt();
  const int d = G.d;
  const int RS = 256 / (d / 4);
  k_bwd_partial<<<(unsigned)blocks, 256, sizeof(float) * 3 * RS * d, s>>>(G);
  k_bwd_finalize<<<(unsigned)fin_blocks, 256, 0, s>>>(G);
  k_bwd_apply<<<gps::grid_for(threads, 256), 256, 0, s>>>(G);
  return gps::launch_status(who);
}

}  // namespace

extern "C" {

size_t gps_block_norm_workspace_floats(int64_t N, int64_t E, int d) {
  if (N < 1 || E < 0 || d < 1) return 0;
  // the largest list: three tasks with up to 3 partial columns each
  return ((size_t)nblocks_for(N) * 2 + (size_t)nblocks_for(std::max<int64_t>(E, 1))) * 3 * (size_t)d + 16;
}

int gps_bn_stats_pair(const float* zA, int64_t RA, const gps_bn* bnA, const float* zB, int64_t RB,
                      const gps_bn* bnB, int d, float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_bn_stats_pair", RA, d)) return rc;
  if (int rc = check_common("gps_bn_stats_pair", RB, d)) return rc;
  if (int rc = check_bn("gps_bn_stats_pair", bnA, true)) return rc;
  if (int rc = check_bn("gps_bn_stats_pair", bnB, true)) return rc;
  GPS_REQUIRE(zA && zB && ws && al16(zA) && al16(zB) && al16(ws), "gps_bn_stats_pair: null/misaligned buffer");
  FwdPlan P{};
  P.g.d = d;
  add_fwd(P, K_LOAD, zA, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0, nullptr, RA, bnA, ws);
  add_fwd(P, K_LOAD, zB, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0, nullptr, RB, bnB, ws);
  return launch_fwd(P, gps::as_stream(stream), "gps_bn_stats_pair");
}

int gps_block_mid_fwd(const float* xt, const float* x, const gps_bn* bn_x, float p, uint64_t seed_x,
                      float* x1, const float* eh, const float* e, const gps_bn* bn_e, uint64_t seed_e,
                      float* e1, const float* ao, float p_attn, uint64_t seed_a, float* za,
                      const gps_bn* bn_local, const gps_bn* bn_attn, int64_t N, int64_t E, int d,
                      float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_block_mid_fwd", N, d)) return rc;
  if (int rc = check_common("gps_block_mid_fwd", E, d)) return rc;
  for (const gps_bn* b : {bn_x, bn_e, bn_local, bn_attn})
    if (int rc = check_bn("gps_block_mid_fwd", b, true)) return rc;
  GPS_REQUIRE(xt && x && x1 && eh && e && e1 && ao && za && ws, "gps_block_mid_fwd: null buffer");
  GPS_REQUIRE(al16(xt) && al16(x) && al16(x1) && al16(eh) && al16(e) && al16(e1) && al16(ao) && al16(za) && al16(ws),
              "gps_block_mid_fwd: buffers must be 16-byte aligned");
  GPS_REQUIRE(p >= 0.f && p < 1.f && p_attn >= 0.f && p_attn < 1.f, "gps_block_mid_fwd: dropout p");
  FwdPlan P{};
  P.g.d = d;
  add_fwd(P, K_BN_ACT, xt, nullptr, x, bn_x, nullptr, 1, p, seed_x, x1, N, bn_local, ws);
  add_fwd(P, K_BN_ACT, eh, nullptr, e, bn_e, nullptr, 1, p, seed_e, e1, E, nullptr, ws);
  add_fwd(P, K_ADD_DROP, x, ao, nullptr, nullptr, nullptr, 0, p_attn, seed_a, za, N, bn_attn, ws);
  return launch_fwd(P, gps::as_stream(stream), "gps_block_mid_fwd");
}

int gps_bn_dual_apply(const float* z1, const gps_bn* bn1, const float* z2, const gps_bn* bn2, int64_t R,
                      int d, float* out, gps_stream_t stream) {
  if (int rc = check_common("gps_bn_dual_apply", R, d)) return rc;
  if (int rc = check_bn("gps_bn_dual_apply", bn1, false)) return rc;
  if (int rc = check_bn("gps_bn_dual_apply", bn2, false)) return rc;
  GPS_REQUIRE(z1 && z2 && out && al16(z1) && al16(z2) && al16(out), "gps_bn_dual_apply: null/misaligned buffer");
  FwdPlan P{};
  P.g.d = d;
  float* none = nullptr;
  add_fwd(P, K_BN_DUAL, z1, z2, nullptr, bn1, bn2, 0, 0.f, 0, out, R, nullptr, none);
  return launch_fwd(P, gps::as_stream(stream), "gps_bn_dual_apply");
}

int gps_add_drop_stats(const float* a, const float* b, int64_t R, int d, float p, uint64_t seed, float* out,
                       const gps_bn* bn, float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_add_drop_stats", R, d)) return rc;
  if (int rc = check_bn("gps_add_drop_stats", bn, true)) return rc;
  GPS_REQUIRE(a && b && out && ws && al16(a) && al16(b) && al16(out) && al16(ws) && p >= 0.f && p < 1.f,
              "gps_add_drop_stats: bad arguments");
  FwdPlan P{};
  P.g.d = d;
  add_fwd(P, K_ADD_DROP, a, b, nullptr, nullptr, nullptr, 0, p, seed, out, R, bn, ws);
  return launch_fwd(P, gps::as_stream(stream), "gps_add_drop_stats");
}

int gps_add_drop_stats_pair(const float* a1, const float* b1, float p1, uint64_t seed1, float* out1,
                            const gps_bn* bn1, const float* a2, const float* b2, float p2, uint64_t seed2,
                            float* out2, const gps_bn* bn2, int64_t R, int d, float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_add_drop_stats_pair", R, d)) return rc;
  if (int rc = check_bn("gps_add_drop_stats_pair", bn1, true)) return rc;
  if (int rc = check_bn("gps_add_drop_stats_pair", bn2, true)) return rc;
  GPS_REQUIRE(a1 && b1 && out1 && a2 && b2 && out2 && ws && al16(a1) && al16(b1) && al16(out1) && al16(a2) &&
                  al16(b2) && al16(out2) && al16(ws) && p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f,
              "gps_add_drop_stats_pair: bad arguments");
  FwdPlan P{};
  P.g.d = d;
  add_fwd(P, K_ADD_DROP, a1, b1, nullptr, nullptr, nullptr, 0, p1, seed1, out1, R, bn1, ws);
  add_fwd(P, K_ADD_DROP, a2, b2, nullptr, nullptr, nullptr, 0, p2, seed2, out2, R, bn2, ws);
  return launch_fwd(P, gps::as_stream(stream), "gps_add_drop_stats_pair");
}

int gps_bn_bwd_drop(const float* z, const float* g_y, const gps_bn* bn, int64_t R, int d, int relu, float p,
                    uint64_t seed, float* g_z, float* g_gamma, float* g_beta, float p2, uint64_t seed2,
                    float* g_drop, float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_bn_bwd_drop", R, d)) return rc;
  if (int rc = check_bn("gps_bn_bwd_drop", bn, false)) return rc;
  GPS_REQUIRE(z && g_y && g_z && g_gamma && g_beta && ws && al16(z) && al16(g_y) && al16(g_z) &&
                  al16(g_gamma) && al16(g_beta) && al16(g_drop) && al16(ws),
              "gps_bn_bwd_drop: null/misaligned buffer");
  BwdGroup G{};
  G.d = d;
  int blocks = 0, fin = 0;
  int64_t threads = 0;
  add_bwd(G, blocks, fin, threads, z, g_y, bn, relu, p, seed, nullptr, nullptr, g_beta, g_gamma, nullptr,
          nullptr, g_z, nullptr, g_drop, p2, seed2, R, ws);
  return launch_bwd(G, blocks, fin, threads, gps::as_stream(stream), "gps_bn_bwd_drop");
}

int gps_bn_dual_bwd(const float* z1, const gps_bn* bn1, const float* z2, const gps_bn* bn2, const float* g_y,
                    int64_t R, int d, float* g_z1, float p1, uint64_t seed1, float* g_sum, float p2,
                    uint64_t seed2, float* g_drop2,
                    float* g_gamma1, float* g_beta1, float* g_gamma2, float* g_beta2, float* ws,
                    gps_stream_t stream) {
  if (int rc = check_common("gps_bn_dual_bwd", R, d)) return rc;
  if (int rc = check_bn("gps_bn_dual_bwd", bn1, false)) return rc;
  if (int rc = check_bn("gps_bn_dual_bwd", bn2, false)) return rc;
  GPS_REQUIRE(z1 && z2 && g_y && g_z1 && g_sum && g_gamma1 && g_beta1 && g_gamma2 && g_beta2 && ws,
              "gps_bn_dual_bwd: null buffer");
  GPS_REQUIRE(al16(z1) && al16(z2) && al16(g_y) && al16(g_z1) && al16(g_sum) && al16(g_drop2) &&
                  al16(g_gamma1) && al16(g_beta1) && al16(g_gamma2) && al16(g_beta2) && al16(ws),
              "gps_bn_dual_bwd: buffers must be 16-byte aligned");
  BwdGroup G{};
  G.d = d;
  int blocks = 0, fin = 0;
  int64_t threads = 0;
  add_bwd(G, blocks, fin, threads, z1, g_y, bn1, 0, 0.f, 0, z2, bn2, g_beta1, g_gamma1, g_beta2, g_gamma2,
          g_z1, g_sum, g_drop2, p2, seed2, R, ws);
  G.t[0].p1x = p1;
  G.t[0].seed1x = seed1;
  return launch_bwd(G, blocks, fin, threads, gps::as_stream(stream), "gps_bn_dual_bwd");
}

int gps_bn_bwd_pair(const float* zA, const float* gA, const gps_bn* bnA, int64_t RA, uint64_t seedA,
                    float* g_zA, float* g_gammaA, float* g_betaA, const float* zB, const float* gB,
                    const gps_bn* bnB, int64_t RB, uint64_t seedB, float* g_zB, float* g_gammaB,
                    float* g_betaB, int d, int relu, float p, float* ws, gps_stream_t stream) {
  if (int rc = check_common("gps_bn_bwd_pair", RA, d)) return rc;
  if (int rc = check_common("gps_bn_bwd_pair", RB, d)) return rc;
  if (int rc = check_bn("gps_bn_bwd_pair", bnA, false)) return rc;
  if (int rc = check_bn("gps_bn_bwd_pair", bnB, false)) return rc;
  GPS_REQUIRE(zA && gA && g_zA && g_gammaA && g_betaA && zB && gB && g_zB && g_gammaB && g_betaB && ws,
              "gps_bn_bwd_pair: null buffer");
  GPS_REQUIRE(al16(zA) && al16(gA) && al16(g_zA) && al16(g_gammaA) && al16(g_betaA) && al16(zB) && al16(gB) &&
                  al16(g_zB) && al16(g_gammaB) && al16(g_betaB) && al16(ws),
              "gps_bn_bwd_pair: buffers must be 16-byte aligned");
  BwdGroup G{};
  G.d = d;
  int blocks = 0, fin = 0;
  int64_t threads = 0;
  add_bwd(G, blocks, fin, threads, zA, gA, bnA, relu, p, seedA, nullptr, nullptr, g_betaA, g_gammaA, nullptr,
          nullptr, g_zA, nullptr, nullptr, 0.f, 0, RA, ws);
  add_bwd(G, blocks, fin, threads, zB, gB, bnB, relu, p, seedB, nullptr, nullptr, g_betaB, g_gammaB, nullptr,
          nullptr, g_zB, nullptr, nullptr, 0.f, 0, RB, ws);
  return launch_bwd(G, blocks, fin, threads, gps::as_stream(stream), "gps_bn_bwd_pair");
}

}  // extern "C"
