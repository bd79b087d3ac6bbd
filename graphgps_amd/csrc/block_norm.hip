// Task-list BatchNorm / residual / dropout kernels of the fused GPS blocks (graphgps_amd/layer/gps_block.py).
//
// Reference stages: graphgps/layer/gatedgcn_layer.py:72-83 (bn_node_x / bn_edge_e, ReLU, dropout, residual),
// graphgps/layer/gps_layer.py:191-194,212-229 (norm1_local, norm1_attn, the h_local + h_attn sum, norm2) and their
// autograd backward.  bn_fused.hip gives every BatchNorm1d its own three launches (partial statistics, finalize, apply)
// plus one launch per residual + dropout: 37 launches per layer.  Round 1 issued the same arithmetic as LISTS of up to
// four independent row-stream tasks per launch (19 launches per layer, 17 in round 2).  Round 3:
//   * no finalize launches: every kernel that produces column partials (batch statistics forward, sum g / sum g*zhat
//     backward) completes them in-launch through csrc/col_tree.hpp (write-through records, arrival tickets, the last
//     arriver of a group / of the tree combines in index order: deterministic, nobody waits);
//   * the backward applies are row-block kernels like the partials (column constants loaded once per thread, two rows in
//     flight) and can CHAIN: while a task writes the gradient it produces, it accumulates the column partials of the next
//     BatchNorm backward that consumes that gradient (norm1_local's g_x1 -> bn_node_x's sum g, sum g*zhat), which removes
//     that BatchNorm's own partial pass;
//   * the C ABI is a generic task list (gps_norm_fwd / gps_norm_bwd_partial / gps_norm_bwd_apply), composed by the host.
// Launches per CustomGatedGCN+Transformer layer: forward statistics of x~ / e^ (a LOAD list; or out of the GatedGCN
// kernel, gps_gatedgcn_fwd_stats) + mid + dual apply + norm2 apply (the statistics of z2 and za come from the ring GEMM
// epilogues), backward 5  =>  9, or 8 with the GatedGCN variant (round 2: 17).
//
// Row kernels keep the lane-owns-4-channels mapping (d % 4 == 0, d <= 1024): a workgroup is RS rows x d/4 lanes
// (RS = 512 / (d/4): 384 threads at d = 384), rows of a block are walked RS at a time, two passes in flight.
#include <algorithm>
#include <cstdlib>

#include "col_tree.hpp"
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

namespace tr = gps::tree;

// Row blocks per task: one per CU.  The blocks of a task are also the level-0 records of its reduction tree, and 256 records
// = 16 groups of 16, so BOTH levels of the tree are ONE burst of <= 16 x 3 loads per thread (MAXI).  The burst lives in
// registers, and the kernel's register allocation is its maximum over all paths: 24-record bursts (512 blocks) cost the
// streaming loop half of its occupancy (133-154 VGPRs -> 3 waves per SIMD; measured 37 us for a 106 MB list).
constexpr int TARGET_BLOCKS = 256;
constexpr int kMaxThreads = 512;
constexpr int kMaxTasks = 4;
constexpr int MAXI = 16;
typedef Vec<4> V4;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// identical to bn_fused.hip / gemm_panel.hip: (row id, element index) -> keep decision
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

// reduce `NVEC` per-thread column vectors over the RS row lanes of the block (fixed order); result valid for rsub == 0
template <int NVEC>
__device__ __forceinline__ void reduce_rows(V4 (&s)[NVEC], float* lds, int d, int RS, int rsub, int c) {
  if (rsub > 0) {
#pragma unroll
    for (int v = 0; v < NVEC; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(rsub * NVEC + v) * d + c + j] = s[v][j];
  }
  __syncthreads();
  if (rsub == 0) {
    for (int q = 1; q < RS; ++q)
#pragma unroll
      for (int v = 0; v < NVEC; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[v][j] += lds[(q * NVEC + v) * d + c + j];
  }
}

// max |.| over a block's stored values -> ONE atomic on the tensor's word (fp32 bit patterns: an unsigned max, order-free).
// The words feed the fp16 form of the ring GEMM (csrc/gemm_panel.hip) that consumes the stored tensor.  Through LDS, not
// wave shuffles: the last wave of a block may be partly inactive (480 threads at d = 384).
__device__ __forceinline__ float vmax4(float mx, const V4& v) {
  return fmaxf(fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fabsf(v[1])), fabsf(v[2])), fabsf(v[3]));     // (a chain: no temporaries)
}
__device__ __forceinline__ void block_amax(float mx, uint32_t* slot, float* lds) {
  uint32_t* w = reinterpret_cast<uint32_t*>(lds);
  const int t = threadIdx.x, n = blockDim.x;
  __syncthreads();                               // whoever used the scratch before is done
  w[t] = __float_as_uint(mx);
  __syncthreads();
  if (t < 64) {                                  // wave 0 is always whole (threads_for >= 256)
    uint32_t m = w[t];
    for (int q = t + 64; q < n; q += 64) m = max(m, w[q]);
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if (t == 0) gps::amax_raise(slot, m);
  }
  __syncthreads();                               // the scratch may be reused
}

// ------------------------------------------------------------------------------------------------
// forward: produce rows (optionally store them, optionally accumulate their column statistics)
// ------------------------------------------------------------------------------------------------
enum { K_LOAD = 0, K_ADD_DROP = 1, K_BN_ACT = 2, K_BN_DUAL = 3 };

struct FwdTask {
  const float *a, *b, *res;
  Bn bn1, bn2;
  float* out;     // produced rows (nullptr: statistics only)
  uint32_t* amax; // max|out| word, or nullptr
  const int32_t* rdev;   // padded batches: the number of REAL rows (device word; rows R_real .. R-1 are padding), or nullptr
  int64_t R;
  uint64_t seed;
  float p;
  int kind, relu, rpb, nblk, block_begin, has_stats;
  tr::Tree tree;  // statistics of the produced rows
};
struct FwdGroup {
  FwdTask t[kMaxTasks];
  const uint64_t* salt;
  int n, d;
};

struct RowIn {
  V4 a, b, r;
};
template <int KIND>
__device__ __forceinline__ RowIn load_in(const FwdTask& T, int64_t row, int c, int d) {
  RowIn in;
  in.a = V4::load(T.a + row * d + c);
  in.b = (KIND == K_ADD_DROP || KIND == K_BN_DUAL) ? V4::load(T.b + row * d + c) : V4::zero();
  in.r = (KIND == K_BN_ACT && T.res) ? V4::load(T.res + row * d + c) : V4::zero();
  return in;
}
template <int KIND, bool RELU, bool DROP>
__device__ __forceinline__ V4 eval_row(const FwdTask& T, const RowIn& in, int64_t r, int c, const Col& c1, const Col& c2,
                                       uint64_t seed, float inv_keep) {
  V4 o;
  if (KIND == K_LOAD) {
    o = in.a;
  } else if (KIND == K_ADD_DROP) {             // a + drop(b)
    const uint32_t rh = DROP ? row_hash((uint32_t)r, seed) : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = in.b[j];
      if (DROP) u = keep_elem(rh, (uint32_t)(c + j), T.p) ? u * inv_keep : 0.0f;
      o[j] = in.a[j] + u;
    }
  } else if (KIND == K_BN_ACT) {               // res + drop(relu(BN(a)))
    const uint32_t rh = DROP ? row_hash((uint32_t)r, seed) : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = (in.a[j] - c1.mu[j]) * c1.rs[j] * c1.ga[j] + c1.be[j];
      if (RELU) u = fmaxf(u, 0.0f);
      if (DROP) u = keep_elem(rh, (uint32_t)(c + j), T.p) ? u * inv_keep : 0.0f;
      o[j] = T.res ? in.r[j] + u : u;
    }
  } else {                                     // BN1(a) + BN2(b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u1 = (in.a[j] - c1.mu[j]) * c1.rs[j] * c1.ga[j] + c1.be[j];
      const float u2 = (in.b[j] - c2.mu[j]) * c2.rs[j] * c2.ga[j] + c2.be[j];
      o[j] = u1 + u2;
    }
  }
  return o;
}

// Returns whether this block wrote a level-0 statistics record (then the kernel calls tr::arrive, ONE copy per kernel).
template <int KIND, bool RELU, bool DROP>
__device__ __forceinline__ bool run_fwd(const FwdTask& T, int d, int lb, uint64_t seed, float* lds) {
  const int L = d >> 2;
  const int RS = blockDim.x / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * 4;
  const int64_t row0 = (int64_t)lb * T.rpb;
  const int64_t row1 = min(T.R, row0 + T.rpb);
  // Padded batches (round 4: a loader pads N / E up to a bucket so that a captured step can be replayed): rows at or past
  // *rdev are computed and stored like any other row but never enter the statistics.
  const int64_t rreal = T.rdev ? min((int64_t)*T.rdev, T.R) : T.R;
  const bool stats = T.has_stats != 0;
  const float inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
  Col c1, c2;
  if (KIND == K_BN_ACT || KIND == K_BN_DUAL) c1 = load_col(T.bn1, c);
  if (KIND == K_BN_DUAL) c2 = load_col(T.bn2, c);
  V4 k = V4::zero();
  V4 s[2] = {V4::zero(), V4::zero()};
  // shift = the block's own first produced row (block-local shifted sums, see bn_fused.hip)
  if (stats) k = eval_row<KIND, RELU, DROP>(T, load_in<KIND>(T, row0, c, d), row0, c, c1, c2, seed, inv_keep);
  auto account = [&](const V4& v) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = v[j] - k[j];
      s[0][j] += t;
      s[1][j] += t * t;
    }
  };
  // two rows per pass, the NEXT pass's rows requested before this pass's results are stored (clamped addresses: the
  // loads are unconditional), so a pass costs issue time, not a memory round trip
  int64_t r = row0 + rsub;
  const int64_t rl = row1 - 1;
  float mx = 0.0f;
  if (r < row1) {
    RowIn i0 = load_in<KIND>(T, r, c, d), i1 = load_in<KIND>(T, min(r + RS, rl), c, d);
    for (; r + RS < row1; r += 2 * RS) {
      const RowIn n0 = load_in<KIND>(T, min(r + 2 * RS, rl), c, d), n1 = load_in<KIND>(T, min(r + 3 * RS, rl), c, d);
      const V4 v0 = eval_row<KIND, RELU, DROP>(T, i0, r, c, c1, c2, seed, inv_keep);
      const V4 v1 = eval_row<KIND, RELU, DROP>(T, i1, r + RS, c, c1, c2, seed, inv_keep);
      if (T.out) { v0.store(T.out + r * d + c); v1.store(T.out + (r + RS) * d + c); }
      if (stats) {
        if (r < rreal) account(v0);
        if (r + RS < rreal) account(v1);
      }
      mx = vmax4(vmax4(mx, v0), v1);
      i0 = n0; i1 = n1;
    }
    if (r < row1) {
      const V4 v0 = eval_row<KIND, RELU, DROP>(T, i0, r, c, c1, c2, seed, inv_keep);
      if (T.out) v0.store(T.out + r * d + c);
      if (stats && r < rreal) account(v0);
      mx = vmax4(mx, v0);
    }
  }
  if (T.amax) block_amax(mx, T.amax, lds);      // block-uniform
  if (!stats) return false;      // block-uniform
  reduce_rows<2>(s, lds, d, RS, rsub, c);
  if (rsub == 0) {
    const int64_t nreal = min(row1, rreal) - row0;
    const float n = nreal > 0 ? (float)nreal : 0.0f;        // a block of padding only: an empty record (count 0)
    float* rec = T.tree.part + (int64_t)lb * 2 * d;
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // (divisions, not a reciprocal: the un-padded records keep the bits they always had)
      tr::st_sc1(rec + c + j, n > 0.0f ? k[j] + s[0][j] / n : 0.0f);
      tr::st_sc1(rec + d + c + j, n > 0.0f ? fmaxf(s[1][j] - s[0][j] * s[0][j] / n, 0.0f) : 0.0f);
    }
    if (c == 0) tr::st_sc1(T.tree.pcnt + lb, n);
  }
  return true;
}

__global__ __launch_bounds__(kMaxThreads) void k_rows_fwd(const FwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const FwdTask& T = G.t[ti];
  const int lb = blockIdx.x - T.block_begin;
  if (lb >= T.nblk) return;
  const uint64_t seed = gps::salted_seed(T.seed, G.salt);
  const bool drop = T.p > 0.0f;
  bool rec;
  switch (T.kind) {
    case K_LOAD: rec = run_fwd<K_LOAD, false, false>(T, G.d, lb, seed, lds); break;
    case K_ADD_DROP:
      if (drop) rec = run_fwd<K_ADD_DROP, false, true>(T, G.d, lb, seed, lds);
      else rec = run_fwd<K_ADD_DROP, false, false>(T, G.d, lb, seed, lds);
      break;
    case K_BN_ACT:
      if (T.relu) {
        if (drop) rec = run_fwd<K_BN_ACT, true, true>(T, G.d, lb, seed, lds);
        else rec = run_fwd<K_BN_ACT, true, false>(T, G.d, lb, seed, lds);
      } else {
        if (drop) rec = run_fwd<K_BN_ACT, false, true>(T, G.d, lb, seed, lds);
        else rec = run_fwd<K_BN_ACT, false, false>(T, G.d, lb, seed, lds);
      }
      break;
    default: rec = run_fwd<K_BN_DUAL, false, false>(T, G.d, lb, seed, lds); break;
  }
  if (rec) tr::arrive<2, tr::STATS, MAXI>(T.tree, lb, G.d, lds);     // block-uniform
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
struct BwdTask {
  const float *z, *g_y;
  Bn bn;                 // primary BN; relu / (p, seed) are the masks applied to ITS output
  const float* z2;       // dual: a second BN (no masks) fed by the same g_y, or nullptr
  Bn bn2;
  float *g_beta, *g_gamma, *g_beta2, *g_gamma2;   // column sums: written by the partial kernel, read by the apply kernel
  float* g_z;            // primary input gradient
  float* g_sum;          // dual: g_z + g_z2
  float* g_drop;         // dropmask(seed2, p2) of g_z (dual: of g_z2), or nullptr
  uint32_t* amax_drop;   // apply kernel: max|g_drop| word, or nullptr
  const int32_t* rdev;   // padded batches: number of REAL rows (device word), or nullptr
  float p1x;             // dual only: dropout (p1x, seed1x) applied to the STORED g_z (g_sum stays unmasked)
  uint64_t seed1x;
  int64_t R;
  uint64_t seed, seed2;
  float p, p2;
  int relu, rpb, nblk, block_begin;
  // chain (apply kernel): the produced primary gradient (before the p1x mask) is the output gradient of another
  // BatchNorm(cz) -> ReLU -> dropout(cp, cseed); its column sums are accumulated here
  const float* cz;
  Bn cbn;
  uint64_t cseed;
  float cp;
  int crelu, has_chain;
  tr::Tree tree;         // partial kernel: sums of this task; apply kernel: sums of the chain
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
  const int RS = blockDim.x / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * 4;
  const int64_t row0 = (int64_t)lb * T.rpb;
  const int64_t row1 = min(T.R, row0 + T.rpb);
  constexpr int NV = DUAL ? 3 : 2;
  V4 s[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) s[v] = V4::zero();
  const Col c1 = load_col(T.bn, c);
  Col c2;
  if (DUAL) c2 = load_col(T.bn2, c);
  const float inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
  auto account = [&](int64_t r, const V4& v, const V4& gy, const V4& v2) __attribute__((always_inline)) {
    V4 g, zh;
    out_grad<RELU, DROP>(v, gy, c1, DROP ? row_hash((uint32_t)r, seed) : 0u, c, T.p, inv_keep, g, zh);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[0][j] += g[j];
      s[1][j] += g[j] * zh[j];
      if (DUAL) s[NV - 1][j] += g[j] * ((v2[j] - c2.mu[j]) * c2.rs[j]);
    }
  };
  struct In { V4 v, g, w; };
  auto ld = [&](int64_t q) __attribute__((always_inline)) {
    In x;
    x.v = V4::load(T.z + q * d + c);
    x.g = V4::load(T.g_y + q * d + c);
    x.w = DUAL ? V4::load(T.z2 + q * d + c) : V4::zero();
    return x;
  };
  int64_t r = row0 + rsub;
  const int64_t rl = row1 - 1;
  if (r < row1) {           // two rows per pass, the next pass prefetched (see run_fwd)
    In a = ld(r), b = ld(min(r + RS, rl));
    for (; r + RS < row1; r += 2 * RS) {
      const In na = ld(min(r + 2 * RS, rl)), nb = ld(min(r + 3 * RS, rl));
      account(r, a.v, a.g, a.w);
      account(r + RS, b.v, b.g, b.w);
      a = na; b = nb;
    }
    if (r < row1) account(r, a.v, a.g, a.w);
  }
  reduce_rows<NV>(s, lds, d, RS, rsub, c);
  if (rsub == 0) {
    float* rec = T.tree.part + (int64_t)lb * NV * d;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) tr::st_sc1(rec + v * d + c + j, s[v][j]);
  }
}

__global__ __launch_bounds__(kMaxThreads) void k_bwd_partial(const BwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const BwdTask& T = G.t[ti];
  const int lb = blockIdx.x - T.block_begin;
  if (lb >= T.nblk) return;
  const uint64_t seed = gps::salted_seed(T.seed, G.salt);
  const bool drop = T.p > 0.0f;
  if (T.z2) {
    run_bwd_partial<false, false, true>(T, G.d, lb, seed, lds);
    tr::arrive<3, tr::SUMS, MAXI>(T.tree, lb, G.d, lds);
    return;
  }
  if (T.relu) {
    if (drop) run_bwd_partial<true, true, false>(T, G.d, lb, seed, lds);
    else run_bwd_partial<true, false, false>(T, G.d, lb, seed, lds);
  } else {
    if (drop) run_bwd_partial<false, true, false>(T, G.d, lb, seed, lds);
    else run_bwd_partial<false, false, false>(T, G.d, lb, seed, lds);
  }
  tr::arrive<2, tr::SUMS, MAXI>(T.tree, lb, G.d, lds);
}

struct ApplyCtx {
  Col c1, c2, cc;
  V4 s1, s2, s3;
  float inv_keep, inv_n, ik1, ik2, cik;
  uint64_t seed, seed2, seed1x, cseed;
};

// one row of an apply task: stores the gradients, returns the primary one (input of the chain)
template <bool RELU, bool DROP, bool DUAL>
__device__ __forceinline__ V4 apply_row(const BwdTask& T, const ApplyCtx& A, int d, int64_t row, int c, const V4& v,
                                        const V4& gy, const V4& v2, float& dmx, float gate) {
  V4 g, zh, o;
  out_grad<RELU, DROP>(v, gy, A.c1, DROP ? row_hash((uint32_t)row, A.seed) : 0u, c, T.p, A.inv_keep, g, zh);
#pragma unroll
  for (int j = 0; j < 4; ++j)      // gate = 0 on the padding rows of a padded batch (1 everywhere else)
    o[j] = gate * (A.c1.ga[j] * A.c1.rs[j] * (g[j] - A.s1[j] * A.inv_n - zh[j] * A.s2[j] * A.inv_n));
  if (DUAL && T.p1x > 0.0f) {
    const uint32_t rh1 = row_hash((uint32_t)row, A.seed1x);
    V4 om;
#pragma unroll
    for (int j = 0; j < 4; ++j) om[j] = keep_elem(rh1, (uint32_t)(c + j), T.p1x) ? o[j] * A.ik1 : 0.0f;
    om.store(T.g_z + row * d + c);
  } else {
    o.store(T.g_z + row * d + c);
  }
  V4 last = o;
  if (DUAL) {
    V4 o2, sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float zh2 = (v2[j] - A.c2.mu[j]) * A.c2.rs[j];
      o2[j] = gate * (A.c2.ga[j] * A.c2.rs[j] * (g[j] - A.s1[j] * A.inv_n - zh2 * A.s3[j] * A.inv_n));
      sum[j] = o[j] + o2[j];
    }
    if (T.g_sum) sum.store(T.g_sum + row * d + c);
    last = o2;
  }
  if (T.g_drop) {
    V4 q = last;
    if (T.p2 > 0.0f) {
      const uint32_t rh2 = row_hash((uint32_t)row, A.seed2);
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = keep_elem(rh2, (uint32_t)(c + j), T.p2) ? last[j] * A.ik2 : 0.0f;
    }
    q.store(T.g_drop + row * d + c);
    dmx = vmax4(dmx, q);
  }
  return o;
}

template <bool RELU, bool DROP, bool DUAL, bool CHAIN>
__device__ __forceinline__ void run_bwd_apply(const BwdTask& T, int d, int lb, const uint64_t* salt, float* lds) {
  const int L = d >> 2;
  const int RS = blockDim.x / L;
  const int rsub = threadIdx.x / L;
  const int c = (threadIdx.x - rsub * L) * 4;
  const int64_t row0 = (int64_t)lb * T.rpb;
  const int64_t row1 = min(T.R, row0 + T.rpb);
  ApplyCtx A;
  A.c1 = load_col(T.bn, c);
  if (DUAL) A.c2 = load_col(T.bn2, c);
  if (CHAIN) A.cc = load_col(T.cbn, c);
  A.s1 = V4::load(T.g_beta + c);
  A.s2 = V4::load(T.g_gamma + c);
  A.s3 = DUAL ? V4::load(T.g_gamma2 + c) : V4::zero();
  A.inv_keep = DROP ? 1.0f / (1.0f - T.p) : 1.0f;
  // Padded batches: the statistics were taken over the real rows, so 1/R is 1/R_real, and the padding rows' gradients are
  // forced to zero HERE -- a zero output gradient does not give a zero input gradient through BatchNorm
  // (- mean(g) - zhat mean(g zhat)), and every contraction over rows downstream (weight gradients, column sums) relies on
  // padding rows being exactly zero.
  const int64_t rreal = T.rdev ? min((int64_t)*T.rdev, T.R) : T.R;
  A.inv_n = 1.0f / (float)rreal;
  A.ik1 = T.p1x > 0.0f ? 1.0f / (1.0f - T.p1x) : 1.0f;
  A.ik2 = T.p2 > 0.0f ? 1.0f / (1.0f - T.p2) : 1.0f;
  A.cik = CHAIN && T.cp > 0.0f ? 1.0f / (1.0f - T.cp) : 1.0f;
  A.seed = gps::salted_seed(T.seed, salt);
  A.seed2 = gps::salted_seed(T.seed2, salt);
  A.seed1x = gps::salted_seed(T.seed1x, salt);
  A.cseed = gps::salted_seed(T.cseed, salt);
  V4 s[2] = {V4::zero(), V4::zero()};
  const bool cdrop = CHAIN && T.cp > 0.0f, crelu = CHAIN && T.crelu != 0;
  auto chain = [&](int64_t row, const V4& o, const V4& cz) __attribute__((always_inline)) {
    const uint32_t rh = cdrop ? row_hash((uint32_t)row, A.cseed) : 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float zh = (cz[j] - A.cc.mu[j]) * A.cc.rs[j];
      float gg = o[j];
      if (cdrop) gg = keep_elem(rh, (uint32_t)(c + j), T.cp) ? gg * A.cik : 0.0f;
      if (crelu) gg = (zh * A.cc.ga[j] + A.cc.be[j]) > 0.0f ? gg : 0.0f;
      s[0][j] += gg;
      s[1][j] += gg * zh;
    }
  };
  struct In { V4 v, g, w, cz; };
  auto ld = [&](int64_t q) __attribute__((always_inline)) {
    In x;
    x.v = V4::load(T.z + q * d + c);
    x.g = V4::load(T.g_y + q * d + c);
    x.w = DUAL ? V4::load(T.z2 + q * d + c) : V4::zero();
    x.cz = CHAIN ? V4::load(T.cz + q * d + c) : V4::zero();
    return x;
  };
  int64_t r = row0 + rsub;
  const int64_t rl = row1 - 1;
  float dmx = 0.0f;
  if (r < row1) {           // one row per pass with the next one prefetched: this kernel's column constants (up to three
    In a = ld(r);           // BatchNorms + their sums) leave no room for more rows in flight at a useful occupancy
    for (; r < row1; r += RS) {
      const In na = ld(min(r + RS, rl));
      const V4 oa = apply_row<RELU, DROP, DUAL>(T, A, d, r, c, a.v, a.g, a.w, dmx, r < rreal ? 1.0f : 0.0f);
      if (CHAIN) chain(r, oa, a.cz);
      a = na;
    }
  }
  if (T.amax_drop) block_amax(dmx, T.amax_drop, lds);      // block-uniform
  if (!CHAIN) return;
  reduce_rows<2>(s, lds, d, RS, rsub, c);
  if (rsub == 0) {
    float* rec = T.tree.part + (int64_t)lb * 2 * d;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) tr::st_sc1(rec + v * d + c + j, s[v][j]);
  }
}

// (two 480-thread blocks per CU = 4 waves per SIMD = 128 VGPRs: two-task lists have 512 blocks, and at 130 registers the
// second block of a CU waited for the first -- the bound keeps the allocator at the 128 the kernel had before the max|.| word)
__global__ __launch_bounds__(kMaxThreads, 3) void k_bwd_apply(const BwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTasks; ++i)
    if (i < G.n && (int)blockIdx.x >= G.t[i].block_begin) ti = i;
  const BwdTask& T = G.t[ti];
  const int lb = blockIdx.x - T.block_begin;
  if (lb >= T.nblk) return;
  const bool drop = T.p > 0.0f;
  if (T.has_chain) {       // block-uniform; the chain's column sums complete in this launch
    if (T.z2) run_bwd_apply<false, false, true, true>(T, G.d, lb, G.salt, lds);
    else if (T.relu) {
      if (drop) run_bwd_apply<true, true, false, true>(T, G.d, lb, G.salt, lds);
      else run_bwd_apply<true, false, false, true>(T, G.d, lb, G.salt, lds);
    } else {
      if (drop) run_bwd_apply<false, true, false, true>(T, G.d, lb, G.salt, lds);
      else run_bwd_apply<false, false, false, true>(T, G.d, lb, G.salt, lds);
    }
    tr::arrive<2, tr::SUMS, MAXI>(T.tree, lb, G.d, lds);
    return;
  }
  if (T.z2) { run_bwd_apply<false, false, true, false>(T, G.d, lb, G.salt, lds); return; }
  if (T.relu) {
    if (drop) run_bwd_apply<true, true, false, false>(T, G.d, lb, G.salt, lds);
    else run_bwd_apply<true, false, false, false>(T, G.d, lb, G.salt, lds);
  } else {
    if (drop) run_bwd_apply<false, true, false, false>(T, G.d, lb, G.salt, lds);
    else run_bwd_apply<false, false, false, false>(T, G.d, lb, G.salt, lds);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }
// (row blocks per task were an A/B handle, GPS_NORM_BLOCKS, through round 6: 256 -- one per CU -- won every sweep,
// profiles/r06_norm_probe_blocks.txt; the switch is gone)
inline int target_blocks() { return TARGET_BLOCKS; }
// (round 6: more row blocks for the tasks that own no reduction tree -- 512 .. 2,048 instead of 256 -- measured slower on
// every launch of the block, tools/norm_probe.py: the launches are bound by their fixed costs, not by blocks in flight)
inline int rows_per_block(int64_t R) { return (int)std::max<int64_t>(8, (R + target_blocks() - 1) / target_blocks()); }
inline int nblocks_for(int64_t R) { const int rpb = rows_per_block(R); return (int)((R + rpb - 1) / rpb); }
inline Bn bn_of(const gps_bn* b) { return Bn{b->mean, b->rstd, b->gamma, b->beta}; }
inline int threads_for(int d) { const int L = d / 4; return std::max(1, kMaxThreads / L) * L; }
inline size_t lds_bytes_for(int d, int nvec) {
  const int threads = threads_for(d), RS = threads / (d / 4);
  const size_t rows = (size_t)RS * nvec * d, scr = (size_t)tr::scratch_floats(3, threads);
  return sizeof(float) * std::max(rows, scr) + 16;
}

int check_common(const char* who, int64_t R, int d) {
  GPS_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024, "%s: d=%d must be a multiple of 4, <= 1024", who, d);
  GPS_REQUIRE(R >= 2 && R < INT32_MAX, "%s: need 2 <= rows < 2^31 (got %lld)", who, (long long)R);
  return GPS_OK;
}
int check_bn(const char* who, const gps_bn* b) {
  GPS_REQUIRE(b && b->gamma && b->beta && b->mean && b->rstd, "%s: incomplete gps_bn", who);
  GPS_REQUIRE(al16(b->gamma) && al16(b->beta) && al16(b->mean) && al16(b->rstd), "%s: gps_bn buffers must be 16-byte aligned", who);
  GPS_REQUIRE((b->running_mean == nullptr) == (b->running_var == nullptr), "%s: running stats", who);
  return GPS_OK;
}

// carve one tree out of the caller's workspace / counters
int take_tree(const char* who, tr::Tree& T, int P, int NV, int mode, int d, float*& ws, float* ws_end, uint32_t* sync,
              int& n_trees) {
  GPS_REQUIRE(P >= 1 && P <= tr::kMaxParts, "%s: %d column partials exceed the tree (%d)", who, P, tr::kMaxParts);
  GPS_REQUIRE(ws && sync, "%s: statistics need a workspace and a counter buffer", who);
  GPS_REQUIRE(n_trees < 2 * kMaxTasks, "%s: too many trees in one launch", who);
  GPS_REQUIRE(ws + tr::floats_for(P, NV, d) <= ws_end, "%s: workspace too small (see gps_norm_tree_floats)", who);
  T = tr::carve(ws, sync + (size_t)n_trees * tr::kSyncWords, P, NV, mode, d);
  ++n_trees;
  return GPS_OK;
}

}  // namespace

extern "C" {

size_t gps_norm_tree_floats(int64_t R, int d) {
  if (R < 1 || d < 1) return 0;
  return tr::floats_for(nblocks_for(std::max<int64_t>(R, 2)), 3, d) + 16;
}

int gps_norm_sync_words(void) { return 2 * kMaxTasks * tr::kSyncWords; }

// Arrival counters of the in-launch reductions (csrc/col_tree.hpp): zero at entry of every launch, zero at exit.
// gps_sync_reset re-zeroes a buffer (a caller that survived a failed launch); gps_sync_nonzero counts the non-zero words
// into *count (device uint32, raised atomically) -- a debugging aid behind GPS_CHECK_TICKS=1 (layer/gps_block.py).
int gps_sync_reset(uint32_t* sync, size_t words, gps_stream_t stream) {
  GPS_REQUIRE(sync || words == 0, "gps_sync_reset: null buffer");
  if (words == 0) return GPS_OK;
  const hipError_t e = hipMemsetAsync(sync, 0, words * sizeof(uint32_t), gps::as_stream(stream));
  GPS_REQUIRE(e == hipSuccess, "gps_sync_reset: %s", hipGetErrorString(e));
  return GPS_OK;
}
}  // extern "C"
namespace {
__global__ void k_sync_nonzero(const uint32_t* sync, size_t words, uint32_t* count) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < words && __hip_atomic_load(sync + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicAdd(count, 1u);
}
}  // namespace
extern "C" {
int gps_sync_nonzero(const uint32_t* sync, size_t words, uint32_t* count, gps_stream_t stream) {
  GPS_REQUIRE((sync && count) || words == 0, "gps_sync_nonzero: null buffer");
  if (words == 0) return GPS_OK;
  k_sync_nonzero<<<gps::grid_for((int64_t)words, 256), 256, 0, gps::as_stream(stream)>>>(sync, words, count);
  return gps::launch_status("gps_sync_nonzero");
}

int gps_norm_fwd(int n, const gps_norm_fwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                 gps_stream_t stream) {
  static const char* who = "gps_norm_fwd";
  GPS_REQUIRE(n >= 1 && n <= kMaxTasks && tasks, "%s: 1..%d tasks per launch", who, kMaxTasks);
  GPS_REQUIRE(al16(ws), "%s: misaligned workspace", who);
  FwdGroup G{};
  G.n = n; G.d = d;
  G.salt = gps::dropout_salt();
  float* wp = ws;
  float* wend = ws + ws_floats;
  int blocks = 0, n_trees = 0;
  for (int i = 0; i < n; ++i) {
    const gps_norm_fwd_task& S = tasks[i];
    if (int rc = check_common(who, S.R, d)) return rc;
    GPS_REQUIRE(S.kind >= K_LOAD && S.kind <= K_BN_DUAL, "%s: task %d: unknown kind %d", who, i, S.kind);
    GPS_REQUIRE(S.a && al16(S.a) && al16(S.b) && al16(S.res) && al16(S.out), "%s: task %d: null / misaligned rows", who, i);
    GPS_REQUIRE((S.kind != K_ADD_DROP && S.kind != K_BN_DUAL) || S.b, "%s: task %d: second operand missing", who, i);
    GPS_REQUIRE(S.p >= 0.f && S.p < 1.f, "%s: task %d: dropout p", who, i);
    GPS_REQUIRE(S.out || S.stats, "%s: task %d produces nothing", who, i);
    FwdTask& T = G.t[i];
    T.a = S.a; T.b = S.b; T.res = S.kind == K_BN_ACT ? S.res : nullptr;
    if (S.kind == K_BN_ACT || S.kind == K_BN_DUAL) {
      if (int rc = check_bn(who, S.bn1)) return rc;
      T.bn1 = bn_of(S.bn1);
    }
    if (S.kind == K_BN_DUAL) {
      if (int rc = check_bn(who, S.bn2)) return rc;
      T.bn2 = bn_of(S.bn2);
    }
    T.out = S.out; T.R = S.R; T.seed = S.seed;
    T.amax = S.out ? S.amax : nullptr;
    T.rdev = S.rdev;
    T.p = (S.kind == K_ADD_DROP || S.kind == K_BN_ACT) ? S.p : 0.f;
    T.kind = S.kind; T.relu = S.relu;
    T.rpb = rows_per_block(S.R); T.nblk = nblocks_for(S.R);
    T.block_begin = blocks;
    blocks += T.nblk;
    if (S.stats) {
      if (int rc = check_bn(who, S.stats)) return rc;
      if (int rc = take_tree(who, T.tree, T.nblk, 2, tr::STATS, d, wp, wend, sync, n_trees)) return rc;
      T.has_stats = 1;
      T.tree.o0 = S.stats->mean; T.tree.o1 = S.stats->rstd;
      T.tree.o2 = S.stats->running_mean; T.tree.o3 = S.stats->running_var;
      T.tree.eps = S.stats->eps; T.tree.momentum = S.stats->momentum;
    }
  }
  k_rows_fwd<<<(unsigned)blocks, threads_for(d), lds_bytes_for(d, 2), gps::as_stream(stream)>>>(G);
  return gps::launch_status(who);
}

static int fill_bwd(const char* who, int n, const gps_norm_bwd_task* tasks, int d, BwdGroup& G, int& blocks) {
  GPS_REQUIRE(n >= 1 && n <= kMaxTasks && tasks, "%s: 1..%d tasks per launch", who, kMaxTasks);
  G.n = n; G.d = d;
  G.salt = gps::dropout_salt();
  blocks = 0;
  for (int i = 0; i < n; ++i) {
    const gps_norm_bwd_task& S = tasks[i];
    if (int rc = check_common(who, S.R, d)) return rc;
    if (int rc = check_bn(who, S.bn)) return rc;
    GPS_REQUIRE(S.z && S.g_y && S.g_gamma && S.g_beta && al16(S.z) && al16(S.g_y) && al16(S.g_gamma) && al16(S.g_beta),
                "%s: task %d: null / misaligned buffer", who, i);
    GPS_REQUIRE(S.p >= 0.f && S.p < 1.f && S.p2 >= 0.f && S.p2 < 1.f && S.p1x >= 0.f && S.p1x < 1.f && S.cp >= 0.f && S.cp < 1.f,
                "%s: task %d: dropout p", who, i);
    BwdTask& T = G.t[i];
    T.z = S.z; T.g_y = S.g_y; T.bn = bn_of(S.bn); T.z2 = S.z2;
    if (S.z2) {
      if (int rc = check_bn(who, S.bn2)) return rc;
      GPS_REQUIRE(al16(S.z2) && S.g_gamma2 && S.g_beta2 && al16(S.g_gamma2) && al16(S.g_beta2), "%s: task %d: dual buffers", who, i);
      GPS_REQUIRE(!S.relu && S.p == 0.f, "%s: task %d: a dual task carries no masks on its BatchNorm outputs", who, i);
      T.bn2 = bn_of(S.bn2);
    }
    T.g_beta = S.g_beta; T.g_gamma = S.g_gamma; T.g_beta2 = S.g_beta2; T.g_gamma2 = S.g_gamma2;
    T.g_z = S.g_z; T.g_sum = S.g_sum; T.g_drop = S.g_drop;
    T.p1x = S.z2 ? S.p1x : 0.f; T.seed1x = S.seed1x;
    T.R = S.R; T.seed = S.seed; T.seed2 = S.seed2; T.p = S.p; T.p2 = S.p2; T.relu = S.relu;
    T.rdev = S.rdev;
    T.rpb = rows_per_block(S.R); T.nblk = nblocks_for(S.R);
    T.block_begin = blocks;
    blocks += T.nblk;
  }
  return GPS_OK;
}

int gps_norm_bwd_partial(int n, const gps_norm_bwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                         gps_stream_t stream) {
  static const char* who = "gps_norm_bwd_partial";
  BwdGroup G{};
  int blocks = 0, n_trees = 0;
  if (int rc = fill_bwd(who, n, tasks, d, G, blocks)) return rc;
  GPS_REQUIRE(al16(ws), "%s: misaligned workspace", who);
  float* wp = ws;
  float* wend = ws + ws_floats;
  for (int i = 0; i < n; ++i) {
    BwdTask& T = G.t[i];
    const bool dual = T.z2 != nullptr;
    if (int rc = take_tree(who, T.tree, T.nblk, dual ? 3 : 2, tr::SUMS, d, wp, wend, sync, n_trees)) return rc;
    T.tree.o0 = T.g_beta; T.tree.o1 = T.g_gamma; T.tree.o2 = T.g_beta2; T.tree.o3 = T.g_gamma2;
  }
  k_bwd_partial<<<(unsigned)blocks, threads_for(d), lds_bytes_for(d, 3), gps::as_stream(stream)>>>(G);
  return gps::launch_status(who);
}

int gps_norm_bwd_apply(int n, const gps_norm_bwd_task* tasks, int d, float* ws, size_t ws_floats, uint32_t* sync,
                       gps_stream_t stream) {
  static const char* who = "gps_norm_bwd_apply";
  BwdGroup G{};
  int blocks = 0, n_trees = 0;
  if (int rc = fill_bwd(who, n, tasks, d, G, blocks)) return rc;
  float* wp = ws;
  float* wend = ws + ws_floats;
  for (int i = 0; i < n; ++i) {
    const gps_norm_bwd_task& S = tasks[i];
    BwdTask& T = G.t[i];
    GPS_REQUIRE(S.g_z && al16(S.g_z) && al16(S.g_sum) && al16(S.g_drop), "%s: task %d: null / misaligned output", who, i);
    T.amax_drop = S.g_drop ? S.amax_drop : nullptr;
    if (S.cz) {
      if (int rc = check_bn(who, S.cbn)) return rc;
      GPS_REQUIRE(al16(ws) && al16(S.cz) && S.cg_gamma && S.cg_beta && al16(S.cg_gamma) && al16(S.cg_beta),
                  "%s: task %d: chain buffers", who, i);
      T.cz = S.cz; T.cbn = bn_of(S.cbn); T.cseed = S.cseed; T.cp = S.cp; T.crelu = S.crelu; T.has_chain = 1;
      if (int rc = take_tree(who, T.tree, T.nblk, 2, tr::SUMS, d, wp, wend, sync, n_trees)) return rc;
      T.tree.o0 = S.cg_beta; T.tree.o1 = S.cg_gamma;
    }
  }
  k_bwd_apply<<<(unsigned)blocks, threads_for(d), lds_bytes_for(d, 2), gps::as_stream(stream)>>>(G);
  return gps::launch_status(who);
}

}  // extern "C"
