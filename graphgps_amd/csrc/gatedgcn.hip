// GatedGCN sparse core: gather -> gate -> segment-reduce -> node update, one launch forward, one launch backward.
//
// Reference semantics: graphgps/layer/gatedgcn_layer.py:67-70 (propagate), :90-107 (message),
// :109-126 (aggregate), :128-136 (update).  Index convention (PyG flow source_to_target):
// j = edge_index[0] (source), i = edge_index[1] (target); sums are keyed by the TARGET.
//
// Mapping to the machine (HBM-bound, no MFMA):
//   * lane = VEC consecutive channels of one node; a node row of d floats is d/VEC consecutive lanes, so every
//     access (node rows Ax/Bx/Dx/Ex, gathered source rows, Ce / e_hat edge rows) is a whole-row coalesced burst;
//   * a workgroup (768 threads) owns a CONTIGUOUS block of `nb` nodes and walks it `npi` = 768/(d/VEC) rows at a
//     time.  The CSR (and, backward, CSC) slices of that block -- contiguous index ranges -- are staged once in
//     LDS with coalesced loads, so the per-edge neighbour ids / edge ids never cost a dependent global round trip;
//   * blockIdx -> node-block map is XCD-aware: the dispatcher deals workgroup b to XCD b % 8, so logical block
//     (b % 8) * (grid / 8) + b / 8 gives every XCD one contiguous eighth of the node range.  Neighbours of a
//     molecule / AST node are a few rows away, i.e. in the SAME XCD's 4 MiB L2: the per-edge re-gathers of
//     Ex_j / Bx_j (round 1: 1.3x the algorithmic read bytes, each XCD fetching its own copy) become L2 hits;
//   * the reduction over a node's incoming edges runs inside the lane over its CSR segment, two edges in flight:
//     no cross-lane traffic, no atomics, summation in ascending original edge id = the order of the reference's
//     CPU scatter_add -> bitwise reproducible.
//
// Backward = ONE launch, two phases around a workgroup barrier:
//   A (target-keyed): num_i and den_i recomputed from the saved e_hat (the forward saves NOTHING besides its outputs),
//     a_i = g_x_i / D_i, b_i = -a_i num_i / D_i, delta_ij = g_e_ij + (a_i Bx_j + b_i) sig(1-sig) -> g_Ce (edge
//     order), g_Dx_i = sum_j delta_ij;
//   B (source-keyed): g_Ex_j = sum_{j->i} delta_ij, g_Bx_j = sum_{j->i} sig_ij a_i.  For an edge whose target lies
//     in this workgroup's node block (94 % on molecules at the default 32-node blocks) phase A's results come through
//     LDS -- delta and sig a_i per CSR slot where the block's slice fits (per-edge stash), else a_i per node row with
//     delta re-read from the g_Ce row this workgroup wrote a moment ago and sig from the e^ row it read (per-node stash,
//     ASTASH: the form the molecule and AST batches run); for the others everything is recomputed from its inputs
//     (aggr_i = x_tilde_i - Ax_i), so no workgroup ever waits for another.  e_hat / g_e / g_Ce cross the HBM interface
//     once each by the algorithm: 12*E*d + 28*N*d bytes (the per-node form's re-reads are L2 hits for the most part:
//     1.28 x by the counters).
//
// Algorithmic HBM bytes (fp32, per layer):  fwd 8*E*d + 20*N*d, bwd 12*E*d + 28*N*d (minus the 8*N*d of
// num / den that are recomputed instead of read); index traffic 4(N+1)+8E per CSR/CSC slice.
#include <algorithm>
#include <cstdlib>

#include "col_tree.hpp"
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

namespace tr = gps::tree;

constexpr int GG_T = 768;        // threads per workgroup (12 wavefronts; one workgroup per CU: 130 - 168 registers)
constexpr int GG_MAXNB = 512;    // node rows per workgroup, upper bound (LDS rowptr slice)
constexpr int GG_MAXE = 1536;    // staged CSR / CSC entries per workgroup; larger slices read the index from global

struct NodeBlock {
  int64_t n0, n1;
  __device__ bool valid() const { return n0 < n1; }
};

// XCD-aware logical block (grid is a multiple of 8).
__device__ __forceinline__ NodeBlock node_block(int64_t N, int nb) {
  const int per = gridDim.x >> 3;
  const int64_t L = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  NodeBlock r;
  r.n0 = L * nb;
  r.n1 = r.n0 + nb < N ? r.n0 + nb : N;
  return r;
}

// Stage rowptr[n0..n1] and, if it fits, idx_a/idx_b[e0..e1) into LDS.  Returns whether the index slice was staged.
__device__ __forceinline__ bool stage_slice(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ia,
                                            const int32_t* __restrict__ ib, const NodeBlock& nbk, int* s_rp,
                                            int* s_a, int* s_b) {
  const int cnt = (int)(nbk.n1 - nbk.n0);
  const int T = blockDim.x;
  for (int t = threadIdx.x; t <= cnt; t += T) s_rp[t] = rowptr[nbk.n0 + t];
  const int e0 = rowptr[nbk.n0], e1 = rowptr[nbk.n1];
  const bool staged = (e1 - e0) <= GG_MAXE;
  if (staged)
    for (int t = threadIdx.x; t < e1 - e0; t += T) {
      s_a[t] = ia[e0 + t];
      s_b[t] = ib[e0 + t];
    }
  return staged;
}

// Per-thread shifted sums of the rows this lane produces (its VEC channels): the batch statistics of x~ (bn_node_x) and
// e^ (bn_edge_e) -- graphgps/layer/gatedgcn_layer.py:72-73 -- fall out of the forward instead of a second pass over both
// tensors.  Shift = the lane's own first value, so the sums stay accurate whatever the column mean is.
template <int VEC>
struct StatAcc {
  Vec<VEC> kx, sx1, sx2, ke, se1, se2;
  float nx, ne;
  __device__ void init() {
    kx = sx1 = sx2 = ke = se1 = se2 = Vec<VEC>::zero();
    nx = ne = 0.f;
  }
  __device__ __forceinline__ void add_x(const Vec<VEC>& v) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      kx[i] = nx == 0.f ? v[i] : kx[i];
      const float t = v[i] - kx[i];
      sx1[i] += t;
      sx2[i] += t * t;
    }
    nx += 1.f;
  }
  __device__ __forceinline__ void add_e(const Vec<VEC>& v) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      ke[i] = ne == 0.f ? v[i] : ke[i];
      const float t = v[i] - ke[i];
      se1[i] += t;
      se2[i] += t * t;
    }
    ne += 1.f;
  }
};

// ---------------------------------------------------------------------------------------------------------
// Edge chunks.  D (1..4) edges of one node: every operand row is requested before the first one is used, so a node
// with <= 4 edges (molecules, ASTs) costs ONE memory round trip; longer segments are walked 4 edges at a time.
// D is a template parameter (switch on the degree) so that each body is straight-line code: with run-time
// predicates the compiler splits the loads over basic blocks and waits between them.
// GATE: the EquivStableLapPE variant (gatedgcn_layer.py:101-104): sigma_ij is multiplied by a per-edge
// scalar r_ij in (0,1) (r_edge[edge id]) before it gates and normalises.
// ---------------------------------------------------------------------------------------------------------
template <int VEC, bool GATE, bool STATS, int D>
__device__ __forceinline__ void fwd_chunk(const float* __restrict__ Bx, const float* __restrict__ Ex, int64_t ld,
                                          const float* __restrict__ Ce, const float* __restrict__ r_edge, int d,
                                          int c, const int* nbr, const int* eids, const Vec<VEC>& dx,
                                          Vec<VEC>& num, Vec<VEC>& den, float* __restrict__ e_hat, StatAcc<VEC>& sa,
                                          bool counted) {
  int64_t id[D];
  Vec<VEC> ex[D], bx[D], ce[D];
  float rr[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const int64_t j = nbr[u];
    id[u] = eids[u];
    ex[u] = Vec<VEC>::load(Ex + j * ld + c);
    bx[u] = Vec<VEC>::load(Bx + j * ld + c);
    ce[u] = Vec<VEC>::load(Ce + id[u] * d + c);
    rr[u] = GATE ? r_edge[id[u]] : 1.0f;
  }
#pragma unroll
  for (int u = 0; u < D; ++u) {
    Vec<VEC> eh;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      eh[v] = (dx[v] + ex[u][v]) + ce[u][v];  // e_ij = Dx_i + Ex_j + Ce            (:96)
      float s = sigmoidf_fast(eh[v]);        //                                     (:97)
      if (GATE) s = s * rr[u];                // sigma_ij * r_ij                     (:101-104)
      num[v] += s * bx[u][v];                 // scatter(sigma*Bx_j)                 (:117-119)
      den[v] += s;                            // scatter(sigma)                      (:121-123)
    }
    eh.store(e_hat + id[u] * d + c);          // self.e = e_ij, returned in edge order (:106,134)
    if (STATS && counted) sa.add_e(eh);
  }
}

template <int VEC, bool GATE, bool STATS>
__device__ __forceinline__ void fwd_rows(const float* __restrict__ Ax, const float* __restrict__ Bx,
                                         const float* __restrict__ Dx, const float* __restrict__ Ex, int64_t ld,
                                         const float* __restrict__ Ce, const int* rp, const int* nbr,
                                         const int* eids, const NodeBlock& blk, int d, int row, int npi, int c,
                                         float* __restrict__ x_tilde, float* __restrict__ e_hat,
                                         const float* __restrict__ r_edge, StatAcc<VEC>& sa, int64_t nreal) {
  for (int64_t node = blk.n0 + row; node < blk.n1; node += npi) {
    const int beg = rp[node - blk.n0], end = rp[node - blk.n0 + 1];
    // padded batches (loader.BucketPadding): padding nodes sit at the end and their edges join padding to padding only, so
    // "target node is real" gates the node's row AND its incoming edges out of the statistics
    const bool counted = node < nreal;
    const Vec<VEC> dx = Vec<VEC>::load(Dx + node * ld + c);
    const Vec<VEC> ax = Vec<VEC>::load(Ax + node * ld + c);
    Vec<VEC> num = Vec<VEC>::zero(), den = Vec<VEC>::zero();
    int k = beg;
    for (; k + 4 < end; k += 4)
      fwd_chunk<VEC, GATE, STATS, 4>(Bx, Ex, ld, Ce, r_edge, d, c, nbr + k, eids + k, dx, num, den, e_hat, sa, counted);
    switch (end - k) {
      case 1: fwd_chunk<VEC, GATE, STATS, 1>(Bx, Ex, ld, Ce, r_edge, d, c, nbr + k, eids + k, dx, num, den, e_hat, sa, counted); break;
      case 2: fwd_chunk<VEC, GATE, STATS, 2>(Bx, Ex, ld, Ce, r_edge, d, c, nbr + k, eids + k, dx, num, den, e_hat, sa, counted); break;
      case 3: fwd_chunk<VEC, GATE, STATS, 3>(Bx, Ex, ld, Ce, r_edge, d, c, nbr + k, eids + k, dx, num, den, e_hat, sa, counted); break;
      case 4: fwd_chunk<VEC, GATE, STATS, 4>(Bx, Ex, ld, Ce, r_edge, d, c, nbr + k, eids + k, dx, num, den, e_hat, sa, counted); break;
      default: break;
    }
    Vec<VEC> xt;
#pragma unroll
    for (int v = 0; v < VEC; ++v) xt[v] = ax[v] + num[v] / (den[v] + 1e-6f);  //          (:125,133)
    xt.store(x_tilde + node * (int64_t)d + c);
    if (STATS && counted) sa.add_x(xt);
  }
}

// Block-level finish of the statistics (STATS kernels): the npi row lanes of every column meet through LDS (merged in
// row order by Chan's formula) and lane row 0 writes the workgroup's RECORD -- (mean, M2) of x~ and of e^ with the two row
// counts -- with plain stores: the records are combined by the launch behind this one (k_gg_stats_finalize).  Round 3 - 5
// completed them in THIS launch through the two-level arrival tree (csrc/col_tree.hpp): its tail -- write-through stores,
// tickets, two dependent combine levels by single workgroups -- cost the HBM-bound kernel 21 us (26 -> 47), as much as the
// separate statistics pass over x~ and e^ it was meant to replace; a kernel boundary + a 96-workgroup combine costs ~6.
struct StatRecords {
  float* part;     // [P][4][d]: mean_x, M2_x, mean_e, M2_e of node block P
  float* cnt;      // [P][2]: real nodes, real edges of the block
};
template <int VEC>
__device__ __forceinline__ void fwd_stats_finish(const StatAcc<VEC>& sa, const StatRecords& T, int64_t L, bool active, int row,
                                                 int npi, int c, int d, float* lds) {
  float* cntx = lds;                     // [npi] nodes per row lane
  float* cnte = lds + npi;               // [npi] edges per row lane
  float* buf = lds + 2 * ((npi + 3) & ~3);   // [npi][2][d]
  Vec<VEC> m[2], q[2];
  float n[2] = {sa.nx, sa.ne};
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    m[0][v] = sa.nx > 0.f ? sa.kx[v] + sa.sx1[v] / sa.nx : 0.f;
    q[0][v] = sa.nx > 0.f ? fmaxf(sa.sx2[v] - sa.sx1[v] * sa.sx1[v] / sa.nx, 0.f) : 0.f;
    m[1][v] = sa.ne > 0.f ? sa.ke[v] + sa.se1[v] / sa.ne : 0.f;
    q[1][v] = sa.ne > 0.f ? fmaxf(sa.se2[v] - sa.se1[v] * sa.se1[v] / sa.ne, 0.f) : 0.f;
  }
  if (active && c == 0) { cntx[row] = sa.nx; cnte[row] = sa.ne; }
#pragma unroll
  for (int s = 0; s < 2; ++s) {          // x~ first, then e^, through the same LDS region
    if (s) __syncthreads();
    if (active && row > 0) {
      m[s].store(buf + (row * 2 + 0) * d + c);
      q[s].store(buf + (row * 2 + 1) * d + c);
    }
    __syncthreads();
    if (active && row == 0) {
      const float* cn = s ? cnte : cntx;
      for (int r2 = 1; r2 < npi; ++r2) {
        const Vec<VEC> m2 = Vec<VEC>::load(buf + (r2 * 2 + 0) * d + c), q2 = Vec<VEC>::load(buf + (r2 * 2 + 1) * d + c);
        const float n2 = cn[r2];
        float nn = n[s];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          nn = n[s];
          tr::chan_merge(nn, m[s][v], q[s][v], n2, m2[v], q2[v]);
        }
        n[s] = nn;
      }
      float* rec = T.part + (L * 4 + 2 * s) * d;
      m[s].store(rec + c);
      q[s].store(rec + d + c);
      if (c == 0) T.cnt[L * 2 + s] = n[s];
    }
  }
}

// The records of P node blocks -> the (mean, rstd) vectors of bn_node_x and bn_edge_e (+ their running statistics).
// A workgroup owns FC columns of ONE of the two statistics; its 256 threads are FC columns x FS record slices: a thread
// merges records sl, sl + FS, ... (every load issued before the first use: one memory round trip for up to 16 x FS
// records, more records repeat), the slices meet through LDS and are merged in slice order.  Combination by Chan's
// formula in a fixed order: deterministic, the accuracy of csrc/col_tree.hpp's STATS records.
constexpr int FC = 8, FS = 32;
struct StatOut {
  float *mean, *rstd, *rmean, *rvar;
  float eps, mom;
};
__global__ __launch_bounds__(FC * FS) void k_gg_stats_finalize(const StatRecords T, int P, int d, const StatOut ox,
                                                               const StatOut oe) {
  __shared__ float sm[FS][3][FC];
  const int per = d / FC;                         // workgroups per statistic (d % FC == 0: d % 8 == 0)
  const int s = blockIdx.x / per;                 // 0: x~, 1: e^
  const int cl = threadIdx.x % FC, sl = threadIdx.x / FC;
  const int c = (blockIdx.x - s * per) * FC + cl;
  float n = 0.f, m = 0.f, q = 0.f;
  for (int base = sl; base < P; base += 16 * FS) {
    float mv[16], qv[16], nv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int L = min(base + u * FS, P - 1);
      mv[u] = T.part[((int64_t)L * 4 + 2 * s) * d + c];
      qv[u] = T.part[((int64_t)L * 4 + 2 * s + 1) * d + c];
      nv[u] = T.cnt[L * 2 + s];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (base + u * FS < P) tr::chan_merge(n, m, q, nv[u], mv[u], qv[u]);
  }
  sm[sl][0][cl] = n; sm[sl][1][cl] = m; sm[sl][2][cl] = q;
  __syncthreads();
  if (sl == 0) {
    for (int r = 1; r < FS; ++r) tr::chan_merge(n, m, q, sm[r][0][cl], sm[r][1][cl], sm[r][2][cl]);
    const StatOut& o = s ? oe : ox;
    tr::stats_out(o.mean, o.rstd, o.rmean, o.rvar, o.eps, o.mom, c, m, q, n);
  }
}

template <int VEC, bool GATE, bool STATS>
__global__ __launch_bounds__(GG_T) void k_gatedgcn_fwd(
    const float* __restrict__ Ax, const float* __restrict__ Bx, const float* __restrict__ Dx,
    const float* __restrict__ Ex, int64_t ld, const float* __restrict__ Ce,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
    const int32_t* __restrict__ eid, int64_t N, int d, float* __restrict__ x_tilde,
    float* __restrict__ e_hat, const float* __restrict__ r_edge, int nb, int npi, const StatRecords stats,
    const int32_t* __restrict__ n_real) {
  __shared__ int s_rp[GG_MAXNB + 1];
  __shared__ int s_src[GG_MAXE], s_eid[GG_MAXE];
  extern __shared__ __attribute__((aligned(16))) float g_fwd_lds[];   // STATS: row-lane exchange + tree scratch
  const NodeBlock blk = node_block(N, nb);
  if (!blk.valid()) return;            // (records exist for the valid node blocks only)
  const int64_t nreal = STATS && n_real ? (int64_t)*n_real : N;
  const bool staged = stage_slice(rowptr, src, eid, blk, s_rp, s_src, s_eid);
  __syncthreads();
  const int lpr = d / VEC;
  const int row = threadIdx.x / lpr;
  const bool active = row < npi;
  if (!STATS && !active) return;
  const int c = (threadIdx.x - row * lpr) * VEC;
  const int e0 = s_rp[0];
  StatAcc<VEC> sa;
  if (STATS) sa.init();
  if (active) {
    if (staged)     // index slices in LDS (workgroup-uniform branch: two copies of the body, one address space each)
      fwd_rows<VEC, GATE, STATS>(Ax, Bx, Dx, Ex, ld, Ce, s_rp, s_src - e0, s_eid - e0, blk, d, row, npi, c, x_tilde, e_hat,
                                 r_edge, sa, nreal);
    else            // a hub-heavy block (> GG_MAXE entries): same code on the global index arrays
      fwd_rows<VEC, GATE, STATS>(Ax, Bx, Dx, Ex, ld, Ce, s_rp, src, eid, blk, d, row, npi, c, x_tilde, e_hat, r_edge, sa, nreal);
  }
  if (STATS) fwd_stats_finish<VEC>(sa, stats, blk.n0 / nb, active, row, npi, c, d, g_fwd_lds);
}

// ---- BatchNorm backward folded into the loads of the output gradients (round 6) -----------------------------------------
// The two gradients this kernel consumes are outputs of BatchNorm backward applies whose ONLY reader it is:
//   g_x~ = dBN_x(g_x1; x~)   (gatedgcn_layer.py:72,75-78: x = x_in + dropout(relu(bn_node_x(x~))))
//   g_e^ = dBN_e(g_e1; e^)   (gatedgcn_layer.py:73,76-79)
// and it already reads x~ and e^.  With FOLD the kernel takes g_x1 / g_e1 and evaluates, per element it loads,
//   g = relu'/dropout mask(row, col) * g_y ;  zhat = (z - mean) rstd ;  out = gate * gamma rstd (g - S1/n - zhat S2/n)
// -- the arithmetic of csrc/block_norm.hip k_bwd_apply, same association -- from six column vectors held in registers
// (S1 = sum g, S2 = sum g zhat: gps_norm_bwd_partial / the chain of an apply).  That removes the bn_node_x apply launch
// (35 MB) and the bn_edge_e half of another (71 MB) per layer and pass.
struct BnFold {
  const float *mean, *rstd, *gamma, *beta, *sum_g, *sum_gz;
  const int32_t* rdev;     // padded batches: number of real rows (device word), or nullptr
  int64_t R;
  uint64_t seed;
  float p;
  int relu;
};
__device__ __forceinline__ uint32_t gg_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Column vectors of one fold: mean, rstd, gamma, beta, S1, S2 over this lane's VEC channels
template <int VEC>
struct FoldVecs {
  Vec<VEC> mu, rs, ga, be, s1, s2;
  __device__ __forceinline__ void load(const BnFold& F, int c) {
    mu = Vec<VEC>::load(F.mean + c); rs = Vec<VEC>::load(F.rstd + c);
    ga = Vec<VEC>::load(F.gamma + c); be = Vec<VEC>::load(F.beta + c);
    s1 = Vec<VEC>::load(F.sum_g + c); s2 = Vec<VEC>::load(F.sum_gz + c);
  }
  __device__ __forceinline__ void load_lds(const float* sF, int d, int c) {     // [6][d] staged by the workgroup
    mu = Vec<VEC>::load(sF + c); rs = Vec<VEC>::load(sF + d + c);
    ga = Vec<VEC>::load(sF + 2 * d + c); be = Vec<VEC>::load(sF + 3 * d + c);
    s1 = Vec<VEC>::load(sF + 4 * d + c); s2 = Vec<VEC>::load(sF + 5 * d + c);
  }
};
// ... and its launch-uniform scalars (the ReLU between the BatchNorm and the dropout is GatedGCN's own: always on)
struct FoldScal {
  float inv_n, inv_keep, p;
  uint32_t seed_lo, seed_hi;
  int rreal;
  __device__ __forceinline__ void init(const BnFold& F, const uint64_t* salt) {
    rreal = (int)(F.rdev ? min((int64_t)*F.rdev, F.R) : F.R);
    inv_n = 1.0f / (float)rreal;
    p = F.p;
    inv_keep = F.p > 0.0f ? 1.0f / (1.0f - F.p) : 1.0f;
    const uint64_t seed = gps::salted_seed(F.seed, salt);
    seed_lo = (uint32_t)seed;
    seed_hi = (uint32_t)(seed >> 32);
  }
};
// identical to block_norm.hip row_hash / keep_elem / out_grad / apply_row
template <int VEC>
__device__ __forceinline__ Vec<VEC> fold_apply(const FoldScal& S, const FoldVecs<VEC>& K, const Vec<VEC>& gy, const Vec<VEC>& z,
                                               int64_t row, int c) {
  const bool drop = S.p > 0.0f;
  const uint32_t rh = drop ? gg_mix32((uint32_t)row ^ S.seed_lo) + S.seed_hi : 0u;
  const float gate = row < (int64_t)S.rreal ? 1.0f : 0.0f;
  Vec<VEC> o;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const float zh = (z[v] - K.mu[v]) * K.rs[v];
    float gg = gy[v];
    if (drop) {
      const uint32_t r = gg_mix32(rh + (uint32_t)(c + v) * 0x9E3779B9U);
      gg = (float)(r >> 8) * (1.0f / 16777216.0f) >= S.p ? gg * S.inv_keep : 0.0f;
    }
    gg = (zh * K.ga[v] + K.be[v]) > 0.0f ? gg : 0.0f;
    o[v] = gate * (K.ga[v] * K.rs[v] * (gg - K.s1[v] * S.inv_n - zh * K.s2[v] * S.inv_n));
  }
  return o;
}
// The pair of folds of one launch.  The edge fold runs in the inner loops: its vectors live in registers.  The node fold
// runs once per node (and per out-of-block target): its vectors are staged in LDS ([6][d] behind the stash) -- both sets in
// registers spilled at 768 threads per workgroup (168 VGPRs).  An empty stand-in keeps the un-folded instantiation as it was.
template <int VEC, int FOLD>
struct Folds {
  FoldScal es;
  const float* sF;
  int d;
  __device__ __forceinline__ Vec<VEC> edge(const Vec<VEC>& gy, const Vec<VEC>& eh, int64_t id, int c) const {
    FoldVecs<VEC> ev;
    ev.load_lds(sF + 6 * d, d, c);
    return fold_apply<VEC>(es, ev, gy, eh, id, c);
  }
  __device__ __forceinline__ Vec<VEC> node(const Vec<VEC>& gy, const Vec<VEC>& xt, int64_t n, int c) const {
    FoldVecs<VEC> xv;
    xv.load_lds(sF, d, c);
    // the node fold's scalars come back from LDS too (once per node): 25 scalar registers spilled with them resident
    const float* q = sF + 12 * d;
    FoldScal xs;
    xs.inv_n = q[0]; xs.inv_keep = q[1]; xs.p = q[2];
    xs.seed_lo = __float_as_uint(q[3]); xs.seed_hi = __float_as_uint(q[4]); xs.rreal = __float_as_int(q[5]);
    return fold_apply<VEC>(xs, xv, gy, xt, n, c);
  }
};
template <int VEC>
struct Folds<VEC, 0> {};

// Backward, one launch (see the header comment):
//   a_i = g_x_i / D_i,  b_i = -a_i * num_i / D_i                 (D_i = den_i + 1e-6, num_i recomputed)
//   delta_ij = g_e_ij + (a_i * Bx_j + b_i) * sig_ij * (1 - sig_ij)
//   g_Ce[eid] = delta_ij ;  g_Dx_i = sum_j delta_ij ;  g_Ax_i = g_x_i
//   g_Ex_j = sum_{j->i} delta_ij ;   g_Bx_j = sum_{j->i} sig_ij * a_i
// Phase A, a node with D <= 4 incoming edges in ONE pass: with t_ij = g_e_ij + a_i Bx_j s'_ij and
// s'_ij = r_ij sig (1 - sig) kept in registers, delta_ij = t_ij + b_i s'_ij once num_i (hence b_i) is known.
template <int VEC, bool GATE, int FOLD, int D>
__device__ __forceinline__ void bwd_a_chunk(const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                            const float* __restrict__ Bx, int64_t ld,
                                            const float* __restrict__ r_edge, int d, int c, const int* nbr,
                                            const int* eids, const Vec<VEC>& gx, Vec<VEC>& gdx, float* g_Ce,
                                            float* __restrict__ sD, float* __restrict__ sS, int slot0, int cap,
                                            float& mce, const Folds<VEC, FOLD>& FK, float* __restrict__ a_row) {
  int id[D];                  // (32-bit: widened at each use -- with the folds every register of this chunk counts)
  Vec<VEC> eh[D], ge[D], bx[D];
  float rr[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const int64_t j = nbr[u];
    id[u] = eids[u];
    eh[u] = Vec<VEC>::load(e_hat + (int64_t)id[u] * d + c);
    ge[u] = Vec<VEC>::load(g_e + (int64_t)id[u] * d + c);
    bx[u] = Vec<VEC>::load(Bx + j * ld + c);
    rr[u] = GATE ? r_edge[id[u]] : 1.0f;
  }
  if constexpr ((FOLD & 2) != 0) {
#pragma unroll
    for (int u = 0; u < D; ++u) ge[u] = FK.edge(ge[u], eh[u], id[u], c);
  }
  // num_i and den_i exactly as the forward summed them (same order, same gate arithmetic): nothing was saved
  Vec<VEC> num = Vec<VEC>::zero(), den = Vec<VEC>::zero();
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float s = sigmoidf_fast(eh[u][v]);
      const float sg = GATE ? s * rr[u] : s;
      num[v] += sg * bx[u][v];
      den[v] += sg;
      eh[u][v] = s;
    }
  Vec<VEC> a, b;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const float inv = 1.0f / (den[v] + 1e-6f);
    a[v] = gx[v] * inv;
    b[v] = -a[v] * (num[v] * inv);
  }
  if (a_row) a.store(a_row);         // (ASTASH: a_i for phase B)
#pragma unroll
  for (int u = 0; u < D; ++u) {
    Vec<VEC> dl, sa;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float s = eh[u][v];
      float sp = s * (1.0f - s);
      if (GATE) sp = sp * rr[u];
      dl[v] = (ge[u][v] + (a[v] * bx[u][v]) * sp) + b[v] * sp;
      gdx[v] += dl[v];
      sa[v] = (GATE ? s * rr[u] : s) * a[v];   // sig_ij a_i: what the source-keyed phase adds to g_Bx_j
      mce = fmaxf(mce, fabsf(dl[v]));          // max|g_Ce|: the record of the GEMMs that consume g_Ce (gps_hip.h)
    }
    dl.store(g_Ce + (int64_t)id[u] * d + c);
    if (slot0 + u < cap) {                    // hand-over to phase B through LDS (no second trip to memory)
      dl.store(sD + (slot0 + u) * d + c);
      sa.store(sS + (slot0 + u) * d + c);
    }
  }
}

// Long segments (more incoming edges than the one-pass form holds): the two passes in CHUNKS -- every operand row of a chunk
// is requested before the first one is used, so a node of degree 4 .. 8 costs 3 .. 5 memory round trips instead of one per
// edge and pass (8 .. 16: with the folds a degree-4 node -- 6 % of a molecule batch's nodes, a third of an AST batch's --
// already takes this path, and a workgroup's barrier waits for its slowest row).  Same values added in the same (ascending
// edge) order as the edge-by-edge loops they replace (the compiler's FMA contraction may differ: results agree to rounding).
template <int VEC, bool GATE, int D>
__device__ __forceinline__ void bwd_a_numden(const float* __restrict__ e_hat, const float* __restrict__ Bx, int64_t ld,
                                             const float* __restrict__ r_edge, int d, int c, const int* nbr, const int* eids,
                                             Vec<VEC>& num, Vec<VEC>& den) {
  Vec<VEC> eh[D], bx[D];
  float rr[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const int64_t j = nbr[u], id = eids[u];
    eh[u] = Vec<VEC>::load(e_hat + id * d + c);
    bx[u] = Vec<VEC>::load(Bx + j * ld + c);
    rr[u] = GATE ? r_edge[id] : 1.0f;
  }
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float s = sigmoidf_fast(eh[u][v]);
      if (GATE) s = s * rr[u];
      num[v] += s * bx[u][v];
      den[v] += s;
    }
}
template <int VEC, bool GATE, int FOLD, int D>
__device__ __forceinline__ void bwd_a_delta(const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                            const float* __restrict__ Bx, int64_t ld, const float* __restrict__ r_edge, int d,
                                            int c, const int* nbr, const int* eids, const Vec<VEC>& a, const Vec<VEC>& b,
                                            Vec<VEC>& gdx, float* g_Ce, float* __restrict__ sD, float* __restrict__ sS,
                                            int slot0, int cap, float& mce, const Folds<VEC, FOLD>& FK) {
  int id[D];
  Vec<VEC> eh[D], ge[D], bx[D];
  float rr[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const int64_t j = nbr[u];
    id[u] = eids[u];
    eh[u] = Vec<VEC>::load(e_hat + (int64_t)id[u] * d + c);
    ge[u] = Vec<VEC>::load(g_e + (int64_t)id[u] * d + c);
    bx[u] = Vec<VEC>::load(Bx + j * ld + c);
    rr[u] = GATE ? r_edge[id[u]] : 1.0f;
  }
  if constexpr ((FOLD & 2) != 0) {
#pragma unroll
    for (int u = 0; u < D; ++u) ge[u] = FK.edge(ge[u], eh[u], id[u], c);
  }
#pragma unroll
  for (int u = 0; u < D; ++u) {
    Vec<VEC> dl, sa;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float s = sigmoidf_fast(eh[u][v]);
      float sp = s * (1.0f - s);
      if (GATE) sp = sp * rr[u];
      dl[v] = (ge[u][v] + (a[v] * bx[u][v]) * sp) + b[v] * sp;   // same association as the one-pass form
      gdx[v] += dl[v];
      sa[v] = (GATE ? s * rr[u] : s) * a[v];
      mce = fmaxf(mce, fabsf(dl[v]));
    }
    dl.store(g_Ce + (int64_t)id[u] * d + c);
    if (slot0 + u < cap) {
      dl.store(sD + (slot0 + u) * d + c);
      sa.store(sS + (slot0 + u) * d + c);
    }
  }
}

template <int VEC, bool GATE, int FOLD, bool ASTASH>
__device__ __forceinline__ void bwd_a_rows(const float* __restrict__ g_x, int64_t ldgx,
                                           const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                           const float* __restrict__ Bx, int64_t ld, const int* rp, const int* nbr,
                                           const int* eids, const NodeBlock& blk, int d, int row, int npi, int c,
                                           float* g_Ce, float* __restrict__ g_Ax, float* __restrict__ g_Dx,
                                           int64_t ldg, const float* __restrict__ r_edge, float* __restrict__ sD,
                                           float* __restrict__ sS, int e0, int cap, float& mce, float& mnode,
                                           const float* __restrict__ x_tilde, const Folds<VEC, FOLD>& FK) {
#define GPS_BWD_A(DD) bwd_a_chunk<VEC, GATE, FOLD, DD>(g_e, e_hat, Bx, ld, r_edge, d, c, nbr + beg, eids + beg, gx, gdx, \
                                                       g_Ce, sD, sS, beg - e0, ASTASH ? 0 : cap, mce, FK, a_row)
  for (int64_t node = blk.n0 + row; node < blk.n1; node += npi) {
    const int beg = rp[node - blk.n0], end = rp[node - blk.n0 + 1];
    Vec<VEC> gx = Vec<VEC>::load(g_x + node * ldgx + c);
    if constexpr ((FOLD & 1) != 0) gx = FK.node(gx, Vec<VEC>::load(x_tilde + node * d + c), node, c);
    Vec<VEC> gdx = Vec<VEC>::zero();
    // ASTASH: a_i = g_x_i / D_i goes to phase B through LDS (cap = rows of that stash, sD = its base)
    float* a_row = ASTASH && node - blk.n0 < cap ? sD + (node - blk.n0) * d + c : nullptr;
    // (with both folds the one-pass form holds three edges: four -- 48 registers of rows on top of the folds' -- spill at the
    // 168 registers per lane of a 768-thread workgroup; a fourth incoming edge takes the two-pass form below)
    switch ((FOLD == 3 && end - beg == 4) ? 5 : end - beg) {
      case 0: break;
      case 1: GPS_BWD_A(1); break;
      case 2: GPS_BWD_A(2); break;
      case 3: GPS_BWD_A(3); break;
      case 4: if constexpr (FOLD != 3) GPS_BWD_A(4); break;
      default: {                               // long segment: num_i / den_i first, then the deltas (rows re-read from L1 / L2)
        Vec<VEC> num = Vec<VEC>::zero(), den = Vec<VEC>::zero();
        int k = beg;
#define GPS_ND(DD) bwd_a_numden<VEC, GATE, DD>(e_hat, Bx, ld, r_edge, d, c, nbr + k, eids + k, num, den)
        for (; k + 4 <= end; k += 4) GPS_ND(4);
        switch (end - k) {
          case 1: GPS_ND(1); break;
          case 2: GPS_ND(2); break;
          case 3: GPS_ND(3); break;
          default: break;
        }
#undef GPS_ND
        Vec<VEC> a, b;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float inv = 1.0f / (den[v] + 1e-6f);
          a[v] = gx[v] * inv;
          b[v] = -a[v] * (num[v] * inv);
        }
        if (a_row) a.store(a_row);
        k = beg;
#define GPS_DL(DD) bwd_a_delta<VEC, GATE, FOLD, DD>(g_e, e_hat, Bx, ld, r_edge, d, c, nbr + k, eids + k, a, b, gdx, g_Ce, sD, sS, \
                                                     k - e0, ASTASH ? 0 : cap, mce, FK)
        for (; k + 2 <= end; k += 2) GPS_DL(2);
        if (k < end) GPS_DL(1);
#undef GPS_DL
      }
    }
    if (g_Ax) gx.store(g_Ax + node * ldg + c);
    gdx.store(g_Dx + node * ldg + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) mnode = fmaxf(mnode, fmaxf(fabsf(gx[v]), fabsf(gdx[v])));
  }
#undef GPS_BWD_A
}

// D entries of a target's incoming segment: sum of the gates sig (e^) in entry order, every row requested before the first use
template <int VEC, bool GATE, int D>
__device__ __forceinline__ void bwd_b_den(const float* __restrict__ e_hat, const int32_t* __restrict__ eids,
                                          const float* __restrict__ r_edge, int d, int c, Vec<VEC>& dn) {
  int64_t id2[D];
#pragma unroll
  for (int q = 0; q < D; ++q) id2[q] = eids[q];
  Vec<VEC> e2[D];
  float r2[D];
#pragma unroll
  for (int q = 0; q < D; ++q) {
    e2[q] = Vec<VEC>::load(e_hat + id2[q] * d + c);
    r2[q] = GATE ? r_edge[id2[q]] : 1.0f;
  }
#pragma unroll
  for (int q = 0; q < D; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float s2 = sigmoidf_fast(e2[q][v]);
      dn[v] += GATE ? s2 * r2[q] : s2;
    }
}

// Phase B, D outgoing edges of one source node.  An edge whose target this workgroup owns finds its delta and
// sig a_i in the LDS stash phase A filled (slot = the edge's position in the block's CSR slice, looked up through the
// target's <= few-entry segment); only edges to other workgroups' nodes -- or beyond the stash -- go back to memory.
template <int VEC, bool GATE, int FOLD, int D>
__device__ __forceinline__ void bwd_b_chunk(const float* __restrict__ g_x, int64_t ldgx,
                                            const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                            const float* __restrict__ Ax, int64_t ld,
                                            const float* __restrict__ x_tilde,
                                            const int32_t* __restrict__ rowptr_g, const int32_t* __restrict__ eid_g,
                                            const float* __restrict__ r_edge, const float* g_Ce, int d, int c,
                                            const int* tgt, const int* eids, const NodeBlock& blk,
                                            const float* __restrict__ Bx, int64_t node, Vec<VEC>& gbx,
                                            Vec<VEC>& gex, const int* rp_d, const int* eids_d, int e0,
                                            const float* __restrict__ sD, const float* __restrict__ sS, int cap,
                                            const Folds<VEC, FOLD>& FK) {
  int64_t ti[D], id[D];
  int slot[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    ti[u] = tgt[u];
    id[u] = eids[u];
    slot[u] = -1;
    if (cap > 0 && ti[u] >= blk.n0 && ti[u] < blk.n1) {
      const int b0 = rp_d[ti[u] - blk.n0], b1 = rp_d[ti[u] - blk.n0 + 1];
      for (int kk = b0; kk < b1; ++kk)
        if (eids_d[kk] == (int)id[u] && kk - e0 < cap) slot[u] = kk - e0;
    }
  }
#pragma unroll
  for (int u = 0; u < D; ++u) {
    if (slot[u] >= 0) {                      // common case: LDS only
      const Vec<VEC> dl = Vec<VEC>::load(sD + slot[u] * d + c);
      const Vec<VEC> sa = Vec<VEC>::load(sS + slot[u] * d + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        gex[v] += dl[v];
        gbx[v] += sa[v];
      }
    } else {                                 // target owned by another workgroup (or stash overflow)
      // The target's segment (its den, as the forward summed it) is walked two entries at a time, both rows requested before
      // the first is used: a dependent memory round trip per PAIR of entries instead of two per entry.  ~10 - 15 % of a molecule
      // batch's edges come through here, and the rows of a workgroup wait for the slowest one at the end of the phase.
      const bool in = ti[u] >= blk.n0 && ti[u] < blk.n1;
      const int s0 = rowptr_g[ti[u]], s1 = rowptr_g[ti[u] + 1];
      const Vec<VEC> eh = Vec<VEC>::load(e_hat + id[u] * d + c);
      Vec<VEC> gx = Vec<VEC>::load(g_x + ti[u] * ldgx + c);
      if constexpr ((FOLD & 1) != 0) gx = FK.node(gx, Vec<VEC>::load(x_tilde + ti[u] * d + c), ti[u], c);
      // den of the target, as the forward summed it: walk the target's own incoming segment (global CSR)
      Vec<VEC> dn = Vec<VEC>::zero();
      int k2 = s0;
      for (; k2 + 2 <= s1; k2 += 2) bwd_b_den<VEC, GATE, 2>(e_hat, eid_g + k2, r_edge, d, c, dn);
      if (k2 < s1) bwd_b_den<VEC, GATE, 1>(e_hat, eid_g + k2, r_edge, d, c, dn);
      Vec<VEC> p0 = Vec<VEC>::load((in ? g_Ce : g_e) + id[u] * d + c);   // own delta row, or g_e to rebuild it
      if constexpr ((FOLD & 2) != 0) { if (!in) p0 = FK.edge(p0, eh, id[u], c); }
      const float rr = GATE ? r_edge[id[u]] : 1.0f;
      Vec<VEC> xt = Vec<VEC>::zero(), axi = Vec<VEC>::zero(), bxj = Vec<VEC>::zero();
      if (!in) {
        xt = Vec<VEC>::load(x_tilde + ti[u] * d + c);
        axi = Vec<VEC>::load(Ax + ti[u] * ld + c);
        bxj = Vec<VEC>::load(Bx + node * ld + c);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float ai = gx[v] * (1.0f / (dn[v] + 1e-6f));
        const float s = sigmoidf_fast(eh[v]);
        float dl = p0[v];
        if (!in) {
          float sp = s * (1.0f - s);
          if (GATE) sp = sp * rr;
          dl = (dl + (ai * bxj[v]) * sp) + (-ai * (xt[v] - axi[v])) * sp;   // b_i = -a_i aggr_i
        }
        gex[v] += dl;
        gbx[v] += (GATE ? s * rr : s) * ai;
      }
    }
  }
}

// Phase B with the a_i stash (ASTASH; blocks whose CSR slice is longer than the delta | sig a stash holds -- AST batches at
// d = 256: ~120 entries per block against 56 slots): an in-block target's a_i row waits in LDS (nb rows instead of two rows
// per EDGE), its delta is the g_Ce row this workgroup stored in phase A and sig comes again from the e^ row phase A read --
// two L2 / L1 hits requested together for all D edges, where the slot-less path below walks the target's segment for its
// den (>= 3 dependent round trips).  Same values, same order: sig a_i is rounded as a product before it is added, as the
// stash held it.  Out-of-block targets take the recomputing path of bwd_b_chunk one edge at a time.
template <int VEC, bool GATE, int FOLD, int D>
__device__ __forceinline__ void bwd_b_chunk_a(const float* __restrict__ g_x, int64_t ldgx,
                                              const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                              const float* __restrict__ Ax, int64_t ld,
                                              const float* __restrict__ x_tilde,
                                              const int32_t* __restrict__ rowptr_g, const int32_t* __restrict__ eid_g,
                                              const float* __restrict__ r_edge, const float* g_Ce, int d, int c,
                                              const int* tgt, const int* eids, const NodeBlock& blk,
                                              const float* __restrict__ Bx, int64_t node, Vec<VEC>& gbx,
                                              Vec<VEC>& gex, const int* rp_d, const int* eids_d, int e0,
                                              const float* __restrict__ sA, int arows, const Folds<VEC, FOLD>& FK) {
  int64_t ti[D], id[D];
  bool hit[D];
  Vec<VEC> dl[D], eh[D];
  float rr[D];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    ti[u] = tgt[u];
    id[u] = eids[u];
    hit[u] = ti[u] >= blk.n0 && ti[u] < blk.n1 && ti[u] - blk.n0 < arows;
    // (another workgroup's g_Ce row is never touched: a miss reads its e^ row twice, straight-line code either way)
    dl[u] = Vec<VEC>::load((hit[u] ? g_Ce : e_hat) + id[u] * d + c);
    eh[u] = Vec<VEC>::load(e_hat + id[u] * d + c);
    rr[u] = GATE ? r_edge[id[u]] : 1.0f;
  }
#pragma unroll
  for (int u = 0; u < D; ++u) {
    if (hit[u]) {
      const Vec<VEC> ai = Vec<VEC>::load(sA + (ti[u] - blk.n0) * d + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float s = sigmoidf_fast(eh[u][v]);
        gex[v] += dl[u][v];
        gbx[v] += __fmul_rn(GATE ? s * rr[u] : s, ai[v]);
      }
    } else {
      bwd_b_chunk<VEC, GATE, FOLD, 1>(g_x, ldgx, g_e, e_hat, Ax, ld, x_tilde, rowptr_g, eid_g, r_edge, g_Ce, d, c, tgt + u,
                                      eids + u, blk, Bx, node, gbx, gex, rp_d, eids_d, e0, nullptr, nullptr, 0, FK);
    }
  }
}

template <int VEC, bool GATE, int FOLD, bool ASTASH>
__device__ __forceinline__ void bwd_b_rows(const float* __restrict__ g_x, int64_t ldgx,
                                           const float* __restrict__ g_e, const float* __restrict__ e_hat,
                                           const float* __restrict__ Ax, const float* __restrict__ Bx, int64_t ld,
                                           const float* __restrict__ x_tilde,
                                           const int32_t* __restrict__ rowptr_g, const int32_t* __restrict__ eid_g,
                                           const int* rq, const int* tgt, const int* eids, const NodeBlock& blk,
                                           int d, int row, int npi, int c, const float* g_Ce,
                                           float* __restrict__ g_Bx, float* __restrict__ g_Ex, int64_t ldg,
                                           const float* __restrict__ r_edge, const int* rp_d, const int* eids_d,
                                           int e0, const float* __restrict__ sD, const float* __restrict__ sS,
                                           int cap, float& mnode, const Folds<VEC, FOLD>& FK) {
#define GPS_BWD_B(DD) bwd_b_chunk<VEC, GATE, FOLD, DD>(g_x, ldgx, g_e, e_hat, Ax, ld, x_tilde, rowptr_g, eid_g, r_edge, g_Ce, d, \
                                                       c, tgt + k, eids + k, blk, Bx, node, gbx, gex, rp_d, eids_d, e0, sD, \
                                                       sS, cap, FK)
#define GPS_BWD_BA(DD) bwd_b_chunk_a<VEC, GATE, FOLD, DD>(g_x, ldgx, g_e, e_hat, Ax, ld, x_tilde, rowptr_g, eid_g, r_edge, g_Ce, \
                                                          d, c, tgt + k, eids + k, blk, Bx, node, gbx, gex, rp_d, eids_d, e0, sD, \
                                                          cap, FK)
  for (int64_t node = blk.n0 + row; node < blk.n1; node += npi) {
    const int beg = rq[node - blk.n0], end = rq[node - blk.n0 + 1];
    Vec<VEC> gbx = Vec<VEC>::zero(), gex = Vec<VEC>::zero();
    int k = beg;
    if constexpr (ASTASH) {      // two edges at a time: four with the recomputing path inlined behind each spill with both folds
      for (; k + 2 <= end; k += 2) GPS_BWD_BA(2);
      if (k < end) GPS_BWD_BA(1);
    } else {
      for (; k + 4 < end; k += 4) GPS_BWD_B(4);
      switch (end - k) {
        case 1: GPS_BWD_B(1); break;
        case 2: GPS_BWD_B(2); break;
        case 3: GPS_BWD_B(3); break;
        case 4: GPS_BWD_B(4); break;
        default: break;
      }
    }
    gbx.store(g_Bx + node * ldg + c);
    gex.store(g_Ex + node * ldg + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) mnode = fmaxf(mnode, fmaxf(fabsf(gbx[v]), fabsf(gex[v])));
  }
#undef GPS_BWD_BA
#undef GPS_BWD_B
}

// FORM = FOLD | 4 * ASTASH (one template argument: the name the traces and bench.py's kernel tables match stays
// k_gatedgcn_bwd<VEC, GATE, n>)
template <int VEC, bool GATE, int FORM>
__global__ __launch_bounds__(GG_T) void k_gatedgcn_bwd(
    const float* __restrict__ g_x, int64_t ldgx, const float* __restrict__ g_e, const float* __restrict__ e_hat,
    const float* __restrict__ Ax, const float* __restrict__ Bx, int64_t ld, const float* __restrict__ x_tilde,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
    const int32_t* __restrict__ eid, const int32_t* __restrict__ rowptr_s, const int32_t* __restrict__ dst,
    const int32_t* __restrict__ eid_s, int64_t N, int d, float* g_Ce, float* __restrict__ g_Ax,
    float* __restrict__ g_Bx, float* __restrict__ g_Dx, float* __restrict__ g_Ex, int64_t ldg,
    const float* __restrict__ r_edge, int nb, int npi, int cap_arg, uint32_t* amax_node, uint32_t* amax_ce,
    const BnFold fold_x, const BnFold fold_e, const uint64_t* salt) {
  constexpr int FOLD = FORM & 3;
  constexpr bool ASTASH = (FORM & 4) != 0;
  __shared__ int s_rp[GG_MAXNB + 1], s_rq[GG_MAXNB + 1];
  __shared__ uint32_t s_amax[2][GG_T / 64];
  __shared__ int s_src[GG_MAXE], s_eid[GG_MAXE], s_dst[GG_MAXE], s_eid2[GG_MAXE];
  extern __shared__ __attribute__((aligned(16))) float g_stash[];   // [2][cap][d]: delta | sig a_i per CSR slot; ASTASH: [cap][d] a_i per node row
  const NodeBlock blk = node_block(N, nb);
  if (!blk.valid()) return;                 // whole workgroup leaves together
  const bool st_d = stage_slice(rowptr, src, eid, blk, s_rp, s_src, s_eid);
  const bool st_s = stage_slice(rowptr_s, dst, eid_s, blk, s_rq, s_dst, s_eid2);
  float* sD = g_stash;
  float* sS = g_stash + (int64_t)cap_arg * d;
  float* sF = g_stash + (ASTASH ? 1 : 2) * (int64_t)cap_arg * d;   // FOLD: [12][d] column vectors of the node fold | the edge fold
  if constexpr (FOLD != 0) {
    if constexpr ((FOLD & 1) != 0) {
      if (threadIdx.x == 0) {
        FoldScal xs;
        xs.init(fold_x, salt);
        float* q = sF + 12 * d;
        q[0] = xs.inv_n; q[1] = xs.inv_keep; q[2] = xs.p;
        q[3] = __uint_as_float(xs.seed_lo); q[4] = __uint_as_float(xs.seed_hi); q[5] = __int_as_float(xs.rreal);
      }
    }
    for (int t = threadIdx.x; t < d; t += blockDim.x) {
      if constexpr ((FOLD & 1) != 0) {
        sF[t] = fold_x.mean[t]; sF[d + t] = fold_x.rstd[t]; sF[2 * d + t] = fold_x.gamma[t];
        sF[3 * d + t] = fold_x.beta[t]; sF[4 * d + t] = fold_x.sum_g[t]; sF[5 * d + t] = fold_x.sum_gz[t];
      }
      if constexpr ((FOLD & 2) != 0) {
        sF[6 * d + t] = fold_e.mean[t]; sF[7 * d + t] = fold_e.rstd[t]; sF[8 * d + t] = fold_e.gamma[t];
        sF[9 * d + t] = fold_e.beta[t]; sF[10 * d + t] = fold_e.sum_g[t]; sF[11 * d + t] = fold_e.sum_gz[t];
      }
    }
  }
  __syncthreads();
  const int lpr = d / VEC;
  const int row = threadIdx.x / lpr;
  const bool active = row < npi;
  const int c = (threadIdx.x - row * lpr) * VEC;
  // ---- phase A: keyed by target ------------------------------------------------------------------
  const int e0 = s_rp[0];
  const int cap = (ASTASH || st_d) ? cap_arg : 0;   // the slot lookup of phase B walks the staged CSR slice (ASTASH: a_i rows)
  float mce = 0.0f, mnode = 0.0f;        // max|g_Ce|, max over the four node gradients: the records of their GEMMs
  Folds<VEC, FOLD> FK;
  if constexpr (FOLD != 0) {
    if constexpr ((FOLD & 2) != 0) FK.es.init(fold_e, salt);
    FK.sF = sF;
    FK.d = d;
  }
  if (active) {
    if (st_d)
      bwd_a_rows<VEC, GATE, FOLD, ASTASH>(g_x, ldgx, g_e, e_hat, Bx, ld, s_rp, s_src - e0, s_eid - e0, blk, d, row, npi, c,
                                          g_Ce, g_Ax, g_Dx, ldg, r_edge, sD, sS, e0, cap, mce, mnode, x_tilde, FK);
    else
      bwd_a_rows<VEC, GATE, FOLD, ASTASH>(g_x, ldgx, g_e, e_hat, Bx, ld, s_rp, src, eid, blk, d, row, npi, c, g_Ce, g_Ax,
                                          g_Dx, ldg, r_edge, sD, sS, e0, ASTASH ? cap : 0, mce, mnode, x_tilde, FK);
  }
  __threadfence_block();
  __syncthreads();            // this workgroup's g_Ce rows are visible to all of its lanes (same CU)
  // ---- phase B: keyed by source ------------------------------------------------------------------
  if (!active && !amax_node) return;
  if (active) {
    const int q0 = s_rq[0];
    if (st_s)
      bwd_b_rows<VEC, GATE, FOLD, ASTASH>(g_x, ldgx, g_e, e_hat, Ax, Bx, ld, x_tilde, rowptr, eid, s_rq, s_dst - q0,
                                          s_eid2 - q0, blk, d, row, npi, c, g_Ce, g_Bx, g_Ex, ldg, r_edge, s_rp, s_eid - e0, e0,
                                          sD, sS, cap, mnode, FK);
    else
      bwd_b_rows<VEC, GATE, FOLD, ASTASH>(g_x, ldgx, g_e, e_hat, Ax, Bx, ld, x_tilde, rowptr, eid, s_rq, dst, eid_s, blk, d,
                                          row, npi, c, g_Ce, g_Bx, g_Ex, ldg, r_edge, s_rp, s_eid - e0, e0, sD, sS, cap, mnode,
                                          FK);
  }
  if (!amax_node) return;                  // (kernel-uniform)
  // one atomic per record and workgroup: through LDS (a wave of the block may hold inactive rows' lanes only)
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t m0 = __float_as_uint(mnode), m1 = __float_as_uint(mce);
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    m0 = max(m0, (uint32_t)__shfl_xor((int)m0, o));
    m1 = max(m1, (uint32_t)__shfl_xor((int)m1, o));
  }
  if ((threadIdx.x & 63) == 0) { s_amax[0][wave] = m0; s_amax[1][wave] = m1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    uint32_t m = 0;
    for (int w = 0; w < nw; ++w) m = max(m, s_amax[threadIdx.x][w]);
    gps::amax_raise(threadIdx.x == 0 ? amax_node : amax_ce, m);
  }
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Launch shape.  Threads per workgroup (GPS_GG_FWD_THREADS / GPS_GG_THREADS) and the number of workgroups
// aimed at (GPS_GG_TARGET_WG; default = the device's CU count: both kernels hold one workgroup per CU, so that is ONE
// dispatch round of the largest node blocks that still fill the chip -- round 6, with the per-node a_i stash behind it:
// P30 x 256 at 32-node blocks 59 -> 53 us backward, 25 -> 23.5 forward, -0.14 ms per step against two rounds of 16-node
// blocks; profiles/r06_gatedgcn_astash.txt) are read once; node rows per workgroup = N / target rounded up to whole
// passes, never more than the LDS rowptr slice holds.
struct Plan { int threads, npi, nb; unsigned grid; };
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
inline Plan plan_for(int64_t N, int lanes_per_row, bool forward) {
  // GPS_GG_FWD_THREADS.  Measured at P30 x 256, d = 384 (us per launch):
  //                                         768 threads   512 threads
  //   isolated, rotating operands              26.3          24.6     (two 8-wave workgroups per CU cover each other)
  //   in the captured step, default schedule   26.0          31.7-34.9   (the forked attention kernel runs beside it)
  //   in the step, GPS_GEMM_MERGE=1            40-45         34
  // The step is what counts: 768.  backward: 768 (one workgroup per CU either way: 140 KB of LDS)
  static const int fwd_threads = env_int("GPS_GG_FWD_THREADS", GG_T);
  static const int bwd_threads = env_int("GPS_GG_THREADS", GG_T);
  static const int cfg_target = []() {
    int t = env_int("GPS_GG_TARGET_WG", 0), dev = 0;
    if (t <= 0 && (hipGetDevice(&dev) != hipSuccess ||
                   hipDeviceGetAttribute(&t, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || t <= 0)) t = 256;
    return t;
  }();
  const int cfg_threads = forward ? fwd_threads : bwd_threads;
  Plan p;
  p.threads = cfg_threads < 64 ? 64 : (cfg_threads > GG_T ? GG_T : (cfg_threads / 64) * 64);
  if (p.threads < lanes_per_row) p.threads = GG_T;
  p.npi = p.threads / lanes_per_row;
  int64_t nb = (N + cfg_target - 1) / cfg_target;
  nb = ((nb + p.npi - 1) / p.npi) * p.npi;
  const int cap = (GG_MAXNB / p.npi) * p.npi;
  if (nb > cap) nb = cap;
  if (nb < p.npi) nb = p.npi;
  p.nb = (int)nb;
  const int64_t blocks = (N + nb - 1) / nb;
  p.grid = (unsigned)(((blocks + 7) / 8) * 8);
  return p;
}

}  // namespace

#define GPS_GG_FWD(GATE, STATS, LDS)                                                                  \
  k_gatedgcn_fwd<VEC, GATE, STATS><<<pl.grid, pl.threads, LDS, s>>>(Ax, Bx, Dx, Ex, ld_node, Ce, rowptr_dst, \
      src_by_dst, eid_by_dst, N, d, x_tilde, e_hat, r_edge, pl.nb, pl.npi, recs, n_real)
#define GPS_GG_BWD(GATE, FOLD, ASTASH)                                                               \
  k_gatedgcn_bwd<VEC, GATE, (FOLD) | ((ASTASH) && VEC == 4 ? 4 : 0)><<<pl.grid, pl.threads, stash_bytes, s>>>(g_x, ld_gx, g_e, e_hat, Ax, Bx, ld_node, x_tilde, \
      rowptr_dst, src_by_dst, eid_by_dst, rowptr_src, dst_by_src, eid_by_src, N, d, g_Ce,             \
      g_Ax == g_x ? nullptr : g_Ax, g_Bx, g_Dx, g_Ex, ld_gnode, r_edge, pl.nb, pl.npi, cap, amax_node, amax_ce, fx, fe, \
      gps::dropout_salt())

extern "C" {

size_t gps_gatedgcn_stats_floats(int64_t N, int d) {
  if (N < 1 || d < 4 || d % 4) return 0;
  const Plan pl = plan_for(N, d / 4, true);
  const size_t P = (size_t)((N + pl.nb - 1) / pl.nb);
  return P * 4 * d + 2 * P + 16;
}

static int gatedgcn_fwd_impl(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                             int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                             const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                             int d, float* x_tilde, float* e_hat, const float* r_edge, const gps_bn* bn_x,
                             const gps_bn* bn_e, float* ws, size_t ws_floats, const int32_t* n_real, gps_stream_t stream);

int gps_gatedgcn_fwd(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                     int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                     const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                     int d, float* x_tilde, float* e_hat, const float* r_edge, gps_stream_t stream) {
  return gatedgcn_fwd_impl(Ax, Bx, Dx, Ex, ld_node, Ce, rowptr_dst, src_by_dst, eid_by_dst, N, E, d, x_tilde, e_hat, r_edge,
                           nullptr, nullptr, nullptr, 0, nullptr, stream);
}

int gps_gatedgcn_fwd_stats(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                           int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                           const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                           int d, float* x_tilde, float* e_hat, const float* r_edge, const gps_bn* bn_x,
                           const gps_bn* bn_e, float* ws, size_t ws_floats, const int32_t* n_real, gps_stream_t stream) {
  GPS_REQUIRE(bn_x && bn_e && bn_x->mean && bn_x->rstd && bn_e->mean && bn_e->rstd && ws,
              "gps_gatedgcn_fwd_stats: null statistics buffer");
  GPS_REQUIRE(N >= 2 && E >= 2 && d % 8 == 0, "gps_gatedgcn_fwd_stats: needs N, E >= 2 and d %% 8 == 0");
  GPS_REQUIRE((bn_x->running_mean == nullptr) == (bn_x->running_var == nullptr) &&
              (bn_e->running_mean == nullptr) == (bn_e->running_var == nullptr), "gps_gatedgcn_fwd_stats: running stats");
  return gatedgcn_fwd_impl(Ax, Bx, Dx, Ex, ld_node, Ce, rowptr_dst, src_by_dst, eid_by_dst, N, E, d, x_tilde, e_hat, r_edge,
                           bn_x, bn_e, ws, ws_floats, n_real, stream);
}

static int gatedgcn_fwd_impl(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                             int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                             const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                             int d, float* x_tilde, float* e_hat, const float* r_edge, const gps_bn* bn_x,
                             const gps_bn* bn_e, float* ws, size_t ws_floats, const int32_t* n_real, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_node >= d, "gps_gatedgcn_fwd: bad sizes N=%lld E=%lld d=%d ld=%lld",
              (long long)N, (long long)E, d, (long long)ld_node);
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(Ax && Bx && Dx && Ex && rowptr_dst && x_tilde, "gps_gatedgcn_fwd: null node buffer");
  GPS_REQUIRE(E == 0 || (Ce && src_by_dst && eid_by_dst && e_hat), "gps_gatedgcn_fwd: null edge buffer");
  auto ok = [&](size_t a) {
    return aligned_to(Ax, a) && aligned_to(Bx, a) && aligned_to(Dx, a) && aligned_to(Ex, a) &&
           aligned_to(Ce, a) && aligned_to(x_tilde, a) && aligned_to(e_hat, a);
  };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ld_node % 4 == 0 && ok(16), ld_node % 2 == 0 && ok(8), {
    GPS_REQUIRE(d / VEC <= GG_T, "gps_gatedgcn_fwd: d=%d too wide for one workgroup pass (%d lanes)", d, d / VEC);
    const Plan pl = plan_for(N, d / VEC, true);
    StatRecords recs{};
    if (bn_x) {
      GPS_REQUIRE(VEC == 4, "gps_gatedgcn_fwd_stats: rows must be 16-byte aligned (d %% 4 == 0, ld %% 4 == 0)");
      const int P = (int)((N + pl.nb - 1) / pl.nb);
      GPS_REQUIRE(aligned_to(ws, 16) && ws_floats >= (size_t)P * 4 * d + 2 * (size_t)P,
                  "gps_gatedgcn_fwd_stats: workspace (gps_gatedgcn_stats_floats)");
      recs.part = ws;
      recs.cnt = ws + (size_t)P * 4 * d;
      // dynamic LDS: [2 npi] counts + [npi][2][d] row-lane exchange
      const size_t lds = sizeof(float) * (2 * (size_t)((pl.npi + 3) & ~3) + (size_t)pl.npi * 2 * d) + 16;
      if (r_edge) GPS_GG_FWD(true, true, lds); else GPS_GG_FWD(false, true, lds);
      if (int rc = gps::launch_status("gps_gatedgcn_fwd_stats")) return rc;
      const StatOut ox{bn_x->mean, bn_x->rstd, bn_x->running_mean, bn_x->running_var, bn_x->eps, bn_x->momentum};
      const StatOut oe{bn_e->mean, bn_e->rstd, bn_e->running_mean, bn_e->running_var, bn_e->eps, bn_e->momentum};
      k_gg_stats_finalize<<<2 * (d / FC), FC * FS, 0, s>>>(recs, P, d, ox, oe);
    } else {
      if (r_edge) GPS_GG_FWD(true, false, 0); else GPS_GG_FWD(false, false, 0);
    }
  });
  return gps::launch_status("gps_gatedgcn_fwd");
}

static int fold_of(const char* who, const gps_bn_bwd_fold* f, int64_t rows, BnFold& out) {
  GPS_REQUIRE(f->bn && f->bn->mean && f->bn->rstd && f->bn->gamma && f->bn->beta && f->sum_g && f->sum_gz,
              "gps_gatedgcn_bwd_bn: %s: incomplete fold (BatchNorm vectors and both column sums)", who);
  GPS_REQUIRE(aligned_to(f->bn->mean, 16) && aligned_to(f->bn->rstd, 16) && aligned_to(f->bn->gamma, 16) &&
              aligned_to(f->bn->beta, 16) && aligned_to(f->sum_g, 16) && aligned_to(f->sum_gz, 16),
              "gps_gatedgcn_bwd_bn: %s: column vectors must be 16-byte aligned", who);
  GPS_REQUIRE(f->p >= 0.0f && f->p < 1.0f, "gps_gatedgcn_bwd_bn: %s: dropout p", who);
  GPS_REQUIRE(f->relu == 1 && rows < INT32_MAX, "gps_gatedgcn_bwd_bn: %s: the fold is BatchNorm -> ReLU -> dropout (relu = 1)", who);
  out = BnFold{f->bn->mean, f->bn->rstd, f->bn->gamma, f->bn->beta, f->sum_g, f->sum_gz, f->rdev, rows, f->seed, f->p, f->relu};
  return GPS_OK;
}

static int gatedgcn_bwd_impl(const float* g_x, int64_t ld_gx, const float* g_e, const float* e_hat, const float* Ax,
                             const float* Bx, int64_t ld_node, const float* x_tilde,
                             const int32_t* rowptr_dst, const int32_t* src_by_dst,
                             const int32_t* eid_by_dst, const int32_t* rowptr_src,
                             const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                             int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                             int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce,
                             const gps_bn_bwd_fold* fold_x, const gps_bn_bwd_fold* fold_e, gps_stream_t stream);

int gps_gatedgcn_bwd(const float* g_x, int64_t ld_gx, const float* g_e, const float* e_hat, const float* Ax,
                     const float* Bx, int64_t ld_node, const float* x_tilde,
                     const int32_t* rowptr_dst, const int32_t* src_by_dst,
                     const int32_t* eid_by_dst, const int32_t* rowptr_src,
                     const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                     int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                     int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce, gps_stream_t stream) {
  return gatedgcn_bwd_impl(g_x, ld_gx, g_e, e_hat, Ax, Bx, ld_node, x_tilde, rowptr_dst, src_by_dst, eid_by_dst, rowptr_src,
                           dst_by_src, eid_by_src, N, E, d, g_Ce, g_Ax, g_Bx, g_Dx, g_Ex, ld_gnode, r_edge, amax_node, amax_ce,
                           nullptr, nullptr, stream);
}

int gps_gatedgcn_bwd_bn(const float* g_x1, int64_t ld_gx, const float* g_e1, const float* e_hat, const float* Ax,
                        const float* Bx, int64_t ld_node, const float* x_tilde,
                        const int32_t* rowptr_dst, const int32_t* src_by_dst,
                        const int32_t* eid_by_dst, const int32_t* rowptr_src,
                        const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                        int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                        int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce,
                        const gps_bn_bwd_fold* fold_x, const gps_bn_bwd_fold* fold_e, gps_stream_t stream) {
  GPS_REQUIRE(!fold_x || g_Ax != g_x1, "gps_gatedgcn_bwd_bn: g_Ax receives the folded gradient and cannot alias g_x1");
  return gatedgcn_bwd_impl(g_x1, ld_gx, g_e1, e_hat, Ax, Bx, ld_node, x_tilde, rowptr_dst, src_by_dst, eid_by_dst, rowptr_src,
                           dst_by_src, eid_by_src, N, E, d, g_Ce, g_Ax, g_Bx, g_Dx, g_Ex, ld_gnode, r_edge, amax_node, amax_ce,
                           fold_x, fold_e, stream);
}

static int gatedgcn_bwd_impl(const float* g_x, int64_t ld_gx, const float* g_e, const float* e_hat, const float* Ax,
                             const float* Bx, int64_t ld_node, const float* x_tilde,
                             const int32_t* rowptr_dst, const int32_t* src_by_dst,
                             const int32_t* eid_by_dst, const int32_t* rowptr_src,
                             const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                             int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                             int64_t ld_gnode, const float* r_edge, uint32_t* amax_node, uint32_t* amax_ce,
                             const gps_bn_bwd_fold* fold_x, const gps_bn_bwd_fold* fold_e, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_node >= d && ld_gnode >= d && ld_gx >= d,
              "gps_gatedgcn_bwd: bad sizes");
  BnFold fx{}, fe{};
  const bool fold = fold_x != nullptr || fold_e != nullptr;
  GPS_REQUIRE(!fold || !r_edge, "gps_gatedgcn_bwd_bn: the folded form is not built with the per-edge gate r_edge");
  if (fold_x) { if (int rc = fold_of("nodes", fold_x, N, fx)) return rc; }
  if (fold_e) { if (int rc = fold_of("edges", fold_e, E, fe)) return rc; }
  GPS_REQUIRE((amax_node == nullptr) == (amax_ce == nullptr), "gps_gatedgcn_bwd: both max|.| records or neither");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(g_x && Ax && Bx && x_tilde && rowptr_dst && rowptr_src && g_Ax && g_Bx && g_Dx && g_Ex,
              "gps_gatedgcn_bwd: null node buffer");
  GPS_REQUIRE(E == 0 || (g_e && e_hat && src_by_dst && eid_by_dst && dst_by_src && eid_by_src && g_Ce),
              "gps_gatedgcn_bwd: null edge buffer");
  GPS_REQUIRE(g_Ax != g_x || ld_gx == ld_gnode, "gps_gatedgcn_bwd: g_Ax aliases g_x with a different stride");
  auto ok = [&](size_t a) {
    return aligned_to(g_x, a) && aligned_to(g_e, a) && aligned_to(e_hat, a) && aligned_to(Ax, a) &&
           aligned_to(Bx, a) && aligned_to(x_tilde, a) && aligned_to(g_Ce, a) &&
           aligned_to(g_Ax, a) && aligned_to(g_Bx, a) && aligned_to(g_Dx, a) && aligned_to(g_Ex, a);
  };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ld_node % 4 == 0 && ld_gnode % 4 == 0 && ld_gx % 4 == 0 && ok(16),
                   ld_node % 2 == 0 && ld_gnode % 2 == 0 && ld_gx % 2 == 0 && ok(8), {
    GPS_REQUIRE(d / VEC <= GG_T, "gps_gatedgcn_bwd: d=%d too wide for one workgroup pass (%d lanes)", d, d / VEC);
    Plan pl = plan_for(N, d / VEC, false);
    // LDS stash of phase A's per-edge results for phase B: [2][cap][d] floats next to the 29 KB of index slices
    static const int stash_kb = env_int("GPS_GG_STASH_KB", 112);
    const size_t fold_bytes = fold ? ((size_t)12 * d + 8) * sizeof(float) : 0;     // the folds' column vectors behind the stash
    int64_t stash_budget = (int64_t)stash_kb * 1024;
    if (fold && stash_budget + (int64_t)fold_bytes > 131072) stash_budget = 131072 - (int64_t)fold_bytes;   // 29 KB static + 128 KB
    GPS_REQUIRE(stash_budget >= 0, "gps_gatedgcn_bwd_bn: d=%d too wide for the folded form", d);
    int cap = (int)(stash_budget / (8LL * d));
    if (cap > GG_MAXE) cap = GG_MAXE;
    // A block's CSR slice is ~nb E / N entries.  Where the per-edge stash cannot hold it (AST batches at d = 256: ~120 entries
    // against 56 slots) the per-NODE stash takes over: a_i rows in LDS, delta and sig from the rows phase A touched
    // (k_gatedgcn_bwd<.., ASTASH>: every in-block edge on a two-load path instead of the segment walk).  16-byte rows only.
    static const int astash_cfg = env_int("GPS_GG_ASTASH", 1);       // 0 never | 1 where the per-edge stash overflows | 2 wherever it fits
    const bool want_astash = astash_cfg && VEC == 4 && (astash_cfg == 2 || (double)pl.nb * (double)E > (double)cap * (double)N);
    // a batch so large that one dispatch round's node blocks outgrow the a_i stash (50k nodes at d = 384: 196 rows against 73):
    // smaller blocks over more rounds keep every in-block edge off the recomputing path
    const int64_t rows_fit = (stash_budget / (4LL * d) / pl.npi) * pl.npi;
    if (want_astash && pl.nb > rows_fit && rows_fit >= pl.npi) {
      pl.nb = (int)rows_fit;
      pl.grid = (unsigned)((((N + pl.nb - 1) / pl.nb + 7) / 8) * 8);
    }
    const bool astash = want_astash && (int64_t)pl.nb * d * 4 <= stash_budget;
    if (astash) cap = pl.nb;
    const size_t stash_bytes = (size_t)cap * d * (astash ? 4 : 8) + fold_bytes;
    if (astash) {               // (VEC == 4 only: the macro maps the other widths onto the per-edge form)
      if (fold_x && fold_e) GPS_GG_BWD(false, 3, true);
      else if (fold_e) GPS_GG_BWD(false, 2, true);
      else if (fold_x) GPS_GG_BWD(false, 1, true);
      else { if (r_edge) GPS_GG_BWD(true, 0, true); else GPS_GG_BWD(false, 0, true); }
    } else {
      if (fold_x && fold_e) GPS_GG_BWD(false, 3, false);
      else if (fold_e) GPS_GG_BWD(false, 2, false);
      else if (fold_x) GPS_GG_BWD(false, 1, false);
      else { if (r_edge) GPS_GG_BWD(true, 0, false); else GPS_GG_BWD(false, 0, false); }
    }
  });
  return gps::launch_status("gps_gatedgcn_bwd");
}

}  // extern "C"
