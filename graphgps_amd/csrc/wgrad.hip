// Weight + bias gradient of a row-wise linear layer on fp32 MFMA, split-K, deterministic.
//
//     gW[m][n] = sum_r g[r][m] * x[r][n]        gb[m] = sum_r g[r][m]
//
// with r over N nodes or E edges (7.5k-15k at PCQM4M sizes) and a small [M, Nn] result
// (384..2688 x 384..768).  This is what autograd derives for the nn.Linear modules of the block
// (graphgps/layer/gatedgcn_layer.py:57-61, graphgps/layer/gps_layer.py:143-144,234-241); through
// rocBLAS/hipBLASLt these long-K / small-output GEMMs run at 45-85 TFLOP/s (profiles/), and the bias
// gradients were separate column-sum launches.
//
// Mapping: both operands are row-major with the contraction index r as the SLOW dimension, which is
// exactly the MFMA A/B register layout (lane l holds A[i = l&31][k = l>>5]): no transposes anywhere.
// Workgroup = 128 x 128 output tile, 4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32), 32-row chunks of
// g and x staged through double-buffered LDS (conflict-free row reads), next chunk's global loads in
// flight while the current one is multiplied.  The row range is split S ways so that tiles * S ~ one
// workgroup per CU; slices write partial tiles that a second kernel sums in slice order (no atomics).
// The bias gradient falls out of the staged g chunk for free.
//
// Grouped form (gps_wgrad_grouped): the five weight gradients of one GPS block (merged A|B|D|E|in_proj,
// C, out_proj, ff1, ff2) in ONE launch + ONE reduce launch.  Separately they cost ~390 us per block of
// which only ~250 us is MFMA time: each launch pays its own prologue (first chunk from HBM), partial-
// tile epilogue, tail and reduce pass, and the small ones split K 28 ways just to fill the chip.
// Grouped, every workgroup gets an equal share (~R*tiles/256 chunk-tiles) of the whole list, so slices
// are long (S = 2..4) and the partial traffic drops 4x.
#include "gps_common.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Exact 3-way split of 8 fp32 values into bf16 pieces (hi + mid + lo == v bit-for-bit: each piece
// takes the next 8 significant bits by truncation, the remainders are exact fp32 subtractions),
// packed as the bf16x8 operand of v_mfma_f32_32x32x16_bf16.  The product a*b is then formed from 6 of
// the 9 piece products -- hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll are below 2^-21 of |a||b| --
// with fp32 accumulation, on the bf16 pipe that runs 16x the fp32-input MFMA rate.
union Pack {
  uint32_t u[4];
  bf16x8 v;
};
__device__ __forceinline__ void split3(const float (&v)[8], Pack& hi, Pack& mid, Pack& lo) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = __float_as_uint(v[j]) & 0xFFFF0000u;
    const float r1 = v[j] - __uint_as_float(h[j]);
    m[j] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m[j]);
    l[j] = __float_as_uint(r2) & 0xFFFF0000u;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi.u[q] = (h[2 * q] >> 16) | h[2 * q + 1];
    mid.u[q] = (m[2 * q] >> 16) | m[2 * q + 1];
    lo.u[q] = (l[2 * q] >> 16) | l[2 * q + 1];
  }
}
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NLD = BK / 8;   // float4 loads per thread per operand per chunk (256 threads x 4 floats = 8 rows)
constexpr int kMaxGroup = 8;

struct Problem {
  const float* g;
  const float* x;
  float* part;        // [S][M][Nn] partial tiles
  float* bias_part;   // [S][M] partial column sums (nullptr: no bias gradient)
  float* gw;          // [M][Nn]
  float* gb;          // [M] or nullptr
  int64_t ldg, ldx, R;
  int M, Nn, tiles_m, tiles_n, S, rows_per_slice;
  int block_begin;    // first workgroup of this problem (grouped launch)
  int64_t out_begin;  // first reduce-thread index of this problem (grouped reduce)
};

struct Group {
  Problem p[kMaxGroup];
  int n;
};

// One (tile, slice) work item of problem P.  SPLIT: contraction on the bf16 pipe via the exact
// 3-way split above (default); otherwise on the fp32-input MFMA.
template <bool SPLIT>
__device__ __forceinline__ void wgrad_tile(const Problem& P, int tile, int slice, float (*As)[BK][BM],
                                           float (*Bs)[BK][BN]) {
  const float* __restrict__ g = P.g;
  const float* __restrict__ x = P.x;
  const int64_t ldg = P.ldg, ldx = P.ldx;
  const int M = P.M, Nn = P.Nn;
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int64_t r_begin = (int64_t)slice * P.rows_per_slice;
  const int64_t r_end = min(P.R, r_begin + P.rows_per_slice);
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: thread loads float4 (row = t/32 + 8*j, col4 = t%32), j = 0..NLD-1, for A and for B
  const int srow = t >> 5, scol = (t & 31) * 4;
  const bool a_col_ok = m0 + scol < M, b_col_ok = n0 + scol < Nn;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 ra[NLD], rb[NLD];
  // branch-free staging loads: out-of-range rows / column quads are clamped to a valid address and
  // zeroed by a select at store time (exec-masked branches around every load serialise their issue,
  // and a select placed right after the load would make the MFMAs wait for it)
  const float* ga = g + m0 + (a_col_ok ? scol : 0);
  const float* xb = x + n0 + (b_col_ok ? scol : 0);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_chunk = [&](int64_t r0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int64_t r = r0 + srow + 8 * j;
      const int64_t rc = r < r_end ? r : r_end - 1;
      ra[j] = *reinterpret_cast<const float4*>(ga + rc * ldg);
      rb[j] = *reinterpret_cast<const float4*>(xb + rc * ldx);
    }
  };
  auto store_chunk = [&](int buf, int64_t r0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const bool ok = r0 + srow + 8 * j < r_end;
      const float4 va = (ok && a_col_ok) ? ra[j] : zero4;
      const float4 vb = (ok && b_col_ok) ? rb[j] : zero4;
      *reinterpret_cast<float4*>(&As[buf][srow + 8 * j][scol]) = va;
      *reinterpret_cast<float4*>(&Bs[buf][srow + 8 * j][scol]) = vb;
      bsum.x += va.x; bsum.y += va.y; bsum.z += va.z; bsum.w += va.w;
    }
  };

  load_chunk(r_begin);
  store_chunk(0, r_begin);
  __syncthreads();
  int buf = 0;
  const int kh = lane >> 5, li = lane & 31;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += BK) {
    const bool more = r0 + BK < r_end;
    if (more) load_chunk(r0 + BK);     // global loads in flight during the MFMAs below
    if constexpr (SPLIT) {
      // lane (li, kh) owns rows 16*ks + 8*kh + j (j = 0..7) of the chunk for its column: any
      // assignment of rows to the instruction's 16 k-slots is valid as long as A and B use the same
      // one (a contraction index is order-free)
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const float* ap = &As[buf][16 * ks + 8 * kh][wm * 64 + li];
        const float* bp = &Bs[buf][16 * ks + 8 * kh][wn * 64 + li];
        float a0[8], a1[8], b0[8], b1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0[j] = ap[j * BM];
          a1[j] = ap[j * BM + 32];
          b0[j] = bp[j * BN];
          b1[j] = bp[j * BN + 32];
        }
        Pack A[2][3], B[2][3];
        split3(a0, A[0][0], A[0][1], A[0][2]);
        split3(a1, A[1][0], A[1][1], A[1][2]);
        split3(b0, B[0][0], B[0][1], B[0][2]);
        split3(b1, B[1][0], B[1][1], B[1][2]);
        // smallest terms first; the four accumulators rotate so no MFMA waits on its predecessor
        constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][TA[term]].v, B[j][TB[term]].v,
                                                                  acc[i][j], 0, 0, 0);
      }
    } else {
    // operands of step ks+1 are fetched from LDS into their own registers BEFORE the MFMAs of
      // step ks issue (sched_barrier: the scheduler sinks the prefetch next to its use otherwise)
      const float* ap = &As[buf][kh][wm * 64 + li];
      const float* bp = &Bs[buf][kh][wn * 64 + li];
      float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
  #pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        if (ks + 1 < BK / 2) {
          na0 = ap[(2 * ks + 2) * BM];
          na1 = ap[(2 * ks + 2) * BM + 32];
          nb0 = bp[(2 * ks + 2) * BN];
          nb1 = bp[(2 * ks + 2) * BN + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      }
    }
    if (more) store_chunk(buf ^ 1, r0 + BK);
    __syncthreads();
    buf ^= 1;
  }

  // partial tile: D[row = (q&3) + 8*(q>>2) + 4*(lane>>5)][col = lane&31]
  float* po = P.part + (int64_t)slice * M * Nn;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < Nn) po[(int64_t)row * Nn + col] = acc[i][j][q];
      }
    }
  if (P.bias_part && tn == 0) {
    // 8 threads (srow = 0..7) share a column quad: reduce through LDS in fixed order
    float* scratch = &As[0][0][0];   // all MFMA reads are done (barrier at loop end)
    *reinterpret_cast<float4*>(&scratch[srow * BM + scol]) = bsum;
    __syncthreads();
    if (t < BM && m0 + t < M) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += scratch[q * BM + t];
      P.bias_part[(int64_t)slice * M + m0 + t] = s;
    }
  }
}

// plain map: consecutive blocks = the tiles of one row slice.  (An XCD-grouped map -- all tiles
// of a slice on one XCD for L2 reuse -- was measured 25-50 % SLOWER here: with ~250 workgroups
// the uneven tiles-per-XCD split costs a second dispatch round on some XCDs.)
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_wgrad(const Group G) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < G.n && (int)blockIdx.x >= G.p[i].block_begin) pi = i;
  const Problem& P = G.p[pi];
  const int local = blockIdx.x - P.block_begin;
  const int tiles = P.tiles_m * P.tiles_n;
  const int slice = local / tiles;
  if (slice >= P.S) return;
  wgrad_tile<SPLIT>(P, local - slice * tiles, slice, As, Bs);
}

// out[i] = sum_s part[s][i] in slice order (one float per thread: enough threads to pull the
// S x M x Nn partials at bandwidth); bias likewise, by the first threads of each problem
__global__ __launch_bounds__(256) void k_wgrad_reduce(const Group G) {
  const int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < G.n && gi >= G.p[i].out_begin) pi = i;
  const Problem& P = G.p[pi];
  const int64_t i = gi - P.out_begin;
  const int64_t total = (int64_t)P.M * P.Nn;
  if (i < total) {
    float a = 0.f;
#pragma unroll 4
    for (int s = 0; s < P.S; ++s) a += P.part[(int64_t)s * total + i];
    P.gw[i] = a;
  }
  if (P.gb && i < P.M) {
    float a = 0.f;
#pragma unroll 4
    for (int s = 0; s < P.S; ++s) a += P.bias_part[(int64_t)s * P.M + i];
    P.gb[i] = a;
  }
}

// Slices for a problem so that one workgroup owns about `chunks_per_block` 32-row chunks.
inline void plan_slices(Problem& p, int64_t chunks_per_block) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Nn + BN - 1) / BN;
  const int64_t chunks = (p.R + BK - 1) / BK;
  int64_t S = (chunks + chunks_per_block - 1) / chunks_per_block;
  const int64_t max_s = (chunks + 3) / 4;            // at least 4 chunks per slice
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int64_t rps = (p.R + S - 1) / S;
  rps = (rps + BK - 1) / BK * BK;
  p.rows_per_slice = (int)rps;
  p.S = (int)((p.R + rps - 1) / rps);
}

// Smallest chunks-per-workgroup for which the whole list fits in ONE dispatch round of kTargetBlocks
// workgroups (one more would cost a second round).  One workgroup per CU: standalone, two per CU are
// faster (one's MFMAs fill the other's chunk-boundary bubble: 291 vs 335 us on the fp32-input path), but
// this kernel runs on the weight-gradient stream NEXT TO the main stream's backward kernels, and two
// workgroups per CU take every vector register of the chip (2 x 248 per lane), locking those kernels out
// until it drains; with one per CU the HBM-bound norm / attention / GatedGCN kernels co-run with it.
// Measured in the training step, same box: 13.60 ms (512 workgroups) vs 13.14-13.33 ms (256).
// GPS_WGRAD_TARGET_BLOCKS overrides (tuning).
static const int kTargetBlocks = [] { const char* e = getenv("GPS_WGRAD_TARGET_BLOCKS"); return e ? atoi(e) : 256; }();
inline int64_t balanced_chunks(const int64_t* R, const int* M, const int* Nn, int n) {
  int64_t work = 0, tiles_total = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t tiles = (int64_t)((M[i] + BM - 1) / BM) * ((Nn[i] + BN - 1) / BN);
    work += ((R[i] + BK - 1) / BK) * tiles;
    tiles_total += tiles;
  }
  int64_t cpb = std::max<int64_t>(4, (work + kTargetBlocks - 1) / kTargetBlocks);
  if (tiles_total >= kTargetBlocks) return INT64_MAX / 4;   // no split-K at all
  for (;; ++cpb) {
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
      const int64_t tiles = (int64_t)((M[i] + BM - 1) / BM) * ((Nn[i] + BN - 1) / BN);
      const int64_t chunks = (R[i] + BK - 1) / BK;
      int64_t S = std::min((chunks + cpb - 1) / cpb, (chunks + 3) / 4);
      if (S < 1) S = 1;
      const int64_t rps = ((R[i] + S - 1) / S + BK - 1) / BK * BK;
      blocks += tiles * ((R[i] + rps - 1) / rps);
    }
    if (blocks <= kTargetBlocks) return cpb;
  }
}

inline size_t problem_ws_floats(const Problem& p) {
  return ((size_t)p.S * ((size_t)p.M * p.Nn + p.M) + 3) / 4 * 4;   // keeps the next one 16-B aligned
}

int check_problem(const char* who, const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R,
                  int M, int Nn, const float* gw) {
  GPS_REQUIRE(R >= 1 && M > 0 && Nn > 0 && ldg >= M && ldx >= Nn, "%s: bad sizes", who);
  GPS_REQUIRE(g && x && gw, "%s: null buffer", who);
  GPS_REQUIRE(M % 4 == 0 && Nn % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(g) % 16 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                  (reinterpret_cast<uintptr_t>(gw) % 16 == 0),
              "%s: dimensions must be multiples of 4 and buffers 16-byte aligned", who);
  return GPS_OK;
}

int launch_group(Group& G, float* ws, hipStream_t s, const char* who) {
  int blocks = 0;
  int64_t outs = 0;
  for (int i = 0; i < G.n; ++i) {
    Problem& p = G.p[i];
    p.part = ws;
    float* bias_part = ws + (size_t)p.S * p.M * p.Nn;
    p.bias_part = p.gb ? bias_part : nullptr;
    ws += problem_ws_floats(p);
    p.block_begin = blocks;
    blocks += p.S * p.tiles_m * p.tiles_n;
    p.out_begin = outs;
    outs += ((int64_t)p.M * p.Nn + 255) / 256 * 256;   // whole reduce blocks per problem
  }
  // GPS_WGRAD_FP32_MFMA=1 keeps the contraction on the fp32-input MFMA (v_mfma_f32_32x32x2_f32)
  static const bool fp32_pipe = [] { const char* e = getenv("GPS_WGRAD_FP32_MFMA"); return e && atoi(e) != 0; }();
  if (fp32_pipe)
    k_wgrad<false><<<(unsigned)blocks, 256, 0, s>>>(G);
  else
    k_wgrad<true><<<(unsigned)blocks, 256, 0, s>>>(G);
  k_wgrad_reduce<<<gps::grid_for(outs, 256), 256, 0, s>>>(G);
  return gps::launch_status(who);
}

}  // namespace

extern "C" {

size_t gps_wgrad_workspace_floats(int64_t R, int M, int Nn) {
  if (R <= 0 || M <= 0 || Nn <= 0) return 0;
  Problem p{};
  p.R = R; p.M = M; p.Nn = Nn;
  plan_slices(p, balanced_chunks(&R, &M, &Nn, 1));
  return problem_ws_floats(p);
}

int gps_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int Nn,
              float* gw, float* gb, float* ws, gps_stream_t stream) {
  if (int rc = check_problem("gps_wgrad", g, ldg, x, ldx, R, M, Nn, gw)) return rc;
  GPS_REQUIRE(ws && reinterpret_cast<uintptr_t>(ws) % 16 == 0, "gps_wgrad: workspace null/misaligned");
  Group G{};
  G.n = 1;
  Problem& p = G.p[0];
  p.g = g; p.x = x; p.gw = gw; p.gb = gb; p.ldg = ldg; p.ldx = ldx; p.R = R; p.M = M; p.Nn = Nn;
  plan_slices(p, balanced_chunks(&R, &M, &Nn, 1));
  return launch_group(G, ws, gps::as_stream(stream), "gps_wgrad");
}

size_t gps_wgrad_grouped_workspace_floats(int n, const gps_wgrad_problem* probs) {
  if (n <= 0 || n > kMaxGroup || !probs) return 0;
  int64_t R[kMaxGroup];
  int M[kMaxGroup], Nn[kMaxGroup];
  for (int i = 0; i < n; ++i) { R[i] = probs[i].R; M[i] = probs[i].M; Nn[i] = probs[i].Nn; }
  const int64_t cpb = balanced_chunks(R, M, Nn, n);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (R[i] <= 0 || M[i] <= 0 || Nn[i] <= 0) return 0;
    Problem p{};
    p.R = R[i]; p.M = M[i]; p.Nn = Nn[i];
    plan_slices(p, cpb);
    total += problem_ws_floats(p);
  }
  return total;
}

int gps_wgrad_grouped(int n, const gps_wgrad_problem* probs, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(n >= 1 && n <= kMaxGroup && probs, "gps_wgrad_grouped: n=%d (1..%d)", n, kMaxGroup);
  GPS_REQUIRE(ws && reinterpret_cast<uintptr_t>(ws) % 16 == 0,
              "gps_wgrad_grouped: workspace null/misaligned");
  Group G{};
  G.n = n;
  int64_t R[kMaxGroup];
  int M[kMaxGroup], Nn[kMaxGroup];
  for (int i = 0; i < n; ++i) {
    const gps_wgrad_problem& q = probs[i];
    if (int rc = check_problem("gps_wgrad_grouped", q.g, q.ldg, q.x, q.ldx, q.R, q.M, q.Nn, q.gw))
      return rc;
    R[i] = q.R; M[i] = q.M; Nn[i] = q.Nn;
  }
  const int64_t cpb = balanced_chunks(R, M, Nn, n);
  for (int i = 0; i < n; ++i) {
    const gps_wgrad_problem& q = probs[i];
    Problem& p = G.p[i];
    p.g = q.g; p.x = q.x; p.gw = q.gw; p.gb = q.gb; p.ldg = q.ldg; p.ldx = q.ldx;
    p.R = q.R; p.M = q.M; p.Nn = q.Nn;
    plan_slices(p, cpb);
  }
  return launch_group(G, ws, gps::as_stream(stream), "gps_wgrad_grouped");
}

}  // extern "C"
