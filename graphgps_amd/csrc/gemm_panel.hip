// Row-panel GEMM for the dense side of the GPS block: C = A W^T (+ bias, + addend, ReLU/dropout epilogues), fp32 in,
// fp32 out, fp32-exact products on the bf16 MFMA pipe.
//
// What it replaces: the rocBLAS / hipBLASLt fp32 GEMMs behind the nn.Linear modules of the block
// (graphgps/layer/gatedgcn_layer.py:57-61 A..E, graphgps/layer/gps_layer.py:104-106 in/out-proj, :143-144,253-257
// FFN) and their input-gradient GEMMs.  Shapes at PCQM4M: M = 7.6k nodes or 15.3k edges, N and K in {384, 768, 2688}.
// Through the libraries these run at 50-119 TFLOP/s (35-76 % of the 157 TFLOP/s fp32-input MFMA peak): K is short
// (12-24 k-steps), so a 128-row tile spends as long in its prologue / epilogue as in its loop, and 7569 / 128 = 60
// row tiles x 3 column tiles leaves 30 % of the CUs idle.
//
// Arithmetic (same as csrc/wgrad.hip): every fp32 operand value is split EXACTLY into three bf16 pieces
// (hi + mid + lo == v bit for bit) and a*b is formed from 6 of the 9 piece products (hh, hm, mh, hl, lh, mm; the
// dropped ones are < 2^-21 |a||b|) with fp32 accumulation in v_mfma_f32_32x32x16_bf16.  Every partial product is
// exact in fp32, so the rounding model is that of an fp32-input MFMA GEMM (measured 7e-7 vs fp64), while the bf16
// pipe issues 16x the fp32-input rate: 6 MFMAs per fp32 one leave a 2.7x higher ceiling (417 TFLOP/s-equivalent).
//
// Data movement (the part round 1's gps_gemm_nt got wrong: it split A again in every one of up to 21 column tiles,
// in producer wavefronts whose VALU work did not hide, and quantised to 128 x 128 tiles):
//   * W is split ONCE per optimizer step by k_split_weights into a k-stage-major image Wp[piece][K/32][N][32] bf16
//     (both W and W^T, for forward and input-gradient), so a column panel's k-stage is ONE contiguous 12 KB block
//     per piece: global -> LDS is a straight 16-byte-per-lane copy, every L2 line used whole;
//   * a workgroup owns a 64-row x 192-column panel for the whole K: 7569 rows -> 119 row tiles, N / 192 = 2, 4 or 14
//     column panels -> 238 / 476 / 1666 workgroups = 0.93, 1.86, 6.5 rounds of 256 CUs (93 % of the chip busy where
//     128 x 128 tiles give 70 %); blocks are numbered panel-major, so the ~256 resident workgroups stream the same
//     one or two weight panels (0.4 MB each) out of L2 while A rows come from HBM once per panel;
//   * A (fp32) is loaded one 64 x 32 stage ahead into registers, split by the loading lane (8 values per lane per
//     stage: ~80 VALU instructions that interleave with the 36 MFMAs of the stage) and written to LDS as three bf16
//     tiles; stages are single-buffered in LDS (61 KB) so that TWO workgroups share a CU and cover each other's
//     staging / barrier phases;
//   * 4 wavefronts as 2 x 2, each 32 x 96 of the panel = 3 accumulator blocks: per 16-wide k-step 3 + 9 16-byte LDS
//     fragment reads (conflict-free at an 80-byte row pitch) feed 18 MFMAs (576 cycles) -- the loop is MFMA-bound.
// Epilogues (all optional, applied to the accumulators before the one store of C):
//   + bias[n]; + Cin[m][n] (residual / gradient accumulation, may alias C); ReLU + dropout(p, seed) keyed
//   (row, column) exactly as csrc/bn_fused.hip's act_drop (gps_layer.py:256); multiplication by the ReLU/dropout mask
//   of a saved activation `mask_src` (the FFN's backward, = gps_act_drop_bwd).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "col_tree.hpp"
#include "gps_common.hpp"

namespace {

namespace tr = gps::tree;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BK = 32;               // contraction elements per k-stage
constexpr int NTHREADS = 256;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// (row, column) dropout mask of the row kernels (csrc/bn_fused.hip: row_hash / keep_elem)
__device__ __forceinline__ uint32_t row_hash(uint32_t rowid, uint64_t seed) {
  return mix32(rowid ^ (uint32_t)seed) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ bool keep_elem(uint32_t rh, uint32_t col, float p_drop) {
  const uint32_t r = mix32(rh + col * 0x9E3779B9U);
  return (float)(r >> 8) * (1.0f / 16777216.0f) >= p_drop;
}

// exact 3-way split of one fp32 into bf16 bit patterns (in the high halves)
__device__ __forceinline__ void split1(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(v) & 0xFFFF0000u;
  const float r1 = v - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(m);
  l = __float_as_uint(r2) & 0xFFFF0000u;
}

// ------------------------------------------------------------------------------------------------------------
// weight images
// ------------------------------------------------------------------------------------------------------------
struct SplitDesc {
  const float* W;       // [rows][cols] fp32, row stride ldw
  int64_t ldw;
  int rows, cols;
  uint16_t* nt;         // image of B[n = row][k = col]  (forward: C = A W^T), or nullptr
  uint16_t* tn;         // image of B[n = col][k = row]  (input gradient: C = G W), or nullptr
  int block_begin;
  int nt_n, nt_k, tn_n, tn_k;   // padded image geometry (rows of the image, contraction length): see rg_npad / rg_kpad
};
constexpr int kMaxSplit = 56;      // 56 x 64-byte descriptors: the launch's kernel arguments stay under 4 KB
struct SplitGroup {
  SplitDesc d[kMaxSplit];
  int n;
};

// image index of element (n, k) of a B matrix with N rows and K columns, piece p
__device__ __forceinline__ int64_t img_index(int p, int n, int k, int N, int K) {
  return (((int64_t)p * (K / BK) + (k / BK)) * N + n) * BK + (k % BK);
}

// One workgroup = a 32-row x 128-column tile of W (one k-stage of the transposed image, four of the direct one).
// Rows are read coalesced and split once; the direct image takes 8-byte stores straight from registers, the
// transposed image goes through an LDS transpose so that its 64-byte (n, 32 k) runs leave as 16-byte stores
// (2-byte scattered stores made this kernel 31 us per layer; it runs once per layer and step).
constexpr int ST_R = 32, ST_C = 128, ST_P = ST_R + 8;      // LDS tile [3][128 columns][32 rows + pad] bf16
__global__ __launch_bounds__(256) void k_split_weights(const SplitGroup G) {
  __shared__ __attribute__((aligned(16))) uint16_t T[3][ST_C * ST_P];
  int di = 0, hi = G.n - 1;            // descriptor of this workgroup: the last one that begins at or before it
  while (di < hi) {
    const int mid = (di + hi + 1) >> 1;
    if ((int)blockIdx.x >= G.d[mid].block_begin) di = mid; else hi = mid - 1;
  }
  const SplitDesc& D = G.d[di];
  const int ctiles = (D.cols + ST_C - 1) / ST_C;
  const int tile = blockIdx.x - D.block_begin;
  const int r0 = (tile / ctiles) * ST_R, c0 = (tile % ctiles) * ST_C;
  const int t = threadIdx.x;
  // 32 rows x 32 float4 column groups = 1024 float4: 4 per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int r = idx >> 5, cq = (idx & 31) * 4;
    const int row = r0 + r, col = c0 + cq;
    const bool ok = row < D.rows && col < D.cols;
    const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(D.W + (int64_t)row * D.ldw + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split1(v[j], h[j], m[j], l[j]);
    // B[n = row][k = col..col+3]: 4 consecutive k of one row -> one 8-byte store per piece; columns between cols and the
    // padded contraction length get the zeros of the half-empty last stage
    if (D.nt && row < D.rows && col < D.nt_k) {
      *reinterpret_cast<u32x2*>(D.nt + img_index(0, row, col, D.nt_n, D.nt_k)) = (u32x2){(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
      *reinterpret_cast<u32x2*>(D.nt + img_index(1, row, col, D.nt_n, D.nt_k)) = (u32x2){(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
      *reinterpret_cast<u32x2*>(D.nt + img_index(2, row, col, D.nt_n, D.nt_k)) = (u32x2){(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
    }
    if (D.tn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T[0][(cq + j) * ST_P + r] = (uint16_t)(h[j] >> 16);
        T[1][(cq + j) * ST_P + r] = (uint16_t)(m[j] >> 16);
        T[2][(cq + j) * ST_P + r] = (uint16_t)(l[j] >> 16);
      }
    }
  }
  if (!D.tn) return;
  __syncthreads();
  // transposed image B[n = col][k = row]: this tile is stage r0 / 32, columns c0 .. c0 + 127, 64 bytes per (piece, n):
  // 3 pieces x 128 n x 4 chunks of 16 bytes = 1536 chunks, 6 per thread
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int idx = t + 256 * i;
    const int p = idx / (ST_C * 4), rem = idx - p * (ST_C * 4);
    const int n = rem >> 2, ch = rem & 3;
    if (c0 + n < D.cols && r0 + ch * 8 < D.tn_k) {       // (rows past D.rows were loaded as zeros: the k padding)
      const u32x4 v = *reinterpret_cast<const u32x4*>(&T[p][n * ST_P + ch * 8]);
      *reinterpret_cast<u32x4*>(D.tn + img_index(p, c0 + n, r0 + ch * 8, D.tn_n, D.tn_k)) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------------------
struct PanelArgs {
  const float* A;
  int64_t lda, M;
  int K, N;
  int Nimg;                // rows of the weight image (N rounded up to whole column panels: rg_npad)
  const uint16_t* Bp;      // weight image [3][ceil(K/32)][Nimg][32]
  const float* bias;       // [N] or nullptr
  const float* Cin;        // [M][N] addend (row stride ldcin) or nullptr; may alias C
  int64_t ldcin;
  float* C;
  int64_t ldc;
  int epilogue;            // 0 none, 1 relu + dropout(p, seed), 2 multiply by the relu/dropout mask of mask_src
  const float* mask_src;   // [M][N] (row stride ldmask): the saved activation whose mask is applied (epilogue 2)
  int64_t ldmask;
  float p_drop;
  uint64_t seed;
  const uint64_t* salt;
  int row_tiles;
  unsigned long long* trace;   // debugging aid (gps_gemm_panel_trace): 4 shader-clock stamps per workgroup, or nullptr
  // epilogue 3 (ring kernel only): C = Cin + dropout(A W^T + bias) and the column statistics of C (a training-mode
  // BatchNorm1d over the M rows), completed in-launch by one csrc/col_tree.hpp tree per 192-column panel whose level-0
  // records are the row tiles.  Tree of panel q: workspace st_ws + q * st_stride, counters st_tick + q * kSyncWords.
  float* st_ws;
  size_t st_stride;
  unsigned* st_tick;
  float *st_mean, *st_rstd, *st_rmean, *st_rvar;
  float st_eps, st_mom;
  // fp16 form (k_gemm_ring16): max|A| and max|B| as fp32 bit patterns (device words written by gps_absmax / a producer)
  const uint32_t* a_amax;
  const uint32_t* w_amax;
  uint32_t* c_amax;        // optional: raised to max|C| over the stored elements (the word of the NEXT GEMM that reads C)
  const int32_t* m_dev;    // epilogue 3, padded batches: the number of REAL rows (device word; rows past it are stored but kept
                           // out of the column statistics), or nullptr
  // epilogue 4 (k_gemm_ring16 only): C is the OUTPUT GRADIENT g of two BatchNorm1d's over the tensors cs_z / cs_z2 (same
  // shape as C: norm1_local + norm1_attn of a GPS block), and their backward column sums S1 = sum_rows g,
  // S2 = sum_rows g zhat (zhat = (z - mean) rstd) leave with it, completed in-launch by one SUMS tree per column panel
  // (st_ws / st_stride / st_tick as epilogue 3) into st_mean (S1), st_rstd (S2), st_rmean (S1 again), st_rvar (S2 of cs_z2).
  const float *cs_z, *cs_z2;
  int64_t cs_ldz, cs_ldz2;
  const float *cs_mu, *cs_rs, *cs_mu2, *cs_rs2;
};

// ------------------------------------------------------------------------------------------------------------
// Ring kernel: the same 64 x 192 panel and the same arithmetic, with the staging rebuilt around LDS-DMA.
//
// Round 2's first kernel (k_gemm_panel, removed in round 6: the ring serves every supported shape since round 3) staged
// through registers (global -> VGPR -> split -> ds_write) into ONE LDS stage behind two barriers; rocprofv3 PMC at the
// block's shapes: MFMA pipe 31 % busy (17 % at N = 384), 48 % of wave time in issue
// stalls, 32 % parked at waitcnt / barriers, 2.7 bank-conflict cycles per LDS instruction (the padded pitch is
// conflict-free for the fragment reads but 2-way for the staging writes).  Here:
//   * every global -> LDS byte moves by global_load_lds_dwordx4 (1 KB per wave-instruction, no VGPR, no ds_write) into
//     a THREE-slot ring: stage s+2 is issued while stage s is multiplied, a counted vmcnt(11) leaves it in flight
//     across the ONE barrier per stage (a plain __syncthreads would drain it);
//   * A travels as raw fp32 (64 rows x 128 B per stage); a wave reads its 32 x 16 fragment as two ds_read_b128 per
//     lane and splits it in registers right before the MFMAs (11 VALU per value pair, in the MFMAs' shadow): no
//     split-then-write pass, no second barrier; the two column-waves of a row block split the same values twice,
//     which costs VALU slots that were idle;
//   * LDS images are lane-linear (the DMA writes base + 16 * lane), so the bank swizzle sits on the per-lane SOURCE
//     address and, identically, on the fragment read: A chunk' = chunk ^ ((row >> 1) & 7) over the 8 chunks of a
//     128-byte row, W chunk' = chunk ^ ((row >> 2) & 3) over the 4 chunks of a 64-byte row -- each ds_read_b128 lane
//     group {0-3,12-15,20-27 | 4-11,16-19,28-31} then covers all sixteen 16-byte slots of the 256-byte bank row;
//   * fragments are double-buffered in registers one 16-wide k-step ahead, the barrier sits BETWEEN the two k-steps of
//     a stage, so every fragment read has 18 MFMAs (576 cycles) of cover and no MFMA waits on LDS latency.
// One workgroup per CU (132 KB of LDS), one wavefront per SIMD.
// ------------------------------------------------------------------------------------------------------------
// A column panel is 64 * NJ columns (NJ = 3: 192, the width the block's d = 384 projections tile with; NJ = 2: 128 for
// d = 256 -- ogbg-code2-GPS.yaml, pcqm4m-GPSdeep --; NJ = 1: 64 for d = 64 and the 7 * 64-wide merged projection).
constexpr int RG_SLOTS = 3;
constexpr int rg_bp(int nj) { return 64 * nj * BK * 2; }                       // one bf16 piece of the W stage: 4 KB per 64 columns
constexpr int rg_a_bytes(int mb) { return 64 * mb * BK * 4; }                  // raw fp32 A stage: 8 KB per 64 rows
constexpr int rg_slot_bytes(int mb, int nj) { return rg_a_bytes(mb) + 3 * rg_bp(nj); }   // NJ = 3: 44 KB (64 rows) / 52 KB (128 rows)
constexpr int rg_lds_bytes(int mb, int nj) { return RG_SLOTS * rg_slot_bytes(mb, nj); }  // NJ = 3: 132 KB / 156 KB
// Geometry of a [N, K] weight (N = output columns, K = contraction), both multiples of 4:
//   * k-stages of 32: KS = ceil(K / 32); when K % 32 != 0 the last stage is partly empty -- the image holds zeros there
//     and the A operand re-reads valid columns (EDGE kernels), so the padding contributes exact zeros;
//   * column panels of 64 * nj: the widest of 192 / 128 / 64 that divides N (192 only with k-stages in threes: its kernel
//     is the round-2 schedule, unchanged, without the left-over stages of the general rotation); when none divides N
//     (304 = 4.75 x 64, 48, 96: GPS-small and the narrow configs) the panel count is rounded up and the last panel's
//     surplus columns are computed and never stored (EDGE kernels), choosing nj by padded width / panel efficiency.
static inline int rg_ks(int64_t K) { return (int)((K + BK - 1) / BK); }
static inline int rg_nj(int64_t N, int64_t K) {
  const bool threes = rg_ks(K) % RG_SLOTS == 0;
  if (N % 192 == 0 && threes) return 3;
  if (K % BK == 0) {                       // whole stages: whole panels wherever a width divides N (the measured rule)
    if (N % 128 == 0) return 2;
    if (N % 64 == 0) return 1;
  }
  int best = 1;                            // EDGE kernels either way: the cheapest padded width
  double best_cost = 1e30;
  for (int nj = 1; nj <= 3; ++nj) {
    if (nj == 3 && !threes) continue;
    const double eff = nj == 3 ? 1.0 : (nj == 2 ? 0.9 : 0.7);
    const double cost = (double)((N + 64 * nj - 1) / (64 * nj) * (64 * nj)) / eff;
    if (cost < best_cost) { best_cost = cost; best = nj; }
  }
  return best;
}
static inline int64_t rg_npad(int64_t N, int64_t K) { const int tn = 64 * rg_nj(N, K); return (N + tn - 1) / tn * tn; }
static inline int64_t rg_kpad(int64_t K) { return (int64_t)rg_ks(K) * BK; }
static inline bool rg_edge(int64_t N, int64_t K) { return rg_npad(N, K) != N || rg_kpad(K) != K; }
// s_waitcnt immediate (gfx9 layout): vmcnt in bits 3:0 and 15:14, expcnt (left open) in 6:4, lgkmcnt in 11:8
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)l, 16, 0, 0);
}

template <int NJ>
struct RingFrag {
  bf16x8 b[NJ][3];     // W fragments of one 16-wide k-step: [column block][piece]
};
struct RingA {
  u32x4 p[3];          // the A fragment of one row block and k-step as bf16 pairs: hi, mid, lo
};

constexpr int kZeroBias = 8192;
__device__ float g_zero_bias[kZeroBias];      // stands in for a null bias, so that the epilogue's bias load is unconditional

// Epilogue of the ring kernel: MB row blocks of 32 per wave.  D[row = (q&3) + 8*(q>>2) + 4*kh][col = li] of each
// 32 x 32 block; 32 lanes = 128 contiguous bytes per row.  FULL (every row of the panel exists: all but the last row
// tile) is straight-line code: a store under a per-row branch made the compiler wait vmcnt(0) -- i.e. for every
// earlier STORE to retire -- before each of the 48 / 96 stores of a lane (measured: 10k / 23k cycles per workgroup,
// a third of its lifetime).
// EPI == 3 additionally accumulates, per lane and column block j, the shifted sums of the values it stores
// (sk = the wave's first row, s1 = sum (v - sk), s2 = sum (v - sk)^2 over the rows this lane owns).
// EDGE: a last column panel that reaches past N (its surplus columns are computed from whatever the image holds there
// and never stored): column indices of the loads are clamped, the stores predicated.
template <int MB, int NJ, int EPI, bool HAS_CIN, bool FULL, bool EDGE>
__device__ __forceinline__ void ring_store(const PanelArgs& P, const f32x16 (&acc)[MB][NJ], int64_t m0, int n0, int wm,
                                           int wn, int li, int kh, float (&sk)[NJ], float (&s1)[NJ], float (&s2)[NJ],
                                           float& amx) {
  const uint64_t seed = gps::salted_seed(P.seed, P.salt);
  const bool drop = EPI != 0 && P.p_drop > 0.0f;
  const float inv_keep = drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const int64_t mreal = (EPI == 3 && P.m_dev) ? min((int64_t)*P.m_dev, P.M) : P.M;
  float bv[NJ];
  float cmu[NJ], crs[NJ], cmu2[NJ], crs2[NJ];        // epilogue 4: the BatchNorms' column constants
#pragma unroll
  for (int j = 0; j < NJ; ++j) {         // never null here: the host passes g_zero_bias
    const int col = n0 + wn * (32 * NJ) + j * 32 + li;
    bv[j] = P.bias[EDGE ? min(col, P.N - 1) : col];
    if (EPI == 4) { cmu[j] = P.cs_mu[col]; crs[j] = P.cs_rs[col]; cmu2[j] = P.cs_mu2[col]; crs2[j] = P.cs_rs2[col]; }
  }
  // One 32-row block at a time: EVERY addend / mask value of the block is requested before the first store.  The addend
  // may alias C (in-place accumulation), so the compiler cannot move a load above an earlier store by itself, and one
  // load -> add -> store round trip per element cost 34-40 us per launch at the block's shapes; a lane only ever
  // re-reads elements it writes itself, so loading ahead is safe.
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float cin[NJ][16], msk[NJ][16], zz[NJ][16], zz2[NJ][16];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col0 = n0 + wn * (32 * NJ) + j * 32 + li;
      const int col = EDGE ? min(col0, P.N - 1) : col0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + (wm * MB + mb) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        const int64_t rc = FULL ? row : (row < P.M ? row : P.M - 1);   // clamped: loads unconditional, the store predicated
        if (HAS_CIN) cin[j][q] = __builtin_nontemporal_load(P.Cin + rc * P.ldcin + col);
        if (EPI == 2) msk[j][q] = __builtin_nontemporal_load(P.mask_src + rc * P.ldmask + col);
        if (EPI == 4) {                 // (both are read again by the apply behind this launch: cached loads)
          zz[j][q] = P.cs_z[rc * P.cs_ldz + col];
          zz2[j][q] = P.cs_z2[rc * P.cs_ldz2 + col];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn * (32 * NJ) + j * 32 + li;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + (wm * MB + mb) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        const int64_t rc = FULL ? row : (row < P.M ? row : P.M - 1);
        float v = acc[mb][j][q] + bv[j];
        if (HAS_CIN && EPI != 3) v += cin[j][q];
        if (EPI == 1) v = fmaxf(v, 0.0f);
        // EPI 2: the saved activation IS the mask -- t = dropout(relu(f1)) is positive exactly where the element was kept
        // and f1 > 0, so no hash is re-evaluated here (12 VALU per element, round 3), only the 1 / (1 - p) scaling
        if (EPI == 2) v = msk[j][q] > 0.0f ? v * inv_keep : 0.0f;
        if (EPI == 1 || EPI == 3) {
          const bool keep = !drop || keep_elem(row_hash((uint32_t)rc, seed), (uint32_t)col, P.p_drop);
          v = keep ? v * inv_keep : 0.0f;
        }
        if (EPI == 3) {                 // the residual is NOT dropped: C = Cin + dropout(product)
          if (HAS_CIN) v += cin[j][q];
          if (mb == 0 && q == 0) sk[j] = __shfl(v, li);      // row 0 of the wave's row range sits in lane li (kh = 0)
          const float t = row < mreal ? v - sk[j] : 0.0f;     // (mreal <= M: also the ragged last tile's guard)
          s1[j] += t;
          s2[j] += t * t;
        }
        if (EPI == 4) {                 // BatchNorm-backward column sums of the gradient this tile stores (sk = S1, s1 / s2 = S2)
          const float g = (FULL || row < P.M) ? v : 0.0f;
          sk[j] += g;
          s1[j] += g * ((zz[j][q] - cmu[j]) * crs[j]);
          s2[j] += g * ((zz2[j][q] - cmu2[j]) * crs2[j]);
        }
        const bool live = (FULL || row < P.M) && (!EDGE || col < P.N);
        amx = live ? fmaxf(amx, fabsf(v)) : amx;        // (one v_max with |.|: dead code wherever the caller drops amx)
        if (live) __builtin_nontemporal_store(v, P.C + row * P.ldc + col);
      }
    }
  }
}
template <int MB, int NJ, int EPI, bool HAS_CIN, bool EDGE>
__device__ __forceinline__ void ring_epilogue(const PanelArgs& P, const f32x16 (&acc)[MB][NJ], int64_t m0, int n0, int wm,
                                              int wn, int li, int kh, float (&sk)[NJ], float (&s1)[NJ], float (&s2)[NJ],
                                              float& amx) {
  // (all workgroup-uniform) only the last column panel of an EDGE shape can be partial; the others take the straight-line stores
  if (EDGE && n0 + 64 * NJ > P.N) ring_store<MB, NJ, EPI, HAS_CIN, false, true>(P, acc, m0, n0, wm, wn, li, kh, sk, s1, s2, amx);
  else if (m0 + 64 * MB <= P.M) ring_store<MB, NJ, EPI, HAS_CIN, true, false>(P, acc, m0, n0, wm, wn, li, kh, sk, s1, s2, amx);   // workgroup-uniform
  else ring_store<MB, NJ, EPI, HAS_CIN, false, false>(P, acc, m0, n0, wm, wn, li, kh, sk, s1, s2, amx);
}

// Column statistics of the panel (EPI == 3): the two row-waves of each column half meet through LDS (the ring is free
// by now), one thread per column merges them (Chan) and writes the workgroup's level-0 record write-through; the tree of
// this column panel (records = row tiles) then completes in-launch (csrc/col_tree.hpp).
template <int MB, int NJ, int NTH = 0>
__device__ __forceinline__ void ring_stats(const PanelArgs& P, int rt, int panel, int64_t m0, int wm, int wn, int li, int kh,
                                           const float (&sk)[NJ], float (&s1)[NJ], float (&s2)[NJ], float* lds) {
  constexpr int TN = 64 * NJ;                         // (shadows the 192-column constant of the register-staged kernel)
  constexpr int ROWS = 32 * MB;                       // rows per wave
  __syncthreads();      // every wave is past its final vmcnt(0): no LDS-DMA of the main loop can still land in the ring
  const int64_t mreal = P.m_dev ? min((int64_t)*P.m_dev, P.M) : P.M;       // padded batches: real rows only
  const int64_t left = mreal - (m0 + (int64_t)wm * ROWS);
  const float nw = left <= 0 ? 0.0f : (left < ROWS ? (float)left : (float)ROWS);
  float* rec = lds + 16;                              // [2 row-waves][2][TN]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s1[j] += __shfl_xor(s1[j], 32);
    s2[j] += __shfl_xor(s2[j], 32);
    if (kh == 0) {
      const int col = wn * (32 * NJ) + j * 32 + li;
      rec[(wm * 2 + 0) * TN + col] = nw > 0.0f ? sk[j] + s1[j] / nw : 0.0f;
      rec[(wm * 2 + 1) * TN + col] = nw > 0.0f ? fmaxf(s2[j] - s1[j] * s1[j] / nw, 0.0f) : 0.0f;
    }
  }
  __syncthreads();
  tr::Tree T{};
  T.P = P.row_tiles; T.NV = 2; T.mode = tr::STATS;
  T.fan = tr::fan_for(T.P);
  T.NG = (T.P + T.fan - 1) / T.fan;
  float* w = P.st_ws + (size_t)panel * P.st_stride;
  T.part = w; w += tr::pad4((size_t)T.P * 2 * TN);
  T.pcnt = w; w += tr::pad4((size_t)2 * T.P);
  T.grp = w; w += tr::pad4((size_t)T.NG * 2 * TN);
  T.gcnt = w;
  T.tick = P.st_tick + (size_t)panel * tr::kSyncWords;
  const int n0 = panel * TN;
  T.o0 = P.st_mean + n0; T.o1 = P.st_rstd + n0;
  T.o2 = P.st_rmean ? P.st_rmean + n0 : nullptr; T.o3 = P.st_rvar ? P.st_rvar + n0 : nullptr;
  T.eps = P.st_eps; T.momentum = P.st_mom;
  const int t = threadIdx.x;
  const int64_t l0 = mreal - m0;                      // (real) rows of this tile that exist; <= 0: a tile of padding only
  const float n0w = l0 <= 0 ? 0.0f : (l0 < ROWS ? (float)l0 : (float)ROWS);
  const float n1w = l0 <= ROWS ? 0.0f : (l0 < 2 * ROWS ? (float)(l0 - ROWS) : (float)ROWS);
  if (t < TN) {
    const float ma = rec[0 * TN + t], qa = rec[1 * TN + t], mb_ = rec[2 * TN + t], qb = rec[3 * TN + t];
    const float nn = n0w + n1w, dl = mb_ - ma, wgt = nn > 0.0f ? n1w / nn : 0.0f;
    tr::st_sc1(T.part + ((int64_t)rt * 2 + 0) * TN + t, ma + dl * wgt);
    tr::st_sc1(T.part + ((int64_t)rt * 2 + 1) * TN + t, qa + qb + dl * dl * n0w * wgt);
    if (t == 0) tr::st_sc1(T.pcnt + rt, nn);
  }
  tr::arrive<2, tr::STATS, 16, NTH>(T, rt, TN, lds);
}

// Column SUMS of the panel (EPI == 4): NV plain sums per column (S1, S2, S2 of the second BatchNorm); the two halves of
// a wave (kh) meet by a shuffle, the two row-waves through LDS, the tile's record goes out write-through and the panel's
// SUMS tree (records = row tiles) completes in-launch.  Rows past M contributed zeros (ring_store).
template <int MB, int NJ, int NV>
__device__ __forceinline__ void ring_sums(const PanelArgs& P, int rt, int panel, int wm, int wn, int li, int kh,
                                          float (&s0)[NJ], float (&s1)[NJ], float (&s2)[NJ], float* lds) {
  constexpr int TN = 64 * NJ;
  __syncthreads();      // every wave is past its final vmcnt(0): no LDS-DMA of the main loop can still land in the ring
  float* rec = lds + 16;                              // [2 row-waves][NV][TN]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s0[j] += __shfl_xor(s0[j], 32);
    s1[j] += __shfl_xor(s1[j], 32);
    if (NV > 2) s2[j] += __shfl_xor(s2[j], 32);
    if (kh == 0) {
      const int col = wn * (32 * NJ) + j * 32 + li;
      rec[(wm * NV + 0) * TN + col] = s0[j];
      rec[(wm * NV + 1) * TN + col] = s1[j];
      if (NV > 2) rec[(wm * NV + 2) * TN + col] = s2[j];
    }
  }
  __syncthreads();
  tr::Tree T{};
  T.P = P.row_tiles; T.NV = NV; T.mode = tr::SUMS;
  T.fan = tr::fan_for(T.P);
  T.NG = (T.P + T.fan - 1) / T.fan;
  float* w = P.st_ws + (size_t)panel * P.st_stride;
  T.part = w; w += tr::pad4((size_t)T.P * NV * TN);
  T.pcnt = w; w += tr::pad4((size_t)2 * T.P);
  T.grp = w; w += tr::pad4((size_t)T.NG * NV * TN);
  T.gcnt = w;
  T.tick = P.st_tick + (size_t)panel * tr::kSyncWords;
  const int n0 = panel * TN;
  T.o0 = P.st_mean + n0; T.o1 = P.st_rstd + n0;
  T.o2 = NV > 2 ? P.st_rmean + n0 : nullptr; T.o3 = NV > 2 ? P.st_rvar + n0 : nullptr;
  const int t = threadIdx.x;
  if (t < TN) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      tr::st_sc1(T.part + ((int64_t)rt * NV + v) * TN + t, rec[(0 * NV + v) * TN + t] + rec[(1 * NV + v) * TN + t]);
  }
  tr::arrive<NV, tr::SUMS, 16>(T, rt, TN, lds);
}

// MB = row blocks of 32 per wave, NJ = column blocks of 32 per wave: the workgroup's panel is (64 * MB) rows x (64 * NJ)
// columns, 2 x 2 waves of (32 * MB) x (32 * NJ).
// MB = 2 halves the W bytes (and the LDS fragment reads) per MFMA: measured with MB = 1 a long-K panel runs at ~2350
// cycles per stage against 1152 of MFMA issue, i.e. at ~19 bytes / cycle / CU through the global -> LDS path, the same
// per-CU rate the 256 x 256 bf16 reference kernels sustain -- the load path, not the matrix pipe, was the bound.
// NJ = 3 is the schedule tuned for the d = 384 block (below, unchanged since round 2); NJ = 2 / 1 deal the same staging
// over their fewer MFMA gaps by rule (the A split is amortised over fewer column blocks: 11 VALU per value pair against
// 6 NJ MFMAs, so NJ = 1 is VALU-bound -- it serves the small widths, where the launches are latency-bound anyway).
// Any number of k-stages >= 1: the static three-slot rotation runs in threes, the one or two stages left over reuse the
// first slots of the rotation; DMA past the last stage re-fetches the last one (never read).
// EDGE: N need not be a whole number of panels (ring_store) and K may end inside a stage -- the last stage's A transfers
// past K then re-read the stage's first chunk (finite values, multiplied by the zeros the image holds there).
template <int MB, int NJ, int EPI, bool HAS_CIN, bool EDGE = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_gemm_ring(const PanelArgs P) {
  constexpr int TNV = 64 * NJ;                  // panel columns
  constexpr int BP = rg_bp(NJ);                 // bytes of one W piece of a stage
  constexpr int A_BYTES = rg_a_bytes(MB), SLOT = rg_slot_bytes(MB, NJ);
  constexpr int NA = 2 * MB;                    // A transfers per wave and stage (8 rows x 128 B each)
  constexpr int NW = 3 * NJ;                    // W transfers per wave and stage (16 rows x 64 B each)
  constexpr int ND = NA + NW;                   // DMA transfers per wave and stage
  constexpr int ND_ODD = ND / 2, ND_EVEN = ND - ND_ODD;   // dealt over the two regions of a stage
  constexpr int G = 6 * MB * NJ;                // MFMAs (= issue gaps) per region
  // first gap of the A split (4 * MB pairs x 3 instalments) and instalments per gap
  constexpr int SPLIT0 = NJ == 3 ? G - 12 * MB - (MB > 1 ? 2 : 0) : (G >= 12 ? 3 : 1);
  constexpr int SPLITQ = NJ == 3 ? 1 : (12 * MB + (G - SPLIT0) - 1) / (G - SPLIT0);
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (P.trace && threadIdx.x == 0) P.trace[4 * (size_t)blockIdx.x + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const int panel = blockIdx.x / P.row_tiles, rt = blockIdx.x - panel * P.row_tiles;   // panel-major numbering
  const int64_t m0 = (int64_t)rt * (64 * MB);
  const int n0 = panel * TNV;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;
  const int KS = (P.K + BK - 1) / BK;

  // ---- DMA sources -------------------------------------------------------------------------------------------
  // A: wave w fills rows 16 MB w .. of the stage, NA instructions of 8 rows x 128 B; lane -> (row, LDS chunk position
  // lane & 7), which fetches source chunk pos ^ ((row >> 1) & 7).  Rows past M re-read row M-1 (never stored).
  const unsigned char* a_src[NA];
  int a_tail[NA];           // EDGE: byte offset to subtract in the last stage so that a chunk past K re-reads a valid one
  const int nvc = EDGE ? (P.K - (KS - 1) * BK) / 4 : 8;     // valid 16-byte chunks of the last stage (K % 4 == 0)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = 8 * NA * wave + 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const int64_t grow = min(m0 + row, P.M - 1);
    a_src[i] = reinterpret_cast<const unsigned char*>(P.A + grow * P.lda) + c * 16;
    a_tail[i] = (EDGE && c >= nvc) ? c * 16 : 0;             // -> chunk 0 of that stage (always valid)
  }
  // W: per piece 4 NJ blocks of 16 rows x 64 B; wave w moves blocks NJ w .. NJ w + NJ - 1 of every piece (NW instructions).
  // lane -> (row lane >> 2 of the block, position lane & 3) fetching chunk pos ^ ((row >> 2) & 3)
  const int64_t stage_stride = (int64_t)P.Nimg * (BK * 2);          // bytes between k-stages of one piece
  const int64_t piece_stride = stage_stride * KS;
  const unsigned char* b_src = reinterpret_cast<const unsigned char*>(P.Bp) + ((int64_t)n0 + 16 * NJ * wave + (lane >> 2)) * (BK * 2) +
                               (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  // DMA transfer g (0 .. ND-1) of stage s into `slot`: the first NA = A rows, then W piece i / NJ, block i % NJ
  auto dma = [&](int g, int s, unsigned char* slot) __attribute__((always_inline)) {
    if (g < NA) {
      if (EDGE) glds16(a_src[g] + (int64_t)s * (BK * 4) - (s == KS - 1 ? a_tail[g] : 0), slot + wave * (NA * 1024) + g * 1024);
      else glds16(a_src[g] + (int64_t)s * (BK * 4), slot + wave * (NA * 1024) + g * 1024);
    } else {
      const int i = g - NA;
      glds16(b_src + s * stage_stride + (i / NJ) * piece_stride + (i % NJ) * 1024,
             slot + A_BYTES + (i / NJ) * BP + wave * (NJ * 1024) + (i % NJ) * 1024);
    }
  };

  // ---- fragment addresses (byte offsets inside a slot) ---------------------------------------------------------
  int a_off[2][2], b_off[2];      // A: row block 0 (block mb is 32 rows = 4096 bytes further)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c0 = ks * 4 + kh * 2, sw = (li >> 1) & 7;
    a_off[ks][0] = (wm * 32 * MB + li) * 128 + ((c0 ^ sw) * 16);
    a_off[ks][1] = (wm * 32 * MB + li) * 128 + (((c0 + 1) ^ sw) * 16);
    b_off[ks] = A_BYTES + (wn * 32 * NJ + li) * 64 + (((ks * 2 + kh) ^ ((li >> 2) & 3)) * 16);
  }

  f32x16 acc[MB][NJ];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mb][j][q] = 0.0f;

  // exact 3-way split of one pair of fp32 values into bf16 pairs, in three instalments (4 + 4 + 3 VALU) so that it can
  // be dealt out over MFMA issue gaps; v_perm_b32 packs the two high halves
  float sv0, sv1, sr0, sr1;     // state carried between the instalments of a pair
  auto split_part = [&](int part, const f32x4 (&raw)[2], int d, RingA& out) __attribute__((always_inline)) {
    if (part == 0) {
      sv0 = raw[d >> 1][(d & 1) * 2];
      sv1 = raw[d >> 1][(d & 1) * 2 + 1];
      sr0 = sv0 - __uint_as_float(__float_as_uint(sv0) & 0xFFFF0000u);
      sr1 = sv1 - __uint_as_float(__float_as_uint(sv1) & 0xFFFF0000u);
    } else if (part == 1) {
      out.p[0][d] = __builtin_amdgcn_perm(__float_as_uint(sv1), __float_as_uint(sv0), 0x07060302u);
      out.p[1][d] = __builtin_amdgcn_perm(__float_as_uint(sr1), __float_as_uint(sr0), 0x07060302u);
      sv0 = sr0 - __uint_as_float(__float_as_uint(sr0) & 0xFFFF0000u);
      sv1 = sr1 - __uint_as_float(__float_as_uint(sr1) & 0xFFFF0000u);
    } else {
      out.p[2][d] = __builtin_amdgcn_perm(__float_as_uint(sv1), __float_as_uint(sv0), 0x07060302u);
    }
  };

  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};   // smallest terms first

  // One region = the G = 6 MB NJ MFMAs of k-step u (operands `ac`, `fc`, already in registers), with the staging of
  // k-step u+1 dealt into their issue gaps (at one wavefront per SIMD the matrix pipe takes an MFMA every 32 cycles,
  // which hides ~5 other instructions; a burst of loads or VALU between two MFMAs is idle pipe time):
  //   gaps 0 .. 3        the raw-A reads and the 3 NJ W-fragment reads of step u+1 (all of them early: the compiler waits
  //                      lgkmcnt(0) at the first use of a raw value, so every read should have landed by then)
  //   gaps 0 ..          the DMA transfers (dma_first .. dma_first + dma_count - 1 of stage dma_stage), one per gap
  //   gaps SPLIT0 ..     the split of the raw A values (4 MB pairs x 3 instalments, SPLITQ per gap)
  // sched_barrier(0) after every gap pins this order.  The MB NJ accumulators rotate, so no MFMA waits on its predecessor.
  f32x4 raw[MB][2];
  auto region = [&](const RingA (&ac)[MB], const RingFrag<NJ>& fc, RingA (&an)[MB], RingFrag<NJ>& fn, const unsigned char* rd_slot,
                    int rd_ks, int dma_stage, unsigned char* dma_slot, int dma_first, int dma_count)
                    __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int term = i / (NJ * MB), mb = (i / NJ) % MB, j = i % NJ;
      acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[mb].p[TA[term]]), fc.b[j][TB[term]],
                                                          acc[mb][j], 0, 0, 0);
      if (i == 0) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          raw[b][0] = *reinterpret_cast<const f32x4*>(rd_slot + a_off[rd_ks][0] + b * 4096);
          raw[b][1] = *reinterpret_cast<const f32x4*>(rd_slot + a_off[rd_ks][1] + b * 4096);
        }
      }
      if (i < 4) {
        // W-fragment reads per gap -- NJ = 3: 1, 3, 3, 2;  NJ = 2: 1, 2, 2, 1;  NJ = 1: 1, 1, 1, 0
        constexpr int first[5] = {0, 1, NJ == 3 ? 4 : (NJ == 2 ? 3 : 2), NJ == 3 ? 7 : (NJ == 2 ? 5 : 3), NW};
#pragma unroll
        for (int k = first[i]; k < first[i + 1]; ++k)
          fn.b[k / 3][k % 3] = *reinterpret_cast<const bf16x8*>(rd_slot + b_off[rd_ks] + (k / 3) * (32 * 64) + (k % 3) * BP);
      }
      if (i < dma_count) dma(dma_first + i, dma_stage, dma_slot);
      if (i >= SPLIT0) {
#pragma unroll
        for (int u = 0; u < SPLITQ; ++u) {
          const int q = (i - SPLIT0) * SPLITQ + u;       // pair q / 3 (row block (q / 3) / 4, dword (q / 3) % 4), instalment q % 3
          if (q < 12 * MB) split_part(q % 3, raw[(q / 3) / 4], (q / 3) % 4, an[(q / 3) / 4]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(ND_EVEN <= G && 12 * MB <= (G - SPLIT0) * SPLITQ, "the staging does not fit the region's issue gaps");

  unsigned char* const slot0 = ring;
  unsigned char* const slot1 = ring + SLOT;
  unsigned char* const slot2 = ring + 2 * SLOT;
  // prologue: stages 0 and 1 whole, the first ND_ODD transfers of stage 2 (stage indices clamped to the last one)
#pragma unroll
  for (int g = 0; g < ND; ++g) dma(g, 0, slot0);
#pragma unroll
  for (int g = 0; g < ND; ++g) dma(g, min(1, KS - 1), slot1);
#pragma unroll
  for (int g = 0; g < ND_ODD; ++g) dma(g, min(2, KS - 1), slot2);
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(ND + ND_ODD, 15));   // this wave's share of stage 0 has landed
  __builtin_amdgcn_s_barrier();                               // ... and every other wave's
  stamp(1);
  RingFrag<NJ> f0, f1;
  RingA a0[MB], a1[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    raw[b][0] = *reinterpret_cast<const f32x4*>(slot0 + a_off[0][0] + b * 4096);
    raw[b][1] = *reinterpret_cast<const f32x4*>(slot0 + a_off[0][1] + b * 4096);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      f0.b[j][p] = *reinterpret_cast<const bf16x8*>(slot0 + b_off[0] + j * (32 * 64) + p * BP);
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int part = 0; part < 3; ++part) split_part(part, raw[b], d, a0[b]);
  __builtin_amdgcn_sched_barrier(0);

  // Stage s (slot CUR) = regions 2s and 2s+1.  Region 2s multiplies k-step 2s while k-step 2s+1 is fetched from CUR and
  // the last ND_EVEN transfers of stage s+2 are issued; then the stage barrier: the counted wait retires this wave's
  // share of stage s+1 (the ND transfers of stage s+2 stay in flight across it), lgkmcnt(0) retires its reads of CUR,
  // so after the barrier CUR may be refilled.  Region 2s+1 multiplies k-step 2s+1, fetches k-step 2s+2 from NXT and
  // issues the first ND_ODD transfers of stage s+3 into CUR.  Past the last stage the DMA re-fetches stage KS-1 into the
  // free slot, which keeps the counted wait uniform.  (The wait is the builtin, not inline asm, so that the compiler's
  // own counter bookkeeping sees it.)
#define GPS_RING_STAGE(S, CUR, NXT, NX2)                                                       \
  region(a0, f0, a1, f1, CUR, 1, min((S) + 2, KS - 1), NX2, ND_ODD, ND_EVEN);                  \
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(ND, 0));                                              \
  __builtin_amdgcn_s_barrier();                                                                \
  __builtin_amdgcn_sched_barrier(0);                                                           \
  region(a1, f1, a0, f0, NXT, 0, min((S) + 3, KS - 1), CUR, 0, ND_ODD);
  int s0 = 0;
  for (; s0 + 3 <= KS; s0 += 3) {                         // the slots of the rotation are static
    GPS_RING_STAGE(s0, slot0, slot1, slot2)
    GPS_RING_STAGE(s0 + 1, slot1, slot2, slot0)
    GPS_RING_STAGE(s0 + 2, slot2, slot0, slot1)
  }
  if constexpr (NJ != 3) {                                // (NJ = 3 is only dispatched with stages in threes: rg_nj)
    if (s0 < KS) {                                        // one or two stages left over (workgroup-uniform)
      GPS_RING_STAGE(s0, slot0, slot1, slot2)
      if (s0 + 1 < KS) { GPS_RING_STAGE(s0 + 1, slot1, slot2, slot0) }
    }
  }
#undef GPS_RING_STAGE
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));         // no DMA may still be writing this workgroup's LDS when it retires
  stamp(2);
  float sk[NJ], s1[NJ], s2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sk[j] = s1[j] = s2[j] = 0.f;
  float amx_unused = 0.f;
  ring_epilogue<MB, NJ, EPI, HAS_CIN, EDGE>(P, acc, m0, n0, wm, wn, li, kh, sk, s1, s2, amx_unused);
  if (EPI == 3) ring_stats<MB, NJ>(P, rt, panel, m0, wm, wn, li, kh, sk, s1, s2, reinterpret_cast<float*>(ring));
  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
  stamp(3);
}

// ------------------------------------------------------------------------------------------------------------
// fp16 form of the ring kernel (round 4): the same panels, LDS images and epilogues, with every fp32 operand value
// carried as TWO fp16 pieces of its power-of-two-scaled value and a*b formed from THREE piece products
// (lo*hi, hi*lo, hi*hi; the dropped lo*lo is < 2^-22 |a||b|) on v_mfma_f32_32x32x16_f16 -- half the matrix-pipe work
// of the 3 x bf16 / 6-product form above, and 6 VALU per value pair for the split instead of 11.
//   * pieces: t = v * 2^e (exact), hi = RN_fp16(t), lo = RN_fp16(t - hi) (the remainder is exact in fp32; both
//     conversions round to nearest, v_cvt_pk_f16_f32): |t - hi - lo| <= 2^-22 |t|, i.e. 22 significant bits per operand,
//     against 24 for fp32 -- below the accumulation error of any fp32 GEMM of K >= 16 (tools/gemm16_emulate.py: max
//     and rms error against fp64 equal to or below the 6-product form at every shape of the block, offsets and
//     gradient-sized operands included; the MFMA rounds three partial sums per k-step into the accumulator, not six);
//   * range: fp16 keeps 11 bits only between 2^-14 and 2^16, so each operand TENSOR is scaled by the power of two
//     that puts its largest magnitude in [2^14, 2^15): e = 141 - biased_exponent(max|v|).  The maxima are device words
//     (fp32 bit patterns: max over |v| is an unsigned integer max, order-free, deterministic) written by gps_absmax or
//     by the operand's producer; the weight image carries its own.  Elements more than 2^17 below their tensor's
//     maximum lose relative precision (lo goes subnormal; the absolute error stays below 2^-39 of the maximum);
//     the accumulators are un-scaled by 2^-(eA + eB) (exact) in the epilogue;
//   * the ring is deeper: a slot is A (raw fp32) + 2 W pieces = 32-40 KB, so FOUR or FIVE slots fit the CU's 160 KB
//     where the 3-piece form had three.  With the MFMA time per stage halved, the bytes in flight -- not the matrix
//     pipe -- decide whether the loop stalls: 2.5-3.5 stages are now outstanding across every barrier.
// ------------------------------------------------------------------------------------------------------------
using gps::amax_be;        // biased exponent of a max|.| record (gps_common.hpp: 8 words, floored at 16)
__device__ __forceinline__ float amax_scale(unsigned be) { return __uint_as_float((268u - be) << 23); }     // 2^(141 - be)
__device__ __forceinline__ float amax_unscale(unsigned be) { return __uint_as_float((be - 14u) << 23); }    // 2^(be - 141)

// (v0, v1) -> packed fp16 pairs hi, lo of the scaled values
__device__ __forceinline__ void split16(float v0, float v1, float s, uint32_t& hi, uint32_t& lo) {
  const f32x2 t = (f32x2){v0, v1} * s;
  const f16x2 h = __builtin_convertvector(t, f16x2);
  const f32x2 r = t - __builtin_convertvector(h, f32x2);      // exact
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// ---- max |v| of up to 56 row-major fp32 matrices in one launch ------------------------------------------------------
struct AbsDesc {
  const float* A;
  int64_t ld, rows;
  int cols;
  int nr;              // rows per workgroup
  uint32_t* slot;
  int block_begin;
};
constexpr int kMaxAbs = 56;
struct AbsGroup {
  AbsDesc d[kMaxAbs];
  int n;
};
__global__ __launch_bounds__(256) void k_absmax(const AbsGroup G) {
  int di = 0, hi = G.n - 1;
  while (di < hi) {
    const int mid = (di + hi + 1) >> 1;
    if ((int)blockIdx.x >= G.d[mid].block_begin) di = mid; else hi = mid - 1;
  }
  const AbsDesc& D = G.d[di];
  const int cq = D.cols >> 2;
  const int64_t r0 = (int64_t)(blockIdx.x - D.block_begin) * D.nr;
  const int64_t nrows = min((int64_t)D.nr, D.rows - r0);
  const int n4 = (int)(nrows * cq);
  const float* base = D.A + r0 * D.ld;
  uint32_t m = 0;
#pragma unroll 8
  for (int idx = threadIdx.x; idx < n4; idx += 256) {
    const int r = idx / cq, c = idx - r * cq;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (int64_t)r * D.ld + 4 * c));
    m = max(max(m, v[0] & 0x7FFFFFFFu), max(v[1] & 0x7FFFFFFFu, max(v[2] & 0x7FFFFFFFu, v[3] & 0x7FFFFFFFu)));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  // ONE atomic per workgroup: atomics on one word serialise at the L2 (~6 ns each -- with one per wavefront a 12 MB
  // matrix took 20 us, 0.6 TB/s, whatever its size)
  __shared__ uint32_t wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) gps::amax_raise(D.slot, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// ---- weight images, fp16 form: [2 pieces][K/32][N'][32], same geometry as the 3-piece image ------------------------------
struct SplitDesc16 {
  SplitDesc d;
  const uint32_t* amax;
};
constexpr int kMaxSplit16 = 48;      // 48 x 72-byte descriptors
struct SplitGroup16 {
  SplitDesc16 d[kMaxSplit16];
  int n;
};
__global__ __launch_bounds__(256) void k_split_weights16(const SplitGroup16 G) {
  __shared__ __attribute__((aligned(16))) uint16_t T[2][ST_C * ST_P];
  int di = 0, hi = G.n - 1;
  while (di < hi) {
    const int mid = (di + hi + 1) >> 1;
    if ((int)blockIdx.x >= G.d[mid].d.block_begin) di = mid; else hi = mid - 1;
  }
  const SplitDesc& D = G.d[di].d;
  const float sc = amax_scale(amax_be(G.d[di].amax));
  const int ctiles = (D.cols + ST_C - 1) / ST_C;
  const int tile = blockIdx.x - D.block_begin;
  const int r0 = (tile / ctiles) * ST_R, c0 = (tile % ctiles) * ST_C;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int r = idx >> 5, cq = (idx & 31) * 4;
    const int row = r0 + r, col = c0 + cq;
    const bool ok = row < D.rows && col < D.cols;
    const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(D.W + (int64_t)row * D.ldw + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t h[2], l[2];
    split16(v[0], v[1], sc, h[0], l[0]);
    split16(v[2], v[3], sc, h[1], l[1]);
    if (D.nt && row < D.rows && col < D.nt_k) {
      *reinterpret_cast<u32x2*>(D.nt + img_index(0, row, col, D.nt_n, D.nt_k)) = (u32x2){h[0], h[1]};
      *reinterpret_cast<u32x2*>(D.nt + img_index(1, row, col, D.nt_n, D.nt_k)) = (u32x2){l[0], l[1]};
    }
    if (D.tn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T[0][(cq + j) * ST_P + r] = (uint16_t)(h[j >> 1] >> (16 * (j & 1)));
        T[1][(cq + j) * ST_P + r] = (uint16_t)(l[j >> 1] >> (16 * (j & 1)));
      }
    }
  }
  if (!D.tn) return;
  __syncthreads();
  // 2 pieces x 128 n x 4 chunks of 16 bytes = 1024 chunks, 4 per thread
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    const int p = idx / (ST_C * 4), rem = idx - p * (ST_C * 4);
    const int n = rem >> 2, ch = rem & 3;
    if (c0 + n < D.cols && r0 + ch * 8 < D.tn_k) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(&T[p][n * ST_P + ch * 8]);
      *reinterpret_cast<u32x4*>(D.tn + img_index(p, c0 + n, r0 + ch * 8, D.tn_n, D.tn_k)) = v;
    }
  }
}

constexpr int r16_slot_bytes(int mb, int nj) { return rg_a_bytes(mb) + 2 * rg_bp(nj); }        // 16-40 KB
constexpr int r16_slots(int mb, int nj) { return 163840 / r16_slot_bytes(mb, nj) > 5 ? 5 : 163840 / r16_slot_bytes(mb, nj); }
constexpr int r16_lds_bytes(int mb, int nj) { return r16_slots(mb, nj) * r16_slot_bytes(mb, nj); }
// ring depth of the PERSISTENT tiles: a tile's k-stages must be whole rotations, and depth is not what bounds the loop
// (profiles/r06_gemm_tiles_occupancy.txt: 3 slots run as fast as 5), so the 128-column panels (5 slots fit) rotate over 4 --
// K = 256 (8 stages: the AST batches' d) persists like K = 384 (12 stages) does on the 192-column panels
constexpr int r16_persist_slots(int mb, int nj) { return r16_slots(mb, nj) > 4 ? 4 : r16_slots(mb, nj); }
// first W-fragment read of gap i when nw reads are dealt over `gaps` gaps: one beside the raw-A reads of gap 0, the rest evenly
constexpr int r16_rd_first(int i, int nw, int gaps) {
  return i <= 0 ? 0 : (i >= gaps ? nw : 1 + ((i - 1) * (nw - 1) + (gaps - 1) - 1) / (gaps - 1));
}

template <int NJ>
struct Ring16Frag {
  u32x4 b[NJ][2];      // W fragments of one 16-wide k-step: [column block][hi, lo] as fp16 pairs
};
struct Ring16A {
  u32x4 p[2];          // the A fragment of one row block and k-step: hi, lo
};
template <int I>
struct IntC { static constexpr int value = I; };

// PERSIST (k_gemm_ring16_pair's first problem when it spans several dispatch rounds): the workgroup walks tiles bid,
// bid + stride, ... < ntiles.  The k-loop issues its DMA S - 1 stages ahead and, past a tile's last stage, used to re-fetch
// that stage to keep the counted waits uniform; here those transfers fetch the NEXT tile's first stages, the last region
// reads and splits its first k-step, and after the epilogue (C stores; one vmcnt(0): loads and stores share the counter) the
// loop simply goes on -- the ~3.6 us from a workgroup's entry to its first landed stage are paid once per workgroup instead
// of once per tile (tools/gemm_trace.py: 20 % of a 128 x 192 tile's life).  Needs KS % S == 0 (the next tile's stage 0 must
// land in slot 0), no edge shape, an epilogue that leaves the ring alone (EPI <= 2; the max|C| record uses the ring as
// scratch once, behind the workgroup's LAST tile, from the running maximum over its tiles).
template <int MB, int NJ, int EPI, bool HAS_CIN, bool EDGE, bool PERSIST = false>
__device__ __forceinline__ void ring16_body(const PanelArgs& P, const int bid, const int stride = 0, const int ntiles = 0) {
  static_assert(!PERSIST || (!EDGE && EPI <= 2), "persistent tiles: whole panels, ring-free epilogue");
  constexpr int TNV = 64 * NJ;
  constexpr int BP = rg_bp(NJ);
  constexpr int A_BYTES = rg_a_bytes(MB), SLOT = r16_slot_bytes(MB, NJ);
  constexpr int S = PERSIST ? r16_persist_slots(MB, NJ) : r16_slots(MB, NJ);          // ring depth
  constexpr int NA = 2 * MB;                    // A transfers per wave and stage
  constexpr int NW = 2 * NJ;                    // W transfers per wave and stage
  constexpr int ND = NA + NW;
  constexpr int ND_ODD = ND / 2, ND_EVEN = ND - ND_ODD;
  constexpr int G = 3 * MB * NJ;                // MFMAs (= issue gaps) per region
  constexpr int RD_GAPS = G < 4 ? G : 4;        // gaps that carry the W-fragment reads of the next k-step
  // the split of the next k-step's raw A values: 4 MB pairs x 3 instalments of 2-3 VALU, SPLITQ instalments per gap in the
  // LAST gaps of the region (the compiler waits lgkmcnt(0) at the first use of a raw value: by then every LDS read of gaps
  // 0 .. 3 should have landed, or the wait is a stall of one LDS round trip per region); regions of fewer than 9 MFMAs
  // (the narrow panels: latency-bound shapes) deal it evenly from gap 1
  constexpr int SPLITQ = G >= 9 ? (12 * MB + (G - 4) - 1) / (G - 4) : (12 * MB + (G - 1) - 1) / (G - 1);
  constexpr int SPLIT0 = G >= 9 ? G - (12 * MB + SPLITQ - 1) / SPLITQ : 1;
  static_assert(ND_EVEN <= G && 12 * MB <= (G - SPLIT0) * SPLITQ, "the staging does not fit the region's issue gaps");
  static_assert((S - 2) * ND + ND_ODD <= 63 && S >= 3, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (P.trace && threadIdx.x == 0) P.trace[4 * (size_t)bid + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  int tile = bid;
  int panel = tile / P.row_tiles, rt = tile - panel * P.row_tiles;   // panel-major numbering
  int64_t m0 = (int64_t)rt * (64 * MB);
  int n0 = panel * TNV;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;
  const int KS = (P.K + BK - 1) / BK;
  const unsigned bea = amax_be(P.a_amax), bew = amax_be(P.w_amax);
  const float sa = amax_scale(bea);

  // ---- DMA sources (as k_gemm_ring, two W pieces) ------------------------------------------------------------------
  const unsigned char* a_src[NA];
  const unsigned char* a_srcN[NA];          // PERSIST: the next tile's sources, pre-biased by -KS stages
  int a_tail[NA];
  const int nvc = EDGE ? (P.K - (KS - 1) * BK) / 4 : 8;
  const int64_t stage_stride = (int64_t)P.Nimg * (BK * 2);
  const int64_t piece_stride = stage_stride * KS;
  const unsigned char* b_src;
  const unsigned char* b_srcN = nullptr;
  auto sources = [&](int64_t tm0, int tn0, const unsigned char* (&as)[NA], const unsigned char*& bs, int64_t bias_stages)
                     __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = 8 * NA * wave + 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const int64_t grow = min(tm0 + row, P.M - 1);
      as[i] = reinterpret_cast<const unsigned char*>(P.A + grow * P.lda) + c * 16 - bias_stages * (BK * 4);
      if (bias_stages == 0) a_tail[i] = (EDGE && c >= nvc) ? c * 16 : 0;
    }
    bs = reinterpret_cast<const unsigned char*>(P.Bp) + ((int64_t)tn0 + 16 * NJ * wave + (lane >> 2)) * (BK * 2) +
         (((lane & 3) ^ ((lane >> 4) & 3)) * 16) - bias_stages * stage_stride;
  };
  sources(m0, n0, a_src, b_src, 0);
  bool has_next = false;                    // (workgroup-uniform)
  auto dma = [&](int g, int s, unsigned char* slot) __attribute__((always_inline)) {
    const bool nx = PERSIST && s >= KS;     // past this tile's last stage: the next tile's first stages (has_next holds)
    if (g < NA) {
      if (EDGE) glds16(a_src[g] + (int64_t)s * (BK * 4) - (s == KS - 1 ? a_tail[g] : 0), slot + wave * (NA * 1024) + g * 1024);
      else glds16((nx ? a_srcN[g] : a_src[g]) + (int64_t)s * (BK * 4), slot + wave * (NA * 1024) + g * 1024);
    } else {
      const int i = g - NA;
      glds16((nx ? b_srcN : b_src) + s * stage_stride + (i / NJ) * piece_stride + (i % NJ) * 1024,
             slot + A_BYTES + (i / NJ) * BP + wave * (NJ * 1024) + (i % NJ) * 1024);
    }
  };

  // ---- fragment addresses (byte offsets inside a slot) ---------------------------------------------------------
  int a_off[2][2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c0 = ks * 4 + kh * 2, sw = (li >> 1) & 7;
    a_off[ks][0] = (wm * 32 * MB + li) * 128 + ((c0 ^ sw) * 16);
    a_off[ks][1] = (wm * 32 * MB + li) * 128 + (((c0 + 1) ^ sw) * 16);
    b_off[ks] = A_BYTES + (wn * 32 * NJ + li) * 64 + (((ks * 2 + kh) ^ ((li >> 2) & 3)) * 16);
  }

  f32x16 acc[MB][NJ];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mb][j][q] = 0.0f;

  // split of one pair of raw values in three instalments of 2 VALU: scale + pack hi | hi back to fp32 | remainder + pack lo
  // (the packed hi travels between the instalments in its own register: read back out of the partly written fragment
  // `out.p[0][d]` it was folded to the FIRST pair's value by the compiler -- caught in the ISA, tools/kernel_regs.py era)
  f32x2 st_hb;
  uint32_t st_hi;
  auto split_part = [&](int part, const f32x4 (&raw)[2], int d, Ring16A& out) __attribute__((always_inline)) {
    const float v0 = raw[d >> 1][(d & 1) * 2], v1 = raw[d >> 1][(d & 1) * 2 + 1];
    if (part == 0) {
      st_hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){v0 * sa, v1 * sa}, f16x2));
      out.p[0][d] = st_hi;
    } else if (part == 1) {
      st_hb = __builtin_convertvector(__builtin_bit_cast(f16x2, st_hi), f32x2);
    } else {
      out.p[1][d] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){fmaf(v0, sa, -st_hb[0]), fmaf(v1, sa, -st_hb[1])}, f16x2));
    }
  };

  constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};     // lo*hi, hi*lo, hi*hi: smallest terms first

  f32x4 raw[MB][2];
  auto region = [&](const Ring16A (&ac)[MB], const Ring16Frag<NJ>& fc, Ring16A (&an)[MB], Ring16Frag<NJ>& fn, const unsigned char* rd_slot,
                    int rd_ks, int dma_stage, unsigned char* dma_slot, int dma_first, int dma_count)
                    __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int term = i / (NJ * MB), mb = (i / NJ) % MB, j = i % NJ;
      acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ac[mb].p[TA[term]]),
                                                         __builtin_bit_cast(f16x8, fc.b[j][TB[term]]), acc[mb][j], 0, 0, 0);
      if (i == 0) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          raw[b][0] = *reinterpret_cast<const f32x4*>(rd_slot + a_off[rd_ks][0] + b * 4096);
          raw[b][1] = *reinterpret_cast<const f32x4*>(rd_slot + a_off[rd_ks][1] + b * 4096);
        }
      }
      if (i < RD_GAPS) {
#pragma unroll
        for (int k = r16_rd_first(i, NW, RD_GAPS); k < r16_rd_first(i + 1, NW, RD_GAPS); ++k)
          fn.b[k / 2][k % 2] = *reinterpret_cast<const u32x4*>(rd_slot + b_off[rd_ks] + (k / 2) * (32 * 64) + (k % 2) * BP);
      }
      if (i < dma_count) dma(dma_first + i, dma_stage, dma_slot);
      if (i >= SPLIT0) {
#pragma unroll
        for (int u = 0; u < SPLITQ; ++u) {
          const int q = (i - SPLIT0) * SPLITQ + u;
          if (q < 12 * MB) split_part(q % 3, raw[(q / 3) / 4], (q / 3) % 4, an[(q / 3) / 4]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: stages 0 .. S-2 whole, the first ND_ODD transfers of stage S-1 (stage indices clamped to the last one)
#pragma unroll
  for (int st = 0; st < S - 1; ++st)
#pragma unroll
    for (int g = 0; g < ND; ++g) dma(g, min(st, KS - 1), ring + st * SLOT);
#pragma unroll
  for (int g = 0; g < ND_ODD; ++g) dma(g, min(S - 1, KS - 1), ring + (S - 1) * SLOT);
  __builtin_amdgcn_s_waitcnt(waitcnt_imm((S - 2) * ND + ND_ODD, 15));   // this wave's share of stage 0 has landed
  __builtin_amdgcn_s_barrier();                                         // ... and every other wave's
  stamp(1);
  Ring16Frag<NJ> f0, f1;
  Ring16A a0[MB], a1[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    raw[b][0] = *reinterpret_cast<const f32x4*>(ring + a_off[0][0] + b * 4096);
    raw[b][1] = *reinterpret_cast<const f32x4*>(ring + a_off[0][1] + b * 4096);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      f0.b[j][p] = *reinterpret_cast<const u32x4*>(ring + b_off[0] + j * (32 * 64) + p * BP);
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int part = 0; part < 3; ++part) split_part(part, raw[b], d, a0[b]);
  __builtin_amdgcn_sched_barrier(0);

  // Stage s sits in slot s % S.  Region 2s multiplies k-step 2s while k-step 2s+1 is fetched from the same slot and the
  // last ND_EVEN transfers of stage s+S-1 are issued; then the stage barrier: the counted wait retires this wave's share
  // of stage s+1 (stages s+2 .. s+S-1 stay in flight across it), lgkmcnt(0) retires its reads of slot s % S, which may
  // be refilled after the barrier.  Region 2s+1 multiplies k-step 2s+1, fetches k-step 2s+2 from the next slot and
  // issues the first ND_ODD transfers of stage s+S into the slot just freed.
  auto stage = [&](auto ic, int s) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    unsigned char* const cur = ring + I * SLOT;
    unsigned char* const nxt = ring + ((I + 1) % S) * SLOT;
    unsigned char* const lst = ring + ((I + S - 1) % S) * SLOT;
    region(a0, f0, a1, f1, cur, 1, (PERSIST && has_next) ? s + S - 1 : min(s + S - 1, KS - 1), lst, ND_ODD, ND_EVEN);
    __builtin_amdgcn_s_waitcnt(waitcnt_imm((S - 2) * ND, 0));
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    region(a1, f1, a0, f0, nxt, 0, (PERSIST && has_next) ? s + S : min(s + S, KS - 1), cur, 0, ND_ODD);
  };
  if (PERSIST) {
    has_next = tile + stride < ntiles;
    if (has_next) {
      const int t2 = tile + stride, p2 = t2 / P.row_tiles;
      sources((int64_t)(t2 - p2 * P.row_tiles) * (64 * MB), p2 * TNV, a_srcN, b_srcN, KS);
    }
  }
  float amx_run = 0.f;                      // max|C| over every tile this workgroup computes (one atomic at its end)
 next_tile:
  int s0 = 0;
  for (; s0 + S <= KS; s0 += S) {
    stage(IntC<0>{}, s0);
    stage(IntC<1>{}, s0 + 1);
    stage(IntC<2>{}, s0 + 2);
    if constexpr (S > 3) stage(IntC<3 % S>{}, s0 + 3);
    if constexpr (S > 4) stage(IntC<4 % S>{}, s0 + 4);
  }
  if (s0 < KS) {                                            // up to S-1 stages left over (workgroup-uniform)
    stage(IntC<0>{}, s0);
    if (s0 + 1 < KS) {
      stage(IntC<1>{}, s0 + 1);
      if (s0 + 2 < KS) {
        stage(IntC<2>{}, s0 + 2);
        if constexpr (S > 4) {
          if (s0 + 3 < KS) stage(IntC<3 % S>{}, s0 + 3);
        }
      }
    }
  }
  if (!(PERSIST && has_next))
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));       // no DMA may still be writing this workgroup's LDS when it retires
  stamp(2);
  // un-scale: exact powers of two, two factors so that neither can leave the fp32 exponent range on its own
  const float ua = amax_unscale(bea), uw = amax_unscale(bew);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mb][j][q] = (acc[mb][j][q] * ua) * uw;
  float sk[NJ], s1[NJ], s2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) sk[j] = s1[j] = s2[j] = 0.f;
  float amx = 0.f;
  ring_epilogue<MB, NJ, EPI, HAS_CIN, EDGE>(P, acc, m0, n0, wm, wn, li, kh, sk, s1, s2, amx);
  amx_run = fmaxf(amx_run, amx);
  if constexpr (PERSIST) {
    if (has_next) {       // the seam: the next tile's first stages are in the ring (or on their way), its first k-step in a0 / f0
      __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));     // stores and DMA share vmcnt: the loop's counted waits must see DMA only
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[mb][j][q] = 0.0f;
      tile += stride;
      panel = tile / P.row_tiles; rt = tile - panel * P.row_tiles;
      m0 = (int64_t)rt * (64 * MB); n0 = panel * TNV;
#pragma unroll
      for (int i = 0; i < NA; ++i) a_src[i] = a_srcN[i] + (int64_t)KS * (BK * 4);
      b_src = b_srcN + (int64_t)KS * stage_stride;
      has_next = tile + stride < ntiles;
      if (has_next) {
        const int t2 = tile + stride, p2 = t2 / P.row_tiles;
        sources((int64_t)(t2 - p2 * P.row_tiles) * (64 * MB), p2 * TNV, a_srcN, b_srcN, KS);
      }
      goto next_tile;
    }
  }
  if (EPI == 3) ring_stats<MB, NJ>(P, rt, panel, m0, wm, wn, li, kh, sk, s1, s2, reinterpret_cast<float*>(ring));
  if (EPI == 4) ring_sums<MB, NJ, 3>(P, rt, panel, wm, wn, li, kh, sk, s1, s2, reinterpret_cast<float*>(ring));
  if (P.c_amax) {                                        // workgroup-uniform: max|C| of this workgroup's tiles -> one atomic
    uint32_t m = __float_as_uint(amx_run);
#pragma unroll
    for (int o = 32; o; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    uint32_t* wmax = reinterpret_cast<uint32_t*>(ring);
    __syncthreads();                                     // the ring (and the statistics scratch) is free
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
    if (t == 0) gps::amax_raise(P.c_amax, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
  }
  // A workgroup retires with its C stores still in flight (the memory system completes them; the end of the kernel orders
  // them before the next launch): draining them here kept the CU from taking its next tile for a store round trip per
  // dispatch round -- only the trace wants the stores-done stamp.
  if (P.trace) {
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
    stamp(3);
  }
}

template <int MB, int NJ, int EPI, bool HAS_CIN, bool EDGE = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_gemm_ring16(const PanelArgs P) {
  ring16_body<MB, NJ, EPI, HAS_CIN, EDGE>(P, (int)blockIdx.x);
}

// One problem of two dispatch rounds or more (AST-sized batches: 792 tiles of 8 k-stages in FF1 and its input gradient) as
// gridDim.x persistent workgroups: ring16_body PERSIST, workgroup b takes tiles b, b + gridDim.x, ...
template <int MB, int NJ, int EPI, bool HAS_CIN>
__global__ __launch_bounds__(NTHREADS, 1) void k_gemm_ring16_persist(const PanelArgs P, const int tiles) {
  ring16_body<MB, NJ, EPI, HAS_CIN, false, true>(P, (int)blockIdx.x, (int)gridDim.x, tiles);
}

// Two independent GEMMs in ONE dispatch (gps_gemm16_panel_pair): workgroups 0 .. split-1 are the tiles of the first problem,
// the rest those of the second.  A dependent dispatch of a replayed graph costs ~4 us before its first workgroup runs, and a
// GEMM of one dispatch round leaves every CU idle through its prologue and its epilogue; paired, the second problem's tiles
// start as the first's retire.  The first problem should be the one with the longer tiles (they are dispatched first).
// TAIL: the last rows of the second problem as a third one with 64-row tiles (gps_gemm16_panel_pair's tail balancing): tiles
// are dispatched in blockIdx order onto CUs as they free up, so when the tile count leaves a last dispatch round mostly empty
// (P30 x 256: 840 + 240 tiles of 128 x 192 on 256 CUs = 4.2 rounds, the fifth 22 % full), re-cutting that round's tiles in
// halves ends the launch half a tile-time earlier.
// PERSIST0: the first problem's tiles are walked by `split` persistent workgroups (ring16_body PERSIST) instead of one
// workgroup each: `tiles0` of them, workgroup b takes b, b + split, ...
template <int MB0, int MB1, int NJ, bool HAS_CIN, bool TAIL = false, bool PERSIST0 = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_gemm_ring16_pair(const PanelArgs P0, const PanelArgs P1, const int split,
                                                                 const PanelArgs P2, const int split2, const int tiles0) {
  if ((int)blockIdx.x < split) {
    if constexpr (PERSIST0) ring16_body<MB0, NJ, 0, HAS_CIN, false, true>(P0, (int)blockIdx.x, split, tiles0);
    else ring16_body<MB0, NJ, 0, HAS_CIN, false>(P0, (int)blockIdx.x);
  }
  else if (!TAIL || (int)blockIdx.x < split2) ring16_body<MB1, NJ, 0, HAS_CIN, false>(P1, (int)blockIdx.x - split);
  else ring16_body<1, NJ, 0, HAS_CIN, false>(P2, (int)blockIdx.x - split2);
}


inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

unsigned long long* g_panel_trace = nullptr;

static int panel_launch(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                        const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue, const float* mask_src,
                        int64_t ldmask, float p_drop, uint64_t seed, const gps_bn* stats, float* ws, size_t ws_floats,
                        uint32_t* sync, gps_stream_t stream, const uint32_t* a_amax = nullptr, const uint32_t* w_amax = nullptr,
                        uint32_t* c_amax = nullptr, const int32_t* m_dev = nullptr);

// a checked, filled launch of the ring kernel: what panel_launch dispatches (and gps_gemm16_panel_pair dispatches two of)
struct PanelPlan {
  PanelArgs P;
  int mb, nj;
  bool edge, f16;
  unsigned grid;
};
static int panel_prepare(PanelPlan& Q, const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N,
                         const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue,
                         const float* mask_src, int64_t ldmask, float p_drop, uint64_t seed, const gps_bn* stats, float* ws,
                         uint32_t* sync, const uint32_t* a_amax, const uint32_t* w_amax, uint32_t* c_amax, const int32_t* m_dev,
                         const gps_gemm_colsums* cs = nullptr);

// 128-row panels (MB = 2) when they still give every CU a workgroup; 64-column panels always use 64 rows (the only
// NJ = 1 instantiation).
static int ring_mb(int64_t M, int N, int K) {
  const int nj = rg_nj(N, K);
  if (nj == 1) return 1;
  const int64_t tiles128 = ((M + 127) / 128) * (rg_npad(N, K) / (64 * nj));
  return tiles128 >= 200 ? 2 : 1;
}

extern "C" {

// Debugging aid, not part of the drop-in surface: subsequent gps_gemm_panel launches of the ring kernel record
// s_memtime at entry / first stage landed / loop done / stores done, 4 x uint64 per workgroup, into `buf` (device
// memory, >= 4 * grid entries); nullptr switches it off.
int gps_gemm_panel_trace(unsigned long long* buf) { g_panel_trace = buf; return GPS_OK; }

// bf16 elements of the image of an [N, K] weight: N padded to whole column panels, K to whole stages (rg_npad / rg_kpad)
size_t gps_gemm_image_elems(int64_t N, int64_t K) { return N > 0 && K > 0 ? (size_t)(3 * rg_npad(N, K) * rg_kpad(K)) : 0; }

// the ring kernel: column panels of 192 / 128 / 64 (the last one partial when none divides N), any number of 32-wide
// k-stages (the last one partly empty when K % 32 != 0); N and K multiples of 4 (16-byte rows / chunks)
int gps_gemm_panel_supported(int64_t N, int64_t K) { return N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0; }

int gps_gemm_split_weights(int n, const gps_gemm_split* descs, gps_stream_t stream) {
  GPS_REQUIRE(n >= 1 && n <= kMaxSplit && descs, "gps_gemm_split_weights: 1..%d weights per launch", kMaxSplit);
  SplitGroup G{};
  G.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const gps_gemm_split& d = descs[i];
    GPS_REQUIRE(d.W && d.rows > 0 && d.cols > 0 && d.ldw >= d.cols && d.cols % 4 == 0 && d.ldw % 4 == 0 && al16(d.W),
                "gps_gemm_split_weights: weight %d: bad shape / alignment", i);
    GPS_REQUIRE(!d.image_nt || (gps_gemm_panel_supported(d.rows, d.cols) && al16(d.image_nt)),
                "gps_gemm_split_weights: weight %d: rows, cols %% 4", i);
    GPS_REQUIRE(!d.image_tn || (gps_gemm_panel_supported(d.cols, d.rows) && al16(d.image_tn)),
                "gps_gemm_split_weights: weight %d: rows, cols %% 4", i);
    G.d[i] = SplitDesc{d.W, d.ldw, d.rows, d.cols, d.image_nt, d.image_tn, blocks,
                       (int)rg_npad(d.rows, d.cols), (int)rg_kpad(d.cols), (int)rg_npad(d.cols, d.rows), (int)rg_kpad(d.rows)};
    blocks += ((d.rows + ST_R - 1) / ST_R) * ((d.cols + ST_C - 1) / ST_C);
  }
  k_split_weights<<<(unsigned)blocks, 256, 0, gps::as_stream(stream)>>>(G);
  return gps::launch_status("gps_gemm_split_weights");
}

size_t gps_gemm_stats_floats(int64_t M, int N, int K) {
  if (M < 1 || !gps_gemm_panel_supported(N, K) || rg_edge(N, K)) return 0;
  const int tn = 64 * rg_nj(N, K), mb = ring_mb(M, N, K);
  const int rt = (int)((M + 64 * mb - 1) / (64 * mb));
  return (size_t)(N / tn) * (tr::floats_for(rt, 2, tn) + 16);
}
int gps_gemm_stats_sync_words(int N) { return N >= 64 && N % 64 == 0 ? (N / 64) * tr::kSyncWords : 0; }   // (>= any panel count)
int gps_gemm_stats_supported(int64_t M, int N, int K) {
  if (!gps_gemm_panel_supported(N, K) || rg_edge(N, K) || M < 2) return 0;   // (whole panels only)
  const int mb = ring_mb(M, N, K);
  return (M + 64 * mb - 1) / (64 * mb) <= tr::kMaxParts;
}

int gps_gemm_panel(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                   const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue, const float* mask_src,
                   int64_t ldmask, float p_drop, uint64_t seed, gps_stream_t stream) {
  GPS_REQUIRE(epilogue >= 0 && epilogue <= 2, "gps_gemm_panel: epilogue");
  return panel_launch(A, lda, M, K, image, N, bias, Cin, ldcin, C, ldc, epilogue, mask_src, ldmask, p_drop, seed, nullptr,
                      nullptr, 0, nullptr, stream);
}

int gps_gemm_panel_stats(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                         const float* Cin, int64_t ldcin, float* C, int64_t ldc, float p_drop, uint64_t seed,
                         const gps_bn* stats, float* ws, size_t ws_floats, uint32_t* sync, gps_stream_t stream) {
  GPS_REQUIRE(gps_gemm_stats_supported(M, N, K), "gps_gemm_panel_stats: shape M=%lld N=%d K=%d not served by the ring kernel",
              (long long)M, N, K);
  GPS_REQUIRE(Cin && stats && stats->mean && stats->rstd && ws && sync && al16(ws), "gps_gemm_panel_stats: null / misaligned buffer");
  GPS_REQUIRE((stats->running_mean == nullptr) == (stats->running_var == nullptr), "gps_gemm_panel_stats: running stats");
  GPS_REQUIRE(ws_floats >= gps_gemm_stats_floats(M, N, K), "gps_gemm_panel_stats: workspace too small (gps_gemm_stats_floats)");
  return panel_launch(A, lda, M, K, image, N, bias, Cin, ldcin, C, ldc, 3, nullptr, 0, p_drop, seed, stats, ws, ws_floats, sync,
                      stream);
}

// ---- fp16 form (k_gemm_ring16): the operand maxima travel as device words ------------------------------------------------
size_t gps_gemm16_image_elems(int64_t N, int64_t K) { return N > 0 && K > 0 ? (size_t)(2 * rg_npad(N, K) * rg_kpad(K)) : 0; }

int gps_absmax(int n, const gps_absmax_desc* descs, gps_stream_t stream) {
  GPS_REQUIRE(n >= 1 && n <= kMaxAbs && descs, "gps_absmax: 1..%d matrices per launch", kMaxAbs);
  AbsGroup G{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const gps_absmax_desc& d = descs[i];
    GPS_REQUIRE(d.A && d.slot && d.rows >= 0 && d.cols > 0 && d.cols % 4 == 0 && d.ld >= d.cols && d.ld % 4 == 0 && al16(d.A),
                "gps_absmax: matrix %d: bad shape / alignment", i);
    if (d.rows == 0) continue;
    const int cq = d.cols / 4;
    int64_t nr = cq >= 1024 ? 1 : 1024 / cq;                   // >= 16 KB per workgroup ...
    nr = std::max<int64_t>(nr, (d.rows + 255) / 256);          // ... and at most 256 workgroups (= atomics) per matrix
    G.d[G.n] = AbsDesc{d.A, d.ld, d.rows, d.cols, (int)nr, d.slot, blocks};
    blocks += (int)((d.rows + nr - 1) / nr);
    ++G.n;
  }
  if (G.n == 0) return GPS_OK;
  k_absmax<<<(unsigned)blocks, 256, 0, gps::as_stream(stream)>>>(G);
  return gps::launch_status("gps_absmax");
}

int gps_gemm16_split_weights(int n, const gps_gemm_split16* descs, gps_stream_t stream) {
  GPS_REQUIRE(n >= 1 && n <= kMaxSplit16 && descs, "gps_gemm16_split_weights: 1..%d weights per launch", kMaxSplit16);
  SplitGroup16 G{};
  G.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const gps_gemm_split16& d = descs[i];
    GPS_REQUIRE(d.W && d.amax && d.rows > 0 && d.cols > 0 && d.ldw >= d.cols && d.cols % 4 == 0 && d.ldw % 4 == 0 && al16(d.W),
                "gps_gemm16_split_weights: weight %d: bad shape / alignment", i);
    GPS_REQUIRE(!d.image_nt || (gps_gemm_panel_supported(d.rows, d.cols) && al16(d.image_nt)),
                "gps_gemm16_split_weights: weight %d: rows, cols %% 4", i);
    GPS_REQUIRE(!d.image_tn || (gps_gemm_panel_supported(d.cols, d.rows) && al16(d.image_tn)),
                "gps_gemm16_split_weights: weight %d: rows, cols %% 4", i);
    G.d[i].d = SplitDesc{d.W, d.ldw, d.rows, d.cols, d.image_nt, d.image_tn, blocks,
                         (int)rg_npad(d.rows, d.cols), (int)rg_kpad(d.cols), (int)rg_npad(d.cols, d.rows), (int)rg_kpad(d.rows)};
    G.d[i].amax = d.amax;
    blocks += ((d.rows + ST_R - 1) / ST_R) * ((d.cols + ST_C - 1) / ST_C);
  }
  k_split_weights16<<<(unsigned)blocks, 256, 0, gps::as_stream(stream)>>>(G);
  return gps::launch_status("gps_gemm16_split_weights");
}

int gps_gemm16_panel(const float* A, int64_t lda, int64_t M, int K, const uint32_t* a_amax, const uint16_t* image,
                     const uint32_t* w_amax, int N, const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc,
                     int epilogue, const float* mask_src, int64_t ldmask, float p_drop, uint64_t seed, uint32_t* c_amax,
                     gps_stream_t stream) {
  GPS_REQUIRE(epilogue >= 0 && epilogue <= 2, "gps_gemm16_panel: epilogue");
  GPS_REQUIRE(a_amax && w_amax, "gps_gemm16_panel: operand maxima");
  return panel_launch(A, lda, M, K, image, N, bias, Cin, ldcin, C, ldc, epilogue, mask_src, ldmask, p_drop, seed, nullptr,
                      nullptr, 0, nullptr, stream, a_amax, w_amax, c_amax);
}

int gps_gemm16_panel_stats(const float* A, int64_t lda, int64_t M, int K, const uint32_t* a_amax, const uint16_t* image,
                           const uint32_t* w_amax, int N, const float* bias, const float* Cin, int64_t ldcin, float* C,
                           int64_t ldc, float p_drop, uint64_t seed, const gps_bn* stats, float* ws, size_t ws_floats,
                           uint32_t* sync, const int32_t* m_dev, gps_stream_t stream) {
  GPS_REQUIRE(gps_gemm_stats_supported(M, N, K), "gps_gemm16_panel_stats: shape M=%lld N=%d K=%d not served by the ring kernel",
              (long long)M, N, K);
  GPS_REQUIRE(a_amax && w_amax, "gps_gemm16_panel_stats: operand maxima");
  GPS_REQUIRE(Cin && stats && stats->mean && stats->rstd && ws && sync && al16(ws), "gps_gemm16_panel_stats: null / misaligned buffer");
  GPS_REQUIRE((stats->running_mean == nullptr) == (stats->running_var == nullptr), "gps_gemm16_panel_stats: running stats");
  GPS_REQUIRE(ws_floats >= gps_gemm_stats_floats(M, N, K), "gps_gemm16_panel_stats: workspace too small (gps_gemm_stats_floats)");
  return panel_launch(A, lda, M, K, image, N, bias, Cin, ldcin, C, ldc, 3, nullptr, 0, p_drop, seed, stats, ws, ws_floats, sync,
                      stream, a_amax, w_amax, nullptr, m_dev);
}

}  // extern "C"

static int panel_prepare(PanelPlan& Q, const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N,
                         const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue,
                         const float* mask_src, int64_t ldmask, float p_drop, uint64_t seed, const gps_bn* stats, float* ws,
                         uint32_t* sync, const uint32_t* a_amax, const uint32_t* w_amax, uint32_t* c_amax, const int32_t* m_dev,
                         const gps_gemm_colsums* cs) {
  GPS_REQUIRE(M >= 1 && gps_gemm_panel_supported(N, K), "gps_gemm_panel: needs N %% 4 == 0 and K %% 4 == 0 (N=%d K=%d)",
              N, K);
  GPS_REQUIRE(A && image && C && lda >= K && ldc >= N && lda % 4 == 0 && al16(A) && al16(image),
              "gps_gemm_panel: null / misaligned buffer");
  GPS_REQUIRE(!Cin || ldcin >= N, "gps_gemm_panel: bad addend stride");
  GPS_REQUIRE(epilogue >= 0 && epilogue <= 4 && (epilogue != 2 || (mask_src && ldmask >= N)), "gps_gemm_panel: epilogue");
  GPS_REQUIRE((epilogue == 4) == (cs != nullptr), "gps_gemm_panel: epilogue 4 carries a gps_gemm_colsums");
  GPS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "gps_gemm_panel: p_drop");
  PanelArgs& P = Q.P;
  P = PanelArgs{};
  P.A = A; P.lda = lda; P.M = M; P.K = K; P.N = N; P.Nimg = (int)rg_npad(N, K); P.Bp = image; P.bias = bias; P.Cin = Cin; P.ldcin = ldcin;
  P.C = C; P.ldc = ldc; P.epilogue = epilogue; P.mask_src = mask_src; P.ldmask = ldmask;
  P.p_drop = (epilogue >= 1 && epilogue <= 3) ? p_drop : 0.0f; P.seed = seed; P.salt = gps::dropout_salt();
  P.trace = g_panel_trace;
  P.a_amax = a_amax; P.w_amax = w_amax; P.c_amax = c_amax; P.m_dev = epilogue == 3 ? m_dev : nullptr;
  const bool f16 = a_amax != nullptr;
  GPS_REQUIRE((a_amax == nullptr) == (w_amax == nullptr), "gps_gemm16_panel: both operand maxima or neither");
  // The ring kernels (LDS-DMA; k_gemm_ring: three bf16 pieces, six products -- exact; k_gemm_ring16: two fp16 pieces, three
  // products under per-tensor scales) serve every supported shape.
  const int mb = ring_mb(M, N, K), nj = rg_nj(N, K);
  const bool edge = rg_edge(N, K);
  GPS_REQUIRE(epilogue < 3 || !edge, "gps_gemm_panel: the statistics epilogues need whole column panels and k-stages");
  unsigned grid;
  {
    if (!P.bias) {
      GPS_REQUIRE(N <= kZeroBias, "gps_gemm_panel: N=%d without a bias exceeds the built-in zero row (%d)", N, kZeroBias);
      static float* zeros = []() { void* p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_bias)) == hipSuccess ? (float*)p : nullptr; }();
      GPS_REQUIRE(zeros, "gps_gemm_panel: zero-bias symbol");
      P.bias = zeros;
    }
    P.row_tiles = (int)((M + 64 * mb - 1) / (64 * mb));
    grid = (unsigned)(P.row_tiles * (P.Nimg / (64 * nj)));
    if (epilogue == 3) {
      P.st_ws = ws;
      P.st_stride = tr::floats_for(P.row_tiles, 2, 64 * nj) + 16;
      P.st_tick = sync;
      P.st_mean = stats->mean; P.st_rstd = stats->rstd; P.st_rmean = stats->running_mean; P.st_rvar = stats->running_var;
      P.st_eps = stats->eps; P.st_mom = stats->momentum;
    }
    if (cs) {      // epilogue 4: BatchNorm-backward column sums of C (checked by gps_gemm16_panel_sums)
      P.st_ws = cs->ws;
      P.st_stride = tr::floats_for(P.row_tiles, 3, 64 * nj) + 16;
      P.st_tick = cs->sync;
      P.st_mean = cs->sum_g; P.st_rstd = cs->sum_gz; P.st_rmean = cs->sum_g2; P.st_rvar = cs->sum_gz2;
      P.cs_z = cs->z; P.cs_ldz = cs->ldz; P.cs_mu = cs->bn->mean; P.cs_rs = cs->bn->rstd;
      P.cs_z2 = cs->z2; P.cs_ldz2 = cs->ldz2; P.cs_mu2 = cs->bn2->mean; P.cs_rs2 = cs->bn2->rstd;
    }
  }
  Q.mb = mb; Q.nj = nj; Q.edge = edge; Q.f16 = f16; Q.grid = grid;
  return GPS_OK;
}

// CUs of the device and GPS_GEMM_SCHED (a mask, default 3: 1 = tail balancing of the pair dispatch, 2 = persistent tiles; A/B
// handles), each read once
static int ring_cus() {
  static const int cus = []() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  return cus;
}
static int ring_sched() {
  static const int sched = []() { const char* v = getenv("GPS_GEMM_SCHED"); return v && *v ? atoi(v) : 3; }();
  return sched;
}

static int panel_launch(const float* A, int64_t lda, int64_t M, int K, const uint16_t* image, int N, const float* bias,
                        const float* Cin, int64_t ldcin, float* C, int64_t ldc, int epilogue, const float* mask_src,
                        int64_t ldmask, float p_drop, uint64_t seed, const gps_bn* stats, float* ws, size_t ws_floats,
                        uint32_t* sync, gps_stream_t stream, const uint32_t* a_amax, const uint32_t* w_amax, uint32_t* c_amax,
                        const int32_t* m_dev) {
  GPS_REQUIRE(M >= 0, "gps_gemm_panel: M");
  if (M == 0) return GPS_OK;
  (void)ws_floats;
  PanelPlan Q;
  if (int rc = panel_prepare(Q, A, lda, M, K, image, N, bias, Cin, ldcin, C, ldc, epilogue, mask_src, ldmask, p_drop, seed, stats,
                             ws, sync, a_amax, w_amax, c_amax, m_dev)) return rc;
  const PanelArgs& P = Q.P;
  const int mb = Q.mb, nj = Q.nj;
  const bool edge = Q.edge, f16 = Q.f16;
  const unsigned grid = Q.grid;
  hipStream_t s = gps::as_stream(stream);
#define GPS_RING_ANY_T(KERNEL, LDS, THREADS)                                                          \
  do {                                                                                                \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),        \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    GPS_REQUIRE(attr == hipSuccess, "gps_gemm_panel: cannot reserve %d bytes of LDS", LDS);           \
    KERNEL<<<grid, THREADS, LDS, s>>>(P);                                                             \
  } while (0)
#define GPS_RING_ANY(KERNEL, LDS) GPS_RING_ANY_T(KERNEL, LDS, NTHREADS)
#define GPS_RING_LAUNCH(MBV, NJV, E, C)                                                               \
  do {                                                                                                \
    if (f16) GPS_RING_ANY((k_gemm_ring16<MBV, NJV, E, C>), r16_lds_bytes(MBV, NJV));             \
    else GPS_RING_ANY((k_gemm_ring<MBV, NJV, E, C>), rg_lds_bytes(MBV, NJV));                         \
  } while (0)
#define GPS_RING_EDGE(MBV, NJV, E, C)                                                                 \
  do {                                                                                                \
    if (f16) GPS_RING_ANY((k_gemm_ring16<MBV, NJV, E, C, true>), r16_lds_bytes(MBV, NJV));            \
    else GPS_RING_ANY((k_gemm_ring<MBV, NJV, E, C, true>), rg_lds_bytes(MBV, NJV));                   \
  } while (0)
#define GPS_RING_SHAPES(E, C)                                                                         \
  do {                                                                                                \
    if (edge) {                                                                                       \
      if (nj == 3) { if (mb == 2) GPS_RING_EDGE(2, 3, E, C); else GPS_RING_EDGE(1, 3, E, C); }        \
      else if (nj == 2) { if (mb == 2) GPS_RING_EDGE(2, 2, E, C); else GPS_RING_EDGE(1, 2, E, C); }   \
      else GPS_RING_EDGE(1, 1, E, C);                                                                 \
    }                                                                                                 \
    else if (nj == 3) { if (mb == 2) GPS_RING_LAUNCH(2, 3, E, C); else GPS_RING_LAUNCH(1, 3, E, C); }      \
    else if (nj == 2) { if (mb == 2) GPS_RING_LAUNCH(2, 2, E, C); else GPS_RING_LAUNCH(1, 2, E, C); } \
    else GPS_RING_LAUNCH(1, 1, E, C);                                                                 \
  } while (0)
#define GPS_PANEL_LAUNCH(E, C) GPS_RING_SHAPES(E, C)
  // Persistent tiles (k_gemm_ring16_persist): 128-row tiles of whole panels over two dispatch rounds or more, k-stages in whole
  // rotations of the 4-slot ring, epilogues 0 - 2.  Same tiles, same arithmetic: bit-identical to the one-workgroup-per-tile launch.
  const int cus = ring_cus();
  if (f16 && !edge && mb == 2 && nj >= 2 && epilogue <= 2 && (ring_sched() & 2) != 0 && cus > 0 && grid >= 2u * (unsigned)cus &&
      ((K + BK - 1) / BK) % r16_persist_slots(2, nj) == 0 && !P.trace) {
#define GPS_RING_PERSIST(NJV, E, C)                                                                                  \
  do {                                                                                                                \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring16_persist<2, NJV, E, C>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, r16_lds_bytes(2, NJV)); \
    GPS_REQUIRE(attr == hipSuccess, "gps_gemm_panel: cannot reserve LDS");                                            \
    k_gemm_ring16_persist<2, NJV, E, C><<<(unsigned)cus, NTHREADS, r16_lds_bytes(2, NJV), s>>>(P, (int)grid);         \
  } while (0)
#define GPS_RING_PERSIST_E(E)                                                                                         \
  do {                                                                                                                \
    if (nj == 3) { if (Cin) GPS_RING_PERSIST(3, E, true); else GPS_RING_PERSIST(3, E, false); }                       \
    else { if (Cin) GPS_RING_PERSIST(2, E, true); else GPS_RING_PERSIST(2, E, false); }                               \
  } while (0)
    if (epilogue == 0) GPS_RING_PERSIST_E(0);
    else if (epilogue == 1) GPS_RING_PERSIST_E(1);
    else GPS_RING_PERSIST_E(2);
#undef GPS_RING_PERSIST_E
#undef GPS_RING_PERSIST
    return gps::launch_status("gps_gemm_panel");
  }
  if (epilogue == 0) { if (Cin) GPS_PANEL_LAUNCH(0, true); else GPS_PANEL_LAUNCH(0, false); }
  else if (epilogue == 1) { if (Cin) GPS_PANEL_LAUNCH(1, true); else GPS_PANEL_LAUNCH(1, false); }
  else if (epilogue == 2) { if (Cin) GPS_PANEL_LAUNCH(2, true); else GPS_PANEL_LAUNCH(2, false); }
  else {      // (never an edge shape: required above)
    if (nj == 3) { if (mb == 2) GPS_RING_LAUNCH(2, 3, 3, true); else GPS_RING_LAUNCH(1, 3, 3, true); }
    else if (nj == 2) { if (mb == 2) GPS_RING_LAUNCH(2, 2, 3, true); else GPS_RING_LAUNCH(1, 2, 3, true); }
    else GPS_RING_LAUNCH(1, 1, 3, true);
  }
#undef GPS_RING_EDGE
#undef GPS_RING_LAUNCH
#undef GPS_RING_ANY
#undef GPS_RING_ANY_T
#undef GPS_RING_SHAPES
#undef GPS_PANEL_LAUNCH
  return gps::launch_status("gps_gemm_panel");
}

extern "C" int gps_gemm16_panel_pair(const gps_gemm16_problem* first, const gps_gemm16_problem* second, gps_stream_t stream) {
  GPS_REQUIRE(first && second, "gps_gemm16_panel_pair: null problem");
  const gps_gemm16_problem* pr[2] = {first, second};
  PanelPlan Q[2];
  bool live[2];
  for (int i = 0; i < 2; ++i) {
    const gps_gemm16_problem& p = *pr[i];
    GPS_REQUIRE(p.M >= 0 && p.a_amax && p.w_amax, "gps_gemm16_panel_pair: problem %d: rows / operand maxima", i);
    live[i] = p.M > 0;
    if (!live[i]) continue;
    if (int rc = panel_prepare(Q[i], p.A, p.lda, p.M, p.K, p.image, p.N, p.bias, p.Cin, p.ldcin, p.C, p.ldc, 0, nullptr, 0, 0.0f, 0,
                               nullptr, nullptr, nullptr, p.a_amax, p.w_amax, p.c_amax, nullptr)) return rc;
  }
  // one dispatch when both problems take the same non-edge fp16-form instantiation family (column panel width, addend or not)
  const bool paired = live[0] && live[1] && !Q[0].edge && !Q[1].edge && Q[0].nj == Q[1].nj && Q[0].nj >= 2 &&
                      (Q[0].P.Cin != nullptr) == (Q[1].P.Cin != nullptr);
  if (!paired) {
    for (int i = 0; i < 2; ++i) {
      const gps_gemm16_problem& p = *pr[i];
      if (!live[i]) continue;
      if (int rc = panel_launch(p.A, p.lda, p.M, p.K, p.image, p.N, p.bias, p.Cin, p.ldcin, p.C, p.ldc, 0, nullptr, 0, 0.0f, 0,
                                nullptr, nullptr, 0, nullptr, stream, p.a_amax, p.w_amax, p.c_amax)) return rc;
    }
    return GPS_OK;
  }
  hipStream_t s = gps::as_stream(stream);
  // Tail balancing (k_gemm_ring16_pair TAIL): when both problems run 128-row tiles and the tile count leaves a last dispatch
  // round at most half full, the rows behind that round's tiles -- the last row tiles of the second problem -- are cut as
  // 64-row tiles instead, so the launch ends half a tile-time earlier.  Same products, same per-element arithmetic: results
  // are bit-identical to the un-balanced dispatch (a tile's rows do not interact).
  const int cus = ring_cus(), sched = ring_sched();
  const bool tail_on = (sched & 1) != 0;
  PanelArgs P2 = Q[1].P;
  unsigned grid2 = 0;
  if (tail_on && cus > 0 && Q[0].mb == 2 && Q[1].mb == 2) {
    const int panels1 = Q[1].P.Nimg / (64 * Q[1].nj);
    const int rem = (int)((Q[0].grid + Q[1].grid) % (unsigned)cus);
    const int r = (rem + panels1 - 1) / panels1;                     // 128-row tiles of problem 1 behind the last round
    if (rem > 0 && 2 * rem <= cus && r >= 1 && r < Q[1].P.row_tiles) {
      const int keep = Q[1].P.row_tiles - r;
      const int64_t row0 = (int64_t)keep * 128, rows2 = Q[1].P.M - row0;
      P2.A = Q[1].P.A + row0 * Q[1].P.lda;
      P2.C = Q[1].P.C + row0 * Q[1].P.ldc;
      if (Q[1].P.Cin) P2.Cin = Q[1].P.Cin + row0 * Q[1].P.ldcin;
      P2.M = rows2;
      P2.row_tiles = (int)((rows2 + 63) / 64);
      grid2 = (unsigned)(P2.row_tiles * panels1);
      Q[1].P.M = row0;
      Q[1].P.row_tiles = keep;
      Q[1].grid = (unsigned)(keep * panels1);
    }
  }
  // Persistent first problem (ring16_body PERSIST): when it spans two dispatch rounds or more, `cus` workgroups walk its tiles
  // and prefetch across them.  128-row tiles only (the instantiations built), whole rotations of the (4-slot) ring.
  const bool persist_on = (sched & 2) != 0;
  const int tiles0 = (int)Q[0].grid;
  const int ks0 = (Q[0].P.K + BK - 1) / BK;
  const bool persist = persist_on && cus > 0 && Q[0].mb == 2 && Q[1].mb == 2 && tiles0 >= 2 * cus &&
                       ks0 % r16_persist_slots(2, Q[0].nj) == 0 && !Q[0].P.trace;
  const unsigned wg0 = persist ? (unsigned)cus : Q[0].grid;
  const unsigned grid = wg0 + Q[1].grid + grid2;
  const int split = (int)wg0, split2 = (int)(wg0 + Q[1].grid);
#define GPS_PAIR(MB0, MB1, NJV, C, T, PS)                                                                           \
  do {                                                                                                              \
    constexpr int LDS = r16_lds_bytes(MB0, NJV) > r16_lds_bytes(MB1, NJV) ? r16_lds_bytes(MB0, NJV) : r16_lds_bytes(MB1, NJV); \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring16_pair<MB0, MB1, NJV, C, T, PS>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);            \
    GPS_REQUIRE(attr == hipSuccess, "gps_gemm16_panel_pair: cannot reserve %d bytes of LDS", LDS);                  \
    k_gemm_ring16_pair<MB0, MB1, NJV, C, T, PS><<<grid, NTHREADS, LDS, s>>>(Q[0].P, Q[1].P, split, P2, split2, tiles0); \
  } while (0)
#define GPS_PAIR_MB(NJV, C)                                                                                         \
  do {                                                                                                              \
    if (Q[0].mb == 2) {                                                                                             \
      if (Q[1].mb == 2) {                                                                                           \
        if (persist) { if (grid2) GPS_PAIR(2, 2, NJV, C, true, true); else GPS_PAIR(2, 2, NJV, C, false, true); }   \
        else { if (grid2) GPS_PAIR(2, 2, NJV, C, true, false); else GPS_PAIR(2, 2, NJV, C, false, false); }         \
      }                                                                                                             \
      else GPS_PAIR(2, 1, NJV, C, false, false);                                                                    \
    }                                                                                                               \
    else { if (Q[1].mb == 2) GPS_PAIR(1, 2, NJV, C, false, false); else GPS_PAIR(1, 1, NJV, C, false, false); }     \
  } while (0)
  const bool cin = Q[0].P.Cin != nullptr;
  if (Q[0].nj == 3) { if (cin) GPS_PAIR_MB(3, true); else GPS_PAIR_MB(3, false); }
  else { if (cin) GPS_PAIR_MB(2, true); else GPS_PAIR_MB(2, false); }
#undef GPS_PAIR_MB
#undef GPS_PAIR
  return gps::launch_status("gps_gemm16_panel_pair");
}

// ---- BatchNorm-backward column sums out of an input-gradient GEMM's epilogue (ABI v10) ----------------------------------------
static int colsums_tiles(int64_t M, int N, int K) { const int mb = ring_mb(M, N, K); return (int)((M + 64 * mb - 1) / (64 * mb)); }
extern "C" int gps_gemm_colsums_supported(int64_t M, int N, int K) {
  if (!gps_gemm_panel_supported(N, K) || rg_edge(N, K) || M < 2 || rg_nj(N, K) < 2) return 0;    // whole panels of 128 / 192 columns
  return colsums_tiles(M, N, K) <= tr::kMaxParts;
}
extern "C" size_t gps_gemm_colsums_floats(int64_t M, int N, int K) {
  if (!gps_gemm_colsums_supported(M, N, K)) return 0;
  const int tn = 64 * rg_nj(N, K);
  return (size_t)(N / tn) * (tr::floats_for(colsums_tiles(M, N, K), 3, tn) + 16);
}
extern "C" int gps_gemm16_panel_sums(const gps_gemm16_problem* prob, const gps_gemm_colsums* sums, gps_stream_t stream) {
  const char* who = "gps_gemm16_panel_sums";
  GPS_REQUIRE(prob && sums && prob->a_amax && prob->w_amax && prob->M >= 0, "%s: null problem / sums / operand maxima", who);
  GPS_REQUIRE(prob->Cin, "%s: the input gradient accumulates onto an addend (Cin)", who);
  const gps_gemm16_problem& p = *prob;
  const gps_gemm_colsums* cs = sums;
  GPS_REQUIRE(gps_gemm_colsums_supported(p.M, p.N, p.K), "%s: shape M=%lld N=%d K=%d not served (whole 128 / 192-column panels, M >= 2)",
              who, (long long)p.M, p.N, p.K);
  GPS_REQUIRE(cs->z && cs->ldz >= p.N && cs->bn && cs->bn->mean && cs->bn->rstd && cs->sum_g && cs->sum_gz &&
              cs->z2 && cs->ldz2 >= p.N && cs->bn2 && cs->bn2->mean && cs->bn2->rstd && cs->sum_g2 && cs->sum_gz2 &&
              cs->ws && cs->sync && al16(cs->ws), "%s: incomplete gps_gemm_colsums", who);
  GPS_REQUIRE(cs->ws_floats >= gps_gemm_colsums_floats(p.M, p.N, p.K), "%s: workspace too small (gps_gemm_colsums_floats)", who);
  PanelPlan Q;
  if (int rc = panel_prepare(Q, p.A, p.lda, p.M, p.K, p.image, p.N, p.bias, p.Cin, p.ldcin, p.C, p.ldc, 4, nullptr, 0, 0.0f, 0,
                             nullptr, nullptr, nullptr, p.a_amax, p.w_amax, p.c_amax, nullptr, sums)) return rc;
  hipStream_t s = gps::as_stream(stream);
#define GPS_SUMS_ONE(MBV, NJV)                                                                                        \
  do {                                                                                                                \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_ring16<MBV, NJV, 4, true>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, r16_lds_bytes(MBV, NJV)); \
    GPS_REQUIRE(attr == hipSuccess, "%s: cannot reserve LDS", who);                                                   \
    k_gemm_ring16<MBV, NJV, 4, true><<<Q.grid, NTHREADS, r16_lds_bytes(MBV, NJV), s>>>(Q.P);                          \
  } while (0)
  if (Q.nj == 3) { if (Q.mb == 2) GPS_SUMS_ONE(2, 3); else GPS_SUMS_ONE(1, 3); }
  else { if (Q.mb == 2) GPS_SUMS_ONE(2, 2); else GPS_SUMS_ONE(1, 2); }
#undef GPS_SUMS_ONE
  return gps::launch_status(who);
}
