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
#include "col_tree.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  const uint32_t* g_amax;   // fp16 form (k_wgrad_stream<true>): max|g| / max|x| words (csrc/gemm_panel.hip gps_absmax), or null
  const uint32_t* x_amax;
};

struct Group {
  Problem p[kMaxGroup];
  int n;
  int xcd_map;   // k_wgrad_stream: work items dealt to workgroups XCD by XCD (below)
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


// ------------------------------------------------------------------------------------------------------------
// Streaming kernel (M and Nn multiples of 128, the shapes of the GPS block): same (tile, row-slice) work items, same
// arithmetic, rebuilt around what rocprofv3 showed for k_wgrad: MFMA pipe 28 % busy with 70k VALU instructions per
// wave -- four waves of 64 x 64 each split two g and two x fragments per k-step (176 VALU) for 24 MFMAs, i.e. 7 VALU
// per MFMA issue gap that hides ~5, and every value was split by two waves.
//   * split-K INSIDE the workgroup: each of the 4 waves owns the whole 128 x 128 tile (16 accumulators, 256 AGPRs) and
//     every fourth 16-row k-step of the slice.  A value is split exactly once per workgroup: 8 fragments (352 VALU) per
//     96 MFMAs = 3.7 per gap.  The four partial tiles are summed through LDS in wave order at the end (deterministic);
//   * a wave stages ONLY its own rows: 16 rows x 128 columns of g and of x per stage arrive by LDS-DMA
//     (global_load_lds_dwordx4, 16 transfers per wave and stage) in a private two-slot ring, so the k-loop has no
//     barrier at all -- the wave's own counted vmcnt orders its DMA against its own reads;
//   * the stage is software-pipelined by hand over four groups of 24 MFMAs (4 accumulators rotating in each, so no MFMA
//     waits on its predecessor): while a group multiplies, the fragments of the next group -- and in the second half of
//     the stage those of the next stage -- are read (ds_read_b32 down a column: the contraction index is the slow one)
//     and split in instalments of <= 4 VALU dealt into the MFMA issue gaps; sched_barrier(0) pins the interleave;
//   * the g half of a slot is refilled as soon as its last fragment has been read (group 1), the x half in group 2:
//     a transfer has more than a full stage (3072 cycles of MFMA issue) to land.
// Rows of the last slice beyond a multiple of 64 go through one extra, unpipelined stage whose DMA sources are selected
// per lane (g rows past the end read a zero row).
// ------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int WS_ROWS = 16;                     // rows per wave and stage: one 16-wide k-step
constexpr int WS_STAGE = 4 * WS_ROWS;           // rows per workgroup stage
constexpr int WS_PART = WS_ROWS * 128 * 4;      // 8 KB: 16 rows x 128 fp32 of one operand
constexpr int WS_SLOT = 2 * WS_PART;            // g | x
constexpr int WS_WAVE = 2 * WS_SLOT;            // two slots per wave
constexpr int WS_LDS = 4 * WS_WAVE;             // 128 KB

__device__ float g_zero_row[128];

struct Pieces {
  u32x4 p[3];      // one fragment as bf16 pairs: hi, mid, lo
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((gbl_void_ptr)g, (lds_void_ptr)l, 16, 0, 0);
}
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

// F16 (round 4): the arithmetic of csrc/gemm_panel.hip's k_gemm_ring16 -- each operand value as two fp16 pieces of its
// scaled value (scale = the power of two that puts the operand TENSOR's max|.| word in [2^14, 2^15)), three piece products
// per value product on v_mfma_f32_32x32x16_f16: 48 MFMAs per stage instead of 96 and 8 VALU per value pair instead of 11.
// The groups keep their shape (4 accumulators rotating, the same fills in the same order), two fill slots per issue gap.
using gps::amax_be;
__device__ __forceinline__ uint32_t cvt_pk_f16_rne(float a, float b) {       // v_cvt_pk_f16_f32 (gfx950): round to nearest even
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2));
}
template <bool F16>
__global__ __launch_bounds__(256, 1) void k_wgrad_stream(const Group G) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  // Work item of this workgroup.  Items are numbered (problem, row slice, g-column tile, x-column tile), x fastest: items
  // that are neighbours in that order read the same 128-column slab of g (and one of the few slabs of x) over the same
  // rows.  Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD with its own L2: dealt out in launch order,
  // the three tiles of one g slab land on three different XCDs and every slab crosses the fabric once per tile (PMC,
  // round 2: 844 MB per launch against 233 MB of operands, 5.4 TB/s at the fabric -- as close to that ceiling as to the
  // issue bound).  Instead XCD c takes the CONTIGUOUS c-th eighth of the item list: the sharers of a slab run at the
  // same time on the same L2.  A bijection on [0, gridDim): one workgroup per CU and the same count per XCD as
  // before (the grouping tried in round 1 changed the count per XCD and paid a second dispatch round for it).
  int bid = blockIdx.x;
  if (G.xcd_map) {
    const int n = gridDim.x, q = n >> 3, r = n & 7, c = bid & 7;
    bid = c * q + min(c, r) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < G.n && bid >= G.p[i].block_begin) pi = i;
  const Problem& P = G.p[pi];
  const int local = bid - P.block_begin;
  const int tiles = P.tiles_m * P.tiles_n;
  const int slice = local / tiles;
  if (slice >= P.S) return;
  const int tile = local - slice * tiles;
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int64_t r_begin = (int64_t)slice * P.rows_per_slice;
  const int64_t r_end = min(P.R, r_begin + P.rows_per_slice);
  const int NS = (int)((r_end - r_begin) / WS_STAGE);           // full stages
  const int tail = (int)((r_end - r_begin) - (int64_t)NS * WS_STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, kh = lane >> 5;
  unsigned char* const my = lds + wave * WS_WAVE;

  // DMA: transfer i (0..7) of an operand = rows 2i, 2i+1 of the wave's 16; lane -> row 2i + (lane >> 5), chunk lane & 31
  const int64_t gstage = (int64_t)WS_STAGE * P.ldg * 4, xstage = (int64_t)WS_STAGE * P.ldx * 4;
  const int64_t grow2 = 2 * P.ldg * 4, xrow2 = 2 * P.ldx * 4;
  const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(P.g + (r_begin + wave * WS_ROWS + kh) * P.ldg + m0) + li * 16;
  const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(P.x + (r_begin + wave * WS_ROWS + kh) * P.ldx + n0) + li * 16;
  auto dma_g = [&](int i, int stage, unsigned char* slot) __attribute__((always_inline)) {
    glds16(gsrc + stage * gstage + i * grow2, slot + i * 1024);
  };
  auto dma_x = [&](int i, int stage, unsigned char* slot) __attribute__((always_inline)) {
    glds16(xsrc + stage * xstage + i * xrow2, slot + WS_PART + i * 1024);
  };
  // a fragment's raw values: lane (li, kh) takes rows 8 kh .. 8 kh + 7 of its column (block blk of 32 columns)
  const int frag_off = (8 * kh) * 512 + li * 4;
  auto read2 = [&](const unsigned char* part, int blk, int j, float (&v)[8]) __attribute__((always_inline)) {
    v[j] = *reinterpret_cast<const float*>(part + frag_off + j * 512 + blk * 128);
    v[j + 1] = *reinterpret_cast<const float*>(part + frag_off + (j + 1) * 512 + blk * 128);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // exact 3-way split of values (2d, 2d+1) of a fragment in three instalments (4 + 4 + 3 VALU); `bs` (g fragments only)
  // accumulates the bias gradient, scaled by `bw` (0 for the prefetch that runs past the last stage)
  float sv0, sv1, sr0, sr1, st_s0, st_s1;
  f32x2 st_hb;
  uint32_t st_hi;
  const unsigned beg = F16 ? amax_be(P.g_amax) : 127u, bex = F16 ? amax_be(P.x_amax) : 127u;
  const float scg = __uint_as_float((268u - beg) << 23), scx = __uint_as_float((268u - bex) << 23);     // 2^(141 - be)
  auto split_part = [&](int part, const float (&raw)[8], int d, Pieces& out, float* bs, float bw) __attribute__((always_inline)) {
    if constexpr (F16) {
      // scalar form on purpose: the two values of a pair come out of two LDS words 512 bytes apart and the compiler's
      // load merger does not always deliver them as one aligned register pair -- written on float2 it emitted v_mov gathers
      // in front of v_pk_mul / v_pk_fma for half of the pairs (9.9 VALU per pair in the ISA); scalars take any registers
      const float v0 = raw[2 * d], v1 = raw[2 * d + 1];
      const float sc = bs ? scg : scx;                       // (g fragments carry the bias-gradient accumulator)
      if (part == 0) {
        if (bs) *bs = fmaf(v0 + v1, bw, *bs);
        st_s0 = v0 * sc;                                     // exact: sc is a power of two
        st_s1 = v1 * sc;
        st_hi = cvt_pk_f16_rne(st_s0, st_s1);
        out.p[0][d] = st_hi;
      } else if (part == 1) {
        st_hb = __builtin_convertvector(__builtin_bit_cast(f16x2, st_hi), f32x2);
      } else {
        out.p[1][d] = cvt_pk_f16_rne(st_s0 - st_hb[0], st_s1 - st_hb[1]);
      }
    } else if (part == 0) {
      sv0 = raw[2 * d];
      sv1 = raw[2 * d + 1];
      if (bs) *bs = fmaf(sv0 + sv1, bw, *bs);
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
  auto split_all = [&](const float (&raw)[8], Pieces& out, float* bs, float bw) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int part = 0; part < 3; ++part) split_part(part, raw, d, out, bs, bw);
  };
  constexpr int NT = F16 ? 3 : 6;                                          // piece products per value product
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};   // smallest terms first
  constexpr int TA16[3] = {1, 0, 0}, TB16[3] = {0, 1, 0};                 // lo*hi, hi*lo, hi*hi
  auto mfma = [&](const Pieces& a, const Pieces& b, int term, f32x16& c) __attribute__((always_inline)) {
    if constexpr (F16)
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[TA16[term]]), __builtin_bit_cast(f16x8, b.p[TB16[term]]), c, 0, 0, 0);
    else
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.p[TA[term]]), __builtin_bit_cast(bf16x8, b.p[TB[term]]), c, 0, 0, 0);
  };

  Pieces A01[2][2], A23[2], B[4];
  float rawA23[2][8], rawB23[2][8], rawA01[2][8], rawB01[2][8];
  unsigned char* const slot0 = my;
  unsigned char* const slot1 = my + WS_SLOT;

  if (NS > 0) {
    // prologue: stages 0 and 1 (stage 1 = stage 0 again when there is only one: harmless)
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_g(i, 0, slot0);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_x(i, 0, slot0);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_g(i, min(1, NS - 1), slot1);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_x(i, min(1, NS - 1), slot1);
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(16, 15));        // stage 0 has landed (this wave's own rows: no barrier)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        read2(slot0, f, j, rawA01[f]);
        read2(slot0 + WS_PART, f, j, rawB01[f]);
        read2(slot0, 2 + f, j, rawA23[f]);
      }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      split_all(rawA01[f], A01[0][f], &bsum[f], 1.0f);
      split_all(rawB01[f], B[f], nullptr, 0.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // One group = 24 MFMAs on 4 accumulators (pairs (ia, jb) given per call), `fill(m)` = what goes into the gap after MFMA m
#define WS_GROUP(AI0, AI1, AI2, AI3, J0, J1, J2, J3, I0, I1, I2, I3, FILL)                              \
  _Pragma("unroll") for (int m = 0; m < 4 * NT; ++m) {                                                   \
    const int term = m >> 2, pr = m & 3;                                                                \
    if (pr == 0) mfma(AI0, B[J0], term, acc[I0][J0]);                                                   \
    if (pr == 1) mfma(AI1, B[J1], term, acc[I1][J1]);                                                   \
    if (pr == 2) mfma(AI2, B[J2], term, acc[I2][J2]);                                                   \
    if (pr == 3) mfma(AI3, B[J3], term, acc[I3][J3]);                                                   \
    if constexpr (F16) { FILL(2 * m); FILL(2 * m + 1); } else { FILL(m); }   /* 24 fill slots per group either way */ \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }

  // Stage s in slot CUR (parity C of the A0/A1 piece sets); NXT holds stage s+1.
  //   group 1: pairs (0,0)(1,0)(0,1)(1,1) | read x blocks 2,3 of s; refill the g half of CUR with stage s+2; split A2, A3
  //   group 2: pairs (2,0)(2,1)(3,0)(3,1) | split B2, B3; refill the x half of CUR; wait for stage s+1; read its A0, A1
  //   group 3: pairs (0,2)(1,2)(0,3)(1,3) | read B0, B1 of s+1; split A0, A1 of s+1 (into the other piece set)
  //   group 4: pairs (2,2)(3,2)(2,3)(3,3) | read A2, A3 of s+1; split B0, B1 of s+1 (B[0], B[1] are free after group 2)
#define WS_STAGE_BODY(S, C, CUR, NXT)                                                                     \
  {                                                                                                       \
    const int st2 = min((S) + 2, NS - 1);                                                                 \
    const float bw_next = (S) + 1 < NS ? 1.0f : 0.0f;                                                     \
    auto fill1 = [&](int m) __attribute__((always_inline)) {                                              \
      if (m < 8) read2(CUR + WS_PART, 2 + m / 4, 2 * (m % 4), rawB23[m / 4]);                              \
      if (m >= 8 && m < 16) dma_g(m - 8, st2, CUR);                                                       \
      split_part(m % 3, rawA23[m / 12], (m % 12) / 3, A23[m / 12], &bsum[2 + m / 12], 1.0f);               \
    };                                                                                                    \
    WS_GROUP(A01[C][0], A01[C][1], A01[C][0], A01[C][1], 0, 0, 1, 1, 0, 1, 0, 1, fill1)                   \
    auto fill2 = [&](int m) __attribute__((always_inline)) {                                              \
      split_part(m % 3, rawB23[m / 12], (m % 12) / 3, B[2 + m / 12], nullptr, 0.0f);                      \
      if (m == 8) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));   /* lgkmcnt(0): every read of CUR's x half is done */ \
      if (m >= 8 && m < 16) dma_x(m - 8, st2, CUR);                                                       \
      if (m == 16) __builtin_amdgcn_s_waitcnt(waitcnt_imm(16, 15)); /* vmcnt(16): stage s+1 has landed */  \
      if (m >= 16) read2(NXT, (m - 16) / 4, 2 * ((m - 16) % 4), rawA01[(m - 16) / 4]);                     \
    };                                                                                                    \
    WS_GROUP(A23[0], A23[0], A23[1], A23[1], 0, 1, 0, 1, 2, 2, 3, 3, fill2)                               \
    auto fill3 = [&](int m) __attribute__((always_inline)) {                                              \
      if (m < 8) read2(NXT + WS_PART, m / 4, 2 * (m % 4), rawB01[m / 4]);                                  \
      split_part(m % 3, rawA01[m / 12], (m % 12) / 3, A01[(C) ^ 1][m / 12], &bsum[m / 12], bw_next);       \
    };                                                                                                    \
    WS_GROUP(A01[C][0], A01[C][1], A01[C][0], A01[C][1], 2, 2, 3, 3, 0, 1, 0, 1, fill3)                   \
    auto fill4 = [&](int m) __attribute__((always_inline)) {                                              \
      if (m < 8) read2(NXT, 2 + m / 4, 2 * (m % 4), rawA23[m / 4]);                                        \
      split_part(m % 3, rawB01[m / 12], (m % 12) / 3, B[m / 12], nullptr, 0.0f);                          \
    };                                                                                                    \
    WS_GROUP(A23[0], A23[0], A23[1], A23[1], 2, 3, 2, 3, 2, 2, 3, 3, fill4)                               \
  }

  for (int s = 0; s < NS; s += 2) {
    WS_STAGE_BODY(s, 0, slot0, slot1)
    if (s + 1 < NS) {
      WS_STAGE_BODY(s + 1, 1, slot1, slot0)
    }
  }
#undef WS_STAGE_BODY

  __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));            // every DMA of this wave has landed, every read returned
  if (tail > 0) {
    // the last slice's rows past a multiple of 64: one unpipelined stage with per-lane source selection
    const int64_t rt = r_begin + (int64_t)NS * WS_STAGE + wave * WS_ROWS + kh;     // row of transfer 0
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row) + li * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t r = rt + 2 * i;
      const bool ok = r < r_end;
      const unsigned char* gs = reinterpret_cast<const unsigned char*>(P.g + (ok ? r : 0) * P.ldg + m0) + li * 16;
      const unsigned char* xs = reinterpret_cast<const unsigned char*>(P.x + (ok ? r : r_end - 1) * P.ldx + n0) + li * 16;
      glds16(ok ? gs : zrow, slot0 + i * 1024);
      glds16(xs, slot0 + WS_PART + i * 1024);
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
    float ra[4][8], rb[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        read2(slot0, f, j, ra[f]);
        read2(slot0 + WS_PART, f, j, rb[f]);
      }
    Pieces TAp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      split_all(ra[f], TAp[f], &bsum[f], 1.0f);
      split_all(rb[f], B[f], nullptr, 0.0f);
    }
#pragma unroll
    for (int term = 0; term < NT; ++term)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma(TAp[i], B[j], term, acc[i][j]);
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
  }
#undef WS_GROUP

  // ---- the four waves' partial tiles summed through LDS, in wave order; two passes of 64 rows (8 blocks) each ------
  // layout [wave][block 0..7][lane][16 floats]: a lane re-reads exactly the slots the same lane of the other waves wrote
  float* const red = reinterpret_cast<float*>(lds);
  float* po = P.part + (int64_t)slice * P.M * P.Nn;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();                                        // LDS is free (k-loop reads done / previous pass read)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = 2 * pass + (b >> 2), j = b & 3;
      float* dst = red + ((wave * 8 + b) * 64 + lane) * 16;
#pragma unroll
      for (int q = 0; q < 16; q += 4)
        *reinterpret_cast<float4*>(dst + q) = make_float4(acc[i][j][q], acc[i][j][q + 1], acc[i][j][q + 2], acc[i][j][q + 3]);
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int b = 2 * wave + bb;                          // this wave sums blocks 2w, 2w+1 of the pass
      const int i = 2 * pass + (b >> 2), j = b & 3;
      float sum[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) sum[q] = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* src = red + ((w * 8 + b) * 64 + lane) * 16;
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          const float4 v = *reinterpret_cast<const float4*>(src + q);
          sum[q] += v.x; sum[q + 1] += v.y; sum[q + 2] += v.z; sum[q + 3] += v.w;
        }
      }
      const int col = n0 + j * 32 + li;
      // fp16 form: back from the scaled operands, two exact power-of-two factors (neither can leave the exponent range alone)
      const float ug = __uint_as_float((beg - 14u) << 23), ux = __uint_as_float((bex - 14u) << 23);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        const float v = F16 ? (sum[q] * ug) * ux : sum[q];
        po[(int64_t)row * P.Nn + col] = v;
      }
    }
  }
  if (P.bias_part && tn == 0) {
    __syncthreads();
    float* sc = reinterpret_cast<float*>(lds);              // [wave][kh][128 columns]
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[(wave * 2 + kh) * 128 + i * 32 + li] = bsum[i];
    __syncthreads();
    if (t < 128) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) a += sc[q * 128 + t];
      P.bias_part[(int64_t)slice * P.M + m0 + t] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Register path (round 6, fp16 form only).  The ablation of k_wgrad_stream (profiles/r06_wgrad_ablation.txt) priced its
// load path: of 115 us, the 16 global -> LDS transfers per wavefront and stage cost 43 and the 64 ds_read_b32 that turn LDS
// rows into MFMA fragments 34 -- the matrix loop itself 62.  A wavefront only ever reads back the rows it staged itself,
// so LDS was a transposition buffer, not a shared one.  Here the rows never touch LDS:
//   * lane (l, kh) loads, per stage, rows 8 kh .. 8 kh + 7 of its wavefront's 16 as eight 16-byte loads per operand at
//     columns 4 l .. 4 l + 3: two fully coalesced 512-byte rows per wave-instruction, 16 instructions per stage like the
//     16 transfers before -- but plain loads, whose issue costs a few cycles, into registers the split reads directly;
//   * the four values of a load ARE four fragments: fragment c of an operand is the column set {4 l + c}.  The contraction
//     does not care which lane carries which column as long as the output is stored accordingly: accumulator (a, b),
//     element (ri, li), is g column 4 ri + a times x column 4 li + b -- so a lane's four b-accumulators are four
//     NEIGHBOURING outputs and the partial tile leaves as 16-byte stores, whole 512-byte rows per wave-instruction;
//   * two register sets alternate by stage parity: stage s+2 is requested (group 3 of stage s) into the set stage s has
//     just finished with, one full stage before its first fragment is split; the compiler's own vmcnt accounting orders
//     the uses (plain loads: no counted waits to maintain by hand);
//   * the MFMA groups, the split's instalments in their issue gaps, the rotation of the piece sets and the cross-wavefront
//     sum are k_wgrad_stream's, so every output element is the same sum of the same products in the same order.
// Registers: 256 accumulators + 128 staging + 80 pieces: the file is full; one wavefront per SIMD.
// ------------------------------------------------------------------------------------------------------------
constexpr int WD_LDS = 4 * 4 * 64 * 16 * 4;       // 64 KB: [wave][b fragment][lane][16 floats] of one a-pass of the epilogue

__global__ __launch_bounds__(256, 1) void k_wgrad_direct(const Group G) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  int bid = blockIdx.x;
  if (G.xcd_map) {
    const int n = gridDim.x, q = n >> 3, r = n & 7, c = bid & 7;
    bid = c * q + min(c, r) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < G.n && bid >= G.p[i].block_begin) pi = i;
  const Problem& P = G.p[pi];
  const int local = bid - P.block_begin;
  const int tiles = P.tiles_m * P.tiles_n;
  const int slice = local / tiles;
  if (slice >= P.S) return;
  const int tile = local - slice * tiles;
  const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int64_t r_begin = (int64_t)slice * P.rows_per_slice;
  const int64_t r_end = min(P.R, r_begin + P.rows_per_slice);
  const int NS = (int)((r_end - r_begin) / WS_STAGE);           // full stages
  const int tail = (int)((r_end - r_begin) - (int64_t)NS * WS_STAGE);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, kh = lane >> 5;

  // row j (0..7) of stage st, this lane: r_begin + 64 st + 16 wave + 8 kh + j, columns 4 li .. 4 li + 3 of the tile's slab
  const float* const gp = P.g + (r_begin + wave * WS_ROWS + 8 * kh) * P.ldg + m0 + 4 * li;
  const float* const xp = P.x + (r_begin + wave * WS_ROWS + 8 * kh) * P.ldx + n0 + 4 * li;
  const int64_t gstage = (int64_t)WS_STAGE * P.ldg, xstage = (int64_t)WS_STAGE * P.ldx;
  f32x4 Rg[2][8], Rx[2][8];
#define WD_LOAD(PAR, ST, I)                                                                                        \
  do {                                                                                                             \
    if ((I) < 8) Rg[PAR][(I) & 7] = *reinterpret_cast<const f32x4*>(gp + (ST) * gstage + ((I) & 7) * P.ldg);       \
    else Rx[PAR][(I) & 7] = *reinterpret_cast<const f32x4*>(xp + (ST) * xstage + ((I) & 7) * P.ldx);               \
  } while (0)

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  float st_s0, st_s1;
  f32x2 st_hb;
  uint32_t st_hi;
  const unsigned beg = amax_be(P.g_amax), bex = amax_be(P.x_amax);
  const float scg = __uint_as_float((268u - beg) << 23), scx = __uint_as_float((268u - bex) << 23);     // 2^(141 - be)
  // the split of one pair of values in three instalments (k_wgrad_stream's fp16 form; scalar on purpose: 6 VALU per pair).
  // (Measured and dropped: the pair as four v_fma_mix{lo,hi}_f16 through inline asm -- 4 VALU per pair, bit-identical, but the
  // hazard s_nops the compiler puts around asm it cannot see into made the kernel 8 % SLOWER.)
  auto split_part = [&](int part, float v0, float v1, int d, Pieces& out, float* bs, float bw) __attribute__((always_inline)) {
    const float sc = bs ? scg : scx;
    if (part == 0) {
      if (bs) *bs = fmaf(v0 + v1, bw, *bs);
      st_s0 = v0 * sc;
      st_s1 = v1 * sc;
      st_hi = cvt_pk_f16_rne(st_s0, st_s1);
      out.p[0][d] = st_hi;
    } else if (part == 1) {
      st_hb = __builtin_convertvector(__builtin_bit_cast(f16x2, st_hi), f32x2);
    } else {
      out.p[1][d] = cvt_pk_f16_rne(st_s0 - st_hb[0], st_s1 - st_hb[1]);
    }
  };
  constexpr int NT = 3;
  constexpr int TA16[3] = {1, 0, 0}, TB16[3] = {0, 1, 0};                 // lo*hi, hi*lo, hi*hi
  auto mfma = [&](const Pieces& a, const Pieces& b, int term, f32x16& c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[TA16[term]]), __builtin_bit_cast(f16x8, b.p[TB16[term]]), c, 0, 0, 0);
  };

  Pieces A01[2][2], A23[2], B[4];

  if (NS > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) WD_LOAD(0, 0, i);
#pragma unroll
    for (int i = 0; i < 16; ++i) WD_LOAD(1, min(1, NS - 1), i);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          split_part(part, Rg[0][2 * d][f], Rg[0][2 * d + 1][f], d, A01[0][f], &bsum[f], 1.0f);
        }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int part = 0; part < 3; ++part) split_part(part, Rx[0][2 * d][f], Rx[0][2 * d + 1][f], d, B[f], nullptr, 0.0f);
    __builtin_amdgcn_sched_barrier(0);
  }

#define WD_GROUP(AI0, AI1, AI2, AI3, J0, J1, J2, J3, I0, I1, I2, I3, FILL)                              \
  _Pragma("unroll") for (int m = 0; m < 4 * NT; ++m) {                                                   \
    const int term = m >> 2, pr = m & 3;                                                                \
    if (pr == 0) mfma(AI0, B[J0], term, acc[I0][J0]);                                                   \
    if (pr == 1) mfma(AI1, B[J1], term, acc[I1][J1]);                                                   \
    if (pr == 2) mfma(AI2, B[J2], term, acc[I2][J2]);                                                   \
    if (pr == 3) mfma(AI3, B[J3], term, acc[I3][J3]);                                                   \
    FILL(2 * m); FILL(2 * m + 1);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }

  // Stage s, its rows in register set C (the next stage's in set C ^ 1):
  //   group 1: pairs (0,0)(1,0)(0,1)(1,1) | split A2, A3 of s
  //   group 2: pairs (2,0)(2,1)(3,0)(3,1) | split B2, B3 of s; request the g rows of stage s+2 into set C (free since group 1)
  //   group 3: pairs (0,2)(1,2)(0,3)(1,3) | request its x rows (free since group 2); split A0, A1 of s+1 (into the other piece set)
  //   group 4: pairs (2,2)(3,2)(2,3)(3,3) | split B0, B1 of s+1 (B[0], B[1] are free after group 2)
#define WD_STAGE_BODY(S, C)                                                                               \
  {                                                                                                       \
    const int st2 = min((S) + 2, NS - 1);                                                                 \
    const float bw_next = (S) + 1 < NS ? 1.0f : 0.0f;                                                     \
    auto fill1 = [&](int m) __attribute__((always_inline)) {                                              \
      const int f = 2 + m / 12, d = (m % 12) / 3;                                                         \
      split_part(m % 3, Rg[C][2 * d][f], Rg[C][2 * d + 1][f], d, A23[m / 12], &bsum[f], 1.0f);             \
    };                                                                                                    \
    WD_GROUP(A01[C][0], A01[C][1], A01[C][0], A01[C][1], 0, 0, 1, 1, 0, 1, 0, 1, fill1)                   \
    auto fill2 = [&](int m) __attribute__((always_inline)) {                                              \
      const int f = 2 + m / 12, d = (m % 12) / 3;                                                         \
      if (m < 8) WD_LOAD(C, st2, m);             /* the g rows of set C are free since group 1 */           \
      split_part(m % 3, Rx[C][2 * d][f], Rx[C][2 * d + 1][f], d, B[f], nullptr, 0.0f);                    \
    };                                                                                                    \
    WD_GROUP(A23[0], A23[0], A23[1], A23[1], 0, 1, 0, 1, 2, 2, 3, 3, fill2)                               \
    auto fill3 = [&](int m) __attribute__((always_inline)) {                                              \
      const int f = m / 12, d = (m % 12) / 3;                                                             \
      if (m < 8) WD_LOAD(C, st2, 8 + m);         /* ... its x rows since group 2 */                          \
      split_part(m % 3, Rg[(C) ^ 1][2 * d][f], Rg[(C) ^ 1][2 * d + 1][f], d, A01[(C) ^ 1][f], &bsum[f], bw_next); \
    };                                                                                                    \
    WD_GROUP(A01[C][0], A01[C][1], A01[C][0], A01[C][1], 2, 2, 3, 3, 0, 1, 0, 1, fill3)                   \
    auto fill4 = [&](int m) __attribute__((always_inline)) {                                              \
      const int f = m / 12, d = (m % 12) / 3;                                                             \
      split_part(m % 3, Rx[(C) ^ 1][2 * d][f], Rx[(C) ^ 1][2 * d + 1][f], d, B[f], nullptr, 0.0f);         \
    };                                                                                                    \
    WD_GROUP(A23[0], A23[0], A23[1], A23[1], 2, 3, 2, 3, 2, 2, 3, 3, fill4)                               \
  }

  for (int s = 0; s < NS; s += 2) {
    WD_STAGE_BODY(s, 0)
    if (s + 1 < NS) {
      WD_STAGE_BODY(s + 1, 1)
    }
  }
#undef WD_STAGE_BODY

  if (tail > 0) {
    // the last slice's rows past a multiple of 64: one unpipelined stage, rows past the end contribute zeros (g) / any
    // finite row (x: the last valid one)
    const int64_t rt = r_begin + (int64_t)NS * WS_STAGE + wave * WS_ROWS + 8 * kh;
    Pieces TAp[4];
    float ra[4][8], rb[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = rt + j;
      const bool ok = r < r_end;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(P.g + (ok ? r : r_end - 1) * P.ldg + m0 + 4 * li);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(P.x + (ok ? r : r_end - 1) * P.ldx + n0 + 4 * li);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        ra[f][j] = ok ? gv[f] : 0.0f;
        rb[f][j] = xv[f];
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          split_part(part, ra[f][2 * d], ra[f][2 * d + 1], d, TAp[f], &bsum[f], 1.0f);
        }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int part = 0; part < 3; ++part) split_part(part, rb[f][2 * d], rb[f][2 * d + 1], d, B[f], nullptr, 0.0f);
#pragma unroll
    for (int term = 0; term < NT; ++term)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma(TAp[i], B[j], term, acc[i][j]);
  }
#undef WD_GROUP
#undef WD_LOAD

  // ---- the four waves' partial tiles summed through LDS in wave order, one g fragment (a) per pass: [wave][b][lane][16].
  // Wave w finishes elements q = 4 w .. 4 w + 3 of all four b: its lane's four b-values of an element are the x columns
  // 4 li .. 4 li + 3 of g column 4 ri + a -- one 16-byte store, 512 contiguous bytes per output row and wave-instruction.
  float* const red = reinterpret_cast<float*>(lds);
  float* po = P.part + (int64_t)slice * P.M * P.Nn;
  const float ug = __uint_as_float((beg - 14u) << 23), ux = __uint_as_float((bex - 14u) << 23);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    __syncthreads();                                        // LDS is free (previous pass read)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float* dst = red + ((wave * 4 + b) * 64 + lane) * 16;
#pragma unroll
      for (int q = 0; q < 16; q += 4)
        *reinterpret_cast<float4*>(dst + q) = make_float4(acc[a][b][q], acc[a][b][q + 1], acc[a][b][q + 2], acc[a][b][q + 3]);
    }
    __syncthreads();
    float sum[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[b][e] = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(red + ((w * 4 + b) * 64 + lane) * 16 + 4 * wave);
        sum[b][0] += v.x; sum[b][1] += v.y; sum[b][2] += v.z; sum[b][3] += v.w;
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = 4 * wave + e;
      const int ri = (q & 3) + 8 * (q >> 2) + 4 * kh;
      const int row = m0 + 4 * ri + a;
      float4 o;
      o.x = (sum[0][e] * ug) * ux; o.y = (sum[1][e] * ug) * ux; o.z = (sum[2][e] * ug) * ux; o.w = (sum[3][e] * ug) * ux;
      *reinterpret_cast<float4*>(po + (int64_t)row * P.Nn + n0 + 4 * li) = o;
    }
  }
  if (P.bias_part && tn == 0) {
    __syncthreads();
    float* sc = reinterpret_cast<float*>(lds);              // [wave][kh][128 columns]
#pragma unroll
    for (int i = 0; i < 4; ++i) sc[(wave * 2 + kh) * 128 + 4 * li + i] = bsum[i];
    __syncthreads();
    if (t < 128) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) a += sc[q * 128 + t];
      P.bias_part[(int64_t)slice * P.M + m0 + t] = a;
    }
  }
}

// out[i] = sum_s part[s][i] in slice order; bias likewise, by the first threads of each problem.  VEC = 4 (round 5): a
// thread owns four neighbouring outputs and pulls each slice with one 16-byte load, all S of them independent -- the
// one-float form moved the 16 MB of partials of a GPS block at 1.3 TB/s (12.7 us per launch, rocprofv3 round 5), a
// quarter of the threads each with four times the bytes in flight is what the memory system wants.  Same additions in
// the same order per output: bit-identical.
template <int VEC>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const Group G) {
  const int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < G.n && gi >= G.p[i].out_begin) pi = i;
  const Problem& P = G.p[pi];
  const int64_t i = gi - P.out_begin;
  const int64_t total = (int64_t)P.M * P.Nn;
  if constexpr (VEC == 4) {
    if (i * 4 < total) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
      for (int s = 0; s < P.S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(P.part + (int64_t)s * total + i * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      *reinterpret_cast<float4*>(P.gw + i * 4) = a;
    }
  } else {
    if (i < total) {
      float a = 0.f;
#pragma unroll 4
      for (int s = 0; s < P.S; ++s) a += P.part[(int64_t)s * total + i];
      P.gw[i] = a;
    }
  }
  if (P.gb && i < P.M) {
    float a = 0.f;
#pragma unroll 4
    for (int s = 0; s < P.S; ++s) a += P.bias_part[(int64_t)s * P.M + i];
    P.gb[i] = a;
  }
}

// Slices for a problem so that one workgroup owns about `chunks_per_block` 32-row chunks.
inline void plan_slices(Problem& p, int64_t chunks_per_block, int quantum = BK) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Nn + BN - 1) / BN;
  const int64_t chunks = (p.R + BK - 1) / BK;
  int64_t S = (chunks + chunks_per_block - 1) / chunks_per_block;
  const int64_t max_s = (chunks + 3) / 4;            // at least 4 chunks per slice
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int64_t rps = (p.R + S - 1) / S;
  rps = (rps + quantum - 1) / quantum * quantum;
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
constexpr int kTargetBlocks = 256;
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

// the streaming kernel takes whole 128 x 128 tiles and 16-byte-aligned rows
inline bool stream_shapes(const int* M, const int* Nn, int n) {
  for (int i = 0; i < n; ++i)
    if (M[i] % 128 != 0 || Nn[i] % 128 != 0) return false;
  return true;
}
inline bool stream_ok(const Group& G) {
  int M[kMaxGroup], Nn[kMaxGroup];
  for (int i = 0; i < G.n; ++i) { M[i] = G.p[i].M; Nn[i] = G.p[i].Nn; }
  if (!stream_shapes(M, Nn, G.n)) return false;
  for (int i = 0; i < G.n; ++i)
    if (G.p[i].rows_per_slice % WS_STAGE != 0) return false;
  return true;
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
  }
  // the reduce takes four outputs per thread when every problem allows 16-byte accesses (always, for the GPS blocks)
  bool vec = true;
  for (int i = 0; i < G.n; ++i) {
    const Problem& p = G.p[i];
    vec = vec && ((int64_t)p.M * p.Nn) % 4 == 0 && p.Nn >= 4 && reinterpret_cast<uintptr_t>(p.part) % 16 == 0 &&
          reinterpret_cast<uintptr_t>(p.gw) % 16 == 0;
  }
  for (int i = 0; i < G.n; ++i) {
    Problem& p = G.p[i];
    p.out_begin = outs;
    const int64_t threads = vec ? (int64_t)p.M * p.Nn / 4 : (int64_t)p.M * p.Nn;
    outs += (threads + 255) / 256 * 256;               // whole reduce blocks per problem
  }
  if (stream_ok(G)) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_stream<false>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    static const hipError_t attr16 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_stream<true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    GPS_REQUIRE(attr == hipSuccess && attr16 == hipSuccess, "%s: cannot reserve %d bytes of LDS", who, WS_LDS);
    G.xcd_map = 1;          // (0: launch order -- the round-3 A/B, profiles/r03_pmc_wgrad_xcdmap0.txt)
    bool f16 = true;                 // every problem of the launch carries its operands' max|.| words
    for (int i = 0; i < G.n; ++i) f16 = f16 && G.p[i].g_amax && G.p[i].x_amax;
    static const bool direct = [] { const char* e = getenv("GPS_WGRAD_DIRECT"); return !(e && *e && atoi(e) == 0); }();
    if (f16 && direct) {
      static const hipError_t attrd = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_direct),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
      GPS_REQUIRE(attrd == hipSuccess, "%s: cannot reserve %d bytes of LDS", who, WD_LDS);
      k_wgrad_direct<<<(unsigned)blocks, 256, WD_LDS, s>>>(G);
    } else if (f16) k_wgrad_stream<true><<<(unsigned)blocks, 256, WS_LDS, s>>>(G);
    else k_wgrad_stream<false><<<(unsigned)blocks, 256, WS_LDS, s>>>(G);
  } else
    k_wgrad<true><<<(unsigned)blocks, 256, 0, s>>>(G);
  if (vec) k_wgrad_reduce<4><<<gps::grid_for(outs, 256), 256, 0, s>>>(G);
  else k_wgrad_reduce<1><<<gps::grid_for(outs, 256), 256, 0, s>>>(G);
  return gps::launch_status(who);
}

}  // namespace

extern "C" {

size_t gps_wgrad_workspace_floats(int64_t R, int M, int Nn) {
  if (R <= 0 || M <= 0 || Nn <= 0) return 0;
  Problem p{};
  p.R = R; p.M = M; p.Nn = Nn;
  plan_slices(p, balanced_chunks(&R, &M, &Nn, 1), stream_shapes(&M, &Nn, 1) ? WS_STAGE : BK);
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
  plan_slices(p, balanced_chunks(&R, &M, &Nn, 1), stream_shapes(&M, &Nn, 1) ? WS_STAGE : BK);
  return launch_group(G, ws, gps::as_stream(stream), "gps_wgrad");
}

int gps_wgrad16(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int Nn, const uint32_t* g_amax,
                const uint32_t* x_amax, float* gw, float* gb, float* ws, gps_stream_t stream) {
  if (int rc = check_problem("gps_wgrad16", g, ldg, x, ldx, R, M, Nn, gw)) return rc;
  GPS_REQUIRE(ws && reinterpret_cast<uintptr_t>(ws) % 16 == 0, "gps_wgrad16: workspace null/misaligned");
  GPS_REQUIRE(g_amax && x_amax, "gps_wgrad16: operand maxima");
  Group G{};
  G.n = 1;
  Problem& p = G.p[0];
  p.g = g; p.x = x; p.gw = gw; p.gb = gb; p.ldg = ldg; p.ldx = ldx; p.R = R; p.M = M; p.Nn = Nn;
  p.g_amax = g_amax; p.x_amax = x_amax;
  plan_slices(p, balanced_chunks(&R, &M, &Nn, 1), stream_shapes(&M, &Nn, 1) ? WS_STAGE : BK);
  return launch_group(G, ws, gps::as_stream(stream), "gps_wgrad16");
}

size_t gps_wgrad_grouped_workspace_floats(int n, const gps_wgrad_problem* probs) {
  if (n <= 0 || n > kMaxGroup || !probs) return 0;
  int64_t R[kMaxGroup];
  int M[kMaxGroup], Nn[kMaxGroup];
  for (int i = 0; i < n; ++i) { R[i] = probs[i].R; M[i] = probs[i].M; Nn[i] = probs[i].Nn; }
  const int64_t cpb = balanced_chunks(R, M, Nn, n);
  const int quantum = stream_shapes(M, Nn, n) ? WS_STAGE : BK;
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (R[i] <= 0 || M[i] <= 0 || Nn[i] <= 0) return 0;
    Problem p{};
    p.R = R[i]; p.M = M[i]; p.Nn = Nn[i];
    plan_slices(p, cpb, quantum);
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
  const int quantum = stream_shapes(M, Nn, n) ? WS_STAGE : BK;
  for (int i = 0; i < n; ++i) {
    const gps_wgrad_problem& q = probs[i];
    Problem& p = G.p[i];
    p.g = q.g; p.x = q.x; p.gw = q.gw; p.gb = q.gb; p.ldg = q.ldg; p.ldx = q.ldx;
    p.R = q.R; p.M = q.M; p.Nn = q.Nn;
    p.g_amax = q.g_amax; p.x_amax = q.x_amax;
    plan_slices(p, cpb, quantum);
  }
  return launch_group(G, ws, gps::as_stream(stream), "gps_wgrad_grouped");
}

}  // extern "C"
