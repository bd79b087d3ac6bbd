// Dense projections of the block, fp32 in / fp32 out, contraction on the bf16 MFMA pipe.
//
//     C[r][m] = sum_k A[r][k] * B[m][k]  (+ bias[m]) (+ Cin[r][m])
//
// i.e. y = x W^T + b for an nn.Linear weight W [out, in] (graphgps/layer/gatedgcn_layer.py:57-61,
// graphgps/layer/gps_layer.py:143-144,234-241,253-257) and, with B = W^T (a [in, out] copy), the input
// gradient g_x = g W of the same layers; `Cin` folds the residual accumulation the backward needs
// (g_h = g_z2 + g_f1 W1) into the epilogue.
//
// Arithmetic: both operands are split EXACTLY into three bf16 pieces (hi + mid + lo == value bit for
// bit: each piece takes the next 8 significant bits by truncation, remainders are exact fp32
// subtractions) and a*b is formed from 6 of the 9 piece products (hh, hm, mh, hl, lh, mm; the dropped
// ml, lm, ll are below 2^-21 |a||b|) with fp32 accumulation in v_mfma_f32_32x32x16_bf16.  Every partial
// product is exact in fp32 (8 x 8 significand bits), so the result carries the same rounding model as an
// fp32-input MFMA GEMM; measured error against fp64 is at or below rocBLAS' fp32 GEMM (tests).  The bf16
// pipe issues 16x the fp32-input MFMA rate, so 6 products are still 2.67x faster at the MFMA limit.
//
// Layout: both operands are contiguous along the contraction index, which is what the instruction's
// A/B fragments want (lane = row, 8 consecutive k), so nothing is ever transposed: 16-byte global loads
// (8 lanes x 16 B = one 128-byte line per row), split once per workgroup while staging, pieces stored in
// LDS as [piece][row][32 k] bf16 with an 80-byte row pitch (ds_read_b128 of 16 consecutive rows hits 16
// distinct 16-byte slots), fragments fetched with one ds_read_b128 each.  128 x 128 tile; 8 waves:
// 4 consumers of 64 x 64 (2 x 2 accumulators of 32 x 32) + 4 producers (loads, split, LDS stores) working
// one chunk ahead in the other LDS buffer.
// Status: parity-tested, NOT on the measured path -- at the block's shapes it matches, not beats, the
// TunableOp-selected library kernels: the 3-piece fragments cost 6 B of LDS read per element per use against
// 32-cycle MFMAs (64 B/clk/CU of the LDS' 128 before bank conflicts), i.e. the kernel is co-limited by LDS
// bandwidth and the split's VALU work, not by the MFMA pipe.
#include "gps_common.hpp"

// The GPS_ABL_* switches below time parts of this kernel with the rest removed (tools/micro/gemm_ablate.sh): their results
// are garbage by design, so a build that defines one must say so -- they can never slip into the library by a stray -D.
#if (defined(GPS_ABL_NO_BARRIER) || defined(GPS_ABL_NO_SPLIT) || defined(GPS_ABL_NO_GLOBAL) || defined(GPS_ABL_NO_LDSWRITE) || \
     defined(GPS_ABL_NO_LDSREAD) || defined(GPS_ABL_NO_MFMA)) && !defined(GPS_ABLATION_BUILD)
#error "GPS_ABL_* ablation switches produce wrong results: define GPS_ABLATION_BUILD as well (timing builds only)"
#endif

#ifdef GPS_ABL_NO_BARRIER   // ablation: no workgroup barriers (results are garbage, timing only)
#define GPS_BARRIER() do {} while (0)
#else
#define GPS_BARRIER() __syncthreads()
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, BK = 32;
constexpr int PITCH = 40;            // bf16 elements per LDS row (32 + 8 pad = 80 bytes)
constexpr int NPASS = TM / 64;       // staging passes: 256 threads = 64 rows x 4 units of 8 k per pass

struct Frag {
  uint32_t u[4];
};

// 8 consecutive fp32 -> three packed bf16x8 (16 bytes each) by TRUNCATION: hi = top 16 bits of v,
// r1 = v - hi (exact), mid = top 16 bits of r1, r2 = r1 - mid (exact, <= 8 significant bits) = lo exactly
// -- hi + mid + lo == v bit for bit.  Full-rate VALU only: v_and / v_pk_add_f32 / v_perm_b32 (the RNE
// variant went through v_cvt_pk_bf16_f32 and a shift/mask pair to convert back; the truncated pieces are
// one bit coarser, the dropped products ml, lm, ll stay below 2^-20 |a||b|).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t top16(float lo_elem, float hi_elem) {     // {bf16(lo_elem), bf16(hi_elem)}
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
#ifdef GPS_ABL_NO_SPLIT   // ablation (tools/micro/gemm_ablate.sh): no VALU split, pieces = raw bits
  hi = __float_as_uint(x0); mid = __float_as_uint(x1); lo = hi;
  return;
#endif
  constexpr uint32_t M = 0xffff0000u;
  hi = top16(x0, x1);
  const f32x2 r1 = f32x2{x0, x1} - f32x2{__uint_as_float(__float_as_uint(x0) & M),
                                         __uint_as_float(__float_as_uint(x1) & M)};
  mid = top16(r1[0], r1[1]);
  const f32x2 r2 = r1 - f32x2{__uint_as_float(__float_as_uint(r1[0]) & M),
                              __uint_as_float(__float_as_uint(r1[1]) & M)};
  lo = top16(r2[0], r2[1]);
}
// component-wise (a conditional between two float4 lvalues is a pointer select: it pins both in scratch)
__device__ __forceinline__ float4 keep_if(bool ok, const float4 v) {
  return make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
}
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, uint4& hi, uint4& mid, uint4& lo) {
  split2(v0.x, v0.y, hi.x, mid.x, lo.x);
  split2(v0.z, v0.w, hi.y, mid.y, lo.y);
  split2(v1.x, v1.y, hi.z, mid.z, lo.z);
  split2(v1.z, v1.w, hi.w, mid.w, lo.w);
}

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  const float* Cin;
  float* C;
  int64_t lda, ldb, ldc, ldcin, R;
  int M, K, tiles_m, tiles_n;
};

// Wave-specialised: waves 0-3 multiply (pure ds_read_b128 + MFMA stream), waves 4-7 produce (global loads,
// exact bf16 split, LDS stores) into the other LDS buffer; each SIMD hosts one of each, so the split's VALU
// work runs under the other wave's MFMAs instead of in a phase of its own.  One barrier per chunk.
constexpr int kLdsBuf = 2 * 3 * TM * PITCH;   // uint16 elements per buffer (A + B, 3 pieces)

__global__ __launch_bounds__(512) void k_gemm_nt(const GemmArgs G) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds_raw[];   // [2 buffers][operand][piece][row][PITCH]
  auto L = [&](int buf, int op, int pc, int row, int k) -> uint16_t* {
    return lds_raw + (size_t)buf * kLdsBuf + ((size_t)(op * 3 + pc) * TM + row) * PITCH + k;
  };
  // XCD-aware tile order: the 8 XCDs take workgroups round-robin, so give each XCD a contiguous run
  // of tiles (same A row-tile, consecutive B column tiles -> the A tile is re-read from that XCD's L2)
  const int ntiles = G.tiles_m * G.tiles_n;
  const int per_xcd = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int tm = tile / G.tiles_n, tn = tile - tm * G.tiles_n;
  const int64_t r0 = (int64_t)tm * TM;
  const int m0 = tn * TN;
  const int K = G.K;
  const int nchunks = (K + BK - 1) / BK;
  const bool producer = threadIdx.x >= 256;
  const int t = threadIdx.x & 255;

  if (producer) {
    // staging: thread -> (row = t/4 + 64*pass, 8 consecutive k = two 16-byte loads, one ds_write_b128 per
    // piece); 4 lanes cover one 128-byte line of a row
    const int srow = t >> 2, sk8 = (t & 3) * 8;
    const float* ap[NPASS];
    const float* bp[NPASS];
    bool a_ok[NPASS], b_ok[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int64_t ra = r0 + srow + 64 * p;
      const int rb = m0 + srow + 64 * p;
      a_ok[p] = ra < G.R;
      b_ok[p] = rb < G.M;
      ap[p] = G.A + (a_ok[p] ? ra : G.R - 1) * G.lda;
      bp[p] = G.B + (int64_t)(b_ok[p] ? rb : G.M - 1) * G.ldb;
    }
    // wave-uniform: interior tiles (every row valid, K a multiple of the chunk) skip the zero-fill selects
    const bool edge = (r0 + TM > G.R) || (m0 + TN > G.M) || (K % BK != 0);
    constexpr int NU = NPASS * 2;       // 16-byte loads per operand per chunk: [pass][half of the 8-k unit]
    float4 sa0[NU], sb0[NU], sa1[NU], sb1[NU];
    auto load_chunk = [&](int c, float4 (&sa)[NU], float4 (&sb)[NU]) {
      // raw loads only; a k-quad past K reads quad 0 and is zeroed at split time
      const int kq0 = c * BK + sk8, kq1 = kq0 + 4;
      const int kk0 = kq0 < K ? kq0 : 0, kk1 = kq1 < K ? kq1 : 0;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
#ifdef GPS_ABL_NO_GLOBAL  // ablation: operands from registers, no global traffic
        sa[2 * p] = make_float4(1.0f + kk0, 2.0f, 3.0f, 4.0f);
        sa[2 * p + 1] = make_float4(1.0f + kk1, 2.0f, 3.0f, 4.0f);
        sb[2 * p] = make_float4(0.5f, 0.25f + kk0, 0.125f, 1.0f);
        sb[2 * p + 1] = make_float4(0.5f, 0.25f + kk1, 0.125f, 1.0f);
#else
        sa[2 * p] = *reinterpret_cast<const float4*>(ap[p] + kk0);
        sa[2 * p + 1] = *reinterpret_cast<const float4*>(ap[p] + kk1);
        sb[2 * p] = *reinterpret_cast<const float4*>(bp[p] + kk0);
        sb[2 * p + 1] = *reinterpret_cast<const float4*>(bp[p] + kk1);
#endif
      }
    };
    auto store_chunk = [&](int c, int buf, const float4 (&sa)[NU], const float4 (&sb)[NU]) {
      const bool k0 = c * BK + sk8 < K, k1 = c * BK + sk8 + 4 < K;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        uint4 h, m, l;
        const int row = srow + 64 * p;
#ifdef GPS_ABL_NO_LDSWRITE  // ablation: staging registers consumed without touching LDS
        if (sa[2 * p].x == 123.456f && sb[2 * p + 1].y == 654.321f) *reinterpret_cast<uint4*>(L(buf, 0, 0, row, sk8)) = h;
        continue;
#endif
        float4 a0 = sa[2 * p], a1 = sa[2 * p + 1], b0 = sb[2 * p], b1 = sb[2 * p + 1];
        if (edge) {
          a0 = keep_if(a_ok[p] && k0, a0);
          a1 = keep_if(a_ok[p] && k1, a1);
          b0 = keep_if(b_ok[p] && k0, b0);
          b1 = keep_if(b_ok[p] && k1, b1);
        }
        split8(a0, a1, h, m, l);
        *reinterpret_cast<uint4*>(L(buf, 0, 0, row, sk8)) = h;
        *reinterpret_cast<uint4*>(L(buf, 0, 1, row, sk8)) = m;
        *reinterpret_cast<uint4*>(L(buf, 0, 2, row, sk8)) = l;
        split8(b0, b1, h, m, l);
        *reinterpret_cast<uint4*>(L(buf, 1, 0, row, sk8)) = h;
        *reinterpret_cast<uint4*>(L(buf, 1, 1, row, sk8)) = m;
        *reinterpret_cast<uint4*>(L(buf, 1, 2, row, sk8)) = l;
      }
    };
    // The global loads are UNCONDITIONAL (past the last chunk they re-read it): the compiler's s_waitcnt
    // bookkeeping merges control-flow paths conservatively, and a load that is issued on one path only
    // collapses the vmcnt it allows at the next use to ~0, i.e. the chunk c+2 prefetch would be waited
    // for before chunk c+1 is even staged.  With a path-independent count the wait is vmcnt(8): exactly
    // the loads of the chunk being staged, the newer 8 stay in flight across the barrier.
    const int last = nchunks - 1;
    load_chunk(0, sa0, sb0);
    load_chunk(min(1, last), sa1, sb1);
    store_chunk(0, 0, sa0, sb0);
    GPS_BARRIER();                                   // buffer 0 ready
    for (int c = 0; c < nchunks; c += 2) {
      // consumers multiply chunk c (buffer 0): stage chunk c+1 into buffer 1, fetch chunk c+2
      load_chunk(min(c + 2, last), sa0, sb0);
      if (c + 1 < nchunks) store_chunk(c + 1, 1, sa1, sb1);
      GPS_BARRIER();
      if (c + 1 >= nchunks) break;
      // consumers multiply chunk c+1 (buffer 1): stage chunk c+2 into buffer 0, fetch chunk c+3
      load_chunk(min(c + 3, last), sa1, sb1);
      if (c + 2 < nchunks) store_chunk(c + 2, 0, sa0, sb0);
      GPS_BARRIER();
    }
    return;
  }

  // ---- consumers --------------------------------------------------------------------------------
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
  // Fragment double buffering at the K16-step level: the ds_read_b128s of step s+1 are in flight while
  // the 24 MFMAs of step s run, so the wave never sits on LDS latency with an idle matrix pipe.  One
  // barrier per chunk, placed after the LAST read of the buffer (it releases that buffer to the producers
  // and, by their arrival, publishes the other one).
  struct Frags {
    Frag A[2][3], B[2][3];
  };
  auto fetch = [&](int buf, int ks, Frags& F) {
#ifdef GPS_ABL_NO_LDSREAD  // ablation: fragments from registers, no LDS reads
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          F.A[i][pc].u[q] = 0x3f803f80u + (uint32_t)(buf + pc + i + q + lane);
          F.B[i][pc].u[q] = 0x3f803f80u + (uint32_t)(buf + pc + i + q);
        }
#else
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 a = *reinterpret_cast<const uint4*>(L(buf, 0, pc, wm * 64 + i * 32 + li, 16 * ks + 8 * kh));
        const uint4 b = *reinterpret_cast<const uint4*>(L(buf, 1, pc, wn * 64 + i * 32 + li, 16 * ks + 8 * kh));
        F.A[i][pc].u[0] = a.x; F.A[i][pc].u[1] = a.y; F.A[i][pc].u[2] = a.z; F.A[i][pc].u[3] = a.w;
        F.B[i][pc].u[0] = b.x; F.B[i][pc].u[1] = b.y; F.B[i][pc].u[2] = b.z; F.B[i][pc].u[3] = b.w;
      }
#endif
  };
  auto mma = [&](const Frags& F) {
    // smallest terms first; the four accumulators rotate so no MFMA waits on its predecessor
    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 av, bv;
          __builtin_memcpy(&av, &F.A[i][TA[term]], 16);
          __builtin_memcpy(&bv, &F.B[j][TB[term]], 16);
#ifdef GPS_ABL_NO_MFMA     // ablation: fragments consumed by one VALU op instead of the MFMA
          acc[i][j][term] += __uint_as_float(F.A[i][TA[term]].u[0] ^ F.B[j][TB[term]].u[1]);
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
#endif
        }
  };
  static_assert(BK == 32, "two K16 steps per chunk");
  Frags F0, F1;
  GPS_BARRIER();                                     // buffer 0 ready
  fetch(0, 0, F0);
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    fetch(buf, 1, F1);                               // in flight under the MFMAs of step 0
    __builtin_amdgcn_sched_barrier(0);
    mma(F0);
    __builtin_amdgcn_sched_barrier(0);
    GPS_BARRIER();                                   // buffer `buf` fully read; the other one is published
    if (c + 1 < nchunks) fetch(buf ^ 1, 0, F0);      // in flight under the MFMAs of step 1
    __builtin_amdgcn_sched_barrier(0);
    mma(F1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // epilogue: D[row = (q&3) + 8*(q>>2) + 4*(lane>>5)][col = lane&31]  (+ bias, + Cin)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = m0 + wn * 64 + j * 32 + li;
      const bool c_ok = col < G.M;
      const float bv = (G.bias && c_ok) ? G.bias[col] : 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = r0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
        if (c_ok && row < G.R) {
          float v = acc[i][j][q] + bv;
          if (G.Cin) v += G.Cin[row * G.ldcin + col];
          G.C[row * G.ldc + col] = v;
        }
      }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

extern "C" {

int gps_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t R, int M, int K,
                const float* bias, const float* Cin, int64_t ldcin, float* C, int64_t ldc,
                gps_stream_t stream) {
  GPS_REQUIRE(R >= 0 && M > 0 && K >= 4 && lda >= K && ldb >= K && ldc >= M, "gps_gemm_nt: bad sizes");
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(A && B && C, "gps_gemm_nt: null buffer");
  GPS_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && al16(A) && al16(B),
              "gps_gemm_nt: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  GPS_REQUIRE(!Cin || ldcin >= M, "gps_gemm_nt: ldcin");
  GemmArgs G{};
  G.A = A; G.B = B; G.bias = bias; G.Cin = Cin; G.C = C;
  G.lda = lda; G.ldb = ldb; G.ldc = ldc; G.ldcin = ldcin; G.R = R; G.M = M; G.K = K;
  G.tiles_m = (int)((R + TM - 1) / TM);
  G.tiles_n = (M + TN - 1) / TN;
  const int ntiles = G.tiles_m * G.tiles_n;
  const int per_xcd = (ntiles + 7) / 8;
  constexpr size_t lds_bytes = 2 * (size_t)kLdsBuf * sizeof(uint16_t);     // 120 KB of the CU's 160 KB
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)lds_bytes);
  GPS_REQUIRE(attr == hipSuccess, "gps_gemm_nt: cannot reserve %zu bytes of LDS", lds_bytes);
  k_gemm_nt<<<(unsigned)(per_xcd * 8), 512, lds_bytes, gps::as_stream(stream)>>>(G);
  return gps::launch_status("gps_gemm_nt");
}

}  // extern "C"
