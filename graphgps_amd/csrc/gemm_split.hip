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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, BK = 32;
constexpr int PITCH = 40;            // bf16 elements per LDS row (32 + 8 pad = 80 bytes)
constexpr int NPASS = TM / 32;       // staging passes: 256 threads = 32 rows x 8 k-quads per pass

struct Frag {
  uint32_t u[4];
};

// 4 fp32 -> three packed bf16x4 (8 bytes each).  Round-to-nearest pieces (v_cvt_pk_bf16_f32 packs two
// at a time): hi = rne(v), r1 = v - hi (exact), mid = rne(r1), r2 = r1 - mid (exact, <= 8 significant
// bits), lo = r2 exactly -- hi + mid + lo == v bit for bit.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(f32x2 v, f32x2& back) {
  const bf16x2 b = __builtin_convertvector(v, bf16x2);
  back = __builtin_convertvector(b, f32x2);
  uint32_t u;
  __builtin_memcpy(&u, &b, 4);
  return u;
}
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
  f32x2 a = {v.x, v.y}, c = {v.z, v.w}, ba, bc;
  hi.x = pack2(a, ba); hi.y = pack2(c, bc);
  a -= ba; c -= bc;
  mid.x = pack2(a, ba); mid.y = pack2(c, bc);
  a -= ba; c -= bc;
  lo.x = pack2(a, ba); lo.y = pack2(c, bc);
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
    // staging: thread -> (row = t/8 + 32*pass, k-quad = t%8)
    const int srow = t >> 3, skq = (t & 7) * 4;
    const float* ap[NPASS];
    const float* bp[NPASS];
    bool a_ok[NPASS], b_ok[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int64_t ra = r0 + srow + 32 * p;
      const int rb = m0 + srow + 32 * p;
      a_ok[p] = ra < G.R;
      b_ok[p] = rb < G.M;
      ap[p] = G.A + (a_ok[p] ? ra : G.R - 1) * G.lda;
      bp[p] = G.B + (int64_t)(b_ok[p] ? rb : G.M - 1) * G.ldb;
    }
    float4 ra0[NPASS], rb0[NPASS], ra1[NPASS], rb1[NPASS];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_chunk = [&](int c, float4 (&ra)[NPASS], float4 (&rb)[NPASS]) {
      // raw loads only; a k-quad past K reads quad 0 and is zeroed at split time
      const int kk = (c * BK + skq < K) ? c * BK + skq : 0;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        ra[p] = *reinterpret_cast<const float4*>(ap[p] + kk);
        rb[p] = *reinterpret_cast<const float4*>(bp[p] + kk);
      }
    };
    auto store_chunk = [&](int c, int buf, const float4 (&ra)[NPASS], const float4 (&rb)[NPASS]) {
      const bool k_ok = c * BK + skq < K;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        uint2 h, m, l;
        const int row = srow + 32 * p;
        split4((a_ok[p] && k_ok) ? ra[p] : zero4, h, m, l);
        *reinterpret_cast<uint2*>(L(buf, 0, 0, row, skq)) = h;
        *reinterpret_cast<uint2*>(L(buf, 0, 1, row, skq)) = m;
        *reinterpret_cast<uint2*>(L(buf, 0, 2, row, skq)) = l;
        split4((b_ok[p] && k_ok) ? rb[p] : zero4, h, m, l);
        *reinterpret_cast<uint2*>(L(buf, 1, 0, row, skq)) = h;
        *reinterpret_cast<uint2*>(L(buf, 1, 1, row, skq)) = m;
        *reinterpret_cast<uint2*>(L(buf, 1, 2, row, skq)) = l;
      }
    };
    load_chunk(0, ra0, rb0);
    if (nchunks > 1) load_chunk(1, ra1, rb1);
    store_chunk(0, 0, ra0, rb0);
    __syncthreads();                                   // buffer 0 ready
    for (int c = 0; c < nchunks; c += 2) {
      // consumers multiply chunk c (buffer 0): stage chunk c+1 into buffer 1, fetch chunk c+2
      if (c + 2 < nchunks) load_chunk(c + 2, ra0, rb0);
      if (c + 1 < nchunks) store_chunk(c + 1, 1, ra1, rb1);
      __syncthreads();
      if (c + 1 >= nchunks) break;
      // consumers multiply chunk c+1 (buffer 1): stage chunk c+2 into buffer 0, fetch chunk c+3
      if (c + 3 < nchunks) load_chunk(c + 3, ra1, rb1);
      if (c + 2 < nchunks) store_chunk(c + 2, 0, ra0, rb0);
      __syncthreads();
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
  auto multiply = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Frag A[2][3], B[2][3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 a = *reinterpret_cast<const uint4*>(L(buf, 0, pc, wm * 64 + i * 32 + li, 16 * ks + 8 * kh));
          const uint4 b = *reinterpret_cast<const uint4*>(L(buf, 1, pc, wn * 64 + i * 32 + li, 16 * ks + 8 * kh));
          A[i][pc].u[0] = a.x; A[i][pc].u[1] = a.y; A[i][pc].u[2] = a.z; A[i][pc].u[3] = a.w;
          B[i][pc].u[0] = b.x; B[i][pc].u[1] = b.y; B[i][pc].u[2] = b.z; B[i][pc].u[3] = b.w;
        }
      // smallest terms first; the four accumulators rotate so no MFMA waits on its predecessor
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            bf16x8 av, bv;
            __builtin_memcpy(&av, &A[i][TA[term]], 16);
            __builtin_memcpy(&bv, &B[j][TB[term]], 16);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i][j], 0, 0, 0);
          }
    }
  };
  __syncthreads();                                     // buffer 0 ready
  for (int c = 0; c < nchunks; c += 2) {
    multiply(0);
    __syncthreads();
    if (c + 1 >= nchunks) break;
    multiply(1);
    __syncthreads();
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
