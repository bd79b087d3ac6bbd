// Weight + bias gradient of a row-wise linear layer on fp32 MFMA, split-K, deterministic.
//
//     gW[m][n] = sum_r g[r][m] * x[r][n]        gb[m] = sum_r g[r][m]
//
// with r over N nodes or E edges (7.5k-15k at PCQM4M sizes) and a small [M, Nn] result
// (384..1536 x 384..768).  This is what autograd derives for the nn.Linear modules of the block
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
#include "gps_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NLD = BK / 8;   // float4 loads per thread per operand per chunk (256 threads x 4 floats = 8 rows)

__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ g, int64_t ldg,
                                               const float* __restrict__ x, int64_t ldx, int64_t R,
                                               int M, int Nn, int tiles_m, int tiles_n, int n_slices,
                                               int rows_per_slice, int want_bias,
                                               float* __restrict__ part,
                                               float* __restrict__ bias_part) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
  // plain map: consecutive blocks = the tiles of one row slice.  (An XCD-grouped map -- all tiles
  // of a slice on one XCD for L2 reuse -- was measured 25-50 % SLOWER here: with ~250 workgroups
  // the uneven tiles-per-XCD split costs a second dispatch round on some XCDs.)
  const int tiles = tiles_m * tiles_n;
  const int slice = blockIdx.x / tiles;
  const int tile = blockIdx.x - slice * tiles;
  if (slice >= n_slices) return;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int64_t r_begin = (int64_t)slice * rows_per_slice;
  const int64_t r_end = min(R, r_begin + rows_per_slice);
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
  auto load_chunk = [&](int64_t r0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int64_t r = r0 + srow + 8 * j;
      const bool ok = r < r_end;
      ra[j] = (ok && a_col_ok) ? *reinterpret_cast<const float4*>(g + r * ldg + m0 + scol)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[j] = (ok && b_col_ok) ? *reinterpret_cast<const float4*>(x + r * ldx + n0 + scol)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      *reinterpret_cast<float4*>(&As[buf][srow + 8 * j][scol]) = ra[j];
      *reinterpret_cast<float4*>(&Bs[buf][srow + 8 * j][scol]) = rb[j];
      bsum.x += ra[j].x; bsum.y += ra[j].y; bsum.z += ra[j].z; bsum.w += ra[j].w;
    }
  };

  load_chunk(r_begin);
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += BK) {
    const bool more = r0 + BK < r_end;
    if (more) load_chunk(r0 + BK);     // global loads in flight during the MFMAs below
    const int kh = lane >> 5, li = lane & 31;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int kk = 2 * ks + kh;
      const float a0 = As[buf][kk][wm * 64 + li];
      const float a1 = As[buf][kk][wm * 64 + 32 + li];
      const float b0 = Bs[buf][kk][wn * 64 + li];
      const float b1 = Bs[buf][kk][wn * 64 + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // partial tile: D[row = (q&3) + 8*(q>>2) + 4*(lane>>5)][col = lane&31]
  float* po = part + (int64_t)slice * M * Nn;
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
  if (want_bias && tn == 0) {
    // 8 threads (srow = 0..7) share a column quad: reduce through LDS in fixed order
    float* scratch = &As[0][0][0];   // all MFMA reads are done (barrier at loop end)
    *reinterpret_cast<float4*>(&scratch[srow * BM + scol]) = bsum;
    __syncthreads();
    if (t < BM && m0 + t < M) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += scratch[q * BM + t];
      bias_part[(int64_t)slice * M + m0 + t] = s;
    }
  }
}

// out[i] = sum_s part[s][i] in slice order (one float per thread: enough threads to pull the
// S x M x Nn partials at bandwidth); bias likewise, by the first blocks
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int S, int64_t total,
                                                      float* __restrict__ out,
                                                      const float* __restrict__ bias_part, int M,
                                                      float* __restrict__ bias_out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) {
    float a = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) a += part[(int64_t)s * total + i];
    out[i] = a;
  }
  if (bias_out && i < M) {
    float a = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) a += bias_part[(int64_t)s * M + i];
    bias_out[i] = a;
  }
}

struct Plan {
  int tiles_m, tiles_n, S, rows_per_slice;
};
inline Plan make_plan(int64_t R, int M, int Nn) {
  Plan p;
  p.tiles_m = (M + BM - 1) / BM;
  p.tiles_n = (Nn + BN - 1) / BN;
  const int tiles = p.tiles_m * p.tiles_n;
  int S = 256 / tiles;                               // ~one workgroup per CU
  const int64_t max_s = (R + 4 * BK - 1) / (4 * BK); // at least 4 chunks per slice
  if (S > max_s) S = (int)max_s;
  if (S < 1) S = 1;
  int64_t rps = (R + S - 1) / S;
  rps = (rps + BK - 1) / BK * BK;
  p.rows_per_slice = (int)rps;
  p.S = (int)((R + rps - 1) / rps);
  return p;
}

}  // namespace

extern "C" {

size_t gps_wgrad_workspace_floats(int64_t R, int M, int Nn) {
  if (R <= 0 || M <= 0 || Nn <= 0) return 0;
  const Plan p = make_plan(R, M, Nn);
  return (size_t)p.S * ((size_t)M * Nn + M);
}

int gps_wgrad(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t R, int M, int Nn,
              float* gw, float* gb, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(R >= 1 && M > 0 && Nn > 0 && ldg >= M && ldx >= Nn, "gps_wgrad: bad sizes");
  GPS_REQUIRE(g && x && gw && ws, "gps_wgrad: null buffer");
  GPS_REQUIRE(M % 4 == 0 && Nn % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(g) % 16 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                  (reinterpret_cast<uintptr_t>(gw) % 16 == 0) && (reinterpret_cast<uintptr_t>(ws) % 16 == 0),
              "gps_wgrad: dimensions must be multiples of 4 and buffers 16-byte aligned");
  const Plan p = make_plan(R, M, Nn);
  float* part = ws;
  float* bias_part = ws + (size_t)p.S * M * Nn;
  hipStream_t s = gps::as_stream(stream);
  const int tiles = p.tiles_m * p.tiles_n;
  const unsigned grid = (unsigned)(p.S * tiles);
  k_wgrad<<<grid, 256, 0, s>>>(g, ldg, x, ldx, R, M, Nn, p.tiles_m, p.tiles_n, p.S, p.rows_per_slice,
                               gb != nullptr, part, bias_part);
  const int64_t total = (int64_t)M * Nn;
  k_wgrad_reduce<<<gps::grid_for(total, 256), 256, 0, s>>>(part, p.S, total, gw, bias_part, M, gb);
  return gps::launch_status("gps_wgrad");
}

}  // extern "C"
