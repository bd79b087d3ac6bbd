// Small dense products of the graph-level heads: a few hundred ROWS (one per graph of the batch), widths of 1 .. 384.
//
// Reference: graphgps/head/san_graph.py:19-42 (FC_layers: Linear + act ... Linear on the pooled embedding [B, dim_in]) and
// what autograd derives for it.  Through rocBLAS / hipBLASLt these shapes land on macro tiles sized for large problems --
// [256 x 384] x [384 x 192] ran as ONE 192 x 256 workgroup, 74 us (profiles/r05_kernel_trace_stats_pcqm4m.txt: the 14
// library GEMMs of the pcqm4m head and its backward cost 0.2 ms per step for 0.1 GFLOP).  Here:
//   * C(i, j) = sum_k A(i, k) B(k, j) with both operands addressed through (row, column) strides, so ONE kernel serves
//     y = x W^T (+ bias, ReLU), g_x = g W and g_W = g^T x (+ g_b = column sums of g) without transposed copies;
//   * the ReLU backward mask of the layer's own output is applied to `g` as it is loaded (g (.) [y > 0]);
//   * exact fp32 products on v_mfma_f32_16x16x4_f32 (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]),
//     one 16 x 16 output tile per workgroup, the contraction split over its 4 wavefronts in batches of 32 and the four
//     partial tiles added in wavefront order through LDS: deterministic, no atomics;
//   * every load is unconditional with a clamped address (a load under a condition becomes a branch and the loads of a
//     batch stop being issued together), out-of-range elements are zeroed after the fact.
// Sized for M * N * K up to a few 1e8: a launch is 3 - 6 us, bound by three dependent L2 round trips, not by flops.
#include "gps_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SmallGemm {
  const float* A; int64_t sai, sak;      // A(i, k) = A[i * sai + k * sak]
  const float* B; int64_t sbk, sbj;      // B(k, j) = B[k * sbk + j * sbj]
  const float* amask; int64_t smi, smk;  // A(i, k) counts only where amask(i, k) > 0 (ReLU backward), or nullptr
  const float* bias;                     // + bias[j], or nullptr
  float* C; int64_t ldc;                 // C[i * ldc + j]
  float* rowsum;                         // rowsum[i] = sum_k A(i, k) (after the mask): the bias gradient of the g^T x form
  int M, N, K, relu, tiles_n;
};

constexpr int KB = 32;          // contraction elements per batch (8 MFMAs)
constexpr int NW = 4;           // wavefronts per workgroup = contraction slices

__global__ __launch_bounds__(64 * NW) void k_small_gemm(const SmallGemm G) {
  __shared__ float part[NW][4][64];
  __shared__ float rsum[NW][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.x / G.tiles_n, tj = blockIdx.x - ti * G.tiles_n;
  const int li = lane & 15, kq = lane >> 4;
  const int i = ti * 16 + li, j = tj * 16 + li;          // this lane's A row / B column
  const bool i_ok = i < G.M, j_ok = j < G.N;
  const float* __restrict__ Ap = G.A + (int64_t)(i_ok ? i : G.M - 1) * G.sai;
  const float* __restrict__ Mp = G.amask ? G.amask + (int64_t)(i_ok ? i : G.M - 1) * G.smi : nullptr;
  const float* __restrict__ Bp = G.B + (int64_t)(j_ok ? j : G.N - 1) * G.sbj;
  const int kl = G.K - 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
  float a[8], b[8], m[8];
  auto load = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = min(k0 + 4 * u + kq, kl);
      a[u] = Ap[(int64_t)k * G.sak];
      b[u] = Bp[(int64_t)k * G.sbk];
      m[u] = Mp ? Mp[(int64_t)k * G.smk] : 1.0f;
    }
  };
  int k0 = wave * KB;
  if (k0 < G.K) load(k0);
  for (; k0 < G.K; k0 += NW * KB) {
    float ca[8], cb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = k0 + 4 * u + kq < G.K;
      ca[u] = ok && i_ok && m[u] > 0.0f ? a[u] : 0.0f;
      cb[u] = ok && j_ok ? b[u] : 0.0f;
    }
    if (k0 + NW * KB < G.K) load(k0 + NW * KB);          // the next batch is in flight while this one is multiplied
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[u], cb[u], acc, 0, 0, 0);
      asum += ca[u];
    }
  }
  // the four contraction slices -> one tile, in wavefront order
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc[r];
  if (G.rowsum && tj == 0) {
    asum += __shfl_xor(asum, 16);
    asum += __shfl_xor(asum, 32);
    if (lane < 16) rsum[wave][lane] = asum;
  }
  __syncthreads();
  if (wave == 0) {
    const float bj = G.bias && j_ok ? G.bias[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = ((part[0][r][lane] + part[1][r][lane]) + part[2][r][lane]) + part[3][r][lane] + bj;
      if (G.relu) v = fmaxf(v, 0.0f);
      const int row = ti * 16 + 4 * kq + r;               // C[i = 4 * (lane >> 4) + r][j = lane & 15]
      if (row < G.M && j_ok) G.C[(int64_t)row * G.ldc + j] = v;
    }
    if (G.rowsum && tj == 0 && lane < 16 && i_ok)
      G.rowsum[i] = ((rsum[0][lane] + rsum[1][lane]) + rsum[2][lane]) + rsum[3][lane];
  }
}

int launch(const char* who, SmallGemm& G, gps_stream_t stream) {
  GPS_REQUIRE(G.M >= 1 && G.N >= 1 && G.K >= 1 && G.A && G.B && G.C, "%s: empty or null operand", who);
  GPS_REQUIRE((int64_t)G.M * G.N <= (int64_t)1 << 26, "%s: %d x %d outputs: not a small product", who, G.M, G.N);
  G.tiles_n = (G.N + 15) / 16;
  const unsigned grid = (unsigned)(((G.M + 15) / 16) * G.tiles_n);
  k_small_gemm<<<grid, 64 * NW, 0, gps::as_stream(stream)>>>(G);
  return gps::launch_status(who);
}

}  // namespace

extern "C" {

// y[M, N] = act(x[M, K] w[N, K]^T + bias)      (nn.Linear forward; relu != 0: ReLU)
int gps_small_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int M, int N, int K,
                         int relu, float* y, int64_t ldy, gps_stream_t stream) {
  SmallGemm G{};
  G.A = x; G.sai = ldx; G.sak = 1;
  G.B = w; G.sbk = 1; G.sbj = ldw;
  G.bias = bias; G.C = y; G.ldc = ldy; G.M = M; G.N = N; G.K = K; G.relu = relu;
  return launch("gps_small_linear_fwd", G, stream);
}

// g' = g (.) [y > 0] when y is given (the layer applied ReLU), else g.  g_x[M, K] = g' w (nullptr: skipped);
// g_w[N, K] = g'^T x and g_b[N] = column sums of g' (nullptr: skipped).  Two launches at most.
int gps_small_linear_bwd(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* x, int64_t ldx,
                         const float* w, int64_t ldw, int M, int N, int K, float* g_x, int64_t ldgx, float* g_w,
                         int64_t ldgw, float* g_b, gps_stream_t stream) {
  if (g_x) {
    SmallGemm G{};
    G.A = g; G.sai = ldg; G.sak = 1;
    G.amask = y; G.smi = ldy; G.smk = 1;
    G.B = w; G.sbk = ldw; G.sbj = 1;
    G.C = g_x; G.ldc = ldgx; G.M = M; G.N = K; G.K = N;
    if (int rc = launch("gps_small_linear_bwd (input gradient)", G, stream)) return rc;
  }
  if (g_w) {
    SmallGemm G{};
    G.A = g; G.sai = 1; G.sak = ldg;
    G.amask = y; G.smi = 1; G.smk = ldy;
    G.B = x; G.sbk = ldx; G.sbj = 1;
    G.C = g_w; G.ldc = ldgw; G.rowsum = g_b; G.M = N; G.N = K; G.K = M;
    if (int rc = launch("gps_small_linear_bwd (weight gradient)", G, stream)) return rc;
  }
  return GPS_OK;
}

}  // extern "C"
