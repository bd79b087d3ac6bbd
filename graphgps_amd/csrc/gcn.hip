// GCN sparse core: symmetric-normalised adjacency product with self loops,
//
//     out_i = dinv_i * ( dinv_i * x_i + sum_{j -> i, j != i} dinv_j * x_j ),   dinv = deg^-1/2,
//     deg_i = 1 + #{edges j -> i with j != i}.
//
// Reference semantics: PyG 2.2 GCNConv (third-party; defaults add_self_loops=True, normalize=True,
// edge_weight=None -> gcn_norm), the local model of `gt.layer_type: GCN+...` constructed at
// graphgps/layer/gps_layer.py:53-55 and called at :176-183 without edge attributes.  gcn_norm replaces
// whatever self loops the input has by exactly one unit-weight loop per node (add_remaining_self_loops) and
// counts every remaining edge -- duplicates included -- in the degree of its TARGET.
// The operator is linear with matrix A^[i][j] = dinv_i dinv_j [j -> i]; its transpose is the same kernel on
// the CSC (source-keyed) half of the graph index, which is how the backward runs.  Same lane/row mapping and
// fixed CSR-order reduction (no atomics, bitwise reproducible) as gatedgcn.hip / gine.hip.
// Algorithmic HBM bytes: 4Nd (rows of x, each gathered row counted once) + 4Nd (out) + 4(N+1) + 4E index.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

__global__ __launch_bounds__(256) void k_gcn_dinv(const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ src, int64_t N,
                                                  float* __restrict__ dinv) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  int deg = 1;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) deg += src[k] != (int32_t)i;
  dinv[i] = 1.0f / sqrtf((float)deg);
}

template <int VEC>
__global__ __launch_bounds__(256) void k_gcn_spmm(const float* __restrict__ x, int64_t ldx,
                                                  const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ nbr,
                                                  const float* __restrict__ dinv, int64_t N, int d,
                                                  float* __restrict__ out) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  const float di = dinv[node];
  const Vec<VEC> xi = Vec<VEC>::load(x + node * ldx + c);
  Vec<VEC> acc;
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = di * xi[v];
  for (int k = beg; k < end; ++k) {
    const int j = nbr[k];
    if (j == (int)node) continue;             // input self loops are replaced by the unit loop above
    const float dj = dinv[j];
    const Vec<VEC> xj = Vec<VEC>::load(x + (int64_t)j * ldx + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += dj * xj[v];
  }
  Vec<VEC> o;
#pragma unroll
  for (int v = 0; v < VEC; ++v) o[v] = di * acc[v];
  o.store(out + node * (int64_t)d + c);
}

// Plain adjacency sum with a weighted self term: out_i = self_w * x_i + sum_{k in segment(i)} x_{nbr[k]}
// (every stored edge counts, self loops and duplicates included) = PyG GINConv's (1 + eps) x_i + sum_j x_j
// before its MLP -- the phi network of SignNet (graphgps/encoder/signnet_pos_encoder.py:70-110) runs it over
// the [N, k * channels] eigenvector features.  Transpose = the same kernel on the CSC half.
template <int VEC>
__global__ __launch_bounds__(256) void k_adj_sum(const float* __restrict__ x, int64_t ldx,
                                                 const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ nbr, float self_w, int64_t N, int d,
                                                 float* __restrict__ out) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const Vec<VEC> xi = Vec<VEC>::load(x + node * ldx + c);
  Vec<VEC> acc;
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = self_w * xi[v];
  for (int k = rowptr[node]; k < rowptr[node + 1]; ++k) {
    const Vec<VEC> xj = Vec<VEC>::load(x + (int64_t)nbr[k] * ldx + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += xj[v];
  }
  acc.store(out + node * (int64_t)d + c);
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" {

int gps_gcn_dinv(const int32_t* rowptr_dst, const int32_t* src_by_dst, int64_t N, int64_t E, float* dinv,
                 gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0, "gps_gcn_dinv: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(rowptr_dst && dinv && (E == 0 || src_by_dst), "gps_gcn_dinv: null buffer");
  k_gcn_dinv<<<gps::grid_for(N, 256), 256, 0, gps::as_stream(stream)>>>(rowptr_dst, src_by_dst, N, dinv);
  return gps::launch_status("gps_gcn_dinv");
}

int gps_gcn_spmm(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* nbr, const float* dinv,
                 int64_t N, int64_t E, int d, float* out, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_x >= d, "gps_gcn_spmm: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(x && rowptr && dinv && out && (E == 0 || nbr), "gps_gcn_spmm: null buffer");
  auto ok = [&](size_t a) { return aligned_to(x, a) && aligned_to(out, a) && (ld_x * sizeof(float)) % a == 0; };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ok(16), ok(8), {
    const unsigned grid = gps::grid_for(N * (int64_t)(d / VEC), 256);
    k_gcn_spmm<VEC><<<grid, 256, 0, s>>>(x, ld_x, rowptr, nbr, dinv, N, d, out);
  });
  return gps::launch_status("gps_gcn_spmm");
}

int gps_adj_sum(const float* x, int64_t ld_x, const int32_t* rowptr, const int32_t* nbr, float self_w, int64_t N,
                int64_t E, int d, float* out, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_x >= d, "gps_adj_sum: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(x && rowptr && out && (E == 0 || nbr), "gps_adj_sum: null buffer");
  auto ok = [&](size_t a) { return aligned_to(x, a) && aligned_to(out, a) && (ld_x * sizeof(float)) % a == 0; };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ok(16), ok(8), {
    const unsigned grid = gps::grid_for(N * (int64_t)(d / VEC), 256);
    k_adj_sum<VEC><<<grid, 256, 0, s>>>(x, ld_x, rowptr, nbr, self_w, N, d, out);
  });
  return gps::launch_status("gps_adj_sum");
}

}  // extern "C"
