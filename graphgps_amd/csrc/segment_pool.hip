// Graph-level pooling as a ptr-segmented reduction (sum / mean) and its backward broadcast.
// Replaces GraphGym's pooling_dict['add'|'mean'] = global_add_pool / global_mean_pool
// (torch_scatter atomics; called from graphgps/head/san_graph.py:35 and
// graphgps/head/ogb_code_graph.py:37).  Deterministic: lanes own channels, nodes are summed
// in index order.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void k_pool_fwd(const float* __restrict__ x,
                                                  const int32_t* __restrict__ ptr, int64_t B, int d,
                                                  int mean, float* __restrict__ out) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t g = t / lanes_per_row;
  if (g >= B) return;
  const int c = (int)(t - g * lanes_per_row) * VEC;
  const int n0 = ptr[g], n1 = ptr[g + 1];
  Vec<VEC> acc = Vec<VEC>::zero();
  for (int n = n0; n < n1; ++n) {
    const Vec<VEC> v = Vec<VEC>::load(x + (int64_t)n * d + c);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
  }
  if (mean) {
    const float cnt = (float)max(n1 - n0, 1);  // PyG clamps the count to >= 1
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / cnt;
  }
  acc.store(out + g * (int64_t)d + c);
}

template <int VEC>
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ g_out,
                                                  const int32_t* __restrict__ ptr,
                                                  const int32_t* __restrict__ node_graph, int64_t N,
                                                  int d, int mean, float* __restrict__ g_x) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t n = t / lanes_per_row;
  if (n >= N) return;
  const int c = (int)(t - n * lanes_per_row) * VEC;
  const int g = node_graph[n];
  Vec<VEC> v = Vec<VEC>::load(g_out + (int64_t)g * d + c);
  if (mean) {
    const float cnt = (float)max(ptr[g + 1] - ptr[g], 1);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = v[k] / cnt;
  }
  v.store(g_x + n * (int64_t)d + c);
}

__global__ void k_node_graph(const int32_t* __restrict__ ptr, int64_t B, int32_t* __restrict__ node_graph) {
  const int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (g >= B) return;
  for (int n = ptr[g]; n < ptr[g + 1]; ++n) node_graph[n] = (int32_t)g;
}

}  // namespace

extern "C" {

int gps_node_graph_from_ptr(const int32_t* ptr, int64_t B, int32_t* node_graph, gps_stream_t stream) {
  GPS_REQUIRE(ptr && B >= 0 && (node_graph || B == 0), "gps_node_graph_from_ptr: bad arguments");
  if (B > 0) k_node_graph<<<gps::grid_for(B, 256), 256, 0, gps::as_stream(stream)>>>(ptr, B, node_graph);
  return gps::launch_status("gps_node_graph_from_ptr");
}

int gps_segment_pool_fwd(const float* x, const int32_t* ptr, int64_t B, int d, int mean, float* out,
                         gps_stream_t stream) {
  GPS_REQUIRE(B >= 0 && d > 0, "gps_segment_pool_fwd: bad sizes");
  if (B == 0) return GPS_OK;
  GPS_REQUIRE(x && ptr && out, "gps_segment_pool_fwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const bool a16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const bool a8 = ((uintptr_t)x % 8 == 0) && ((uintptr_t)out % 8 == 0);
  GPS_DISPATCH_VEC(d, a16, a8, {
    k_pool_fwd<VEC><<<gps::grid_for(B * (int64_t)(d / VEC), 256), 256, 0, s>>>(x, ptr, B, d, mean, out);
  });
  return gps::launch_status("gps_segment_pool_fwd");
}

int gps_segment_pool_bwd(const float* g_out, const int32_t* ptr, const int32_t* node_graph,
                         int64_t N, int d, int mean, float* g_x, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && d > 0, "gps_segment_pool_bwd: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(g_out && ptr && node_graph && g_x, "gps_segment_pool_bwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const bool a16 = ((uintptr_t)g_out % 16 == 0) && ((uintptr_t)g_x % 16 == 0);
  const bool a8 = ((uintptr_t)g_out % 8 == 0) && ((uintptr_t)g_x % 8 == 0);
  GPS_DISPATCH_VEC(d, a16, a8, {
    k_pool_bwd<VEC><<<gps::grid_for(N * (int64_t)(d / VEC), 256), 256, 0, s>>>(g_out, ptr, node_graph,
                                                                               N, d, mean, g_x);
  });
  return gps::launch_status("gps_segment_pool_bwd");
}

}  // extern "C"
