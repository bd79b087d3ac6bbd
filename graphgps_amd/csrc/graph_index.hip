// Per-batch graph index: CSR by target, CSC by source, segment ptr, attention tile map.
// Integer-only, HBM/latency-bound; built once per batch and reused by every layer's forward
// and backward.  Deterministic: the slot a thread claims inside a segment comes from an
// atomic cursor (arbitrary), then each segment is sorted by original edge id, which is exactly
// the stable counting-sort order (== numpy argsort(kind='stable')) and the order in which the
// reference's CPU scatter accumulates (graphgps/layer/gatedgcn_layer.py:117-123).
#include <algorithm>

#include "gps_common.hpp"

namespace {

__global__ void k_histogram(const int64_t* __restrict__ ei, int64_t N, int64_t E,
                            int32_t* __restrict__ rowptr_dst, int32_t* __restrict__ rowptr_src) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], t = ei[E + e];
  if ((uint64_t)s < (uint64_t)N && (uint64_t)t < (uint64_t)N) {
    atomicAdd(&rowptr_dst[t + 1], 1);
    atomicAdd(&rowptr_src[s + 1], 1);
  }
}

// In-place inclusive scan of a[0..n) by ONE workgroup (blockIdx.x selects the array).
__global__ __launch_bounds__(1024) void k_scan(int32_t* a0, int32_t* a1, int64_t n) {
  int32_t* a = blockIdx.x == 0 ? a0 : a1;
  __shared__ int32_t wave_tot[16];
  __shared__ int32_t carry_s;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    int64_t i = base + threadIdx.x;
    int32_t v = i < n ? a[i] : 0;
    for (int off = 1; off < 64; off <<= 1) {
      int32_t u = __shfl_up(v, off);
      if (lane >= off) v += u;
    }
    if (lane == 63) wave_tot[wid] = v;
    __syncthreads();
    int32_t prefix = carry_s;
    for (int w = 0; w < wid; ++w) prefix += wave_tot[w];
    if (i < n) a[i] = v + prefix;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = v + prefix;
    __syncthreads();
  }
}

__global__ void k_fill(const int64_t* __restrict__ ei, int64_t N, int64_t E,
                       const int32_t* __restrict__ rowptr_dst, const int32_t* __restrict__ rowptr_src,
                       int32_t* __restrict__ cur_dst, int32_t* __restrict__ cur_src,
                       int32_t* __restrict__ eid_by_dst, int32_t* __restrict__ eid_by_src) {
  int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], t = ei[E + e];
  if ((uint64_t)s < (uint64_t)N && (uint64_t)t < (uint64_t)N) {
    eid_by_dst[rowptr_dst[t] + atomicAdd(&cur_dst[t], 1)] = (int32_t)e;
    eid_by_src[rowptr_src[s] + atomicAdd(&cur_src[s], 1)] = (int32_t)e;
  }
}

__device__ inline void sort_segment(int32_t* a, int n) {
  if (n <= 24) {  // molecules: degree <= 4
    for (int i = 1; i < n; ++i) {
      int32_t v = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
      a[j + 1] = v;
    }
    return;
  }
  // heapsort for hub nodes
  for (int start = n / 2 - 1; start >= 0; --start) {
    int root = start;
    for (;;) {
      int child = 2 * root + 1;
      if (child >= n) break;
      if (child + 1 < n && a[child] < a[child + 1]) ++child;
      if (a[root] >= a[child]) break;
      int32_t t = a[root]; a[root] = a[child]; a[child] = t;
      root = child;
    }
  }
  for (int end = n - 1; end > 0; --end) {
    int32_t t = a[0]; a[0] = a[end]; a[end] = t;
    int root = 0;
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && a[child] < a[child + 1]) ++child;
      if (a[root] >= a[child]) break;
      int32_t u = a[root]; a[root] = a[child]; a[child] = u;
      root = child;
    }
  }
}

// thread n < N: sort target segment n; thread N <= n < 2N: sort source segment n-N.
__global__ void k_sort_and_resolve(const int64_t* __restrict__ ei, int64_t N, int64_t E,
                                   const int32_t* __restrict__ rowptr_dst,
                                   const int32_t* __restrict__ rowptr_src,
                                   int32_t* __restrict__ eid_by_dst, int32_t* __restrict__ src_by_dst,
                                   int32_t* __restrict__ eid_by_src, int32_t* __restrict__ dst_by_src) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= 2 * N) return;
  const bool by_dst = t < N;
  const int64_t n = by_dst ? t : t - N;
  const int32_t* rp = by_dst ? rowptr_dst : rowptr_src;
  int32_t* eid = by_dst ? eid_by_dst : eid_by_src;
  int32_t* other = by_dst ? src_by_dst : dst_by_src;
  const int64_t* other_row = by_dst ? ei : ei + E;  // the endpoint that is NOT the key
  const int beg = rp[n], end = rp[n + 1];
  sort_segment(eid + beg, end - beg);
  for (int k = beg; k < end; ++k) other[k] = (int32_t)other_row[eid[k]];
}

__global__ void k_ptr_from_batch(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                                 int32_t* __restrict__ ptr) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > N) return;
  // ptr[g] = first node index whose graph id is >= g
  int64_t prev = i == 0 ? -1 : batch[i - 1];
  int64_t cur = i == N ? B : batch[i];
  if (cur > B) cur = B;
  for (int64_t g = prev + 1; g <= cur; ++g) ptr[g] = (int32_t)i;
}

__global__ void k_tile_map(const int32_t* __restrict__ ptr, int64_t B, int64_t max_tiles,
                           int32_t* __restrict__ tile_graph, int32_t* __restrict__ tile_row0) {
  int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (g >= B) return;
  const int n0 = ptr[g], n1 = ptr[g + 1];
  int64_t t = (n0 >> 4) + g;  // slots of consecutive graphs never overlap
  for (int r = n0; r < n1 && t < max_tiles; r += 16, ++t) {
    tile_graph[t] = (int32_t)g;
    tile_row0[t] = r;
  }
}

// Dispatch order of the block-form attention kernels (csrc/sattn.hip: one wavefront per (graph, head), every wavefront of the
// launch resident at once, so a SIMD's time is the SUM of the work of the wavefronts it happens to hold and the launch ends
// with the most loaded SIMD).  Slot t of `order` names the graph whose H wavefronts are dispatched t-th.  Graphs are ranked by
// size (descending, ties by id) and dealt to the slots in a snake over rows of `cols` -- the slots t, t + cols, t + 2 cols, ...
// share CUs -- so every CU gets a mix of long and short graphs: at P30 x 256 the most loaded SIMD holds 23 - 25 units of
// tile-pair work instead of 36 - 43 (the mean is 23).  O(B^2) comparisons; beyond 4,096 graphs the identity.
__global__ __launch_bounds__(256) void k_graph_order(const int32_t* __restrict__ ptr, int B, int cols, int32_t* __restrict__ order) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B) return;
  if (B > 4096) { order[g] = g; return; }
  const int n = ptr[g + 1] - ptr[g];
  int rank = 0;
  for (int j = 0; j < B; ++j) {
    const int m = ptr[j + 1] - ptr[j];
    rank += (m > n) || (m == n && j < g);
  }
  const int row = rank / cols, pos = rank - row * cols;
  const int width = min(cols, B - row * cols);                  // (the last row may be partial)
  const int col = (row & 1) ? width - 1 - pos : pos;
  order[row * cols + col] = g;
}

}  // namespace

extern "C" {

size_t gps_graph_index_workspace_bytes(int64_t N, int64_t E) {
  (void)E;
  return N > 0 ? sizeof(int32_t) * 2 * (size_t)N : 0;
}

int gps_graph_index_build(const int64_t* edge_index, int64_t N, int64_t E, int32_t* rowptr_dst,
                          int32_t* src_by_dst, int32_t* eid_by_dst, int32_t* rowptr_src,
                          int32_t* dst_by_src, int32_t* eid_by_src, void* ws, size_t ws_bytes,
                          gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX,
              "gps_graph_index_build: N=%lld E=%lld out of int32 range", (long long)N, (long long)E);
  GPS_REQUIRE(rowptr_dst && rowptr_src, "gps_graph_index_build: null rowptr");
  GPS_REQUIRE(E == 0 || (edge_index && src_by_dst && eid_by_dst && dst_by_src && eid_by_src),
              "gps_graph_index_build: null edge buffer");
  GPS_REQUIRE(ws_bytes >= gps_graph_index_workspace_bytes(N, E) && (ws || N == 0),
              "gps_graph_index_build: workspace too small (%zu < %zu)", ws_bytes,
              gps_graph_index_workspace_bytes(N, E));
  hipStream_t s = gps::as_stream(stream);
  // The three zero-filled arrays (both rowptr vectors, the two fill cursors at the head of the workspace) cost a fill node
  // each in a replayed step (~5 us apiece).  A caller that lays them out back to back -- rowptr_dst | rowptr_src | ws,
  // what ops.build_graph_index does since round 5 -- gets ONE fill.
  const bool packed = rowptr_src == rowptr_dst + (N + 1) && ws == static_cast<void*>(rowptr_src + (N + 1));
  // (fill KERNELS, not hipMemsetAsync: a memset node stalls a replayed graph for ~80 - 100 us, gps_common.hpp fill_words)
  if (packed && N > 0 && E > 0) {
    gps::fill_words(rowptr_dst, 0u, (size_t)(2 * (N + 1) + 2 * N), s);
  } else {
    gps::fill_words(rowptr_dst, 0u, (size_t)(N + 1), s);
    gps::fill_words(rowptr_src, 0u, (size_t)(N + 1), s);
    if (N > 0 && E > 0) gps::fill_words(ws, 0u, (size_t)(2 * N), s);
  }
  if (N == 0 || E == 0) return gps::launch_status("gps_graph_index_build");
  int32_t* cur_dst = static_cast<int32_t*>(ws);
  int32_t* cur_src = cur_dst + N;
  k_histogram<<<gps::grid_for(E, 256), 256, 0, s>>>(edge_index, N, E, rowptr_dst, rowptr_src);
  k_scan<<<2, 1024, 0, s>>>(rowptr_dst, rowptr_src, N + 1);
  k_fill<<<gps::grid_for(E, 256), 256, 0, s>>>(edge_index, N, E, rowptr_dst, rowptr_src, cur_dst,
                                                cur_src, eid_by_dst, eid_by_src);
  k_sort_and_resolve<<<gps::grid_for(2 * N, 256), 256, 0, s>>>(
      edge_index, N, E, rowptr_dst, rowptr_src, eid_by_dst, src_by_dst, eid_by_src, dst_by_src);
  return gps::launch_status("gps_graph_index_build");
}

int gps_segment_ptr_from_batch(const int64_t* batch, int64_t N, int64_t B, int32_t* ptr,
                               gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && B >= 0 && N < INT32_MAX && ptr && (batch || N == 0),
              "gps_segment_ptr_from_batch: bad arguments");
  k_ptr_from_batch<<<gps::grid_for(N + 1, 256), 256, 0, gps::as_stream(stream)>>>(batch, N, B, ptr);
  return gps::launch_status("gps_segment_ptr_from_batch");
}

int gps_attn_graph_order(const int32_t* ptr, int64_t B, int H, int32_t* order, gps_stream_t stream) {
  GPS_REQUIRE(ptr && order && B >= 1 && H >= 1, "gps_attn_graph_order: bad arguments");
  static const int cus = []() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n > 0 ? n : 256;
  }();
  // graphs per "row" of the chip: workgroup b of a fully resident grid runs on CU b % cus (tools/micro/hwid_probe.hip), a
  // graph's H wavefronts are H / 4 consecutive 4-wavefront workgroups
  const int cols = (int)std::max<int64_t>(1, (int64_t)cus * 4 / H);
  k_graph_order<<<gps::grid_for(B, 256), 256, 0, gps::as_stream(stream)>>>(ptr, (int)std::min<int64_t>(B, INT32_MAX), cols, order);
  return gps::launch_status("gps_attn_graph_order");
}

int gps_attn_tile_map(const int32_t* ptr, int64_t B, int64_t max_tiles, int32_t* tile_graph,
                      int32_t* tile_row0, gps_stream_t stream) {
  GPS_REQUIRE(ptr && tile_graph && tile_row0 && B >= 0 && max_tiles >= 0,
              "gps_attn_tile_map: bad arguments");
  hipStream_t s = gps::as_stream(stream);
  if (max_tiles > 0 && tile_row0 == tile_graph + max_tiles) {      // back to back: one fill launch instead of two
    gps::fill_words(tile_graph, 0xFFFFFFFFu, (size_t)(2 * max_tiles), s);
  } else if (max_tiles > 0) {
    gps::fill_words(tile_graph, 0xFFFFFFFFu, (size_t)max_tiles, s);
    gps::fill_words(tile_row0, 0xFFFFFFFFu, (size_t)max_tiles, s);
  }
  if (B > 0) k_tile_map<<<gps::grid_for(B, 256), 256, 0, s>>>(ptr, B, max_tiles, tile_graph, tile_row0);
  return gps::launch_status("gps_attn_tile_map");
}

}  // extern "C"
