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

// Round 5 -- the forward over ROW SLICES.  k_pool_fwd above gives one lane group to a whole graph: 32 code2 graphs of
// ~800 rows are 8 workgroups on 256 CUs walking 800 dependent-free but serial rows each (273 us for 26 MB = 1.2 % of the HBM
// rate).  Here a graph's rows are cut into slices of POOL_SL rows and every slice gets a workgroup: slot
// floor(ptr[g] / SL) + g is the first slice of graph g (the tile-map numbering of csrc/graph_index.hip: provably
// non-overlapping, needs no host knowledge of the graph sizes; floor(N / SL) + B slots in all), a workgroup finds its
// (graph, slice) by a binary search over ptr, R = 256 / (d / VEC) row-lanes walk the slice's rows round-robin and are
// summed in row-lane order through LDS.  A graph of one slice is written straight to `out`; the partial rows of longer
// graphs go to the workspace and k_pool_merge adds them in slice order.  Fixed partition, fixed order: deterministic.
constexpr int POOL_SL = 32;

template <int VEC>
__global__ __launch_bounds__(256) void k_pool_slices(const float* __restrict__ x, const int32_t* __restrict__ ptr,
                                                     int64_t B, int d, int mean, float* __restrict__ out,
                                                     float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [R][d]
  const int b = blockIdx.x;
  // largest g with floor(ptr[g] / SL) + g <= b   (slot numbers increase strictly with g)
  int lo = 0, hi = (int)B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (ptr[mid] / POOL_SL + mid <= b) lo = mid; else hi = mid - 1;
  }
  const int g = lo;
  const int n0 = ptr[g], n1 = ptr[g + 1];
  const int k = b - (n0 / POOL_SL + g);
  const int r0 = n0 + k * POOL_SL;
  if (r0 >= n1) return;                                  // a slot past the graph's last slice (workgroup-uniform)
  const int r1 = min(r0 + POOL_SL, n1);
  const int L = d / VEC;                                 // lanes per row
  const int Lc = L < 256 ? L : 256;
  const int R = 256 / Lc;                                // rows walked side by side
  const int lane = threadIdx.x % Lc, r = threadIdx.x / Lc;
  const bool single = n1 - n0 <= POOL_SL;
  for (int cl = lane; cl < L; cl += Lc) {                // (one trip unless d / VEC > 256)
    const int c = cl * VEC;
    Vec<VEC> acc = Vec<VEC>::zero();
    if (r < R) {
#pragma unroll 4
      for (int n = r0 + r; n < r1; n += R) {
        const Vec<VEC> v = Vec<VEC>::load(x + (int64_t)n * d + c);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += v[q];
      }
      if (r > 0) acc.store(lds + (r - 1) * d + c);
    }
    __syncthreads();
    if (r == 0) {
      for (int q2 = 1; q2 < R; ++q2) {
        const Vec<VEC> p = Vec<VEC>::load(lds + (q2 - 1) * d + c);
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] += p[q];
      }
      if (single) {
        if (mean) {
          const float cnt = (float)(n1 - n0);
#pragma unroll
          for (int q = 0; q < VEC; ++q) acc[q] = acc[q] / cnt;
        }
        acc.store(out + (int64_t)g * d + c);
      } else {
        acc.store(part + (int64_t)b * d + c);
      }
    }
    __syncthreads();
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_pool_merge(const float* __restrict__ part, const int32_t* __restrict__ ptr,
                                                    int64_t B, int d, int mean, float* __restrict__ out) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t g = t / lanes_per_row;
  if (g >= B) return;
  const int c = (int)(t - g * lanes_per_row) * VEC;
  const int n0 = ptr[g], n1 = ptr[g + 1];
  const int n = n1 - n0;
  if (n > 0 && n <= POOL_SL) return;                     // written by k_pool_slices
  Vec<VEC> acc = Vec<VEC>::zero();                        // (an empty graph pools to zero)
  const int64_t slot0 = n0 / POOL_SL + g;
  const int ns = (n + POOL_SL - 1) / POOL_SL;
#pragma unroll 8
  for (int k = 0; k < ns; ++k) {            // (independent loads, one add chain: the unroll keeps 8 rows in flight)
    const Vec<VEC> v = Vec<VEC>::load(part + (slot0 + k) * d + c);
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] += v[q];
  }
  if (mean) {
    const float cnt = (float)max(n, 1);                   // PyG clamps the count to >= 1
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = acc[q] / cnt;
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


// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of an embedding lookup with a large vocabulary (ASTNodeEncoder's 10,030-entry attribute table,
// graphgps/encoder/ast_encoder.py:35-83): g_w[t] = sum of the gradient rows of every lookup of token t.
// ATen: radix sort + sum_and_scatter (0.4 ms per [25k, 256] lookup on MI355X, 1.2 ms when one token takes half of
// the lookups, order of summation not fixed).  Here the lookups arrive grouped by token (a STABLE sort of the token
// ids, done by the caller: `tok` ascending, `perm` = original positions) and the sum is a segmented reduction over
// fixed units of UNIT consecutive entries:
//   1. one lane group (d / 4 lanes, one wavefront at d = 256) per unit walks its entries in order; a run of equal
//      tokens that begins and ends inside the unit is complete and goes straight to g_w; the (at most two) runs that
//      touch a neighbouring unit leave a partial row: slot 0 = continues a run of the previous unit, slot 1 = begins
//      here and continues into the next;
//   2. for every run that spans units, the workgroup of the unit where it BEGINS sums the partial rows of the
//      units it covers, in unit order (row lanes take units round-robin, combined in lane order).
// Deterministic (fixed partition, fixed order), no atomics.  g_w must be zero on entry (tokens without lookups).
constexpr int EMB_UNIT = 64;

template <int L>     // lanes per row (d = 4 L floats); UNIT entries per lane group, 256 / L lane groups per block
__global__ __launch_bounds__(256) void k_embed_grad_units(const float* __restrict__ g, const int64_t* __restrict__ tok,
                                                          const int64_t* __restrict__ perm, int64_t n, int d,
                                                          float* __restrict__ g_w, float* __restrict__ part,
                                                          int32_t* __restrict__ ptok) {
  typedef Vec<4> V4;
  const int lane = threadIdx.x % L, grp = threadIdx.x / L;
  const int64_t u = (int64_t)blockIdx.x * (256 / L) + grp;
  const int64_t ub = u * EMB_UNIT;
  if (ub >= n) return;
  const int64_t ue = ub + EMB_UNIT < n ? ub + EMB_UNIT : n;
  const int c = lane * 4;
  const int64_t prev_tok = ub > 0 ? tok[ub - 1] : -1, next_tok = ue < n ? tok[ue] : -1;
  int32_t pt0 = -1, pt1 = -1;
  V4 acc = V4::zero();
  int64_t cur = tok[ub], run_start = ub;
  auto flush = [&](bool cont_next) __attribute__((always_inline)) {
    const bool cont_prev = run_start == ub && prev_tok == cur;
    if (!cont_prev && !cont_next) {
      acc.store(g_w + cur * d + c);
    } else if (cont_prev) {
      acc.store(part + (u * 2 + 0) * d + c);
      pt0 = (int32_t)cur;
    } else {
      acc.store(part + (u * 2 + 1) * d + c);
      pt1 = (int32_t)cur;
    }
  };
  for (int64_t k0 = ub; k0 < ue; k0 += 8) {       // 8 gradient rows requested before the first is used
    int64_t t8[8];
    V4 v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = k0 + j < ue ? k0 + j : ue - 1;
      t8[j] = tok[k];
      v8[j] = V4::load(g + perm[k] * d + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (k0 + j < ue) {
        if (t8[j] != cur) {
          flush(false);
          cur = t8[j];
          run_start = k0 + j;
          acc = V4::zero();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += v8[j][q];
      }
    }
  }
  flush(next_tok == cur);
  if (lane == 0) {
    ptok[u * 2 + 0] = pt0;
    ptok[u * 2 + 1] = pt1;
  }
}

template <int L>
__global__ __launch_bounds__(1024) void k_embed_grad_merge(const float* __restrict__ part, const int32_t* __restrict__ ptok,
                                                           int64_t U, int d, float* __restrict__ g_w) {
  typedef Vec<4> V4;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [RS][d]
  const int64_t u = blockIdx.x;
  const int t = ptok[u * 2 + 1];
  if (t < 0) return;                                // no run begins in this unit and runs on (block-uniform)
  constexpr int RS = 1024 / L;
  const int lane = threadIdx.x % L, r = threadIdx.x / L;
  const int c = lane * 4;
  V4 acc = V4::zero();
  for (int64_t v = u + 1 + r; v < U && ptok[v * 2] == t; v += RS) {   // the run covers consecutive units
    const V4 p = V4::load(part + (v * 2) * d + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += p[q];
  }
  acc.store(lds + r * d + c);
  __syncthreads();
  if (r == 0) {
    V4 tot = V4::load(part + (u * 2 + 1) * d + c);
    for (int q2 = 0; q2 < RS; ++q2) {
      const V4 p = V4::load(lds + q2 * d + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) tot[q] += p[q];
    }
    tot.store(g_w + (int64_t)t * d + c);
  }
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

size_t gps_embedding_grad_workspace_bytes(int64_t n, int d) {
  if (n < 1 || d < 1) return 0;
  const int64_t U = (n + EMB_UNIT - 1) / EMB_UNIT;
  return (size_t)U * 2 * d * sizeof(float) + (size_t)U * 2 * sizeof(int32_t) + 64;
}
int gps_embedding_grad_supported(int d) { return d == 64 || d == 128 || d == 256; }

int gps_embedding_grad(const float* g, const int64_t* tok_sorted, const int64_t* perm, int64_t n, int64_t V, int d,
                       float* g_w, void* ws, size_t ws_bytes, gps_stream_t stream) {
  GPS_REQUIRE(n >= 0 && V >= 1 && gps_embedding_grad_supported(d), "gps_embedding_grad: d=%d must be 64, 128 or 256", d);
  if (n == 0) return GPS_OK;
  GPS_REQUIRE(g && tok_sorted && perm && g_w && ws && (uintptr_t)g % 16 == 0 && (uintptr_t)g_w % 16 == 0 && (uintptr_t)ws % 16 == 0,
              "gps_embedding_grad: null / misaligned buffer");
  GPS_REQUIRE(ws_bytes >= gps_embedding_grad_workspace_bytes(n, d), "gps_embedding_grad: workspace too small");
  const int64_t U = (n + EMB_UNIT - 1) / EMB_UNIT;
  float* part = static_cast<float*>(ws);
  int32_t* ptok = reinterpret_cast<int32_t*>(part + (size_t)U * 2 * d);
  hipStream_t s = gps::as_stream(stream);
#define GPS_EMB(LV)                                                                                              \
  do {                                                                                                           \
    constexpr int per = 256 / LV;                                                                                \
    k_embed_grad_units<LV><<<(unsigned)((U + per - 1) / per), 256, 0, s>>>(g, tok_sorted, perm, n, d, g_w, part, ptok); \
    k_embed_grad_merge<LV><<<(unsigned)U, 1024, sizeof(float) * (1024 / LV) * d, s>>>(part, ptok, U, d, g_w);    \
  } while (0)
  if (d == 256) GPS_EMB(64); else if (d == 128) GPS_EMB(32); else GPS_EMB(16);
#undef GPS_EMB
  return gps::launch_status("gps_embedding_grad");
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

size_t gps_segment_pool_workspace_bytes(int64_t N, int64_t B, int d) {
  if (N < 0 || B < 1 || d < 1) return 0;
  return (size_t)(N / POOL_SL + B) * d * sizeof(float) + 64;
}

int gps_segment_pool_fwd_sliced(const float* x, const int32_t* ptr, int64_t N, int64_t B, int d, int mean, float* out,
                                void* ws, size_t ws_bytes, gps_stream_t stream) {
  GPS_REQUIRE(B >= 0 && N >= 0 && d > 0, "gps_segment_pool_fwd_sliced: bad sizes");
  if (B == 0) return GPS_OK;
  GPS_REQUIRE(ptr && out && ws && (x || N == 0), "gps_segment_pool_fwd_sliced: null buffer");
  GPS_REQUIRE(ws_bytes >= gps_segment_pool_workspace_bytes(N, B, d) && (uintptr_t)ws % 16 == 0,
              "gps_segment_pool_fwd_sliced: workspace too small / misaligned");
  GPS_REQUIRE(N / POOL_SL + B < (1ll << 31), "gps_segment_pool_fwd_sliced: too many slices");
  hipStream_t s = gps::as_stream(stream);
  float* part = static_cast<float*>(ws);
  const unsigned slots = (unsigned)(N / POOL_SL + B);
  const bool a16 = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  const bool a8 = ((uintptr_t)x % 8 == 0) && ((uintptr_t)out % 8 == 0);
  GPS_DISPATCH_VEC(d, a16, a8, {
    const int L = d / VEC, Lc = L < 256 ? L : 256;
    const size_t lds = sizeof(float) * (size_t)(256 / Lc) * d;
    k_pool_slices<VEC><<<slots, 256, lds, s>>>(x, ptr, B, d, mean, out, part);
    k_pool_merge<VEC><<<gps::grid_for(B * (int64_t)(d / VEC), 256), 256, 0, s>>>(part, ptr, B, d, mean, out);
  });
  return gps::launch_status("gps_segment_pool_fwd_sliced");
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
