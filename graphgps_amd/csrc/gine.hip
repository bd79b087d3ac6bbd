// GINE sparse core: out_i = (1+eps) x_i + sum_{j->i} relu(x_j + e_ji).
//
// Reference semantics: PyG 2.2 GINEConv (third-party), constructed at
// graphgps/layer/gps_layer.py:62-69 and called at :183-185; the message form is mirrored
// in-tree at graphgps/layer/gine_conv_layer.py:70-84.  Same lane/row mapping and the same
// deterministic CSR-order segment reduction as gatedgcn.hip.
// Algorithmic HBM bytes per layer: fwd 4Ed + 8Nd, bwd 8Ed + 12Nd.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

// GATE: GINEConvESLapPE (graphgps/layer/gine_conv_layer.py:70-84): relu(x_j + e_ji) * r_ij with a
// per-edge scalar r_edge[edge id].
template <int VEC, bool GATE>
__global__ __launch_bounds__(256) void k_gine_fwd(const float* __restrict__ x,
                                                  const float* __restrict__ e,
                                                  const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ src,
                                                  const int32_t* __restrict__ eid, int64_t N, int d,
                                                  float one_plus_eps, float* __restrict__ out,
                                                  const float* __restrict__ r_edge) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  Vec<VEC> acc = Vec<VEC>::zero();
  for (int k = beg; k < end; ++k) {
    const Vec<VEC> xj = Vec<VEC>::load(x + (int64_t)src[k] * d + c);
    const Vec<VEC> ee = Vec<VEC>::load(e + (int64_t)eid[k] * d + c);
    const float rr = GATE ? r_edge[eid[k]] : 1.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float m = fmaxf(xj[v] + ee[v], 0.0f);
      acc[v] += GATE ? m * rr : m;
    }
  }
  const Vec<VEC> xi = Vec<VEC>::load(x + node * (int64_t)d + c);
  Vec<VEC> o;
#pragma unroll
  for (int v = 0; v < VEC; ++v) o[v] = acc[v] + one_plus_eps * xi[v];
  o.store(out + node * (int64_t)d + c);
}

// target-keyed: g_e[eid] = g_out[i] * [x_j + e > 0]
template <int VEC, bool GATE>
__global__ __launch_bounds__(256) void k_gine_bwd_dst(const float* __restrict__ g_out,
                                                      const float* __restrict__ x,
                                                      const float* __restrict__ e,
                                                      const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ src,
                                                      const int32_t* __restrict__ eid, int64_t N,
                                                      int d, float* __restrict__ g_e,
                                                      const float* __restrict__ r_edge) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  const Vec<VEC> go = Vec<VEC>::load(g_out + node * (int64_t)d + c);
  for (int k = beg; k < end; ++k) {
    const int64_t id = eid[k];
    const Vec<VEC> xj = Vec<VEC>::load(x + (int64_t)src[k] * d + c);
    const Vec<VEC> ee = Vec<VEC>::load(e + id * d + c);
    Vec<VEC> ge;
    const float rr = GATE ? r_edge[id] : 1.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) ge[v] = (xj[v] + ee[v]) > 0.0f ? (GATE ? go[v] * rr : go[v]) : 0.0f;
    ge.store(g_e + id * d + c);
  }
}

// source-keyed: g_x[j] = (1+eps) g_out[j] + sum_{j->.} g_e[eid]
template <int VEC>
__global__ __launch_bounds__(256) void k_gine_bwd_src(const float* __restrict__ g_out,
                                                      const float* __restrict__ g_e,
                                                      const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ eid, int64_t N,
                                                      int d, float one_plus_eps,
                                                      float* __restrict__ g_x) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  Vec<VEC> acc = Vec<VEC>::zero();
  for (int k = beg; k < end; ++k) {
    const Vec<VEC> ge = Vec<VEC>::load(g_e + (int64_t)eid[k] * d + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += ge[v];
  }
  const Vec<VEC> go = Vec<VEC>::load(g_out + node * (int64_t)d + c);
  Vec<VEC> o;
#pragma unroll
  for (int v = 0; v < VEC; ++v) o[v] = acc[v] + one_plus_eps * go[v];
  o.store(g_x + node * (int64_t)d + c);
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" {

int gps_gine_fwd(const float* x, const float* e, const int32_t* rowptr_dst,
                 const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E, int d,
                 float eps, float* out, const float* r_edge, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0, "gps_gine_fwd: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(x && rowptr_dst && out && (E == 0 || (e && src_by_dst && eid_by_dst)),
              "gps_gine_fwd: null buffer");
  auto ok = [&](size_t a) { return aligned_to(x, a) && aligned_to(e, a) && aligned_to(out, a); };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ok(16), ok(8), {
    const unsigned grid = gps::grid_for(N * (int64_t)(d / VEC), 256);
    if (r_edge)
      k_gine_fwd<VEC, true><<<grid, 256, 0, s>>>(x, e, rowptr_dst, src_by_dst, eid_by_dst, N, d, 1.0f + eps, out, r_edge);
    else
      k_gine_fwd<VEC, false><<<grid, 256, 0, s>>>(x, e, rowptr_dst, src_by_dst, eid_by_dst, N, d, 1.0f + eps, out, r_edge);
  });
  return gps::launch_status("gps_gine_fwd");
}

int gps_gine_bwd(const float* g_out, const float* x, const float* e, const int32_t* rowptr_dst,
                 const int32_t* src_by_dst, const int32_t* eid_by_dst, const int32_t* rowptr_src,
                 const int32_t* eid_by_src, int64_t N, int64_t E, int d, float eps, float* g_x,
                 float* g_e, const float* r_edge, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0, "gps_gine_bwd: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(g_out && x && rowptr_dst && rowptr_src && g_x &&
                  (E == 0 || (e && src_by_dst && eid_by_dst && eid_by_src && g_e)),
              "gps_gine_bwd: null buffer");
  auto ok = [&](size_t a) {
    return aligned_to(g_out, a) && aligned_to(x, a) && aligned_to(e, a) && aligned_to(g_x, a) &&
           aligned_to(g_e, a);
  };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ok(16), ok(8), {
    const unsigned grid = gps::grid_for(N * (int64_t)(d / VEC), 256);
    if (r_edge)
      k_gine_bwd_dst<VEC, true><<<grid, 256, 0, s>>>(g_out, x, e, rowptr_dst, src_by_dst, eid_by_dst, N, d, g_e, r_edge);
    else
      k_gine_bwd_dst<VEC, false><<<grid, 256, 0, s>>>(g_out, x, e, rowptr_dst, src_by_dst, eid_by_dst, N, d, g_e, r_edge);
    k_gine_bwd_src<VEC><<<grid, 256, 0, s>>>(g_out, g_e, rowptr_src, eid_by_src, N, d, 1.0f + eps, g_x);
  });
  return gps::launch_status("gps_gine_bwd");
}

}  // extern "C"
