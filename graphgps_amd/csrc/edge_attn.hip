// SAN edge attention over the real edges: score -> (segment softmax | exp-clamp) -> weighted sum of source values.
//
// Reference: graphgps/layer/san_layer.py:44-92 (SANLayer: w = exp(clamp(s, -5, 5)), h = sum w V / (sum w + 1e-6)) and
// graphgps/layer/san2_layer.py:11-33,65-105 (SAN2Layer: w = pyg_softmax(s, target) = exp(s - max) / (sum + 1e-16)),
// with  s_ij = sum_c K_j[h,c] Q_i[h,c] E_ij[h,c] / sqrt(d_h)  per head h -- there: 3 index_select gathers, an
// [E, H, d_h] product, scatter_max + scatter_add (softmax) and 1-2 scatter adds, ~12 [E, H*d_h] temporaries.
// Here: the same gather-gate-segment-reduce shape as csrc/gatedgcn.hip.  Lane = 4 consecutive channels of one target
// node's row (a head = d_h / 4 consecutive lanes: the score's channel sum is a butterfly over those lanes), the
// reduction over a node's incoming edges runs inside the lane over its CSR segment (ascending edge id, no atomics)
// with the online softmax, so scores, weights and messages never touch memory.  The complement-graph ("fake edge")
// half of the full-graph variants stays on the caller's side (graphgps_amd/layer/san_layers.py).
//
// Backward = two launches: target-keyed (recompute s and the weight, ds, g_Q, g_E; per-edge scalars ds and weight to
// a [E, H] scratch) and source-keyed (g_V, g_K from the scratch scalars).  Deterministic.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

typedef Vec<4> V4;

// sum over the LH = d_h / 4 lanes of one head (LH a power of two <= 16, heads aligned inside the wavefront)
__device__ __forceinline__ float head_sum(float v, int LH) {
  for (int m = 1; m < LH; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float dot4(const V4& a, const V4& b, const V4& c) {
  return a[0] * b[0] * c[0] + a[1] * b[1] * c[1] + a[2] * b[2] * c[2] + a[3] * b[3] * c[3];
}
__device__ __forceinline__ float dot4(const V4& a, const V4& b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// SOFTMAX: SAN2 (per-target softmax, eps 1e-16); else SAN (exp of the score clamped to [-5, 5], sum returned in z)
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void k_edge_attn_fwd(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, int64_t ld,
    const float* __restrict__ Ee, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
    const int32_t* __restrict__ eid, int64_t N, int H, int D, float scale, float* __restrict__ wv,
    float* __restrict__ z, float* __restrict__ mx, float* __restrict__ lsum) {
  const int HD = H * D, L = HD / 4, LH = D / 4;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / L;
  const bool live = node < N;                    // dead lanes still take part in the shuffles
  const int64_t nd = live ? node : N - 1;
  const int c = (int)(t - node * L) * 4;
  const int h = c / D;
  const int beg = rowptr[nd], end = live ? rowptr[nd + 1] : beg;
  const V4 q = V4::load(Q + nd * ld + c);
  V4 acc = V4::zero();
  float m = -INFINITY, l = 0.0f;
  // every lane of a wavefront walks max(deg) iterations so that the head butterflies stay converged
  int deg = end - beg;
  int maxdeg = deg;
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
  for (int it = 0; it < maxdeg; ++it) {
    const bool on = it < deg;
    const int kk = beg + it;
    float part = 0.0f;
    V4 vj = V4::zero();
    if (on) {
      const int64_t j = src[kk], id = eid[kk];
      const V4 kj = V4::load(K + j * ld + c);
      const V4 ee = V4::load(Ee + id * HD + c);
      vj = V4::load(V + j * ld + c);
      part = dot4(kj, q, ee);
    }
    const float s = head_sum(part, LH) * scale;
    if (on) {
      if (SOFTMAX) {
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn);       // first edge: exp(-inf) = 0
        const float p = __expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = acc[v] * alpha + p * vj[v];
        m = mn;
      } else {
        const float w = __expf(fminf(fmaxf(s, -5.0f), 5.0f));
        l += w;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += w * vj[v];
      }
    }
  }
  if (!live) return;
  if (SOFTMAX) {
    const float inv = deg > 0 ? 1.0f / (l + 1e-16f) : 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] *= inv;
    if ((c % D) == 0) {
      mx[nd * H + h] = m;
      lsum[nd * H + h] = l;
    }
  } else if ((c % D) == 0) {
    z[nd * H + h] = l;
  }
  acc.store(wv + nd * (int64_t)HD + c);
}

// Backward, target-keyed: ds_e, weight_e -> scratch [E, H]; g_Q (node rows), g_E (edge rows).
//   SAN2: p_e = exp(s_e - m_i) / (l_i + 1e-16), ds_e = p_e (g_wv_i . V_j - g_wv_i . wv_i)
//   SAN : w_e = exp(clamp(s_e)),  ds_e = (g_wv_i . V_j + g_z_i) w_e [-5 < s_e < 5]
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void k_edge_attn_bwd_dst(
    const float* __restrict__ g_wv, const float* __restrict__ g_z, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ V, int64_t ld, const float* __restrict__ Ee,
    const float* __restrict__ wv, const float* __restrict__ mx, const float* __restrict__ lsum,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid, int64_t N,
    int H, int D, float scale, float* __restrict__ g_Q, float* __restrict__ g_E, float* __restrict__ ds_out,
    float* __restrict__ w_out) {
  const int HD = H * D, L = HD / 4, LH = D / 4;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / L;
  const bool live = node < N;
  const int64_t nd = live ? node : N - 1;
  const int c = (int)(t - node * L) * 4;
  const int h = c / D;
  const int beg = rowptr[nd], end = live ? rowptr[nd + 1] : beg;
  const V4 q = V4::load(Q + nd * ld + c);
  const V4 gw = V4::load(g_wv + nd * (int64_t)HD + c);
  float m = 0.0f, inv = 0.0f, delta = 0.0f, gz = 0.0f;
  if (SOFTMAX) {
    m = mx[nd * H + h];
    inv = 1.0f / (lsum[nd * H + h] + 1e-16f);
    const V4 o = V4::load(wv + nd * (int64_t)HD + c);
    delta = head_sum(dot4(gw, o), LH);
  } else {
    gz = g_z ? g_z[nd * H + h] : 0.0f;
  }
  V4 gq = V4::zero();
  int deg = end - beg, maxdeg = deg;
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
  for (int it = 0; it < maxdeg; ++it) {
    const bool on = it < deg;
    const int kk = beg + it;
    float part = 0.0f, dpart = 0.0f;
    V4 kj = V4::zero(), ee = V4::zero();
    int64_t id = 0;
    if (on) {
      const int64_t j = src[kk];
      id = eid[kk];
      kj = V4::load(K + j * ld + c);
      ee = V4::load(Ee + id * HD + c);
      const V4 vj = V4::load(V + j * ld + c);
      part = dot4(kj, q, ee);
      dpart = dot4(gw, vj);
    }
    const float s = head_sum(part, LH) * scale;
    const float dp = head_sum(dpart, LH);
    if (on) {
      float w, ds;
      if (SOFTMAX) {
        w = __expf(s - m) * inv;
        ds = w * (dp - delta);
      } else {
        w = __expf(fminf(fmaxf(s, -5.0f), 5.0f));
        ds = (s > -5.0f && s < 5.0f) ? (dp + gz) * w : 0.0f;
      }
      const float dss = ds * scale;
      V4 ge;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        gq[v] += dss * kj[v] * ee[v];
        ge[v] = dss * kj[v] * q[v];
      }
      ge.store(g_E + id * HD + c);
      if ((c % D) == 0) {
        ds_out[id * H + h] = dss;
        w_out[id * H + h] = w;
      }
    }
  }
  if (live) gq.store(g_Q + nd * (int64_t)HD + c);
}

// Backward, source-keyed: g_V_j = sum_{j->i} w_e g_wv_i ;  g_K_j = sum_{j->i} ds_e Q_i o E_e
__global__ __launch_bounds__(256) void k_edge_attn_bwd_src(
    const float* __restrict__ g_wv, const float* __restrict__ Q, int64_t ld, const float* __restrict__ Ee,
    const float* __restrict__ ds_in, const float* __restrict__ w_in, const int32_t* __restrict__ rowptr_s,
    const int32_t* __restrict__ dst, const int32_t* __restrict__ eid_s, int64_t N, int H, int D,
    float* __restrict__ g_K, float* __restrict__ g_V) {
  const int HD = H * D, L = HD / 4;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / L;
  if (node >= N) return;
  const int c = (int)(t - node * L) * 4;
  const int h = c / D;
  V4 gk = V4::zero(), gv = V4::zero();
  for (int kk = rowptr_s[node]; kk < rowptr_s[node + 1]; ++kk) {
    const int64_t i = dst[kk], id = eid_s[kk];
    const float dss = ds_in[id * H + h], w = w_in[id * H + h];
    const V4 qi = V4::load(Q + i * ld + c);
    const V4 ee = V4::load(Ee + id * HD + c);
    const V4 gw = V4::load(g_wv + i * (int64_t)HD + c);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      gk[v] += dss * qi[v] * ee[v];
      gv[v] += w * gw[v];
    }
  }
  gk.store(g_K + node * (int64_t)HD + c);
  gv.store(g_V + node * (int64_t)HD + c);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }
inline bool head_ok(int D) { return D == 4 || D == 8 || D == 16 || D == 32 || D == 64; }

}  // namespace

extern "C" {

int gps_edge_attn_supported(int H, int D) { return H > 0 && head_ok(D); }

int gps_edge_attn_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* Ee,
                      const int32_t* rowptr_dst, const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N,
                      int64_t E, int H, int D, float scale, int softmax, float* wv, float* z, float* mx, float* lsum,
                      gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && gps_edge_attn_supported(H, D) && ld >= (int64_t)H * D && ld % 4 == 0,
              "gps_edge_attn_fwd: bad sizes (H=%d, head dim %d must be 4, 8, 16, 32 or 64)", H, D);
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(Q && K && V && rowptr_dst && wv && (E == 0 || (Ee && src_by_dst && eid_by_dst)),
              "gps_edge_attn_fwd: null buffer");
  GPS_REQUIRE(softmax ? (mx && lsum) : (z != nullptr), "gps_edge_attn_fwd: missing statistics buffer");
  GPS_REQUIRE(al16(Q) && al16(K) && al16(V) && al16(Ee) && al16(wv), "gps_edge_attn_fwd: 16-byte alignment");
  const int64_t work = N * (int64_t)(H * D / 4);
  hipStream_t s = gps::as_stream(stream);
  if (softmax)
    k_edge_attn_fwd<true><<<gps::grid_for(work, 256), 256, 0, s>>>(Q, K, V, ld, Ee, rowptr_dst, src_by_dst, eid_by_dst,
                                                                   N, H, D, scale, wv, z, mx, lsum);
  else
    k_edge_attn_fwd<false><<<gps::grid_for(work, 256), 256, 0, s>>>(Q, K, V, ld, Ee, rowptr_dst, src_by_dst, eid_by_dst,
                                                                    N, H, D, scale, wv, z, mx, lsum);
  return gps::launch_status("gps_edge_attn_fwd");
}

int gps_edge_attn_bwd(const float* g_wv, const float* g_z, const float* Q, const float* K, const float* V, int64_t ld,
                      const float* Ee, const float* wv, const float* mx, const float* lsum,
                      const int32_t* rowptr_dst, const int32_t* src_by_dst, const int32_t* eid_by_dst,
                      const int32_t* rowptr_src, const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N,
                      int64_t E, int H, int D, float scale, int softmax, float* g_Q, float* g_K, float* g_V,
                      float* g_E, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && gps_edge_attn_supported(H, D) && ld >= (int64_t)H * D && ld % 4 == 0,
              "gps_edge_attn_bwd: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(g_wv && Q && K && V && rowptr_dst && rowptr_src && g_Q && g_K && g_V, "gps_edge_attn_bwd: null node buffer");
  GPS_REQUIRE(E == 0 || (Ee && g_E && ws && src_by_dst && eid_by_dst && dst_by_src && eid_by_src),
              "gps_edge_attn_bwd: null edge buffer (ws needs 2 * E * H floats)");
  GPS_REQUIRE(!softmax || (wv && mx && lsum), "gps_edge_attn_bwd: missing saved statistics");
  const int64_t work = N * (int64_t)(H * D / 4);
  hipStream_t s = gps::as_stream(stream);
  float* ds = ws;
  float* w = ws ? ws + E * (int64_t)H : nullptr;
  if (softmax)
    k_edge_attn_bwd_dst<true><<<gps::grid_for(work, 256), 256, 0, s>>>(g_wv, g_z, Q, K, V, ld, Ee, wv, mx, lsum,
                                                                       rowptr_dst, src_by_dst, eid_by_dst, N, H, D,
                                                                       scale, g_Q, g_E, ds, w);
  else
    k_edge_attn_bwd_dst<false><<<gps::grid_for(work, 256), 256, 0, s>>>(g_wv, g_z, Q, K, V, ld, Ee, wv, mx, lsum,
                                                                        rowptr_dst, src_by_dst, eid_by_dst, N, H, D,
                                                                        scale, g_Q, g_E, ds, w);
  k_edge_attn_bwd_src<<<gps::grid_for(work, 256), 256, 0, s>>>(g_wv, Q, ld, Ee, ds, w, rowptr_src, dst_by_src,
                                                               eid_by_src, N, H, D, g_K, g_V);
  return gps::launch_status("gps_edge_attn_bwd");
}

}  // extern "C"
