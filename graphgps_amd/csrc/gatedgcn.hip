// GatedGCN sparse core: gather -> gate -> segment-reduce -> node update, in one pass.
//
// Reference semantics: graphgps/layer/gatedgcn_layer.py:67-70 (propagate), :90-107 (message),
// :109-126 (aggregate), :128-136 (update).  Index convention (PyG flow source_to_target):
// j = edge_index[0] (source), i = edge_index[1] (target); sums are keyed by the TARGET.
//
// Mapping to the machine (HBM-bound, no MFMA): lane = VEC consecutive channels of one node, a
// node row of d floats is d/VEC consecutive lanes, so every access (node rows Ax/Bx/Dx/Ex, the
// gathered source rows, the Ce/e_hat edge rows) is a whole-row coalesced burst.  The reduction
// over a node's incoming edges runs sequentially inside the lane over the CSR segment: no
// cross-lane traffic, no atomics, and the summation order (ascending original edge id) is the
// order of the reference's CPU scatter_add -> results are reproducible run to run.
//
// Algorithmic HBM bytes (fp32, per layer):  fwd 8*E*d + 20*N*d (+8*N*d saved
// aggr/den in training), bwd 12*E*d + 28*N*d; index traffic 4(N+1)+8E per pass.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

// GATE: the EquivStableLapPE variant (gatedgcn_layer.py:101-104): sigma_ij is multiplied by a per-edge
// scalar r_ij in (0,1) (r_edge[edge id]) before it gates and normalises.
template <int VEC, bool SAVE, bool GATE>
__global__ __launch_bounds__(256) void k_gatedgcn_fwd(
    const float* __restrict__ Ax, const float* __restrict__ Bx, const float* __restrict__ Dx,
    const float* __restrict__ Ex, int64_t ld, const float* __restrict__ Ce,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
    const int32_t* __restrict__ eid, int64_t N, int d, float* __restrict__ x_tilde,
    float* __restrict__ e_hat, float* __restrict__ aggr_out, float* __restrict__ den_out,
    const float* __restrict__ r_edge) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  const Vec<VEC> dx = Vec<VEC>::load(Dx + node * ld + c);
  Vec<VEC> num = Vec<VEC>::zero(), den = Vec<VEC>::zero();
  for (int k = beg; k < end; ++k) {
    const int64_t j = src[k];
    const int64_t id = eid[k];
    const Vec<VEC> ex = Vec<VEC>::load(Ex + j * ld + c);
    const Vec<VEC> bx = Vec<VEC>::load(Bx + j * ld + c);
    const Vec<VEC> ce = Vec<VEC>::load(Ce + id * d + c);
    Vec<VEC> eh;
    const float rr = GATE ? r_edge[id] : 1.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      eh[v] = (dx[v] + ex[v]) + ce[v];  // e_ij = Dx_i + Ex_j + Ce            (:96)
      float s = sigmoidf_fast(eh[v]);  //                                     (:97)
      if (GATE) s = s * rr;             // sigma_ij * r_ij                     (:101-104)
      num[v] += s * bx[v];                    // scatter(sigma*Bx_j)           (:117-119)
      den[v] += s;                            // scatter(sigma)                (:121-123)
    }
    eh.store(e_hat + id * d + c);  // self.e = e_ij, returned in edge order    (:106,134)
  }
  const Vec<VEC> ax = Vec<VEC>::load(Ax + node * ld + c);
  Vec<VEC> xt, ag;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    ag[v] = num[v] / (den[v] + 1e-6f);  //                                    (:125)
    xt[v] = ax[v] + ag[v];              //                                    (:133)
  }
  xt.store(x_tilde + node * (int64_t)d + c);
  if (SAVE) {
    ag.store(aggr_out + node * (int64_t)d + c);
    den.store(den_out + node * (int64_t)d + c);
  }
}

// Backward pass 1, keyed by TARGET i:
//   a_i = g_x_i / D_i,  b_i = -g_x_i * aggr_i / D_i           (D_i = den_i + 1e-6)
//   delta_ij = g_e_ij + (a_i * Bx_j + b_i) * sig_ij * (1 - sig_ij)
//   g_Ce[eid] = delta_ij ;  g_Dx_i = sum_j delta_ij ;  g_Ax_i = g_x_i
template <int VEC, bool GATE>
__global__ __launch_bounds__(256) void k_gatedgcn_bwd_dst(
    const float* __restrict__ g_x, const float* __restrict__ g_e, const float* __restrict__ e_hat,
    const float* __restrict__ Bx, int64_t ld, const float* __restrict__ aggr,
    const float* __restrict__ den, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ src, const int32_t* __restrict__ eid, int64_t N, int d,
    float* __restrict__ g_Ce, float* __restrict__ g_Ax, float* __restrict__ g_Dx, int64_t ldg,
    const float* __restrict__ r_edge) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  const Vec<VEC> gx = Vec<VEC>::load(g_x + node * (int64_t)d + c);
  const Vec<VEC> ag = Vec<VEC>::load(aggr + node * (int64_t)d + c);
  const Vec<VEC> dn = Vec<VEC>::load(den + node * (int64_t)d + c);
  Vec<VEC> a, b, gdx = Vec<VEC>::zero();
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const float inv = 1.0f / (dn[v] + 1e-6f);
    a[v] = gx[v] * inv;
    b[v] = -a[v] * ag[v];
  }
  for (int k = beg; k < end; ++k) {
    const int64_t j = src[k];
    const int64_t id = eid[k];
    const Vec<VEC> eh = Vec<VEC>::load(e_hat + id * d + c);
    const Vec<VEC> ge = Vec<VEC>::load(g_e + id * d + c);
    const Vec<VEC> bx = Vec<VEC>::load(Bx + j * ld + c);
    Vec<VEC> dl;
    const float rr = GATE ? r_edge[id] : 1.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float s = sigmoidf_fast(eh[v]);
      float gs = a[v] * bx[v] + b[v];          // gradient wrt the (gated) sigma
      if (GATE) gs = gs * rr;
      dl[v] = ge[v] + gs * (s * (1.0f - s));
      gdx[v] += dl[v];
    }
    dl.store(g_Ce + id * d + c);
  }
  gx.store(g_Ax + node * ldg + c);
  gdx.store(g_Dx + node * ldg + c);
}

// Backward pass 2, keyed by SOURCE j (reads the delta written by pass 1):
//   g_Ex_j = sum_{j->i} delta_ij ;   g_Bx_j = sum_{j->i} sig_ij * a_i
template <int VEC, bool GATE>
__global__ __launch_bounds__(256) void k_gatedgcn_bwd_src(
    const float* __restrict__ g_x, const float* __restrict__ e_hat, const float* __restrict__ den,
    const float* __restrict__ delta, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ dst, const int32_t* __restrict__ eid, int64_t N, int d,
    float* __restrict__ g_Bx, float* __restrict__ g_Ex, int64_t ldg, const float* __restrict__ r_edge) {
  const int lanes_per_row = d / VEC;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t node = t / lanes_per_row;
  if (node >= N) return;
  const int c = (int)(t - node * lanes_per_row) * VEC;
  const int beg = rowptr[node], end = rowptr[node + 1];
  Vec<VEC> gbx = Vec<VEC>::zero(), gex = Vec<VEC>::zero();
  for (int k = beg; k < end; ++k) {
    const int64_t i = dst[k];
    const int64_t id = eid[k];
    const Vec<VEC> dl = Vec<VEC>::load(delta + id * d + c);
    const Vec<VEC> eh = Vec<VEC>::load(e_hat + id * d + c);
    const Vec<VEC> gx = Vec<VEC>::load(g_x + i * d + c);
    const Vec<VEC> dn = Vec<VEC>::load(den + i * d + c);
    const float rr = GATE ? r_edge[id] : 1.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float s = sigmoidf_fast(eh[v]);
      if (GATE) s = s * rr;
      gex[v] += dl[v];
      gbx[v] += s * (gx[v] / (dn[v] + 1e-6f));
    }
  }
  gbx.store(g_Bx + node * ldg + c);
  gex.store(g_Ex + node * ldg + c);
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

#define GPS_GG_FWD(SAVE, GATE)                                                                      \
  k_gatedgcn_fwd<VEC, SAVE, GATE><<<gps::grid_for(work, 256), 256, 0, s>>>(                         \
      Ax, Bx, Dx, Ex, ld_node, Ce, rowptr_dst, src_by_dst, eid_by_dst, N, d, x_tilde, e_hat, aggr, den, r_edge)
#define GPS_GG_BWD(GATE)                                                                           \
  do {                                                                                             \
    k_gatedgcn_bwd_dst<VEC, GATE><<<gps::grid_for(work, 256), 256, 0, s>>>(                        \
        g_x, g_e, e_hat, Bx, ld_node, aggr, den, rowptr_dst, src_by_dst, eid_by_dst, N, d, g_Ce,   \
        g_Ax, g_Dx, ld_gnode, r_edge);                                                             \
    k_gatedgcn_bwd_src<VEC, GATE><<<gps::grid_for(work, 256), 256, 0, s>>>(                        \
        g_x, e_hat, den, g_Ce, rowptr_src, dst_by_src, eid_by_src, N, d, g_Bx, g_Ex, ld_gnode, r_edge); \
  } while (0)

extern "C" {

int gps_gatedgcn_fwd(const float* Ax, const float* Bx, const float* Dx, const float* Ex,
                     int64_t ld_node, const float* Ce, const int32_t* rowptr_dst,
                     const int32_t* src_by_dst, const int32_t* eid_by_dst, int64_t N, int64_t E,
                     int d, float* x_tilde, float* e_hat, float* aggr, float* den,
                     const float* r_edge, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_node >= d, "gps_gatedgcn_fwd: bad sizes N=%lld E=%lld d=%d ld=%lld",
              (long long)N, (long long)E, d, (long long)ld_node);
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(Ax && Bx && Dx && Ex && rowptr_dst && x_tilde, "gps_gatedgcn_fwd: null node buffer");
  GPS_REQUIRE(E == 0 || (Ce && src_by_dst && eid_by_dst && e_hat), "gps_gatedgcn_fwd: null edge buffer");
  GPS_REQUIRE((aggr == nullptr) == (den == nullptr), "gps_gatedgcn_fwd: aggr/den must both be set or both NULL");
  const bool save = aggr != nullptr;
  auto ok = [&](size_t a) {
    return aligned_to(Ax, a) && aligned_to(Bx, a) && aligned_to(Dx, a) && aligned_to(Ex, a) &&
           aligned_to(Ce, a) && aligned_to(x_tilde, a) && aligned_to(e_hat, a) &&
           aligned_to(aggr, a) && aligned_to(den, a);
  };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ld_node % 4 == 0 && ok(16), ld_node % 2 == 0 && ok(8), {
    const int64_t work = N * (int64_t)(d / VEC);
    if (save) { if (r_edge) GPS_GG_FWD(true, true); else GPS_GG_FWD(true, false); }
    else { if (r_edge) GPS_GG_FWD(false, true); else GPS_GG_FWD(false, false); }
  });
  return gps::launch_status("gps_gatedgcn_fwd");
}

int gps_gatedgcn_bwd(const float* g_x, const float* g_e, const float* e_hat, const float* Bx,
                     int64_t ld_node, const float* aggr, const float* den,
                     const int32_t* rowptr_dst, const int32_t* src_by_dst,
                     const int32_t* eid_by_dst, const int32_t* rowptr_src,
                     const int32_t* dst_by_src, const int32_t* eid_by_src, int64_t N, int64_t E,
                     int d, float* g_Ce, float* g_Ax, float* g_Bx, float* g_Dx, float* g_Ex,
                     int64_t ld_gnode, const float* r_edge, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && E >= 0 && d > 0 && ld_node >= d && ld_gnode >= d, "gps_gatedgcn_bwd: bad sizes");
  if (N == 0) return GPS_OK;
  GPS_REQUIRE(g_x && Bx && aggr && den && rowptr_dst && rowptr_src && g_Ax && g_Bx && g_Dx && g_Ex,
              "gps_gatedgcn_bwd: null node buffer");
  GPS_REQUIRE(E == 0 || (g_e && e_hat && src_by_dst && eid_by_dst && dst_by_src && eid_by_src && g_Ce),
              "gps_gatedgcn_bwd: null edge buffer");
  auto ok = [&](size_t a) {
    return aligned_to(g_x, a) && aligned_to(g_e, a) && aligned_to(e_hat, a) && aligned_to(Bx, a) &&
           aligned_to(aggr, a) && aligned_to(den, a) && aligned_to(g_Ce, a) && aligned_to(g_Ax, a) &&
           aligned_to(g_Bx, a) && aligned_to(g_Dx, a) && aligned_to(g_Ex, a);
  };
  hipStream_t s = gps::as_stream(stream);
  GPS_DISPATCH_VEC(d, ld_node % 4 == 0 && ld_gnode % 4 == 0 && ok(16),
                   ld_node % 2 == 0 && ld_gnode % 2 == 0 && ok(8), {
    const int64_t work = N * (int64_t)(d / VEC);
    if (r_edge) GPS_GG_BWD(true); else GPS_GG_BWD(false);
  });
  return gps::launch_status("gps_gatedgcn_bwd");
}

}  // extern "C"
