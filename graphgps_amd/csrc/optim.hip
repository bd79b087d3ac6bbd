// Gradient-norm clip + AdamW over a flat fp32 parameter arena: two launches per optimizer step.
//
// Replaces, for the GPS training step, the caller side of the hot path
//   torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.optim.clip_grad_norm_value)
//   optimizer.step()                      (graphgps/train/custom_train.py:33-37)
// with optimizer = torch.optim.AdamW(params, lr=base_lr, weight_decay=weight_decay)
// (graphgps/optimizer/extra_optimizers.py:21-24; betas (0.9, 0.999), eps 1e-8).
//
// Layout: all parameters live in ONE flat buffer p (graphgps_amd/optim.py re-points every
// nn.Parameter into it), gradients in a flat buffer g of the same layout, moments m, v likewise.
// A chunk table cuts the arena into <= kChunk-element pieces that never straddle a parameter, so a
// parameter that received no gradient this step (active[param] == 0) is skipped exactly like
// torch skips `p.grad is None`.  HBM-bound: 4 reads + 3 writes per element = 28 B/param.
//
//   K1 k_sqnorm : per-chunk sum of g^2 (deterministic in-block tree)        -> ws[chunk]
//                 (+ advances the device-resident global and per-parameter step counters)
//   K2 k_adamw  : every block re-reduces ws[] in the same fixed order (fp64) -> clip coefficient,
//                 then updates its chunk.  No host sync, no atomics, replayable from a hipGraph.
#include "gps_common.hpp"
#include "vec.hpp"

namespace {

constexpr int kChunk = 4096;   // floats per block (256 threads x 4 float4)
constexpr int kThreads = 256;

// hyper[] slots (device DOUBLES owned by the caller: torch derives step_size and the bias
// corrections in Python doubles, and 1 - 0.999f is already off by 1.3e-5 relative)
enum { H_LR = 0, H_B1, H_B2, H_EPS, H_WD, H_MAXNORM, H_STEP, H_NORM, H_COUNT };

__device__ __forceinline__ float block_sum(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < kThreads / 64; ++i) t += sh[i];
  return t;  // valid on thread 0
}

__global__ __launch_bounds__(kThreads) void k_sqnorm(const float* __restrict__ g,
                                                     const int64_t* __restrict__ chunk_off,
                                                     const int32_t* __restrict__ chunk_len,
                                                     const int32_t* __restrict__ chunk_param,
                                                     const uint8_t* __restrict__ active,
                                                     double* __restrict__ hyper,
                                                     float* __restrict__ pstep,
                                                     float* __restrict__ ws) {
  __shared__ float sh[kThreads / 64];
  const int c = blockIdx.x;
  const int prm = chunk_param[c];
  const bool on = active[prm] != 0;
  if (threadIdx.x == 0) {
    if (c == 0) hyper[H_STEP] += 1.0;
    // torch keeps one step counter PER PARAMETER (a parameter without a gradient does not
    // advance): the first chunk of every active parameter advances it
    if (on && (c == 0 || chunk_param[c - 1] != prm)) pstep[prm] += 1.0f;
  }
  float acc = 0.f;
  if (on) {
    const float* gp = g + chunk_off[c];
    const int len = chunk_len[c];
    if ((reinterpret_cast<uintptr_t>(gp) & 15) == 0) {
      const int n4 = len >> 2;
      const float4* g4 = reinterpret_cast<const float4*>(gp);
      for (int i = threadIdx.x; i < n4; i += kThreads) {
        const float4 x = g4[i];
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      }
      for (int i = (n4 << 2) + threadIdx.x; i < len; i += kThreads) acc += gp[i] * gp[i];
    } else {
      for (int i = threadIdx.x; i < len; i += kThreads) acc += gp[i] * gp[i];
    }
  }
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) ws[c] = t;
}

struct AdamCoef {
  float decay, one_m_b1, b2, one_m_b2, step_size, inv_bc2_sqrt, eps, clip;
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, const AdamCoef& k) {
  g *= k.clip;
  p *= k.decay;
  m += k.one_m_b1 * (g - m);
  v = k.b2 * v + k.one_m_b2 * g * g;
  const float denom = sqrtf(v) * k.inv_bc2_sqrt + k.eps;
  p -= k.step_size * (m / denom);
}

__global__ __launch_bounds__(kThreads) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const int64_t* __restrict__ chunk_off,
                                                    const int32_t* __restrict__ chunk_len,
                                                    const int32_t* __restrict__ chunk_param,
                                                    const uint8_t* __restrict__ active,
                                                    int64_t n_chunks, double* __restrict__ hyper,
                                                    const float* __restrict__ pstep,
                                                    const float* __restrict__ ws) {
  __shared__ double shd[kThreads / 64];
  __shared__ AdamCoef shk;
  const int c = blockIdx.x;
  const bool on = active[chunk_param[c]] != 0;
  if (!on && c != 0) return;   // block 0 always publishes the norm
  // total ||g||^2: every block sums ws[] in the same order -> identical coefficient everywhere
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n_chunks; i += kThreads) acc += static_cast<double>(ws[i]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < kThreads / 64; ++i) tot += shd[i];
    const float norm = static_cast<float>(sqrt(tot));
    const double lr = hyper[H_LR], b1 = hyper[H_B1], b2 = hyper[H_B2];
    const float maxn = static_cast<float>(hyper[H_MAXNORM]);
    const double step = static_cast<double>(pstep[chunk_param[c]]);
    const double bc1 = 1.0 - pow(b1, step);
    const double bc2 = 1.0 - pow(b2, step);
    AdamCoef k;
    k.decay = static_cast<float>(1.0 - lr * hyper[H_WD]);
    k.one_m_b1 = static_cast<float>(1.0 - b1);
    k.b2 = static_cast<float>(b2);
    k.one_m_b2 = static_cast<float>(1.0 - b2);
    k.step_size = static_cast<float>(lr / bc1);
    k.inv_bc2_sqrt = static_cast<float>(1.0 / sqrt(bc2));
    k.eps = static_cast<float>(hyper[H_EPS]);
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to <= 1
    k.clip = maxn > 0.f ? fminf(1.0f, maxn / (norm + 1e-6f)) : 1.0f;
    shk = k;
    if (c == 0) hyper[H_NORM] = static_cast<double>(norm);
  }
  __syncthreads();
  if (!on) return;
  const AdamCoef k = shk;
  const int64_t off = chunk_off[c];
  const int len = chunk_len[c];
  float* pp = p + off;
  const float* gp = g + off;
  float* mp = m + off;
  float* vp = v + off;
  int done = 0;
  if ((reinterpret_cast<uintptr_t>(pp) & 15) == 0) {   // p, g, m, v share the arena layout
    const int n4 = len >> 2;
    for (int i = threadIdx.x; i < n4; i += kThreads) {
      float4 P = reinterpret_cast<float4*>(pp)[i];
      const float4 G = reinterpret_cast<const float4*>(gp)[i];
      float4 M = reinterpret_cast<float4*>(mp)[i];
      float4 V = reinterpret_cast<float4*>(vp)[i];
      adam_update(P.x, G.x, M.x, V.x, k);
      adam_update(P.y, G.y, M.y, V.y, k);
      adam_update(P.z, G.z, M.z, V.z, k);
      adam_update(P.w, G.w, M.w, V.w, k);
      reinterpret_cast<float4*>(pp)[i] = P;
      reinterpret_cast<float4*>(mp)[i] = M;
      reinterpret_cast<float4*>(vp)[i] = V;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < len; i += kThreads) {
    float P = pp[i], M = mp[i], V = vp[i];
    adam_update(P, gp[i], M, V, k);
    pp[i] = P;
    mp[i] = M;
    vp[i] = V;
  }
}

}  // namespace

extern "C" {

int gps_optim_chunk(void) { return kChunk; }

int gps_adamw_step(float* p, const float* g, float* m, float* v, const int64_t* chunk_off,
                   const int32_t* chunk_len, const int32_t* chunk_param, const uint8_t* active,
                   int64_t n_chunks, double* hyper, float* pstep, float* ws, gps_stream_t stream) {
  GPS_REQUIRE(n_chunks >= 0 && n_chunks < (int64_t(1) << 31), "gps_adamw_step: n_chunks=%lld",
              static_cast<long long>(n_chunks));
  if (n_chunks == 0) return GPS_OK;
  GPS_REQUIRE(p && g && m && v && chunk_off && chunk_len && chunk_param && active && hyper && pstep && ws,
              "gps_adamw_step: null pointer argument");
  hipStream_t s = gps::as_stream(stream);
  k_sqnorm<<<static_cast<unsigned>(n_chunks), kThreads, 0, s>>>(g, chunk_off, chunk_len, chunk_param,
                                                                 active, hyper, pstep, ws);
  k_adamw<<<static_cast<unsigned>(n_chunks), kThreads, 0, s>>>(p, g, m, v, chunk_off, chunk_len,
                                                                chunk_param, active, n_chunks, hyper,
                                                                pstep, ws);
  return gps::launch_status("gps_adamw_step");
}

}  // extern "C"
