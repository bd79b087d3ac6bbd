// Sum-of-embeddings encoders over small vocabularies (round 5): OGB's AtomEncoder / BondEncoder as the reference registers
// them (graphgps/encoder/atom_encoder... via ogb.graphproppred.mol_encoder: `x_embedding += emb[i](x[:, i])` over the 9 atom /
// 3 bond feature columns), the AST type + depth tables (graphgps/encoder/ast_encoder.py:35-83) and the TypeDict encoders
// (graphgps/encoder/type_dict_encoder.py).
//   forward   out[r, :] = sum_i table_i[feats[r, i], :]     -- ONE launch, the tables read in place (a few KB each: cache
//             hits), columns added in index order i = 0 .. k-1, i.e. exactly the reference's summation order.  Rounds 1-4
//             built a multi-hot matrix (zeros, index add, scatter_, pad, cat of the tables) and multiplied it by the
//             stacked tables: 6 launches, and in the replayed pcqm4m step ~90 us of idle time in front of four of the ATen
//             index kernels of the EDGE encoder (profiles/r05_*timeline*).
//   backward  the table gradients stay ONE GEMM multihot^T g (deterministic, what the multi-hot form was built for); the
//             multi-hot matrix it needs is written by ONE launch here (zeros and ones in the same pass).
#include "gps_common.hpp"

namespace {

constexpr int kMaxTables = 16;
struct EmbedArgs {
  const float* tab[kMaxTables];   // table i: [vocab_i, emb] row-major, contiguous
  int vocab[kMaxTables];
  int off[kMaxTables];            // first multi-hot column of table i
  const int64_t* feats;           // [R, k] (row stride ld)
  int64_t ld, R;
  int k, emb, vpad;
};

// thread = (row, 4 consecutive channels)
__global__ __launch_bounds__(256) void k_embed_sum(const EmbedArgs A, float* __restrict__ out) {
  const int q = A.emb >> 2;
  const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int64_t r = t / q;
  if (r >= A.R) return;
  const int c = (int)(t - r * q) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t id[kMaxTables];
#pragma unroll
  for (int i = 0; i < kMaxTables; ++i)
    if (i < A.k) {                               // (kernel-uniform) every index of the row requested before the first use
      const int64_t v = A.feats[r * A.ld + i];
      id[i] = v < 0 ? 0 : (v >= A.vocab[i] ? A.vocab[i] - 1 : v);      // nn.Embedding raises on these; never fault here
    }
#pragma unroll
  for (int i = 0; i < kMaxTables; ++i)
    if (i < A.k) {
      const float4 w = *reinterpret_cast<const float4*>(A.tab[i] + id[i] * A.emb + c);
      acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
  *reinterpret_cast<float4*>(out + r * (int64_t)A.emb + c) = acc;
}

// thread = (row, 4 consecutive multi-hot columns): 1.0 where some feature of the row lands, 0.0 elsewhere
__global__ __launch_bounds__(256) void k_multihot_fill(const EmbedArgs A, float* __restrict__ out) {
  const int q = A.vpad >> 2;
  const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
  const int64_t r = t / q;
  if (r >= A.R) return;
  const int c = (int)(t - r * q) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < kMaxTables; ++i)
    if (i < A.k) {
      const int64_t f = A.feats[r * A.ld + i];
      const int col = A.off[i] + (int)(f < 0 ? 0 : (f >= A.vocab[i] ? A.vocab[i] - 1 : f)) - c;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = col == j ? 1.0f : v[j];
    }
  *reinterpret_cast<float4*>(out + r * (int64_t)A.vpad + c) = make_float4(v[0], v[1], v[2], v[3]);
}

int fill_args(EmbedArgs& A, const int64_t* feats, int64_t ld, int64_t R, int k, const float* const* tables, const int* vocab,
              int emb) {
  GPS_REQUIRE(k >= 1 && k <= kMaxTables && R >= 0 && ld >= k && emb >= 4 && emb % 4 == 0,
              "gps_embed: 1 <= k <= %d feature columns, emb %% 4 == 0 (k=%d emb=%d)", kMaxTables, k, emb);
  GPS_REQUIRE(feats || R == 0, "gps_embed: null feature matrix");
  int off = 0;
  for (int i = 0; i < k; ++i) {
    GPS_REQUIRE(vocab[i] >= 1 && (!tables || (tables[i] && (uintptr_t)tables[i] % 16 == 0)), "gps_embed: table %d", i);
    A.tab[i] = tables ? tables[i] : nullptr;
    A.vocab[i] = vocab[i];
    A.off[i] = off;
    off += vocab[i];
  }
  A.feats = feats; A.ld = ld; A.R = R; A.k = k; A.emb = emb;
  A.vpad = (off + 3) / 4 * 4;
  return GPS_OK;
}

}  // namespace

extern "C" {

int gps_embed_sum(const int64_t* feats, int64_t ld, int64_t R, int k, const float* const* tables, const int* vocab, int emb,
                  float* out, gps_stream_t stream) {
  EmbedArgs A{};
  const int rc = fill_args(A, feats, ld, R, k, tables, vocab, emb);
  if (rc != GPS_OK) return rc;
  GPS_REQUIRE(tables, "gps_embed_sum: null table list");
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(out && (uintptr_t)out % 16 == 0, "gps_embed_sum: null / misaligned output");
  k_embed_sum<<<gps::grid_for(R * (int64_t)(emb / 4), 256), 256, 0, gps::as_stream(stream)>>>(A, out);
  return gps::launch_status("gps_embed_sum");
}

int gps_multihot_columns(int k, const int* vocab) {
  if (k < 1 || k > kMaxTables || !vocab) return 0;
  int off = 0;
  for (int i = 0; i < k; ++i) off += vocab[i];
  return (off + 3) / 4 * 4;
}

int gps_multihot_fill(const int64_t* feats, int64_t ld, int64_t R, int k, const int* vocab, float* out, gps_stream_t stream) {
  EmbedArgs A{};
  const int rc = fill_args(A, feats, ld, R, k, nullptr, vocab, 4);
  if (rc != GPS_OK) return rc;
  if (R == 0) return GPS_OK;
  GPS_REQUIRE(out && (uintptr_t)out % 16 == 0, "gps_multihot_fill: null / misaligned output");
  k_multihot_fill<<<gps::grid_for(R * (int64_t)(A.vpad / 4), 256), 256, 0, gps::as_stream(stream)>>>(A, out);
  return gps::launch_status("gps_multihot_fill");
}

}  // extern "C"
