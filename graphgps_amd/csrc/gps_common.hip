// Error plumbing of the C ABI (include/gps_hip.h).
#include "gps_common.hpp"

namespace gps {

static thread_local char g_err[512] = "";
static const uint64_t* g_dropout_salt = nullptr;

const uint64_t* dropout_salt() { return g_dropout_salt; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
__global__ __launch_bounds__(256) void k_fill_words(uint32_t* __restrict__ dst, uint32_t value, size_t words) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= words && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    *reinterpret_cast<uint4*>(dst + i) = make_uint4(value, value, value, value);
  } else {
    for (size_t k = i; k < words && k < i + 4; ++k) dst[k] = value;
  }
}
}  // namespace

void fill_words(void* dst, uint32_t value, size_t words, hipStream_t stream) {
  if (words == 0) return;
  const size_t threads = (words + 3) / 4;
  k_fill_words<<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(static_cast<uint32_t*>(dst), value, words);
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return GPS_ELAUNCH;
  }
  return GPS_OK;
}

}  // namespace gps

extern "C" {
int gps_abi_version(void) { return 11; }
const char* gps_last_error(void) { return gps::g_err; }
int gps_set_dropout_salt(const uint64_t* device_salt) {
  gps::g_dropout_salt = device_salt;
  return GPS_OK;
}
}
