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
int gps_abi_version(void) { return 9; }
const char* gps_last_error(void) { return gps::g_err; }
int gps_set_dropout_salt(const uint64_t* device_salt) {
  gps::g_dropout_salt = device_salt;
  return GPS_OK;
}
}
