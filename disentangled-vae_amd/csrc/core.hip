// Library bookkeeping: error string, ABI version, device probe.
#include <stdarg.h>
#include "common.hpp"

namespace dvae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dvae

extern "C" int dvae_abi_version(void) { return DVAE_ABI_VERSION; }
extern "C" const char* dvae_last_error(void) { return dvae::g_err; }
extern "C" int dvae_build_has_diag(void) { return dvae::kDiagBuild ? 1 : 0; }
extern "C" int dvae_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
