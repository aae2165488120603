// Thread-local error string + ABI version for libssm_hip.so.
#include "ssm_common.h"

namespace ssm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ssm

extern "C" int ssm_abi_version(void) { return 1; }
extern "C" const char *ssm_last_error_string(void) { return ssm::g_err; }
extern "C" void ssm_plane_dims(int H, int W, int *Hp, int *Wp) {
    if (Hp) *Hp = H + 2 * SSM_PADY;
    if (Wp) *Wp = (W + 2 * SSM_PADX + 3) / 4 * 4;
}
