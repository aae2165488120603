// Thread-local error string + ABI version for libssm_hip.so.
#include "ssm_common.h"

#include <cstdlib>

namespace ssm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
std::atomic<int> &splitk_switch(int which) {
    static std::atomic<int> sw[2] = {{-1}, {-1}};
    std::atomic<int> &v = sw[which ? 1 : 0];
    if (v.load(std::memory_order_relaxed) < 0) {
        const char *e = getenv(which ? "SSM_CONV_SPLITK" : "SSM_WINO_SPLITK");
        int init = e ? (atoi(e) != 0) : 1, expected = -1;
        v.compare_exchange_strong(expected, init);
    }
    return v;
}
}  // namespace ssm

extern "C" int ssm_abi_version(void) { return 1; }
extern "C" const char *ssm_last_error_string(void) { return ssm::g_err; }
extern "C" void ssm_plane_dims(int H, int W, int *Hp, int *Wp) {
    if (Hp) *Hp = H + 2 * SSM_PADY;
    if (Wp) *Wp = (W + 2 * SSM_PADX + 3) / 4 * 4;
}
// Run-time twin of $SSM_WINO_SPLITK / $SSM_CONV_SPLITK: wino / conv = 1 on, 0 off, -1 leave as is.  Returns the previous state as
// (wino | conv << 1).  Plans made before the call keep their split (the Python side caches KS per packed filter).
extern "C" int ssm_splitk_enable(int wino, int conv) {
    const int prev = ssm::splitk_switch(0).load() | (ssm::splitk_switch(1).load() << 1);
    if (wino >= 0) ssm::splitk_switch(0).store(wino != 0);
    if (conv >= 0) ssm::splitk_switch(1).store(conv != 0);
    return prev;
}
