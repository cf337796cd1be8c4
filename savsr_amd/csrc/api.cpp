// Version / error plumbing of the C ABI.
#include "common.hpp"

namespace savsr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dynamic_lds(const void* fn, int bytes, const char* what) {
    constexpr int MAX_DEV = 64, MAX_FN = 64;
    struct Slot { const void* fn; unsigned long long devmask; };
    static Slot slots[MAX_FN] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("%s: hipGetDevice failed: %s", what, hipGetErrorString(e)); return (int)e; }
    Slot* s = nullptr;
    for (int i = 0; i < MAX_FN; ++i) {
        if (slots[i].fn == fn) { s = &slots[i]; break; }
        if (!slots[i].fn) { slots[i].fn = fn; s = &slots[i]; break; }
    }
    if (s && dev < MAX_DEV && (s->devmask >> dev & 1ull)) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%d bytes LDS) failed: %s", what, bytes, hipGetErrorString(e)); return (int)e; }
    if (s && dev < MAX_DEV) s->devmask |= 1ull << dev;
    return 0;
}
}  // namespace savsr

extern "C" {
const char* savsr_version(void) { return "savsr_hip 0.3 (gfx950, split-bf16 MFMA, channel-last)"; }
const char* savsr_last_error(void) { return savsr::g_err; }
int savsr_abi_version(void) { return SAVSR_ABI_VERSION; }
}
