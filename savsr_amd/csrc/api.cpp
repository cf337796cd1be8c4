// Version / error plumbing of the C ABI.
#include "common.hpp"

#include <mutex>

namespace savsr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dynamic_lds(const void* fn, int bytes, const char* what) {
    // (function, device) -> largest dynamic-LDS size the attribute has been set to.  The table is the only mutable state
    // shared between calls, so it sits behind a mutex (one engine per device and thread in a DataParallel-style process);
    // the attribute is set again whenever a caller asks for more than the recorded size.
    constexpr int MAX_DEV = 64, MAX_FN = 64;
    struct Slot { const void* fn; int bytes[MAX_DEV]; };
    static Slot slots[MAX_FN] = {};
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("%s: hipGetDevice failed: %s", what, hipGetErrorString(e)); return (int)e; }
    std::lock_guard<std::mutex> lock(mu);
    Slot* s = nullptr;
    for (int i = 0; i < MAX_FN; ++i) {
        if (slots[i].fn == fn) { s = &slots[i]; break; }
        if (!slots[i].fn) { slots[i].fn = fn; s = &slots[i]; break; }
    }
    const bool tracked = s && dev >= 0 && dev < MAX_DEV;
    if (tracked && s->bytes[dev] >= bytes) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%d bytes LDS) failed: %s", what, bytes, hipGetErrorString(e)); return (int)e; }
    if (tracked) s->bytes[dev] = bytes;
    return 0;
}
}  // namespace savsr

extern "C" {
const char* savsr_version(void) { return "savsr_hip 0.4 (gfx950, split-bf16 MFMA, channel-last)"; }
const char* savsr_last_error(void) { return savsr::g_err; }
int savsr_abi_version(void) { return SAVSR_ABI_VERSION; }
#ifndef SAVSR_SOURCE_HASH
#define SAVSR_SOURCE_HASH "unknown"
#endif
#ifndef SAVSR_SATU_HASH
#define SAVSR_SATU_HASH "unknown"
#endif
const char* savsr_source_hash(void) { return SAVSR_SOURCE_HASH; }
const char* savsr_source_hash_satu(void) { return SAVSR_SATU_HASH; }
int savsr_prepare_device(void) {
    if (int rc = savsr::conv_prepare_device()) return rc;
    return savsr::satu_prepare_device();
}
}
