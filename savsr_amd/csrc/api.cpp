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
}  // namespace savsr

extern "C" {
const char* savsr_version(void) { return "savsr_hip 0.2 (gfx950, split-bf16 MFMA, channel-last)"; }
const char* savsr_last_error(void) { return savsr::g_err; }
int savsr_abi_version(void) { return SAVSR_ABI_VERSION; }
}
