// capi.hip — version / error plumbing of the C ABI (include/pixelpick_hip.h).
#include "pp_common.h"

namespace pp {

char* err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace pp

extern "C" {

int pp_version(void) { return 100; }  // 0.1.0

const char* pp_last_error(void) { return pp::err_buf(); }

}
