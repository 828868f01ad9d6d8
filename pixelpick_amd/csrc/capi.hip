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

EventHook& event_hook()
{
    static thread_local EventHook h{nullptr, nullptr, 0, 0};   // per calling thread, like every other debugging knob
    return h;
}

}  // namespace pp

extern "C" {

void pp_debug_set_kernel_events(void** starts, void** stops, int n)
{
    pp::EventHook& h = pp::event_hook();
    h.start = reinterpret_cast<hipEvent_t*>(starts);
    h.stop = reinterpret_cast<hipEvent_t*>(stops);
    h.n = (starts && stops) ? n : 0;
    h.i = 0;
}

int pp_version(void) { return 100; }  // 0.1.0

const char* pp_last_error(void) { return pp::err_buf(); }

}
