// trace_log.hpp -- the text the reference prints while it runs when `trace = TRUE` (control_int[1] != 0):
//     iter %3d: ssr = %g, par = (...)          callback, src/nls.c:980-995 (once per iteration of the final solve)
//     mstart ssr* = ..., NSP = ..., par = (...)  src/nls_mstart.c:331-337 (every accepted stationary point)
//     multi-start algorithm finished ...        src/nls.c:510-517
//     IRLS iter: %3d, weighted ssr: ...         src/nls_irls.c:466-472
//     the summary block                         src/nls.c:610-630 (src/nls_large.c:259-273 on the large path)
// The reference prints with Rprintf from inside the loop.  The core runs below C++ frames that an R long jump must not
// cross (Rprintf checks for user interrupts), so the lines are COLLECTED here, in the order the reference prints them, and
// the binding prints the text after the call has returned (gslnls_trace_text; SURVEY.md 5: "print after the fact").
// Host only; one log per process -- the boundary is called from the R main thread (threading).
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>

namespace gslnls
{

inline std::string g_trace_log;
inline bool g_trace_on = false;

// The core's parameter order may differ from the caller's (a formula matched against a hand-written device model up to
// parameter order: gslnls_lower_formula's par_order); the printed vectors are the caller's.  g_trace_inv[k] = device index
// of the caller's k-th parameter; empty = identity.  Set by gslnls_trace_set_order before a call, consumed by that call.
inline std::vector<int> g_trace_inv;

inline void trace_begin(bool on)
{
    g_trace_log.clear();
    g_trace_on = on;
}

inline int trace_index(int k, int p) { return (int)g_trace_inv.size() == p ? g_trace_inv[k] : k; }

inline void trace_printf(const char *fmt, ...)
{
    if (!g_trace_on)
        return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_trace_log += buf;
}

// "%g, %g, ..., %g)\n"
inline void trace_vector(const double *x, int p, long stride = 1)
{
    if (!g_trace_on)
        return;
    for (int k = 0; k < p; ++k)
        trace_printf((k < p - 1) ? "%g, " : "%g)\n", x[(size_t)trace_index(k, p) * stride]);
}

} // namespace gslnls
