// rtc_embed.cpp -- the text of the device headers, carried inside libgslnls_hip.so for the in-process compiler
// (rtc_host.hpp): the kernels a new formula is compiled into are the very templates this library was built from.
// Plain .incbin of the files next to this one (the Makefile lists them as prerequisites).
namespace gslnls
{
struct RtcHeader
{
    const char *name;
    const char *begin, *end;
};
}

#define GSLNLS_EMBED(sym, file)                                                                     \
    __asm__(".section .rodata\n"                                                                    \
            ".global gslnls_rtc_" #sym "_begin\n.global gslnls_rtc_" #sym "_end\n"                  \
            "gslnls_rtc_" #sym "_begin:\n.incbin \"" file "\"\n"                                    \
            "gslnls_rtc_" #sym "_end:\n.byte 0\n.previous\n");                                      \
    extern "C" const char gslnls_rtc_##sym##_begin[], gslnls_rtc_##sym##_end[];

GSLNLS_EMBED(prelude, "rtc_prelude.hpp")
GSLNLS_EMBED(lm_core, "lm_core.hpp")
GSLNLS_EMBED(lm_decide, "lm_decide.hpp")
GSLNLS_EMBED(devmath, "devmath.hpp")
GSLNLS_EMBED(models, "models.hpp")
GSLNLS_EMBED(rowops, "rowops.hpp")
GSLNLS_EMBED(dense_kernels, "dense_kernels.hpp")
GSLNLS_EMBED(wide_core, "wide_core.hpp")
GSLNLS_EMBED(wide_kernels, "wide_kernels.hpp")
GSLNLS_EMBED(bd_model_kernels, "bd_model_kernels.hpp")

namespace gslnls
{
const RtcHeader *rtc_embedded_headers(int *count)
{
    static const RtcHeader h[] = {
        {"rtc_prelude.hpp", gslnls_rtc_prelude_begin, gslnls_rtc_prelude_end},
        {"lm_core.hpp", gslnls_rtc_lm_core_begin, gslnls_rtc_lm_core_end},
        {"lm_decide.hpp", gslnls_rtc_lm_decide_begin, gslnls_rtc_lm_decide_end},
        {"devmath.hpp", gslnls_rtc_devmath_begin, gslnls_rtc_devmath_end},
        {"models.hpp", gslnls_rtc_models_begin, gslnls_rtc_models_end},
        {"rowops.hpp", gslnls_rtc_rowops_begin, gslnls_rtc_rowops_end},
        {"dense_kernels.hpp", gslnls_rtc_dense_kernels_begin, gslnls_rtc_dense_kernels_end},
        {"wide_core.hpp", gslnls_rtc_wide_core_begin, gslnls_rtc_wide_core_end},
        {"wide_kernels.hpp", gslnls_rtc_wide_kernels_begin, gslnls_rtc_wide_kernels_end},
        {"bd_model_kernels.hpp", gslnls_rtc_bd_model_kernels_begin, gslnls_rtc_bd_model_kernels_end},
    };
    *count = (int)(sizeof h / sizeof h[0]);
    return h;
}
}
