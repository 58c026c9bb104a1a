// rtc_prelude.hpp -- what the device headers take from the C / C++ standard headers, for the in-process
// compiler.  hiprtc (comgr's clang) compiles device code only and finds no standard library on a deployment
// box (no <cstddef>, <type_traits>, <math.h>): its built-in prelude supplies the HIP runtime and the math
// functions; the handful of macros and typedefs below are the rest.  Included instead of the standard headers
// when __HIPCC_RTC__ is defined (lm_core.hpp, devmath.hpp, dense_kernels.hpp).
#pragma once
#if defined(__HIPCC_RTC__)
#ifndef INFINITY
#define INFINITY __builtin_huge_val()
#endif
#ifndef NAN
#define NAN __builtin_nan("")
#endif
#ifndef DBL_EPSILON
#define DBL_EPSILON 2.2204460492503131e-16
#endif
#ifndef DBL_MAX
#define DBL_MAX 1.7976931348623157e+308
#endif
#ifndef offsetof
#define offsetof(t, m) __builtin_offsetof(t, m)
#endif
typedef unsigned long uintptr_t;
#endif
