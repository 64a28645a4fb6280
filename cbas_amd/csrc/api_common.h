// Error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/cbas_mi355x.h"
#ifndef CBAS_BUILD_DEBUG
#define CBAS_BUILD_DEBUG 0
#endif
#if CBAS_BUILD_DEBUG
#include "../../include/cbas_mi355x_debug.h"      // bring-up / test entry points: debug build only
#endif

extern thread_local char g_cbas_err[512];

static inline int cbas_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cbas_err, sizeof(g_cbas_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return cbas_fail(CBAS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define LAUNCH_TRY(expr)                                                                       \
    do {                                                                                       \
        int _r = (expr);                                                                       \
        if (_r != 0)                                                                           \
            return cbas_fail(_r == -1 ? CBAS_EINVAL : CBAS_EHIP, "%s failed (%d; hip: %s) (%s:%d)", \
                             #expr, _r, hipGetErrorString(hipGetLastError()), __FILE__, __LINE__); \
    } while (0)
