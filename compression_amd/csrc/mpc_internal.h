// Internal definitions shared by the libmpc translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mpc.h"

struct mpc_ctx {
    int device;
    hipStream_t stream;
    char err[512];
    void *scratch = nullptr;      // library-owned device scratch, grown on demand (mpc_ctx_scratch)
    size_t scratch_bytes = 0;
};

// Device scratch of at least `bytes` owned by the context (contents undefined; valid until the next call
// that asks for more).  Returns nullptr and sets the context's error text on failure.
void *mpc_ctx_scratch(mpc_ctx *ctx, size_t bytes);

#define MPC_FAIL(ctx, code, ...)                                   \
    do {                                                           \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                             \
    } while (0)

#define MPC_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            MPC_FAIL(ctx, MPC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                     __FILE__, __LINE__);                                                    \
    } while (0)

#define MPC_CHECK_LAUNCH(ctx, name)                                                         \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess)                                                              \
            MPC_FAIL(ctx, MPC_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define MPC_REQUIRE(ctx, cond, ...)                      \
    do {                                                 \
        if (!(cond)) MPC_FAIL(ctx, MPC_EINVAL, __VA_ARGS__); \
    } while (0)

static inline int64_t mpc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
