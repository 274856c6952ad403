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
    hipEvent_t scratch_done = nullptr;     // recorded after the last kernel that used the scratch ...
    hipStream_t scratch_stream = nullptr;  // ... on this stream
};

// Device scratch of at least `bytes` owned by the context (contents undefined; valid until the next call
// that asks for more).  Returns nullptr and sets the context's error text on failure.  If the previous user ran on
// another stream, the context's current stream is made to wait for it first.  The caller enqueues its kernels on
// ctx->stream and then calls mpc_ctx_scratch_release.
void *mpc_ctx_scratch(mpc_ctx *ctx, size_t bytes);
void mpc_ctx_scratch_release(mpc_ctx *ctx);
// Looks up this device's instance of the static scratch of mpc_rans_decode_noalloc (rans.hip); the context's device
// must be current.  Called by mpc_create.
int mpc_rans_static_init(mpc_ctx *ctx);

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

// Every entry point that takes a context runs with the context's device current and puts the caller's device
// back on return, so a process may hold contexts on several GPUs (and torch's current device is left alone).
struct mpc_device_guard {
    int prev = -1;
    bool switched = false;
    explicit mpc_device_guard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~mpc_device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
    mpc_device_guard(const mpc_device_guard &) = delete;
    mpc_device_guard &operator=(const mpc_device_guard &) = delete;
};
#define MPC_ENTER(ctx)                  \
    if (!(ctx)) return MPC_EINVAL;      \
    mpc_device_guard mpc_guard__((ctx)->device)

constexpr int MPC_MAX_DEVICES = 64;     // per-device flags (kernel attributes set, static scratch addresses)

static inline int64_t mpc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device functions shared by the table kernels (entropy_tables.hip) and the coder (rans.hip) -------------------
// float -> int16 the way a vectorised x86 cast does it: through int32, keeping the low 16 bits.
__device__ __forceinline__ int16_t mpc_wrap_i16(float v) { return (int16_t)(uint16_t)((uint32_t)(int32_t)v & 0xffffu); }

// torch.distributions.Laplace(0, scale).cdf(x) = 0.5 - 0.5 * sign(x) * expm1(-|x| / scale)
__device__ __forceinline__ float mpc_laplace_cdf(float x, float scale) {
    float sg = (x > 0.0f) ? 0.5f : ((x < 0.0f) ? -0.5f : 0.0f);
    float em = (float)expm1((double)__fdiv_rn(-fabsf(x), scale));
    return __fsub_rn(0.5f, __fmul_rn(sg, em));
}

// sigma.clamp(1e-10, 1e10) of abstract_model.py:405 (a NaN becomes the lower bound)
__device__ __forceinline__ float mpc_laplace_scale(float sigma) { return fminf(fmaxf(sigma, 1e-10f), 1e10f); }

// Entry s of the int16 CDF row of Laplace(0, scale) at the targets first + s, "add arange, then round" order
// (abstract_model.py:59-73 applied to :409-416), as the unsigned 16-bit value the coder compares with: the ONE
// definition of a row entry, used by the kernel that materialises rows and by both sides of the coder that does not,
// so that encoder and decoder see the same integers.  factor = 2^precision - (L - 1).
__device__ __forceinline__ uint32_t mpc_laplace_row_entry(float scale, float first, int s, float factor) {
    const float c = mpc_laplace_cdf(first + (float)s, scale);
    return (uint32_t)(uint16_t)mpc_wrap_i16(rintf(__fadd_rn(__fmul_rn(c, factor), (float)s)));
}
