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

// expm1(q) for q <= 0 in double, to about 2^-40 relative: q = k ln 2 + r with |r| <= ln 2 / 2, the Taylor polynomial of
// expm1(r) to r^11 evaluated pairwise (a dependent chain of 12 operations where the general routine has several
// times that: the coder's decoder waits for exactly this chain once per symbol), 2^k (1 + p) - 1 as one fma.  Rounded
// to float it equals the library's expm1 rounded to float for every float argument q <= 0
// (tools/micro/expm1_check.hip walks all 2^31 of them).  NaN stays NaN; below -40 the result is -1 in double (the
// argument is clamped there, without a branch: exp(-40) is far below half a unit in the last place of 1).
__device__ __forceinline__ double mpc_expm1_nonpos(double q) {
    q = q < -40.0 ? -40.0 : q;
    const double k = rint(q * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, q);
    r = fma(k, -1.90821492927058770002e-10, r);
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = fma(r, 1.0 / 6.0, 0.5), a1 = fma(r, 1.0 / 120.0, 1.0 / 24.0), a2 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
    const double a3 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0), a4 = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2);
    const double poly = fma(a4, r8, fma(b1, r4, b0));
    const double p = fma(r2, poly, r);
    const double s = ldexp(1.0, (int)k);
    return fma(s, p, s - 1.0);
}

// torch.distributions.Laplace(0, scale).cdf(x) = 0.5 - 0.5 * sign(x) * expm1(-|x| / scale)
__device__ __forceinline__ float mpc_laplace_cdf(float x, float scale) {
    float sg = (x > 0.0f) ? 0.5f : ((x < 0.0f) ? -0.5f : 0.0f);
    float em = (float)mpc_expm1_nonpos((double)__fdiv_rn(-fabsf(x), scale));
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
