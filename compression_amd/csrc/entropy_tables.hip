// Learned-CDF kernels: Bitparm chain evaluation (float CDF tables), int16 renormalisation,
// code-length LUT, Laplace conditional.  All of them are one-thread-per-output elementwise kernels
// that run once per model (tables) or on small tensors (hyperprior), so none is on the roofline.
//
// Numerics.  The reference evaluates these in fp32 with one rounding per PyTorch op
// (src/compression/models/bitEstimator.py:136-141).  The kernels keep exactly those rounding points
// (no FMA contraction: explicit __fmul_rn/__fadd_rn) and evaluate each transcendental in fp64 before
// rounding to fp32, i.e. they return the correctly rounded fp32 value of exp/log1p/tanh/log2/expm1.
// PyTorch's CPU build uses Sleef's <=1-ulp vector routines instead, so individual table entries can
// differ by one fp32 ulp from a CPU-built table; everything downstream of the float table
// (mpc_renorm_cast_cdf, the integer part of the path) is exact integer / IEEE arithmetic.
#include "mpc_internal.h"

namespace {

__device__ __forceinline__ float f_softplus(float h) {
    // F.softplus(beta=1, threshold=20): h > 20 ? h : log1p(exp(h)), exp and log1p each rounded to fp32
    if (h > 20.0f) return h;
    float e = (float)exp((double)h);
    return (float)log1p((double)e);
}
__device__ __forceinline__ float f_tanh(float x) { return (float)tanh((double)x); }
__device__ __forceinline__ float f_sigmoid(float x) {
    // 1 / (1 + exp(-x)) with fp32 roundings after exp, add and divide
    float e = (float)exp(-(double)x);
    return __fdiv_rn(1.0f, __fadd_rn(1.0f, e));
}

// Per-row derived parameters: {softplus(h1), b1, tanh(a1), ..., softplus(h4), b4, 0}
constexpr int PREP = 12;

__global__ void bitparm_prepare_kernel(const float *h1, const float *b1, const float *a1, const float *h2,
                                       const float *b2, const float *a2, const float *h3, const float *b3,
                                       const float *a3, const float *h4, const float *b4, int64_t n_rows,
                                       float *__restrict__ prep) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float *p = prep + r * PREP;
    p[0] = f_softplus(h1[r]); p[1] = b1[r]; p[2] = f_tanh(a1[r]);
    p[3] = f_softplus(h2[r]); p[4] = b2[r]; p[5] = f_tanh(a2[r]);
    p[6] = f_softplus(h3[r]); p[7] = b3[r]; p[8] = f_tanh(a3[r]);
    p[9] = f_softplus(h4[r]); p[10] = b4[r]; p[11] = 0.0f;
}

__device__ __forceinline__ float bitparm_chain(float x, const float *__restrict__ p) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float t = __fadd_rn(__fmul_rn(x, p[3 * i]), p[3 * i + 1]);
        x = __fadd_rn(t, __fmul_rn(f_tanh(t), p[3 * i + 2]));
    }
    return f_sigmoid(__fadd_rn(__fmul_rn(x, p[9]), p[10]));
}

__global__ void bitparm_table_kernel(const float *__restrict__ prep, int64_t n_rows, float first, int L,
                                     float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * L) return;
    int64_t r = i / L;
    int s = (int)(i % L);
    out[i] = bitparm_chain(first + (float)s, prep + r * PREP);
}

__global__ void bitparm_eval_kernel(const float *__restrict__ prep, int64_t n_rows, const float *__restrict__ x,
                                    int64_t n_per_row, int row_minor, float offset, float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_per_row) return;
    int64_t r = row_minor ? i % n_rows : i / n_per_row;
    out[i] = bitparm_chain(__fadd_rn(x[i], offset), prep + r * PREP);
}

__device__ __forceinline__ int16_t wrap_i16(float v) { return mpc_wrap_i16(v); }

__global__ void renorm_cast_kernel(const float *__restrict__ cdf, int64_t n, int L, float factor, int order,
                                   int16_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = (int)(i % L);
    float scaled = __fmul_rn(cdf[i], factor);
    if (order == MPC_ORDER_ROUND_THEN_ADD) {
        out[i] = (int16_t)(wrap_i16(rintf(scaled)) + (int16_t)s);
    } else {
        out[i] = wrap_i16(rintf(__fadd_rn(scaled, (float)s)));
    }
}

__device__ __forceinline__ float code_length(float p, int form) {
    float q = __fadd_rn(p, 1e-10f);
    float b;
    if (form == MPC_BITS_LOG2) {
        b = -(float)log2((double)q);
    } else {
        float l = (float)log((double)q);
        b = __fdiv_rn(__fmul_rn(-1.0f, l), 0.6931471805599453f);
    }
    if (b != b) return b;                       // torch.clamp keeps NaN
    return fminf(fmaxf(b, 0.0f), 50.0f);
}

__global__ void bits_lut_kernel(const float *__restrict__ table_f, int K, int M, int L, int form,
                                float *__restrict__ lut) {
    int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (c*S + s)*K + k
    int S = L - 1;
    if (o >= (int64_t)M * S * K) return;
    int k = (int)(o % K);
    int64_t cs = o / K;
    int s = (int)(cs % S), c = (int)(cs / S);
    const float *row = table_f + ((int64_t)k * M + c) * L;
    lut[o] = code_length(__fsub_rn(row[s + 1], row[s]), form);
}

__device__ __forceinline__ float laplace_cdf(float x, float scale) { return mpc_laplace_cdf(x, scale); }

__global__ void laplace_bits_kernel(const float *__restrict__ y, const float *__restrict__ sigma, int64_t n,
                                    float *__restrict__ bits, double *__restrict__ total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double b = 0.0;
    if (i < n) {
        float s = fminf(fmaxf(sigma[i], 1e-10f), 1e10f);
        float p = __fsub_rn(laplace_cdf(__fadd_rn(y[i], 0.5f), s), laplace_cdf(__fsub_rn(y[i], 0.5f), s));
        float v = code_length(p, MPC_BITS_LN);
        if (bits) bits[i] = v;
        b = (double)v;
    }
    if (total) {
        for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(total, b);
    }
}

__global__ void prob_bits_kernel(const float *__restrict__ hi, const float *__restrict__ lo, int64_t n,
                                 float *__restrict__ bits, double *__restrict__ total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double b = 0.0;
    if (i < n) {
        float v = code_length(__fsub_rn(hi[i], lo[i]), MPC_BITS_LN);
        if (bits) bits[i] = v;
        b = (double)v;
    }
    if (total) {
        for (int off = 32; off > 0; off >>= 1) b += __shfl_down(b, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(total, b);
    }
}

// ---- network-form prior competition ---------------------------------------------------------------------------
// bits[k][loc] = sum_c clamp(-log2(P_kc(v + .5) - P_kc(v - .5) + 1e-10), 0, 50) with v = y[loc][c] (+ noise[loc][c]):
// the K priors' Bitparm chains evaluated at the latent itself, which need not be an integer (training-mode forward:
// uniform-noise quantisation, src/compression/models/manynets_compressor.py:514-530, 547-554).  One wavefront per
// location, lane k = prior k; the channel sum keeps the association of ATen's sum over dim 2 of the [K,B,M,h,w]
// tensor (16 sequential adds into a level-0 accumulator, drained into level 1, every 256 channels into level 2), and
// the minimum follows torch.min (NaN wins, ties go to the lower index).
struct Cand {
    float v;
    int k;
};
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    bool an = a.v != a.v, bn = b.v != b.v;
    bool a_wins;
    if (an || bn) a_wins = (an && bn) ? (a.k < b.k) : an;
    else a_wins = (a.v < b.v) || (a.v == b.v && a.k < b.k);
    return a_wins ? a : b;
}

__global__ __launch_bounds__(256) void prior_bits_network_kernel(const float *__restrict__ prep, int K, int M,
                                                                  const float *__restrict__ y, const float *__restrict__ noise,
                                                                  int64_t n_loc, float *__restrict__ y_entropy,
                                                                  float *__restrict__ total_bits, int32_t *__restrict__ idx,
                                                                  float *__restrict__ minbits) {
    const int lane = threadIdx.x & 63;
    const int64_t loc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (loc >= n_loc) return;
    const bool active = lane < K;
    const float *prow = prep + (int64_t)(active ? lane : 0) * M * PREP;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    for (int c = 0; c < M; ++c) {
        float v = y[loc * M + c];
        if (noise) v = __fadd_rn(v, noise[loc * M + c]);
        if (y_entropy && lane == 0) y_entropy[loc * M + c] = v;
        const float *p = prow + (int64_t)c * PREP;
        float pr = __fsub_rn(bitparm_chain(__fadd_rn(v, 0.5f), p), bitparm_chain(__fsub_rn(v, 0.5f), p));
        acc0 = __fadd_rn(acc0, code_length(pr, MPC_BITS_LOG2));
        if (((c + 1) & 15) == 0 && c + 1 <= (M & ~15)) {
            acc1 = __fadd_rn(acc1, acc0);
            acc0 = 0.0f;
            if (((c + 1) & 0xF0) == 0) {
                acc2 = __fadd_rn(acc2, acc1);
                acc1 = 0.0f;
            }
        }
    }
    float total = __fadd_rn(__fadd_rn(acc0, acc1), acc2);
    if (total_bits && active) total_bits[(int64_t)lane * n_loc + loc] = total;
    Cand best;
    best.v = active ? total : __builtin_inff();
    best.k = active ? lane : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(best.v, off, 64);
        o.k = __shfl_xor(best.k, off, 64);
        best = better(best, o);
    }
    if (lane == 0) {
        if (idx) idx[loc] = best.k;
        if (minbits) minbits[loc] = best.v;
    }
}

__global__ void laplace_rows_kernel(const float *__restrict__ sigma, int64_t n, float first, int L, float factor,
                                    int16_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * L) return;
    int64_t e = i / L;
    int s = (int)(i % L);
    out[i] = (int16_t)(uint16_t)mpc_laplace_row_entry(mpc_laplace_scale(sigma[e]), first, s, factor);
}

inline unsigned blocks_for(int64_t n) { return (unsigned)mpc_cdiv(n, 256); }

}  // namespace

// The prepared parameter image of one call: stream-ordered memory that is given back on every way out of the entry
// point, the error returns included.
struct PrepBuffer {
    float *ptr = nullptr;
    hipStream_t stream = nullptr;
    ~PrepBuffer() {
        if (ptr) (void)hipFreeAsync(ptr, stream);
    }
};

static int make_prep(mpc_ctx *ctx, const float *const params[11], int64_t n_rows, PrepBuffer &prep) {
    for (int i = 0; i < 11; ++i) MPC_REQUIRE(ctx, params[i] != nullptr, "bitparm: params[%d] is NULL", i);
    prep.stream = ctx->stream;
    MPC_HIP(ctx, hipMallocAsync((void **)&prep.ptr, sizeof(float) * PREP * n_rows, ctx->stream));
    bitparm_prepare_kernel<<<dim3(blocks_for(n_rows)), dim3(256), 0, ctx->stream>>>(
        params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8],
        params[9], params[10], n_rows, prep.ptr);
    MPC_CHECK_LAUNCH(ctx, "bitparm_prepare");
    return MPC_OK;
}

extern "C" {

int mpc_bitparm_cdf_table(mpc_ctx *ctx, const float *const params[11], int64_t n_rows, float first_target,
                          int L, float *table_f) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, params && table_f && n_rows > 0 && L > 1, "bitparm_cdf_table: bad arguments");
    PrepBuffer prep;
    int rc = make_prep(ctx, params, n_rows, prep);
    if (rc) return rc;
    bitparm_table_kernel<<<dim3(blocks_for(n_rows * L)), dim3(256), 0, ctx->stream>>>(prep.ptr, n_rows, first_target, L, table_f);
    MPC_CHECK_LAUNCH(ctx, "bitparm_table");
    return MPC_OK;
}

int mpc_bitparm_cdf_eval(mpc_ctx *ctx, const float *const params[11], int64_t n_rows, const float *x,
                         int64_t n_per_row, int row_minor, float offset, float *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, params && x && out && n_rows > 0 && n_per_row > 0, "bitparm_cdf_eval: bad arguments");
    PrepBuffer prep;
    int rc = make_prep(ctx, params, n_rows, prep);
    if (rc) return rc;
    bitparm_eval_kernel<<<dim3(blocks_for(n_rows * n_per_row)), dim3(256), 0, ctx->stream>>>(prep.ptr, n_rows, x, n_per_row, row_minor, offset, out);
    MPC_CHECK_LAUNCH(ctx, "bitparm_eval");
    return MPC_OK;
}

int mpc_prior_bits_network(mpc_ctx *ctx, const float *const params[11], int K, int M, const float *y, const float *noise,
                           int64_t n_loc, float *y_entropy, float *total_bits, int32_t *idx, float *minbits) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, params && y && n_loc > 0 && M > 0 && K > 0 && K <= 64 && (total_bits || idx || minbits),
                "prior_bits_network: bad arguments (K must be 64 or less)");
    PrepBuffer prep;
    int rc = make_prep(ctx, params, (int64_t)K * M, prep);
    if (rc) return rc;
    prior_bits_network_kernel<<<dim3((unsigned)mpc_cdiv(n_loc, 4)), dim3(256), 0, ctx->stream>>>(prep.ptr, K, M, y, noise, n_loc, y_entropy,
                                                                                                 total_bits, idx, minbits);
    MPC_CHECK_LAUNCH(ctx, "prior_bits_network");
    return MPC_OK;
}

int mpc_renorm_cast_cdf(mpc_ctx *ctx, const float *cdf, int64_t n_rows, int L, int precision, int order,
                        int16_t *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, cdf && out && n_rows > 0 && L > 1 && precision >= 8 && precision <= 16 &&
                         (order == MPC_ORDER_ROUND_THEN_ADD || order == MPC_ORDER_ADD_THEN_ROUND),
                "renorm_cast_cdf: bad arguments");
    float factor = (float)(1 << precision) - (float)(L - 1);
    renorm_cast_kernel<<<dim3(blocks_for(n_rows * L)), dim3(256), 0, ctx->stream>>>(cdf, n_rows * L, L, factor, order, out);
    MPC_CHECK_LAUNCH(ctx, "renorm_cast");
    return MPC_OK;
}

int mpc_bits_lut(mpc_ctx *ctx, const float *table_f, int K, int M, int L, int form, float *lut) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, table_f && lut && K > 0 && M > 0 && L > 1 && (form == MPC_BITS_LOG2 || form == MPC_BITS_LN),
                "bits_lut: bad arguments");
    bits_lut_kernel<<<dim3(blocks_for((int64_t)M * (L - 1) * K)), dim3(256), 0, ctx->stream>>>(table_f, K, M, L, form, lut);
    MPC_CHECK_LAUNCH(ctx, "bits_lut");
    return MPC_OK;
}

int mpc_laplace_bits(mpc_ctx *ctx, const float *yhat, const float *sigma, int64_t n, float *bits, double *total_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, yhat && sigma && n > 0 && (bits || total_dev), "laplace_bits: bad arguments");
    laplace_bits_kernel<<<dim3(blocks_for(n)), dim3(256), 0, ctx->stream>>>(yhat, sigma, n, bits, total_dev);
    MPC_CHECK_LAUNCH(ctx, "laplace_bits");
    return MPC_OK;
}

int mpc_prob_bits(mpc_ctx *ctx, const float *cdf_hi, const float *cdf_lo, int64_t n, float *bits, double *total_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, cdf_hi && cdf_lo && n > 0 && (bits || total_dev), "prob_bits: bad arguments");
    prob_bits_kernel<<<dim3(blocks_for(n)), dim3(256), 0, ctx->stream>>>(cdf_hi, cdf_lo, n, bits, total_dev);
    MPC_CHECK_LAUNCH(ctx, "prob_bits");
    return MPC_OK;
}

int mpc_laplace_cdf_rows(mpc_ctx *ctx, const float *sigma, int64_t n, float first_target, int L, int precision,
                         int16_t *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sigma && out && n > 0 && L > 1 && precision >= 8 && precision <= 16, "laplace_cdf_rows: bad arguments");
    float factor = (float)(1 << precision) - (float)(L - 1);
    laplace_rows_kernel<<<dim3(blocks_for(n * L)), dim3(256), 0, ctx->stream>>>(sigma, n, first_target, L, factor, out);
    MPC_CHECK_LAUNCH(ctx, "laplace_rows");
    return MPC_OK;
}

}  // extern "C"
