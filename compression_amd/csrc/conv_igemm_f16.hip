// fp16-operand / fp32-accumulate variant of the implicit-GEMM transform kernel
// (v_mfma_f32_32x32x16_f16), used for the synthesis transform g_s, whose output only has to be within
// 1e-2 of the fp32 reference (the analysis transform must stay on the exact f32 path: its output is
// rounded to symbols).
//
// Same formulation, orientation and work decomposition as conv_igemm.hip:
//   D[cout][pixel] += W_tap[cout][ci] * in[ci][src(pixel, tap)],   wave = 32 pixels x all Cout,
// with a K chunk of (1 tap x 64 input channels) = 4 MFMA k-steps of 16.  Per k-step a lane holds
// 8 consecutive input channels of ITS pixel (one 16-byte global load) and reads 8 consecutive-k
// weights of its output channel from LDS (one ds_read_b128).  Activations are fp16 NHWC in HBM, so the
// layer-to-layer traffic is half of the f32 path's.
//
// Fused GDN / IGDN: norm[c'] = beta[c'] + sum_c gamma[c'][c] * y[c]^2 is a second fp16 MFMA whose B
// operand is built from the lane's own accumulators: registers 8s..8s+7 of tile f, squared, are the
// 8 k-values of k-step s for this lane half, in the order  c = 32f + 16s + 8(j>>2) + 4h + (j&3);
// gamma is stored pre-permuted in that order.  y^2 is scaled by 2^-8 and saturated before the
// conversion to fp16 (|y| up to 4096 is represented; smaller terms that underflow are negligible
// against beta), the scale is undone on the fp32 sum.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float SQ_SCALE = 0.00390625f;      // 2^-8
constexpr float SQ_UNSCALE = 256.0f;
constexpr float HALF_MAX = 65504.0f;

__device__ __forceinline__ half8 as_half8(const float4 &v) { return __builtin_bit_cast(half8, v); }

__device__ __forceinline__ _Float16 sat_half(float v) { return (_Float16)fminf(fmaxf(v, -HALF_MAX), HALF_MAX); }

template <int NF, int EPI, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv_igemm_f16_kernel(const Params p) {
    constexpr int COUT = NF * 32;
    constexpr int CHUNK_F4 = COUT * 8;                    // 16-byte slots per weight chunk (Cout x 64 halfs)
    constexpr int NT = 64 * NW;                           // threads per block (NW waves of 32 pixels)
    constexpr int STAGE = CHUNK_F4 / NT;
    extern __shared__ float4 lds[];                       // 2 x CHUNK_F4

    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const Phase ph = p.ph[blockIdx.y];

    const int64_t blk = xcd_block();
    if (blk * (32 * NW) >= p.P) return;                 // padding block of the rounded-up grid
    const int64_t pix = blk * (32 * NW) + wave * 32 + j;
    const bool pvalid = pix < p.P;
    int b = 0, by = 0, bx = 0;
    if (pvalid) {
        int64_t hw = (int64_t)p.Hb * p.Wb;
        b = (int)(pix / hw);
        int rem = (int)(pix - (int64_t)b * hw);
        by = rem / p.Wb;
        bx = rem - by * p.Wb;
    }
    const _Float16 *in_b = (const _Float16 *)p.in + (int64_t)b * p.Hi * p.Wi * p.Cs;

    auto load_b = [&](int chunk, float4 (&dst)[4]) {
        int tap = chunk / p.cin_blocks;
        int cb = chunk - tap * p.cin_blocks;
        int2 d = p.tapv[ph.tap_off + tap];                // uniform index: a scalar load from the kernel arguments
        int iy = by * p.sy + d.x, ix = bx * p.sx + d.y;
        bool ok = pvalid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const _Float16 *src = in_b + ((int64_t)iy * p.Wi + ix) * p.Cs + cb * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *(const float4 *)(src + 16 * ks);
            dst[ks] = v;
        }
    };
    const float4 *wsrc = (const float4 *)((const _Float16 *)p.wpk + ph.w_off);
    // GDN: gamma is packed as NF/2 more "weight chunks" (64 input channels of the mix each) that go through the same
    // LDS staging as the convolution weights
    const float4 *gsrc = (const float4 *)p.gamma;
    auto load_w = [&](const float4 *chunk_src, float4 (&dst)[STAGE]) {
        const float4 *s = chunk_src + tid;
#pragma unroll
        for (int r = 0; r < STAGE; ++r) dst[r] = s[r * NT];
    };
    auto store_w = [&](int buf, const float4 (&src)[STAGE]) {
        float4 *d = lds + buf * CHUNK_F4 + tid;
#pragma unroll
        for (int r = 0; r < STAGE; ++r) d[r * NT] = src[r];
    };

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    float4 bcur[4], bnext[4], wreg[STAGE];
    load_w(wsrc, wreg);
    load_b(0, bcur);
    store_w(0, wreg);
    __syncthreads();

    const int nch = ph.nchunks;
    for (int ch = 0; ch < nch; ++ch) {
        const int nxt = (ch + 1 < nch) ? ch + 1 : ch;     // unconditional prefetch keeps everything in registers
        load_w((ch + 1 < nch || EPI != EPI_GDN) ? wsrc + (int64_t)nxt * CHUNK_F4 : gsrc, wreg);   // after the last chunk: gamma
        load_b(nxt, bnext);
        const float4 *wl = lds + (ch & 1) * CHUNK_F4 + h * COUT + j;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 bv = as_half8(bcur[ks]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const half8 av = as_half8(wl[(ks * 2) * COUT + f * 32]);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[f], 0, 0, 0);
            }
        }
        store_w((ch + 1) & 1, wreg);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bcur[ks] = bnext[ks];
    }

    // ---- epilogue: acc[f][r] = channel 32f + (r&3) + 8(r>>2) + 4h of pixel `pix` -----------------
    const int oy = by * p.my + ph.qy, ox = bx * p.mx + ph.qx;

    if constexpr (EPI == EPI_STACKED) {
        if (pvalid) {
            float *out_f = (float *)p.out;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                int q = row / p.cout, co = row - q * p.cout;
                if (q < 4) {
                    float v = acc[0][r] + p.bias[co];
                    if (p.flags & MPC_OUT_CLAMP01) v = fminf(fmaxf(v, 0.0f), 1.0f);
                    int yy = by * 2 + (q >> 1), xx = bx * 2 + (q & 1);
                    out_f[(((int64_t)b * p.cout + co) * p.Ho + yy) * p.Wo + xx] = v;
                }
            }
        }
        return;
    } else {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bb = *(const float4 *)(p.bias + 32 * f + 8 * g + 4 * h);
                acc[f][4 * g + 0] += bb.x;
                acc[f][4 * g + 1] += bb.y;
                acc[f][4 * g + 2] += bb.z;
                acc[f][4 * g + 3] += bb.w;
            }
        if constexpr (EPI == EPI_GDN) {
            const bool inverse = p.act == MPC_ACT_IGDN;
            // norm[c'] = sum_c gamma[c'][c] * y[c]^2 as NF/2 more iterations of the K loop.  The B operand is built from
            // the lane's own accumulators: registers 8s..8s+7 of tile f, squared, are the 8 k-values of one k-step, in the
            // order c = 32f + 16s + 8(jj>>2) + 4h + (jj&3); k-step ks of gamma chunk kc is (f, s) = (2kc + (ks>>1), ks&1).
            // y^2 is scaled by 2^-8 and saturated before the conversion to fp16; the scale is undone on the fp32 sum.
            f32x16 nrm[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[f][r] = 0.0f;
#pragma unroll
            for (int kc = 0; kc < NF / 2; ++kc) {
                load_w(gsrc + (int64_t)(kc + 1 < NF / 2 ? kc + 1 : kc) * CHUNK_F4, wreg);
                const float4 *wl = lds + ((nch + kc) & 1) * CHUNK_F4 + h * COUT + j;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    half8 bv;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        float y = acc[2 * kc + (ks >> 1)][8 * (ks & 1) + jj];
                        bv[jj] = (_Float16)fminf(y * y * SQ_SCALE, HALF_MAX);
                    }
#pragma unroll
                    for (int fo = 0; fo < NF; ++fo)
                        nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(wl[(ks * 2) * COUT + fo * 32]), bv, nrm[fo], 0, 0, 0);
                }
                store_w((nch + kc + 1) & 1, wreg);
                __syncthreads();
            }
            // y * (beta + norm)^(-1/2) (or ^(+1/2)) through the wave-private transposing store (2 bytes per element:
            // all 192 channels in one round); v_rsq_f32 is far more accurate than the fp16 output
            TransStore ts;
            ts.init((char *)lds, wave, lane, j, h,
                    pvalid ? (long long)(((int64_t)b * p.Ho + oy) * p.Wo + ox) : -1ll);
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 be = *(const float4 *)(p.beta + 32 * f + 8 * g + 4 * h);
                    const float bev[4] = {be.x, be.y, be.z, be.w};
                    half4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float a = nrm[f][4 * g + t] * SQ_UNSCALE + bev[t];
                        float rs = __builtin_amdgcn_rsqf(a);
                        float y = acc[f][4 * g + t];
                        o[t] = sat_half(inverse ? y * (a * rs) : y * rs);
                    }
                    *(half4 *)ts.at(j, f * 64 + (8 * g + 4 * h) * 2) = o;
                }
            ts.template flush<NF * 4>((char *)p.out, (long long)COUT * 2, 0);
        } else {
            if (!pvalid) return;
            _Float16 *orow = (_Float16 *)p.out + (((int64_t)b * p.Ho + oy) * p.Wo + ox) * COUT;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int c0 = 32 * f + 8 * g + 4 * h;
                    half4 o;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float v = acc[f][4 * g + t];
                        if (p.act == MPC_ACT_RELU) v = fmaxf(v, 0.0f);
                        o[t] = sat_half(v);
                    }
                    *(half4 *)(orow + c0) = o;
                }
        }
    }
}

template <int NF, int EPI, int NW>
int launch_nw(mpc_ctx *ctx, const Params &p, int nphase) {
    size_t lds = (size_t)2 * NF * 32 * 8 * sizeof(float4);
    if (EPI == EPI_GDN && lds < (size_t)NW * TS_WAVE_BYTES) lds = (size_t)NW * TS_WAVE_BYTES;
    auto kern = conv_igemm_f16_kernel<NF, EPI, NW>;
    static bool attr_set[MPC_MAX_DEVICES] = {};        // per device (the attribute lives in the device's code object)
    bool &attr_done = attr_set[ctx->device];
    if (lds > 48 * 1024 && !attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    int64_t bx = xcd_grid(mpc_cdiv(p.P, 32 * NW));
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many pixels");
    kern<<<dim3((unsigned)bx, (unsigned)nphase, 1), dim3(64 * NW), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "conv_igemm_f16");
    return MPC_OK;
}

template <int NF, int EPI>
int launch(mpc_ctx *ctx, const Params &p, int nphase) {
    if constexpr (NF >= 2) {                  // the one-tile weight chunk is too small to spread over 512 threads
        if (wide_blocks(p, nphase)) return launch_nw<NF, EPI, 8>(ctx, p, nphase);
    }
    return launch_nw<NF, EPI, 4>(ctx, p, nphase);
}

}  // namespace

int mpc_conv_launch_f16(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    const bool gdn = c->desc.act == MPC_ACT_GDN || c->desc.act == MPC_ACT_IGDN;
    if (mpc_deconv_patch_f16_supported(p, c)) {
        static const bool off = [] { const char *e = getenv("MPC_F16_PATCH"); return e && atoi(e) == 0; }();   // development switch
        if (!off) return mpc_deconv_launch_patch_f16(ctx, p, c);
    }
    if (c->stacked) return c->wpk_rgb16 ? mpc_conv_launch_rgb16(ctx, p, c) : launch<1, EPI_STACKED>(ctx, p, c->nphase);
    if (gdn) {
        if (c->nf == 4) return launch<4, EPI_GDN>(ctx, p, c->nphase);
        if (c->nf == 6) return launch<6, EPI_GDN>(ctx, p, c->nphase);
    } else {
        if (c->nf == 4) return launch<4, EPI_PLAIN>(ctx, p, c->nphase);
        if (c->nf == 6) return launch<6, EPI_PLAIN>(ctx, p, c->nphase);
    }
    MPC_FAIL(ctx, MPC_EINVAL, "conv_forward (f16): unsupported layer shape (cout/32 = %d)", c->nf);
}
