// Split-precision ("f16x3") variant of the implicit-GEMM transform kernel for the analysis transform:
// fp32-grade results on the fp16 matrix pipe.
//
// Every fp32 operand v (pre-scaled by a power of two so that it sits well inside fp16's range) is
// carried as hi = fp16(v), lo = fp16(v - hi): 22 significant bits.  A product is evaluated as
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (three v_mfma_f32_32x32x16_f16, fp32 accumulate)
// dropping only a_lo*b_lo (2^-22 relative).  Products of fp16 values are exact in the fp32
// accumulator, and the accumulation chain is K/16 steps long instead of K, so the result is as close
// to the exact dot product as the f32-MFMA kernel's (whose 4800-step fma chain dominates its own
// error).  What has to hold for the symbols: latents that differ from the reference only when they
// sit within ~1e-6 of a rounding tie.
//
// Formats:
//   activations "S16": [pixel][C/8][hi x8 | lo x8] fp16, values scaled by ACT_SCALE = 2^8
//                      (one 32-byte read per lane per k-step brings both halves);
//   weights: scaled by a per-layer power of two (max |w| -> [2^13, 2^14)), packed per K chunk
//            (1 tap x 32 input channels) as [kstep 2][h 2][hi|lo][cout][8] fp16.
// GDN stays on the exact f32 MFMA (4 % of the MACs) on the unscaled fp32 accumulators.
// Same orientation / work decomposition / LDS weight staging as conv_igemm.hip.
#include <stdlib.h>

#include "conv_x3_epilogue.h"

namespace {

// RGB = first layer: the input is the fp32 NHWC image padded to 4 channels; the K axis is the im2col index
// k = tap*4 + c (8 taps per chunk, 25 taps -> 4 chunks; the weights of c = 3 and of the padding taps are
// zero), gathered with one 16-byte load per tap and split on the fly.
template <int NF, int EPI, bool RGB, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv_igemm_x3_kernel(const Params p) {
    constexpr int COUT = NF * 32;
    constexpr int CHUNK_F4 = COUT * 8;                    // 16-byte slots per weight chunk (2*2*2*COUT)
    constexpr int NT = 64 * NW;                           // threads per block (NW waves of 32 pixels)
    constexpr int STAGE = CHUNK_F4 / NT;
    extern __shared__ float4 lds[];                       // 2 x CHUNK_F4

    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const Phase ph = p.ph[blockIdx.y];
    const int2 *taps = p.taps + ph.tap_off;
    const int cg = blockIdx.z * COUT;

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
    // S16 input: 2*Cs halfs per pixel
    const _Float16 *in_b = (const _Float16 *)p.in + (int64_t)b * p.Hi * p.Wi * p.Cs * 2;

    auto load_b = [&](int chunk, float4 (&dst)[4]) {      // [ks][hi|lo]
        if constexpr (RGB) {
            // k = tap*4 + c: the lane's 8 k-values of a k-step are two whole padded pixels (taps)
            const float4 *img = (const float4 *)p.in + (int64_t)b * p.Hi * p.Wi;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    int tap = chunk * 8 + ks * 4 + h * 2 + t;
                    int2 d = taps[tap < ph.ntaps ? tap : 0];
                    int iy = by * p.sy + d.x, ix = bx * p.sx + d.y;
                    bool ok = pvalid && tap < ph.ntaps && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
                    float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) px = img[(int64_t)iy * p.Wi + ix];
                    v[4 * t + 0] = px.x; v[4 * t + 1] = px.y; v[4 * t + 2] = px.z; v[4 * t + 3] = px.w;
                }
                half8 vh, vl;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    float sv = v[jj] * X3_ACT_SCALE;
                    _Float16 hh = (_Float16)sv;
                    vh[jj] = hh;
                    vl[jj] = (_Float16)(sv - (float)hh);
                }
                dst[2 * ks] = __builtin_bit_cast(float4, vh);
                dst[2 * ks + 1] = __builtin_bit_cast(float4, vl);
            }
        } else {
#ifdef MPC_X3_K_BLOCK_MAJOR
            int cb = chunk / ph.ntaps;                     // K walked as (channel block, tap): see wseq below
            int tap = chunk - cb * ph.ntaps;
#else
            int tap = chunk / p.cin_blocks;
            int cb = chunk - tap * p.cin_blocks;
#endif
            int2 d = taps[tap];
            int iy = by * p.sy + d.x, ix = bx * p.sx + d.y;
            bool ok = pvalid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            const _Float16 *src = in_b + ((int64_t)iy * p.Wi + ix) * p.Cs * 2 + (cb * 4 + h) * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float4 vh = make_float4(0.f, 0.f, 0.f, 0.f), vl = vh;
                if (ok) {
                    vh = *(const float4 *)(src + ks * 32);
                    vl = *(const float4 *)(src + ks * 32 + 8);
                }
                dst[2 * ks] = vh;
                dst[2 * ks + 1] = vl;
            }
        }
    };
    const float4 *wsrc = (const float4 *)((const _Float16 *)p.wpk + (int64_t)blockIdx.z * p.w_group + ph.w_off);
    // GDN channel mix on the fp16 pipe: gamma is packed as NF more "weight chunks" (one per 32 input channels
    // of the mix) that go through the same LDS staging as the convolution weights
    const bool gdn_mix16 = (EPI == EPI_GDN) && p.y2_scale > 0.0f;
    const float4 *gsrc = (const float4 *)p.gamma_x3;
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

    // Position in the packed weights of the c-th chunk of the K loop.  The weights are packed tap-major
    // (chunk = tap * channel blocks + block), which is also the default walk.  Built with
    // -DMPC_X3_K_BLOCK_MAJOR the loop walks (channel block, tap) instead: one pass over the taps of a 32-channel
    // block re-reads a six times smaller set of input bytes (an L2-sized one at the benchmark's shapes); only the
    // summation order changes.  Not the default until it has been measured.
    auto wseq = [&](int c) -> int64_t {
#ifdef MPC_X3_K_BLOCK_MAJOR
        if constexpr (!RGB) {
            const int cb = c / ph.ntaps;
            return (int64_t)(c - cb * ph.ntaps) * p.cin_blocks + cb;
        }
#endif
        return c;
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
        // unconditional prefetch (keeps everything in registers); after the last chunk: the first gamma chunk
        const int nxt = (ch + 1 < nch) ? ch + 1 : ch;
        load_w((ch + 1 < nch || !gdn_mix16) ? wsrc + wseq(nxt) * CHUNK_F4 : gsrc, wreg);
        load_b(nxt, bnext);
        // slot index ((ks*2 + h)*2 + part)*COUT + cout
        const float4 *wl = lds + (ch & 1) * CHUNK_F4 + (h * 2) * COUT + j;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 bh = as_half8(bcur[2 * ks]), bl = as_half8(bcur[2 * ks + 1]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const half8 ah = as_half8(wl[(ks * 4) * COUT + f * 32]);
                const half8 al = as_half8(wl[(ks * 4 + 1) * COUT + f * 32]);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[f], 0, 0, 0);
            }
        }
        store_w((ch + 1) & 1, wreg);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) bcur[q] = bnext[q];
    }

    // ---- epilogue: undo the operand scales, add bias ---------------------------------------------
    x3_unscale_bias<NF>(p, acc, cg, h);
    const int oy = by * p.my + ph.qy, ox = bx * p.mx + ph.qx;
    const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
    const bool out_split = (p.flags & MPC_OUT_SPLIT) != 0;

    if constexpr (EPI == EPI_GDN) {
        const bool inverse = p.act == MPC_ACT_IGDN;
        const long long pixel_bytes = (long long)COUT * 4;
        if (gdn_mix16) {
            // norm[c'] = sum_c gamma[c'][c] * y[c]^2 as NF more iterations of the K loop: chunk kc holds gamma[:, 32kc..32kc+31]
            // (hi/lo, scaled) in the weight-chunk layout; the B operand is built from the lane's own accumulators, whose
            // registers 8s..8s+7 of tile kc are the 8 k-values of k-step s:  c = 32kc + 16s + 8(jj>>2) + 4h + (jj&3).
            f32x16 nrm[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[f][r] = 0.0f;
#pragma unroll
            for (int kc = 0; kc < NF; ++kc) {
                load_w(gsrc + (int64_t)(kc + 1 < NF ? kc + 1 : kc) * CHUNK_F4, wreg);
                const float4 *wl = lds + ((nch + kc) & 1) * CHUNK_F4 + (h * 2) * COUT + j;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    half8 bh, bl;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        float y = acc[kc][8 * s + jj];
                        float q = fminf(y * y * p.y2_scale, 65504.0f);
                        _Float16 hh = (_Float16)q;
                        bh[jj] = hh;
                        bl[jj] = (_Float16)(q - (float)hh);
                    }
#pragma unroll
                    for (int fo = 0; fo < NF; ++fo) {
                        const half8 ah = as_half8(wl[(s * 4) * COUT + fo * 32]);
                        const half8 al = as_half8(wl[(s * 4 + 1) * COUT + fo * 32]);
                        nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, nrm[fo], 0, 0, 0);
                        nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, nrm[fo], 0, 0, 0);
                        nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, nrm[fo], 0, 0, 0);
                    }
                }
                store_w((nch + kc + 1) & 1, wreg);
                __syncthreads();
            }
            x3_gdn_store<NF>(p, acc, nrm, lds, wave, lane, j, h, pvalid ? (long long)opix : -1ll);
            return;
        }
        // exact f32 channel mix (layers created without an input bound), one output tile at a time
        // outputs go through the wave-private transposing store (4 bytes per element: 96 channels per round)
        TransStore ts;
        ts.init((char *)lds, wave, lane, j, h, pvalid ? (long long)opix : -1ll);
#pragma unroll 1
        for (int fo = 0; fo < NF; ++fo) {
            f32x16 nrm;
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[r] = 0.0f;
            const float *grow = (const float *)p.gamma + (int64_t)(32 * fo + j) * COUT + 4 * h;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float4 ga[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) ga[g] = *(const float4 *)(grow + 32 * f + 8 * g);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float av[4] = {ga[g].x, ga[g].y, ga[g].z, ga[g].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float y = acc[f][4 * g + t];
                        nrm = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], y * y, nrm, 0, 0, 0);
                    }
                }
            }
            f32x16 yv;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (f == fo) yv = acc[f];
            {
                const int rbase = (fo % 3) * 128;             // byte offset of this tile inside the round's row
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int c0 = 32 * fo + 8 * g + 4 * h;
                    float4 be = *(const float4 *)(p.beta + c0);
                    const float bev[4] = {be.x, be.y, be.z, be.w};
                    float o[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float n = sqrtf(nrm[4 * g + t] + bev[t]);
                        float y = yv[4 * g + t];
                        o[t] = inverse ? y * n : y / n;
                    }
                    if (out_split) {
                        // block g of the tile: [hi x8 | lo x8]; this lane half owns elements 4h..4h+3 of each
                        half4 hi4, lo4;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float sv = fminf(fmaxf(o[t] * X3_ACT_SCALE, -65504.0f), 65504.0f);
                            _Float16 hh = (_Float16)sv;
                            hi4[t] = hh;
                            lo4[t] = (_Float16)(sv - (float)hh);
                        }
                        *(half4 *)ts.at(j, rbase + g * 32 + 8 * h) = hi4;
                        *(half4 *)ts.at(j, rbase + g * 32 + 16 + 8 * h) = lo4;
                    } else {
                        *(float4 *)ts.at(j, rbase + (8 * g + 4 * h) * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
                if (fo % 3 == 2 || fo == NF - 1) ts.flush((char *)p.out, pixel_bytes, (fo / 3) * TS_ROW_BYTES, (fo % 3 + 1) * 8);
            }
        }
    } else {
        if (!pvalid) return;
        x3_plain_store<NF>(p, acc, cg, h, opix);
    }
}

template <int NF, int EPI, bool RGB, int NW>
int launch_nw(mpc_ctx *ctx, const Params &p, int nphase, int groups) {
    size_t lds = (size_t)2 * NF * 32 * 8 * sizeof(float4);
    if (EPI == EPI_GDN && lds < (size_t)NW * TS_WAVE_BYTES) lds = (size_t)NW * TS_WAVE_BYTES;
    auto kern = conv_igemm_x3_kernel<NF, EPI, RGB, NW>;
    static bool attr_set[MPC_MAX_DEVICES] = {};        // per device (the attribute lives in the device's code object)
    bool &attr_done = attr_set[ctx->device];
    if (lds > 48 * 1024 && !attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    int64_t bx = xcd_grid(mpc_cdiv(p.P, 32 * NW));
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many pixels");
    kern<<<dim3((unsigned)bx, (unsigned)nphase, (unsigned)groups), dim3(64 * NW), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "conv_igemm_x3");
    return MPC_OK;
}

template <int NF, int EPI, bool RGB>
int launch(mpc_ctx *ctx, const Params &p, int nphase, int groups) {
    if (wide_blocks(p, nphase * groups)) return launch_nw<NF, EPI, RGB, 8>(ctx, p, nphase, groups);
    return launch_nw<NF, EPI, RGB, 4>(ctx, p, nphase, groups);
}

}  // namespace

int mpc_conv_launch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    if (mpc_conv_patch_x3_supported(p, c)) {
        // MPC_X3_PATCH=0 (development switch) keeps this file's kernel
        static const bool off = [] { const char *e = getenv("MPC_X3_PATCH"); return e && atoi(e) == 0; }();
        if (!off) return mpc_conv_launch_patch_x3(ctx, p, c);
    }
    if (mpc_conv_rgb_patch_x3_supported(p, c)) {
        static const bool off = [] { const char *e = getenv("MPC_X3_RGB_PATCH"); return e && atoi(e) == 0; }();   // development switch
        if (!off) return mpc_conv_launch_rgb_patch_x3(ctx, p, c);
    }
    const bool gdn = c->desc.act == MPC_ACT_GDN || c->desc.act == MPC_ACT_IGDN;
    const int np = c->nphase, ng = c->groups;
    if (c->cin4) {
        if (gdn && c->nf == 4) return launch<4, EPI_GDN, true>(ctx, p, np, ng);
        if (gdn && c->nf == 6) return launch<6, EPI_GDN, true>(ctx, p, np, ng);
        if (!gdn && c->nf == 4) return launch<4, EPI_PLAIN, true>(ctx, p, np, ng);
        if (!gdn && c->nf == 6) return launch<6, EPI_PLAIN, true>(ctx, p, np, ng);
    } else if (gdn) {
        if (c->nf == 4) return launch<4, EPI_GDN, false>(ctx, p, np, ng);
        if (c->nf == 6) return launch<6, EPI_GDN, false>(ctx, p, np, ng);
    } else {
        if (c->nf == 4) return launch<4, EPI_PLAIN, false>(ctx, p, np, ng);
        if (c->nf == 6) return launch<6, EPI_PLAIN, false>(ctx, p, np, ng);
    }
    MPC_FAIL(ctx, MPC_EINVAL, "conv_forward (f16x3): unsupported layer shape (cout/32 = %d)", c->nf);
}
