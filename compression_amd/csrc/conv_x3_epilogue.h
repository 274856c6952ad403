// Epilogue pieces shared by the split-precision (f16x3) convolution kernels: undoing the operand scales, the GDN
// normalisation with full-line stores, the plain store (fp32 / split format / rounded / int16 symbols).
#pragma once
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 as_half8(const float4 &v) { return __builtin_bit_cast(half8, v); }

__device__ __forceinline__ void split_store4(_Float16 *blk, int off, const float (&v)[4]) {
    // v already scaled; hi at blk[off..off+3], lo at blk[8+off..]
    half4 hi, lo;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float s = fminf(fmaxf(v[t], -65504.0f), 65504.0f);
        _Float16 hh = (_Float16)s;
        hi[t] = hh;
        lo[t] = (_Float16)(s - (float)hh);
    }
    *(half4 *)(blk + off) = hi;
    *(half4 *)(blk + 8 + off) = lo;
}

// acc <- acc * unscale + bias (cg = first output channel of this workgroup's group)
template <int NF>
__device__ __forceinline__ void x3_unscale_bias(const Params &p, f32x16 (&acc)[NF], int cg, int h) {
    const float unscale = p.x3_unscale;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 bb = *(const float4 *)(p.bias + cg + 32 * f + 8 * g + 4 * h);
            acc[f][4 * g + 0] = acc[f][4 * g + 0] * unscale + bb.x;
            acc[f][4 * g + 1] = acc[f][4 * g + 1] * unscale + bb.y;
            acc[f][4 * g + 2] = acc[f][4 * g + 2] * unscale + bb.z;
            acc[f][4 * g + 3] = acc[f][4 * g + 3] * unscale + bb.w;
        }
}

// y * (beta + norm)^(-1/2) (or ^(+1/2) for the inverse), written through the wave-private transposing store in
// rounds of 96 channels (384 bytes per pixel).  nrm holds the scaled channel mix; v_rsq_f32 is good to 1 ulp.
// lds: the workgroup's LDS base (free by now); opix < 0: this lane's pixel is not stored.
template <int NF>
__device__ __forceinline__ void x3_gdn_store(const Params &p, const f32x16 (&acc)[NF], const f32x16 (&nrm)[NF], float4 *lds,
                                             int wave, int lane, int j, int h, long long opix) {
    constexpr int COUT = NF * 32;
    const bool inverse = p.act == MPC_ACT_IGDN;
    const bool out_split = (p.flags & MPC_OUT_SPLIT) != 0;
    const long long pixel_bytes = (long long)COUT * 4;
    TransStore ts;
    ts.init((char *)lds, wave, lane, j, h, opix);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int rbase = (f % 3) * 128;              // byte offset of this tile inside the round's row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 be = *(const float4 *)(p.beta + 32 * f + 8 * g + 4 * h);
            const float bev[4] = {be.x, be.y, be.z, be.w};
            float o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float a = nrm[f][4 * g + t] * p.gdn_unscale + bev[t];
                float rs = __builtin_amdgcn_rsqf(a);
                float y = acc[f][4 * g + t];
                o[t] = inverse ? y * (a * rs) : y * rs;
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
        if (f % 3 == 2) ts.template flush<24>((char *)p.out, pixel_bytes, (f / 3) * TS_ROW_BYTES);
        else if (f == NF - 1) ts.template flush<(NF % 3) * 8>((char *)p.out, pixel_bytes, (f / 3) * TS_ROW_BYTES);
    }
}

// No GDN: activation, then fp32 / split-format / rounded / int16-symbol outputs straight from the accumulators.
template <int NF>
__device__ __forceinline__ void x3_plain_store(const Params &p, const f32x16 (&acc)[NF], int cg, int h, int64_t opix) {
    const bool out_split = (p.flags & MPC_OUT_SPLIT) != 0;
    float *out_f = (float *)p.out;
    int sym_clamped = 0;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int c0 = cg + 32 * f + 8 * g + 4 * h;
            float o[4], orr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = acc[f][4 * g + t];
                if (p.act == MPC_ACT_RELU) v = fmaxf(v, 0.0f);
                o[t] = v;
                orr[t] = rintf(v);
            }
            if (out_split) {
                float os[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) os[t] = o[t] * X3_ACT_SCALE;
                split_store4((_Float16 *)p.out + (opix * p.cout + (c0 & ~7)) * 2, c0 & 7, os);
            } else {
                if (out_f) *(float4 *)(out_f + opix * p.cout + c0) = make_float4(o[0], o[1], o[2], o[3]);
                if (p.out_round) *(float4 *)(p.out_round + opix * p.cout + c0) = make_float4(orr[0], orr[1], orr[2], orr[3]);
                if (p.out_sym) store_symbols4(p, opix * p.cout + c0, orr, sym_clamped);
            }
        }
    if (sym_clamped && p.n_clamped) atomicAdd(p.n_clamped, sym_clamped);
}
