// Transform layers (g_a / g_s / h_a / h_s / Testolina) as implicit-GEMM kernels on the CDNA4 matrix
// cores, fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32).
//
// Why fp32 MFMA: the quantised latent must be bit-exact against the fp32 reference up to rounding
// ties (a latent within ~1e-6 of x.5).  bf16/fp16 operands flip 5e-3 / 6e-4 of the symbols; the f32
// MFMA is an exact fp32 fma chain, so only summation order differs from the reference.
//
// Formulation (one kernel for Conv2d and ConvTranspose2d):
//   out[co][p'] = act( bias[co] + sum_{tap} sum_{ci} W_tap[co][ci] * in[ci][src(p, tap)] )
// over a "base grid" of pixels p.  For Conv2d the base grid is the output grid and
// src = p*stride + (ky - pad); for ConvTranspose2d with stride 2 the base grid is the INPUT grid and
// each of the 4 output parities (phase q) is its own stride-1 problem with 9/6/6/4 of the 25 taps,
// writing p' = 2p + q (no zero-insertion, no wasted MACs).
//
// Orientation: D[co][pixel] = A[co][k] * B[k][pixel] with A = weights, B = activations.  In the MFMA
// result a lane owns ONE pixel (column) and 16 output channels (rows) of each 32x32 tile, so
//   * the GDN / IGDN channel mix  norm[c'] = beta[c'] + sum_c gamma[c'][c] * y[c]^2  is a second MFMA
//     whose B operand is the square of the lane's own accumulator registers -- no LDS, no shuffles;
//   * a lane's 4 consecutive registers are 4 consecutive channels: 16-byte NHWC stores.
//
// Work decomposition: workgroup = 4 waves = 128 base pixels x all Cout; wave = 32 pixels x all Cout
// (Cout/32 accumulator tiles).  K loop in chunks of (1 tap x 32 input channels):
//   * weights of the chunk (Cout x 32 floats, pre-packed in exactly the LDS image order) are staged
//     global -> registers -> LDS, double buffered, one barrier per chunk, shared by the 4 waves;
//   * activations go global -> registers directly: a lane needs 16 B of ITS pixel per 8-deep k-step,
//     nothing is shared between lanes, and an f32 MFMA (64 cycles per 32x32x2) leaves ~1500 cycles
//     per k-step to cover the latency, so an LDS halo tile would buy nothing here.
#include "conv_common.h"

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MPC_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == MPC_ACT_EXP) return (float)exp((double)v);
    return v;
}

// NF = Cout/32 accumulator tiles per wave; CIN4 = first layer (3 input channels padded to 4, two taps
// per 8-deep k-step); EPI = epilogue kind.
template <int NF, bool CIN4, int EPI>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const Params p) {
    constexpr int COUT = NF * 32;
    constexpr int CHUNK_F4 = COUT * 8;                    // float4 per weight chunk (Cout x 32 floats)
    constexpr int STAGE = CHUNK_F4 / 256;                 // float4 per thread per chunk
    extern __shared__ float4 lds[];                       // 2 x CHUNK_F4

    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const Phase ph = p.ph[blockIdx.y];
    const int2 *taps = p.taps + ph.tap_off;
    const int cg = blockIdx.z * COUT;                     // first output channel of this block's cout group

    const int64_t blk = xcd_block();
    if (blk * PIX_PER_BLOCK >= p.P) return;           // padding block of the rounded-up grid
    const int64_t pix = blk * PIX_PER_BLOCK + wave * 32 + j;
    const bool pvalid = pix < p.P;
    int b = 0, by = 0, bx = 0;
    if (pvalid) {
        int64_t hw = (int64_t)p.Hb * p.Wb;
        b = (int)(pix / hw);
        int rem = (int)(pix - (int64_t)b * hw);
        by = rem / p.Wb;
        bx = rem - by * p.Wb;
    }
    const float *in_b = (const float *)p.in + (int64_t)b * p.Hi * p.Wi * p.Cs;
    float *const out_f = (float *)p.out;
    const bool in_abs = (p.flags & MPC_IN_ABS) != 0;

    // ---- loaders ---------------------------------------------------------------------------
    auto load_b = [&](int chunk, float4 (&dst)[4]) {
        if constexpr (CIN4) {
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                int tap = chunk * 8 + k8 * 2 + h;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pvalid && tap < ph.ntaps) {
                    int2 d = taps[tap];
                    int iy = by * p.sy + d.x, ix = bx * p.sx + d.y;
                    if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                        v = *(const float4 *)(in_b + ((int64_t)iy * p.Wi + ix) * p.Cs);
                }
                dst[k8] = v;
            }
        } else {
            int tap = chunk / p.cin_blocks;
            int cb = chunk - tap * p.cin_blocks;
            int2 d = taps[tap];
            int iy = by * p.sy + d.x, ix = bx * p.sx + d.y;
            bool ok = pvalid && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            const float *src = in_b + ((int64_t)iy * p.Wi + ix) * p.Cs + cb * 32 + 4 * h;
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *(const float4 *)(src + 8 * k8);
                if (in_abs) v = make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
                dst[k8] = v;
            }
        }
    };
    const float4 *wsrc = (const float4 *)((const float *)p.wpk + (int64_t)blockIdx.z * p.w_group + ph.w_off);
    auto load_w = [&](int chunk, float4 (&dst)[STAGE]) {
        const float4 *s = wsrc + (int64_t)chunk * CHUNK_F4 + tid;
#pragma unroll
        for (int r = 0; r < STAGE; ++r) dst[r] = s[r * 256];
    };
    auto store_w = [&](int buf, const float4 (&src)[STAGE]) {
        float4 *d = lds + buf * CHUNK_F4 + tid;
#pragma unroll
        for (int r = 0; r < STAGE; ++r) d[r * 256] = src[r];
    };

    // ---- main loop ---------------------------------------------------------------------------
    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    float4 bcur[4], bnext[4], wreg[STAGE];
    load_w(0, wreg);
    load_b(0, bcur);
    store_w(0, wreg);
    __syncthreads();

    const int nch = ph.nchunks;
    for (int ch = 0; ch < nch; ++ch) {
        // Prefetch the next chunk (the last iteration harmlessly re-fetches its own chunk: keeping the
        // loads and the LDS store unconditional keeps wreg / bnext in registers).
        const int nxt = (ch + 1 < nch) ? ch + 1 : ch;
        load_w(nxt, wreg);
        load_b(nxt, bnext);
        const float4 *wl = lds + (ch & 1) * CHUNK_F4 + h * COUT + j;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const float bv[4] = {bcur[k8].x, bcur[k8].y, bcur[k8].z, bcur[k8].w};
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float4 a = wl[(k8 * 2) * COUT + f * 32];
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[f], 0, 0, 0);
            }
        }
        store_w((ch + 1) & 1, wreg);
        __syncthreads();
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) bcur[k8] = bnext[k8];
    }

    // ---- epilogue ----------------------------------------------------------------------------
    // acc[f][r] = channel 32f + (r&3) + 8(r>>2) + 4h of pixel `pix`
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int c0 = 32 * f + 8 * g + 4 * h;
            float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI != EPI_STACKED) bb = *(const float4 *)(p.bias + cg + c0);
            acc[f][4 * g + 0] += bb.x;
            acc[f][4 * g + 1] += bb.y;
            acc[f][4 * g + 2] += bb.z;
            acc[f][4 * g + 3] += bb.w;
        }

    const int oy = by * p.my + ph.qy, ox = bx * p.mx + ph.qx;

    if constexpr (EPI == EPI_STACKED) {
        // rows = (phase q, channel co), q-major; only NF == 1.  Output is NCHW.
        if (pvalid) {
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
    } else if constexpr (EPI == EPI_GDN) {
        const bool inverse = p.act == MPC_ACT_IGDN;
        float *orow = out_f + (((int64_t)b * p.Ho + oy) * p.Wo + ox) * COUT;
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
            // out channel 32fo + 8g + 4h + t needs y of the same channel: acc[fo] (static index needed)
            f32x16 yv;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if (f == fo) yv = acc[f];
            if (pvalid) {
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
                    if (p.flags & MPC_OUT_SPLIT) {
                        // split-fp16 ("S16") output for a following f16x3 layer: [pixel][C/8][hi x8 | lo x8]
                        _Float16 *blk = (_Float16 *)p.out + ((((int64_t)b * p.Ho + oy) * p.Wo + ox) * COUT + (c0 & ~7)) * 2;
                        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                        half4_t hi4, lo4;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float sv = fminf(fmaxf(o[t] * X3_ACT_SCALE, -65504.0f), 65504.0f);
                            _Float16 hh = (_Float16)sv;
                            hi4[t] = hh;
                            lo4[t] = (_Float16)(sv - (float)hh);
                        }
                        *(half4_t *)(blk + (c0 & 7)) = hi4;
                        *(half4_t *)(blk + 8 + (c0 & 7)) = lo4;
                    } else {
                        *(float4 *)(orow + c0) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    } else {
        if (!pvalid) return;
        const bool nchw = (p.flags & MPC_OUT_NCHW) != 0;
        const bool clamp01 = (p.flags & MPC_OUT_CLAMP01) != 0;
        const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
        int sym_clamped = 0;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int c0 = cg + 32 * f + 8 * g + 4 * h;
                float o[4], orr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = apply_act(acc[f][4 * g + t], p.act);
                    if (clamp01) v = fminf(fmaxf(v, 0.0f), 1.0f);
                    o[t] = v;
                    orr[t] = rintf(v);
                }
                if (nchw) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        int64_t a = (((int64_t)b * p.cout + c0 + t) * p.Ho + oy) * p.Wo + ox;
                        if (out_f) out_f[a] = o[t];
                        if (p.out_round) p.out_round[a] = orr[t];
                    }
                } else {
                    if (out_f) *(float4 *)(out_f + opix * p.cout + c0) = make_float4(o[0], o[1], o[2], o[3]);
                    if (p.out_round) *(float4 *)(p.out_round + opix * p.cout + c0) = make_float4(orr[0], orr[1], orr[2], orr[3]);
                    if (p.out_sym) store_symbols4(p, opix * p.cout + c0, orr, sym_clamped);
                }
            }
        if (sym_clamped && p.n_clamped) atomicAdd(p.n_clamped, sym_clamped);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight packing (device side, runs once per layer at mpc_conv_create)
// ---------------------------------------------------------------------------------------------
struct PackDesc {
    int kind, cin, cout, ksize, stride, pad;
    int cout_pad;         // 32 * NF: rows per cout group
    int64_t w_group;      // packed floats per cout group
    int cin4;             // first-layer path
    int f16;              // fp16 operands: chunk = tap x 64 input channels, 8 halfs per slot
    int stacked;          // last-layer path: rows = (phase, co)
    int nphase;
    int ntaps[MAX_PHASES], nchunks[MAX_PHASES], tap_off[MAX_PHASES], qy[MAX_PHASES], qx[MAX_PHASES];
    int64_t w_off[MAX_PHASES];
    int2 taps[4 * MAX_TAPS];     // (dy, dx)
    int2 kidx[4 * MAX_TAPS];     // (ky, kx) of the tap; stacked: unused (derived per phase)
};

// Packed image of one chunk: [kstep (4)][h (2)][cout_pad][t (E)] with E = 4 floats (f32 path, chunk = 32 input
// channels) or 8 halfs (f16 path, chunk = 64 input channels).
__global__ void pack_weights_kernel(const PackDesc d, const float *__restrict__ w, void *__restrict__ out,
                                    int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int E = d.f16 ? 8 : 4;
    const int group = (int)(i / d.w_group);
    const int64_t ig = i - (int64_t)group * d.w_group;
    int phase = 0;
    while (phase + 1 < d.nphase && ig >= d.w_off[phase + 1]) ++phase;
    int64_t l = ig - d.w_off[phase];
    const int chunk_elems = d.cout_pad * 8 * E;
    int chunk = (int)(l / chunk_elems);
    int r = (int)(l % chunk_elems);
    int t = r % E;
    int row = (r / E) % d.cout_pad;
    int kh = (r / E) / d.cout_pad;        // kstep*2 + h
    int k8 = kh >> 1, hh = kh & 1;
    int tap, ci;
    if (d.cin4) {
        tap = chunk * 8 + k8 * 2 + hh;
        ci = t;
    } else {
        int cb = d.cin / (8 * E);
        tap = chunk / cb;
        ci = (chunk % cb) * 8 * E + k8 * 2 * E + hh * E + t;
    }
    float v = 0.0f;
    if (tap < d.ntaps[phase] && ci < d.cin) {
        int2 dd = d.taps[d.tap_off[phase] + tap];
        int co = group * d.cout_pad + row, ky, kx;
        bool ok = true;
        if (d.stacked) {
            int q = row / d.cout;
            co = row % d.cout;
            ok = q < 4;
            // ConvTranspose2d k5 s2 p2: oy = 2*iy - 2 + ky with oy = 2*by + qy, iy = by + dy
            ky = (q >> 1) + 2 - 2 * dd.x;
            kx = (q & 1) + 2 - 2 * dd.y;
            ok = ok && ky >= 0 && ky < d.ksize && kx >= 0 && kx < d.ksize;
        } else {
            int2 kk = d.kidx[d.tap_off[phase] + tap];
            ky = kk.x;
            kx = kk.y;
            ok = co < d.cout;
        }
        if (ok) {
            int64_t src = (d.kind == MPC_CONV)
                              ? (((int64_t)co * d.cin + ci) * d.ksize + ky) * d.ksize + kx
                              : (((int64_t)ci * d.cout + co) * d.ksize + ky) * d.ksize + kx;
            v = w[src];
        }
    }
    if (d.f16) ((_Float16 *)out)[i] = (_Float16)v;
    else ((float *)out)[i] = v;
}

// fp16 gamma as cout/64 chunks in the weight-chunk layout of the fp16 kernel (chunk kc = 64 input channels of
// the mix, 4 k-steps): out[kc][(ks*2 + h)*cout + c'][jj] <- gamma_eff[c'][32f + 16s + 8(jj>>2) + 4h + (jj&3)]
// with f = 2kc + (ks>>1), s = ks&1: the k order of the accumulator registers
// A second image follows it for the kernel that multiplies with v_mfma_f32_16x16x32_f16 (deconv5s2_patch_f16.hip), where
// lane group kg = 2s + h holds, of a 32-channel block, channels 16(jj>>2) + 4kg + (jj&3): same slots, the roles of s
// and jj>>2 exchanged.
__global__ void pack_gamma_f16_kernel(const float *__restrict__ gamma_eff, int cout, _Float16 *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * cout) return;
    int kc = i / (cout * 64), r = i % (cout * 64);
    int jj = r & 7, cp = (r >> 3) % cout, sl = (r >> 3) / cout;
    int hh = sl & 1, ks = sl >> 1;
    int c = 32 * (2 * kc + (ks >> 1)) + 16 * (ks & 1) + 8 * (jj >> 2) + 4 * hh + (jj & 3);
    out[i] = (_Float16)gamma_eff[(int64_t)cp * cout + c];
    int c16 = 32 * (2 * kc + (ks >> 1)) + 16 * (jj >> 2) + 8 * (ks & 1) + 4 * hh + (jj & 3);
    out[(int64_t)cout * cout + i] = (_Float16)gamma_eff[(int64_t)cp * cout + c16];
}

// max |w| as the bit pattern of a non-negative float (monotone under unsigned compare)
__global__ void absmax_kernel(const float *__restrict__ w, int64_t n, unsigned *__restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// max over output channels of sum |w[co, :, :, :]|, as float bits (one block per output channel)
__global__ void l1norm_max_kernel(const float *__restrict__ w, int kind, int cin, int cout, int kk, unsigned *__restrict__ out) {
    __shared__ float part[4];
    const int co = blockIdx.x;
    float a = 0.0f;
    for (int i = threadIdx.x; i < cin * kk; i += blockDim.x) {
        int ci = i / kk, t = i - ci * kk;
        int64_t src = (kind == MPC_CONV) ? ((int64_t)co * cin + ci) * kk + t : ((int64_t)ci * cout + co) * kk + t;
        a += fabsf(w[src]);
    }
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(part[0] + part[1] + part[2] + part[3]));
}

// f16x3 GDN: gamma_eff * scale as fp16 hi/lo, as cout/32 chunks in the weight-chunk layout of the f16x3 kernel
// (chunk kc = 32 input channels of the mix): out[kc][((s*2 + h)*2 + part)*cout + c'][jj] <-
// gamma_eff[c'][32kc + 16s + 8(jj>>2) + 4h + (jj&3)], the k order of the accumulator registers
__global__ void pack_gamma_x3_kernel(const float *__restrict__ gamma_eff, int cout, float scale, _Float16 *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cout * cout * 2) return;
    int kc = i / (cout * 64), r = i % (cout * 64);
    int jj = r & 7, cp = (r >> 3) % cout, sl = (r >> 3) / cout;
    int part = sl & 1, hh = (sl >> 1) & 1, s = sl >> 2;
    int c = 32 * kc + 16 * s + 8 * (jj >> 2) + 4 * hh + (jj & 3);
    float v = gamma_eff[(int64_t)cp * cout + c] * scale;
    _Float16 hi = (_Float16)v;
    out[i] = part ? (_Float16)(v - (float)hi) : hi;
}

// f16x3 weights: chunk = 1 tap x 32 input channels, image [kstep 2][h 2][hi|lo][cout_pad][8] fp16 of w * scale
__global__ void pack_weights_x3_kernel(const PackDesc d, const float *__restrict__ w, float scale,
                                       _Float16 *__restrict__ out, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int group = (int)(i / d.w_group);
    const int64_t ig = i - (int64_t)group * d.w_group;
    int phase = 0;
    while (phase + 1 < d.nphase && ig >= d.w_off[phase + 1]) ++phase;
    int64_t l = ig - d.w_off[phase];
    const int chunk_elems = d.cout_pad * 64;
    int chunk = (int)(l / chunk_elems);
    int r = (int)(l % chunk_elems);
    int jj = r & 7;
    int row = (r >> 3) % d.cout_pad;
    int sl = (r >> 3) / d.cout_pad;       // (ks*2 + h)*2 + part
    int part = sl & 1, hh = (sl >> 1) & 1, ks = sl >> 2;
    int tap, ci;
    if (d.cin4) {                          // first layer: K axis = tap*4 + c (8 taps of the padded pixel per chunk)
        int k = chunk * 32 + ks * 16 + hh * 8 + jj;
        tap = k >> 2;
        ci = k & 3;
    } else {
        int cb = d.cin / 32;
        tap = chunk / cb;
        ci = (chunk % cb) * 32 + ks * 16 + hh * 8 + jj;
    }
    int co = group * d.cout_pad + row;
    float v = 0.0f;
    if (tap < d.ntaps[phase] && ci < d.cin && co < d.cout) {
        int2 kk = d.kidx[d.tap_off[phase] + tap];
        int64_t src = (d.kind == MPC_CONV)
                          ? (((int64_t)co * d.cin + ci) * d.ksize + kk.x) * d.ksize + kk.y
                          : (((int64_t)ci * d.cout + co) * d.ksize + kk.x) * d.ksize + kk.y;
        v = w[src] * scale;
    }
    _Float16 hi = (_Float16)v;
    out[i] = part ? (_Float16)(v - (float)hi) : hi;
}

// GDN reparametrisation (src/compression/models/GDN.py:75-81): max(x, bound)^2 - pedestal
__global__ void gdn_reparam_kernel(const float *__restrict__ x, int64_t n, float bound, float pedestal,
                                   float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = fmaxf(x[i], bound);
    out[i] = __fsub_rn(__fmul_rn(v, v), pedestal);
}

}  // namespace

namespace {

#ifndef MPC_NARROW_BELOW
#define MPC_NARROW_BELOW 256
#endif
constexpr int MPC_NARROW_BELOW_WORKGROUPS = MPC_NARROW_BELOW;     // fewer workgroups than CUs: split the output channels over more of them

template <int NF, bool CIN4, int EPI>
int launch(mpc_ctx *ctx, const Params &p, int nphase, int groups) {
    size_t lds = (size_t)2 * NF * 32 * 32 * sizeof(float);
    auto kern = conv_igemm_kernel<NF, CIN4, EPI>;
    static bool attr_set[MPC_MAX_DEVICES] = {};        // per device (the attribute lives in the device's code object)
    bool &attr_done = attr_set[ctx->device];
    if (lds > 48 * 1024 && !attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    int64_t bx = xcd_grid(mpc_cdiv(p.P, PIX_PER_BLOCK));
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many pixels");
    kern<<<dim3((unsigned)bx, (unsigned)nphase, (unsigned)groups), dim3(256), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "conv_igemm");
    return MPC_OK;
}

}  // namespace

extern "C" {

int mpc_conv_create(mpc_ctx *ctx, const mpc_conv_desc *desc, const float *weight, const float *bias,
                    const float *beta, const float *gamma, mpc_conv **out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, desc && weight && bias && out, "conv_create: NULL argument");
    const mpc_conv_desc &d = *desc;
    const bool gdn = d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN;
    MPC_REQUIRE(ctx, d.kind == MPC_CONV || d.kind == MPC_DECONV, "conv_create: bad kind");
    MPC_REQUIRE(ctx, (d.ksize == 3 || d.ksize == 5) && d.pad == (d.ksize - 1) / 2, "conv_create: ksize must be 3 or 5 with pad (k-1)/2");
    MPC_REQUIRE(ctx, d.stride == 1 || d.stride == 2, "conv_create: stride must be 1 or 2");
    MPC_REQUIRE(ctx, d.kind == MPC_CONV ? d.out_pad == 0 : d.out_pad == d.stride - 1, "conv_create: out_pad must be stride-1 for deconv, 0 for conv");
    MPC_REQUIRE(ctx, !gdn || (beta && gamma), "conv_create: GDN needs beta and gamma");
    const bool cin4 = d.cin <= 4;
    const bool stacked = d.kind == MPC_DECONV && d.stride == 2 && d.cout * 4 <= 32;
    const bool f16 = (d.flags & MPC_F16) != 0;
    const bool x3 = (d.flags & MPC_F16X3) != 0;
    MPC_REQUIRE(ctx, !(x3 && f16), "conv_create: MPC_F16 and MPC_F16X3 are exclusive");
    MPC_REQUIRE(ctx, !x3 || (!stacked && (cin4 ? d.cin == 3 : d.cin % 32 == 0) && (d.flags & MPC_IN_ABS) == 0),
                "conv_create: the f16x3 path needs cin == 3 or a multiple of 32, cout a multiple of 32 and no MPC_IN_ABS");
    MPC_REQUIRE(ctx, !x3 || d.act == MPC_ACT_NONE || d.act == MPC_ACT_RELU || d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN,
                "conv_create: the f16x3 path supports no activation, ReLU, GDN and IGDN");
    MPC_REQUIRE(ctx, !f16 || (!cin4 && d.cin % 64 == 0), "conv_create: the fp16 path needs cin to be a multiple of 64");
    MPC_REQUIRE(ctx, !f16 || (d.flags & MPC_IN_ABS) == 0, "conv_create: MPC_IN_ABS is not available on the fp16 path");
    MPC_REQUIRE(ctx, !f16 || d.act == MPC_ACT_NONE || d.act == MPC_ACT_RELU || d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN,
                "conv_create: the fp16 path supports no activation, ReLU, GDN and IGDN");
    MPC_REQUIRE(ctx, cin4 ? (d.kind == MPC_CONV) : (d.cin % 32 == 0), "conv_create: cin must be <= 4 (Conv2d only) or a multiple of 32");
    MPC_REQUIRE(ctx, stacked || d.cout % 32 == 0, "conv_create: cout must be a multiple of 32 (or <= 8 for a stride-2 ConvTranspose2d)");
    MPC_REQUIRE(ctx, !(stacked && d.ksize != 5), "conv_create: small-cout ConvTranspose2d needs ksize 5");
    // accumulator tiles per wave (nf) and cout groups (one group per blockIdx.z): 128/160/192 channels
    // run as one group, 256/320 as two (GDN needs every channel of a pixel in one wave: one group only)
    const int tiles = stacked ? 1 : d.cout / 32;
    MPC_REQUIRE(ctx, tiles == 1 || tiles == 4 || tiles == 5 || tiles == 6 || tiles == 8 || tiles == 10 || tiles == 12,
                "conv_create: cout must be 128, 160, 192, 256, 320 or 384");
    const int groups = tiles > 6 ? 2 : 1;   // (f16x3 uses the same split: 256 -> 2 x 128)
    const int nf = tiles / groups;
    MPC_REQUIRE(ctx, !gdn || groups == 1, "conv_create: fused GDN supports cout <= 192");
    MPC_REQUIRE(ctx, !(stacked && (gdn || d.act != MPC_ACT_NONE)), "conv_create: small-cout layer supports no activation");

    MPC_REQUIRE(ctx, !f16 || groups == 1, "conv_create: the fp16 path supports cout <= 192");

    mpc_conv *c = new mpc_conv();
    c->desc = d; c->device = ctx->device; c->nf = nf; c->groups = groups; c->cin4 = cin4; c->stacked = stacked; c->f16 = f16; c->x3 = x3;

    PackDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.kind = d.kind; pd.cin = d.cin; pd.cout = d.cout; pd.ksize = d.ksize; pd.stride = d.stride; pd.pad = d.pad;
    pd.cout_pad = nf * 32; pd.cin4 = cin4; pd.stacked = stacked; pd.f16 = f16;
    int ntap_total = 0;
    auto add_phase = [&](int qy, int qx) { pd.qy[pd.nphase] = qy; pd.qx[pd.nphase] = qx; pd.tap_off[pd.nphase] = ntap_total; pd.ntaps[pd.nphase] = 0; return pd.nphase++; };
    auto add_tap = [&](int ph, int dy, int dx, int ky, int kx) { pd.taps[ntap_total] = make_int2(dy, dx); pd.kidx[ntap_total] = make_int2(ky, kx); ++ntap_total; ++pd.ntaps[ph]; };
    if (d.kind == MPC_CONV) {
        // iy = oy*s - pad + ky
        c->sy = c->sx = d.stride; c->my = c->mx = 1;
        int ph = add_phase(0, 0);
        for (int ky = 0; ky < d.ksize; ++ky)
            for (int kx = 0; kx < d.ksize; ++kx) add_tap(ph, ky - d.pad, kx - d.pad, ky, kx);
    } else if (d.stride == 1) {
        // oy = iy - pad + ky  ->  iy = oy + pad - ky
        c->sy = c->sx = 1; c->my = c->mx = 1;
        int ph = add_phase(0, 0);
        for (int ky = 0; ky < d.ksize; ++ky)
            for (int kx = 0; kx < d.ksize; ++kx) add_tap(ph, d.pad - ky, d.pad - kx, ky, kx);
    } else if (stacked) {
        // all four parities as rows of one problem over the 3x3 input neighbourhood
        c->sy = c->sx = 1; c->my = c->mx = 2;
        int ph = add_phase(0, 0);
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) add_tap(ph, dy, dx, 0, 0);
    } else {
        // oy = 2*iy - pad + ky (pad = 2, k = 5); parity q: ky = q + 2 - 2*dy in [0, 4]
        c->sy = c->sx = 1; c->my = c->mx = 2;
        const int order[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};   // 9, 6, 6, 4 taps: heavy first
        for (auto &q : order) {
            int ph = add_phase(q[0], q[1]);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int ky = q[0] + d.pad - 2 * dy, kx = q[1] + d.pad - 2 * dx;
                    if (ky >= 0 && ky < d.ksize && kx >= 0 && kx < d.ksize) add_tap(ph, dy, dx, ky, kx);
                }
        }
    }
    c->nphase = pd.nphase;
    MPC_REQUIRE(ctx, ntap_total <= MAX_TAPS, "conv_create: too many taps");   // (cannot trigger: at most 25)
    c->ntaps_total = ntap_total;
    for (int t = 0; t < ntap_total; ++t) c->taps_host[t] = pd.taps[t];
    int64_t w_total = 0;
    for (int ph = 0; ph < pd.nphase; ++ph) {
        pd.nchunks[ph] = cin4 ? (pd.ntaps[ph] + 7) / 8 : pd.ntaps[ph] * (d.cin / (f16 ? 64 : 32));
        pd.w_off[ph] = w_total;
        w_total += (int64_t)pd.nchunks[ph] * pd.cout_pad * ((f16 || x3) ? 64 : 32);
        c->ph[ph].ntaps = pd.ntaps[ph]; c->ph[ph].nchunks = pd.nchunks[ph];
        c->ph[ph].qy = pd.qy[ph]; c->ph[ph].qx = pd.qx[ph];
        c->ph[ph].w_off = pd.w_off[ph]; c->ph[ph].tap_off = pd.tap_off[ph];
    }

    pd.w_group = w_total;
    c->w_group = w_total;
    w_total *= groups;

    hipError_t e = hipSuccess;
    auto fail = [&](const char *what) {
        snprintf(ctx->err, sizeof(ctx->err), "conv_create: %s: %s", what, hipGetErrorString(e));
        mpc_conv_destroy(c);
        return MPC_EHIP;
    };
    if ((e = hipMalloc((void **)&c->wpk, ((f16 || x3) ? sizeof(_Float16) : sizeof(float)) * w_total)) != hipSuccess) return fail("hipMalloc(weights)");
    if ((e = hipMalloc((void **)&c->bias, sizeof(float) * (nf * groups * 32 + 64))) != hipSuccess) return fail("hipMalloc(bias)");
    if ((e = hipMalloc((void **)&c->taps, sizeof(int2) * ntap_total)) != hipSuccess) return fail("hipMalloc(taps)");
    if ((e = hipMemcpyAsync(c->taps, pd.taps, sizeof(int2) * ntap_total, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail("copy taps");
    if ((e = hipMemsetAsync(c->bias, 0, sizeof(float) * (nf * groups * 32 + 64), ctx->stream)) != hipSuccess) return fail("memset bias");
    if ((e = hipMemcpyAsync(c->bias, bias, sizeof(float) * d.cout, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess) return fail("copy bias");
    if (x3) {
        // per-layer power-of-two weight scale: max |w| * scale in [2^13, 2^14)
        unsigned *mx = nullptr, mx_h = 0;
        const int64_t nw = (int64_t)d.cin * d.cout * d.ksize * d.ksize;
        if ((e = hipMalloc((void **)&mx, sizeof(unsigned))) != hipSuccess) return fail("hipMalloc(absmax)");
        (void)hipMemsetAsync(mx, 0, sizeof(unsigned), ctx->stream);
        absmax_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(weight, nw, mx);
        e = hipMemcpyAsync(&mx_h, mx, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(mx);
        if (e != hipSuccess) return fail("absmax");
        float amax;
        memcpy(&amax, &mx_h, sizeof(float));
        int ex = 0;
        if (amax > 0.0f && amax < 1e30f) (void)frexpf(amax, &ex);      // amax = m * 2^ex, m in [0.5, 1)
        c->w_scale = ldexpf(1.0f, 14 - ex);
        pack_weights_x3_kernel<<<dim3((unsigned)mpc_cdiv(w_total, 256)), dim3(256), 0, ctx->stream>>>(pd, weight, c->w_scale, (_Float16 *)c->wpk, w_total);
    } else {
        pack_weights_kernel<<<dim3((unsigned)mpc_cdiv(w_total, 256)), dim3(256), 0, ctx->stream>>>(pd, weight, c->wpk, w_total);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail("pack_weights launch");
    if (!f16 && !x3 && !cin4 && !stacked && !gdn && tiles > 1) {
        // The same weights once more with ONE 32-channel tile per cout group.  A workgroup's time is its K loop, nf
        // tiles wide, however few pixels the launch has; a launch of fewer workgroups than the GPU has CUs (the
        // hyperprior networks work on 1/64 .. 1/1024 of the image's pixels) runs its groups side by side instead:
        // nf times the workgroups, each 1/nf as long.  Every output channel is summed in the same order either way.
        PackDesc pn = pd;
        pn.cout_pad = 32;
        int64_t group_total = 0;
        for (int ph = 0; ph < pn.nphase; ++ph) {
            pn.w_off[ph] = group_total;
            group_total += (int64_t)pn.nchunks[ph] * 32 * 32;
        }
        pn.w_group = c->w_group_narrow = group_total;
        const int64_t narrow_total = group_total * tiles;
        if ((e = hipMalloc((void **)&c->wpk_narrow, sizeof(float) * narrow_total)) != hipSuccess) return fail("hipMalloc(narrow weights)");
        pack_weights_kernel<<<dim3((unsigned)mpc_cdiv(narrow_total, 256)), dim3(256), 0, ctx->stream>>>(pn, weight, c->wpk_narrow, narrow_total);
        if ((e = hipGetLastError()) != hipSuccess) return fail("pack_weights launch (narrow)");
    }
    if (f16 && stacked && mpc_conv_rgb16_supported(d.cin, d.cout)) {
        if ((e = hipMalloc(&c->wpk_rgb16, mpc_conv_rgb16_image_bytes(d.cin))) != hipSuccess) return fail("hipMalloc(rgb16 weights)");
        if ((e = mpc_conv_rgb16_pack(weight, d.cin, d.cout, c->wpk_rgb16, ctx->stream)) != hipSuccess) return fail("pack_rgb16 launch");
    }
    if (gdn) {
        const float pedestal = 0x1p-36f;
        const float beta_bound = (float)sqrt(1e-6 + 0x1p-36);
        const float gamma_bound = 0x1p-18f;
        if ((e = hipMalloc((void **)&c->beta, sizeof(float) * d.cout)) != hipSuccess) return fail("hipMalloc(beta)");
        if ((e = hipMalloc((void **)&c->gamma, sizeof(float) * d.cout * d.cout)) != hipSuccess) return fail("hipMalloc(gamma)");
        gdn_reparam_kernel<<<dim3((unsigned)mpc_cdiv(d.cout, 256)), dim3(256), 0, ctx->stream>>>(beta, d.cout, beta_bound, pedestal, c->beta);
        gdn_reparam_kernel<<<dim3((unsigned)mpc_cdiv((int64_t)d.cout * d.cout, 256)), dim3(256), 0, ctx->stream>>>(gamma, (int64_t)d.cout * d.cout, gamma_bound, pedestal, (float *)c->gamma);
        if (x3 && d.in_bound > 0.0f) {
            // split-precision GDN: bound |conv output| by in_bound * max_co sum|w| + max|b| to scale y^2 into fp16
            unsigned *tmp = nullptr, host[3] = {0, 0, 0};
            const int kk = d.ksize * d.ksize;
            if ((e = hipMalloc((void **)&tmp, 3 * sizeof(unsigned))) != hipSuccess) return fail("hipMalloc(bounds)");
            (void)hipMemsetAsync(tmp, 0, 3 * sizeof(unsigned), ctx->stream);
            l1norm_max_kernel<<<dim3((unsigned)d.cout), dim3(256), 0, ctx->stream>>>(weight, d.kind, d.cin, d.cout, kk, tmp);
            absmax_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(bias, d.cout, tmp + 1);
            absmax_kernel<<<dim3(64), dim3(256), 0, ctx->stream>>>((const float *)c->gamma, (int64_t)d.cout * d.cout, tmp + 2);
            e = hipMemcpyAsync(host, tmp, sizeof(host), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            (void)hipFree(tmp);
            if (e != hipSuccess) return fail("gdn bounds");
            float l1, bmax, gmax;
            memcpy(&l1, &host[0], 4); memcpy(&bmax, &host[1], 4); memcpy(&gmax, &host[2], 4);
            const float ybound = d.in_bound * l1 + bmax;
            int ey = 0, eg = 0;
            (void)frexpf(ybound * ybound, &ey);                       // y^2 < 2^ey
            if (gmax > 0.0f) (void)frexpf(gmax, &eg);
            const int sy = 15 - ey;                                    // y^2 * 2^sy < 2^15
            if (ybound > 0.0f && gmax > 0.0f && sy >= -20 && sy <= 20) {
                c->y2_scale = ldexpf(1.0f, sy);
                c->gamma_scale = ldexpf(1.0f, 14 - eg);
                if ((e = hipMalloc(&c->gamma_x3, sizeof(_Float16) * 2 * d.cout * d.cout)) != hipSuccess) return fail("hipMalloc(gamma x3)");
                pack_gamma_x3_kernel<<<dim3((unsigned)mpc_cdiv((int64_t)2 * d.cout * d.cout, 256)), dim3(256), 0, ctx->stream>>>(
                    (const float *)c->gamma, d.cout, c->gamma_scale, (_Float16 *)c->gamma_x3);
            }
        }
        if (f16) {
            void *g16 = nullptr;
            if ((e = hipMalloc(&g16, sizeof(_Float16) * 2 * d.cout * d.cout)) != hipSuccess) return fail("hipMalloc(gamma f16)");   // (two operand orders)
            pack_gamma_f16_kernel<<<dim3((unsigned)mpc_cdiv((int64_t)d.cout * d.cout, 256)), dim3(256), 0, ctx->stream>>>((const float *)c->gamma, d.cout, (_Float16 *)g16);
            if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { (void)hipFree(g16); return fail("pack gamma"); }
            (void)hipFree(c->gamma);
            c->gamma = g16;
        }
        if ((e = hipGetLastError()) != hipSuccess) return fail("gdn_reparam launch");
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail("synchronize");
    *out = c;
    return MPC_OK;
}

int mpc_conv_destroy(mpc_conv *c) {
    if (!c) return MPC_OK;
    mpc_device_guard guard(c->device);
    (void)hipFree(c->wpk); (void)hipFree(c->wpk_narrow); (void)hipFree(c->bias); (void)hipFree(c->beta); (void)hipFree(c->gamma); (void)hipFree(c->taps);
    (void)hipFree(c->gamma_x3);
    (void)hipFree(c->wpk_rgb16);
    delete c;
    return MPC_OK;
}

int mpc_conv_out_hw(const mpc_conv *c, int H, int W, int *Ho, int *Wo) {
    if (!c || !Ho || !Wo || H <= 0 || W <= 0) return MPC_EINVAL;
    const mpc_conv_desc &d = c->desc;
    if (d.kind == MPC_CONV) {
        *Ho = (H + 2 * d.pad - d.ksize) / d.stride + 1;
        *Wo = (W + 2 * d.pad - d.ksize) / d.stride + 1;
    } else {
        *Ho = (H - 1) * d.stride - 2 * d.pad + d.ksize + d.out_pad;
        *Wo = (W - 1) * d.stride - 2 * d.pad + d.ksize + d.out_pad;
    }
    return MPC_OK;
}

}  // extern "C"

// Body of mpc_conv_forward and mpc_conv_forward_symbols (sym == nullptr: no symbol output).
static int conv_forward_impl(mpc_ctx *ctx, const mpc_conv *c, const void *in, int B, int H, int W, int flags, void *out,
                             float *out_round, int16_t *sym, int min_feat, int n_sym_values, int32_t *n_clamped_dev,
                             const float *score_target = nullptr, double *score_sse = nullptr) {
    MPC_REQUIRE(ctx, c && in && B > 0 && H > 0 && W > 0, "conv_forward: bad arguments");
    MPC_REQUIRE(ctx, c->device == ctx->device, "conv_forward: layer was created on another device");
    const mpc_conv_desc &d = c->desc;
    const bool want_round = (flags & MPC_OUT_ROUND) != 0;
    MPC_REQUIRE(ctx, sym || (want_round ? out_round != nullptr : out != nullptr), "conv_forward: output pointer is NULL");
    if (sym) {
        MPC_REQUIRE(ctx, d.act == MPC_ACT_NONE && !c->f16 && !c->stacked && !(flags & (MPC_OUT_NCHW | MPC_OUT_CLAMP01 | MPC_OUT_SPLIT)),
                    "conv_forward_symbols: needs a layer without activation on the f32 or f16x3 path, NHWC output");
        MPC_REQUIRE(ctx, n_sym_values > 0 && n_sym_values <= 32767, "conv_forward_symbols: bad symbol range");
        if (n_clamped_dev) MPC_HIP(ctx, hipMemsetAsync(n_clamped_dev, 0, sizeof(int32_t), ctx->stream));
    }
    const bool gdn = d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN;
    MPC_REQUIRE(ctx, !(gdn && (flags & (MPC_OUT_ROUND | MPC_OUT_NCHW | MPC_OUT_CLAMP01))), "conv_forward: GDN layers write NHWC only");
    MPC_REQUIRE(ctx, !(c->stacked && want_round), "conv_forward: rounding is not available on the small-cout layer");
    MPC_REQUIRE(ctx, !c->stacked || (flags & MPC_OUT_NCHW), "conv_forward: the small-cout layer writes NCHW only (pass MPC_OUT_NCHW)");
    Params p;
    memset(&p, 0, sizeof(p));
    int Ho, Wo;
    mpc_conv_out_hw(c, H, W, &Ho, &Wo);
    p.in = in; p.out = out; p.out_round = want_round ? out_round : nullptr;
    p.out_sym = sym; p.sym_min_feat = (float)min_feat; p.sym_max = n_sym_values - 1; p.n_clamped = n_clamped_dev;
    p.zeros = c->bias + c->nf * c->groups * 32;
    p.score_target = score_target; p.score_sse = score_sse;
    p.wpk = c->wpk; p.bias = c->bias; p.beta = c->beta; p.gamma = c->gamma; p.taps = c->taps;
    p.B = B; p.Hi = H; p.Wi = W; p.Cs = c->cin4 ? 4 : d.cin;
    p.Ho = Ho; p.Wo = Wo; p.sy = c->sy; p.sx = c->sx; p.my = c->my; p.mx = c->mx;
    p.Hb = (c->my == 2) ? H : Ho; p.Wb = (c->mx == 2) ? W : Wo;
    p.P = (int64_t)B * p.Hb * p.Wb;
    p.cin_blocks = c->cin4 ? 1 : d.cin / 32;
    p.cout = d.cout; p.act = d.act; p.flags = (flags & ~MPC_OUT_ROUND) | (d.flags & MPC_IN_ABS);
    for (int i = 0; i < c->nphase; ++i) p.ph[i] = c->ph[i];
    for (int t = 0; t < c->ntaps_total; ++t) p.tapv[t] = c->taps_host[t];
    MPC_REQUIRE(ctx, (int64_t)B * H * W * p.Cs < (1ll << 40), "conv_forward: input too large");

    p.w_group = c->w_group;
    const int np = c->nphase, ng = c->groups;
    if (c->x3) {
        MPC_REQUIRE(ctx, !(flags & (MPC_OUT_NCHW | MPC_OUT_CLAMP01)), "conv_forward: the f16x3 path writes NHWC fp32 or the split format only");
        MPC_REQUIRE(ctx, !((flags & MPC_OUT_SPLIT) && want_round), "conv_forward: rounding needs the fp32 output");
        p.cin_blocks = c->cin4 ? 1 : d.cin / 32;
        p.x3_unscale = 1.0f / (c->w_scale * X3_ACT_SCALE);
        p.gamma_x3 = c->gamma_x3;
        p.y2_scale = c->gamma_x3 ? c->y2_scale : 0.0f;
        p.gdn_unscale = c->gamma_x3 ? 1.0f / (c->gamma_scale * c->y2_scale) : 0.0f;
        return mpc_conv_launch_x3(ctx, p, c);
    }
    MPC_REQUIRE(ctx, !(flags & MPC_OUT_SPLIT) || (gdn && !c->f16), "conv_forward: MPC_OUT_SPLIT is available on fp32 GDN layers and f16x3 layers");
    if (c->f16) {
        MPC_REQUIRE(ctx, !want_round && !(flags & MPC_OUT_NCHW && !c->stacked), "conv_forward: the fp16 path writes fp16 NHWC (fp32 NCHW on the small-cout layer) and has no rounding output");
        p.cin_blocks = d.cin / 64;
        return mpc_conv_launch_f16(ctx, p, c);
    }
    if (c->stacked) return launch<1, false, EPI_STACKED>(ctx, p, np, ng);
    if (gdn) {
        if (c->cin4) {
            if (c->nf == 4) return launch<4, true, EPI_GDN>(ctx, p, np, ng);
            if (c->nf == 5) return launch<5, true, EPI_GDN>(ctx, p, np, ng);
            if (c->nf == 6) return launch<6, true, EPI_GDN>(ctx, p, np, ng);
        } else {
            if (c->nf == 4) return launch<4, false, EPI_GDN>(ctx, p, np, ng);
            if (c->nf == 5) return launch<5, false, EPI_GDN>(ctx, p, np, ng);
            if (c->nf == 6) return launch<6, false, EPI_GDN>(ctx, p, np, ng);
        }
    } else if (!c->cin4) {
        if (c->wpk_narrow && mpc_cdiv(p.P, PIX_PER_BLOCK) * np * ng < MPC_NARROW_BELOW_WORKGROUPS) {
            Params pn = p;                                     // one 32-channel tile per group (see mpc_conv_create)
            pn.wpk = c->wpk_narrow;
            pn.w_group = c->w_group_narrow;
            for (int i = 0; i < np; ++i) pn.ph[i].w_off = p.ph[i].w_off / c->nf;
            return launch<1, false, EPI_PLAIN>(ctx, pn, np, c->nf * ng);
        }
        if (c->nf == 4) return launch<4, false, EPI_PLAIN>(ctx, p, np, ng);
        if (c->nf == 5) return launch<5, false, EPI_PLAIN>(ctx, p, np, ng);
        if (c->nf == 6) return launch<6, false, EPI_PLAIN>(ctx, p, np, ng);
    } else {
        if (c->nf == 4) return launch<4, true, EPI_PLAIN>(ctx, p, np, ng);
        if (c->nf == 5) return launch<5, true, EPI_PLAIN>(ctx, p, np, ng);
        if (c->nf == 6) return launch<6, true, EPI_PLAIN>(ctx, p, np, ng);
    }
    MPC_FAIL(ctx, MPC_EINVAL, "conv_forward: unsupported layer shape (cout/32 = %d, cin4 = %d, gdn = %d)", c->nf, (int)c->cin4, (int)gdn);
}

extern "C" {

int mpc_conv_forward(mpc_ctx *ctx, const mpc_conv *c, const void *in, int B, int H, int W, int flags, void *out,
                     float *out_round) {
    MPC_ENTER(ctx);
    return conv_forward_impl(ctx, c, in, B, H, W, flags, out, out_round, nullptr, 0, 1, nullptr);
}

int mpc_conv_scored_supported(const mpc_conv *c, int H, int W) {
    return c && c->f16 && c->stacked && c->wpk_rgb16 && c->desc.kind == MPC_DECONV && mpc_conv_rgb16_rows_supported(H, W) ? 1 : 0;
}

int mpc_conv_forward_scored(mpc_ctx *ctx, const mpc_conv *c, const void *in, int B, int H, int W, float *xhat_nchw,
                            const float *target_nchw, double *sse_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, c && xhat_nchw && target_nchw && sse_dev, "conv_forward_scored: NULL argument");
    MPC_REQUIRE(ctx, mpc_conv_scored_supported(c, H, W),
                "conv_forward_scored: needs the fp16 small-cout transposed layer and an input width that is a multiple of 32 "
                "(otherwise: mpc_conv_forward, then mpc_clamp_sse_batched)");
    MPC_HIP(ctx, hipMemsetAsync(sse_dev, 0, sizeof(double) * B, ctx->stream));
    return conv_forward_impl(ctx, c, in, B, H, W, MPC_OUT_NCHW | MPC_OUT_CLAMP01, xhat_nchw, nullptr, nullptr, 0, 1, nullptr,
                             target_nchw, sse_dev);
}

int mpc_conv_forward_symbols(mpc_ctx *ctx, const mpc_conv *c, const void *in, int B, int H, int W, int min_feat,
                             int n_sym_values, float *out_round, int16_t *sym, int32_t *n_clamped_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sym != nullptr, "conv_forward_symbols: sym is NULL");
    return conv_forward_impl(ctx, c, in, B, H, W, out_round ? MPC_OUT_ROUND : 0, nullptr, out_round, sym, min_feat,
                             n_sym_values, n_clamped_dev);
}

}  // extern "C"
