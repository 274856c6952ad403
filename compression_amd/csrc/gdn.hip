// GDN / IGDN as a layer of its own (src/compression/models/GDN.py:65-95): for an NCHW fp32 tensor
//   norm[b,i,p] = sqrt(beta_eff[i] + sum_j gamma_eff[i,j] * x[b,j,p]^2),   y = x / norm  (GDN)  or  x * norm  (IGDN)
// with beta_eff = max(beta, sqrt(1e-6 + 2^-36))^2 - 2^-36 and gamma_eff = max(gamma, 2^-18)^2 - 2^-36 (:75-81).
// In the transforms this normalisation never runs by itself (it continues the K loop of the producing convolution);
// this kernel exists so that the GDN module is callable like the reference's and has a parity test of its own.
//
// The channel mix is a [C x C] x [C x pixels] product on the exact fp32 matrix pipe (v_mfma_f32_32x32x2_f32: an fp32
// fma chain over j in ascending order).  A wave owns 32 consecutive pixels of one image and NF tiles of 32 output
// channels; in NCHW the 32 pixels of a channel are 128 contiguous bytes, so the B operand (x^2) is read straight
// from memory, one coalesced line per half wave and k-step.  gamma_eff is staged through LDS transposed, 32 input
// channels at a time ([j][i]: the 32 lanes of an A fragment read consecutive floats), and reparametrised on the way.
#include "mpc_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int GDN_WAVES = 4;
constexpr int GDN_JCHUNK = 32;              // input channels per staged chunk of gamma

template <int NF>
__global__ __launch_bounds__(64 * GDN_WAVES) void gdn_forward_kernel(const float *__restrict__ x, const float *__restrict__ beta,
                                                                     const float *__restrict__ gamma, int C, int64_t HW, int inverse,
                                                                     float beta_bound, float gamma_bound, float pedestal,
                                                                     float *__restrict__ out) {
    constexpr int CO = NF * 32;                                   // output channels of this workgroup
    __shared__ float g_t[GDN_JCHUNK][CO + 1];                     // gamma_eff[i0 + i][j0 + j] at [j][i]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, kk = lane >> 5;
    const int b = blockIdx.z, i0 = blockIdx.y * CO;
    const int64_t p = ((int64_t)blockIdx.x * GDN_WAVES + wave) * 32 + n;
    const bool pv = p < HW;
    const float *xb = x + (int64_t)b * C * HW;

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    for (int j0 = 0; j0 < C; j0 += GDN_JCHUNK) {
        __syncthreads();                                          // the previous chunk has been read
        for (int e = threadIdx.x; e < CO * GDN_JCHUNK; e += 64 * GDN_WAVES) {
            const int i = e / GDN_JCHUNK, j = e - i * GDN_JCHUNK;
            const float v = fmaxf(gamma[(int64_t)(i0 + i) * C + j0 + j], gamma_bound);
            g_t[j][i] = __fsub_rn(__fmul_rn(v, v), pedestal);
        }
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < GDN_JCHUNK / 2; ++ks) {
            const int j = 2 * ks + kk;
            const float xv = pv ? xb[(int64_t)(j0 + j) * HW + p] : 0.0f;
            const float sq = __fmul_rn(xv, xv);
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(g_t[j][32 * f + n], sq, acc[f], 0, 0, 0);
        }
    }
    if (!pv) return;
    // acc[f][r]: output channel i0 + 32 f + 8 (r / 4) + 4 kk + (r % 4) of pixel p
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * f + 8 * (r >> 2) + 4 * kk + (r & 3);
            const float bv = fmaxf(beta[i], beta_bound);
            const float be = __fsub_rn(__fmul_rn(bv, bv), pedestal);
            const float nrm = __fsqrt_rn(__fadd_rn(acc[f][r], be));
            const float xv = xb[(int64_t)i * HW + p];
            out[((int64_t)b * C + i) * HW + p] = inverse ? __fmul_rn(xv, nrm) : __fdiv_rn(xv, nrm);
        }
}

template <int NF>
int launch(mpc_ctx *ctx, const float *x, const float *beta, const float *gamma, int B, int C, int64_t HW, int inverse, float *out) {
    const dim3 grid((unsigned)mpc_cdiv(HW, 32 * GDN_WAVES), (unsigned)(C / (32 * NF)), (unsigned)B);
    gdn_forward_kernel<NF><<<grid, dim3(64 * GDN_WAVES), 0, ctx->stream>>>(x, beta, gamma, C, HW, inverse, (float)sqrt(1e-6 + 0x1p-36),
                                                                             0x1p-18f, 0x1p-36f, out);
    MPC_CHECK_LAUNCH(ctx, "gdn_forward");
    return MPC_OK;
}

}  // namespace

extern "C" int mpc_gdn_forward(mpc_ctx *ctx, const float *x, const float *beta, const float *gamma, int B, int C, int64_t HW,
                               int inverse, float *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, x && beta && gamma && out, "gdn_forward: null pointer");
    MPC_REQUIRE(ctx, B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 32 == 0, "gdn_forward: B = %d, C = %d (a multiple of 32), HW = %lld", B, C,
                (long long)HW);
    MPC_REQUIRE(ctx, mpc_cdiv(HW, 32 * GDN_WAVES) < (1ll << 31), "gdn_forward: too many pixels per image");
    const int tiles = C / 32;                    // output tiles of 32 channels: the most per workgroup that divide them, up to 6
    if (tiles % 6 == 0) return launch<6>(ctx, x, beta, gamma, B, C, HW, inverse, out);
    if (tiles % 5 == 0) return launch<5>(ctx, x, beta, gamma, B, C, HW, inverse, out);
    if (tiles % 4 == 0) return launch<4>(ctx, x, beta, gamma, B, C, HW, inverse, out);
    if (tiles % 3 == 0) return launch<3>(ctx, x, beta, gamma, B, C, HW, inverse, out);
    if (tiles % 2 == 0) return launch<2>(ctx, x, beta, gamma, B, C, HW, inverse, out);
    return launch<1>(ctx, x, beta, gamma, B, C, HW, inverse, out);
}
