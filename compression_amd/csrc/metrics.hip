// Image-quality statistics of the reference's test harness on the device: SSIM / contrast-structure sums with a
// separable Gaussian window (src/common/extlibs/pt_ms_ssim.py:26-84) and the 2x average pooling between the
// MS-SSIM scales (:203-207).  PSNR comes from mpc_clamp_sse.  HBM-bound: two planes read once per scale.
#include "mpc_internal.h"

namespace {

constexpr int MAX_WIN = 15;
constexpr int TH = 16, TW = 64;            // output tile of one workgroup

struct Window {
    int size;
    float w[MAX_WIN];
};

// One workgroup = one TH x TW tile of the valid output of one (image, channel) plane.  The (TH+size-1) x
// (TW+size-1) input tiles of X and Y go to LDS, the horizontal pass writes the five filtered rows
// (x, y, xx, yy, xy) back to LDS, the vertical pass finishes the five local moments per output pixel.
__global__ __launch_bounds__(256) void ssim_stats_kernel(const float *__restrict__ X, const float *__restrict__ Y, int H, int W,
                                                         Window win, float C1, float C2, double *__restrict__ ssim_sum,
                                                         double *__restrict__ cs_sum) {
    extern __shared__ float sm[];
    const int S = win.size, IH = TH + S - 1, IW = TW + S - 1;
    float *tx = sm, *ty = sm + IH * IW;                    // input tiles
    float *hp = sm + 2 * IH * IW;                          // horizontal pass: [5][IH][TW]
    const int plane = blockIdx.z;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int OH = H - S + 1, OW = W - S + 1;
    const float *xp = X + (int64_t)plane * H * W, *yp = Y + (int64_t)plane * H * W;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        int r = i / IW, c = i - r * IW;
        int gy = oy0 + r, gx = ox0 + c;
        bool ok = gy < H && gx < W;
        tx[i] = ok ? xp[(int64_t)gy * W + gx] : 0.0f;
        ty[i] = ok ? yp[(int64_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < IH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
        for (int k = 0; k < S; ++k) {
            float wv = win.w[k], xv = tx[r * IW + c + k], yv = ty[r * IW + c + k];
            a += wv * xv;
            b += wv * yv;
            aa += wv * (xv * xv);
            bb += wv * (yv * yv);
            ab += wv * (xv * yv);
        }
        hp[0 * IH * TW + i] = a;
        hp[1 * IH * TW + i] = b;
        hp[2 * IH * TW + i] = aa;
        hp[3 * IH * TW + i] = bb;
        hp[4 * IH * TW + i] = ab;
    }
    __syncthreads();
    double s_ssim = 0.0, s_cs = 0.0;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        if (oy0 + r >= OH || ox0 + c >= OW) continue;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < S; ++k) {
            float wv = win.w[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += wv * hp[q * IH * TW + (r + k) * TW + c];
        }
        float mu1 = m[0], mu2 = m[1];
        float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        float cs = (2.0f * s12 + C2) / (s1 + s2 + C2);
        float ss = ((2.0f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
        s_ssim += (double)ss;
        s_cs += (double)cs;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s_ssim += __shfl_down(s_ssim, off, 64);
        s_cs += __shfl_down(s_cs, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&ssim_sum[plane], s_ssim);
        atomicAdd(&cs_sum[plane], s_cs);
    }
}

// F.avg_pool2d(kernel 2, stride 2, padding (H%2, W%2), count_include_pad): zero padding, always divided by 4
__global__ void avg_pool2_kernel(const float *__restrict__ in, int64_t planes, int H, int W, int ph, int pw, int Ho, int Wo,
                                 float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    int ox = (int)(i % Wo);
    int oy = (int)((i / Wo) % Ho);
    int64_t pl = i / ((int64_t)Wo * Ho);
    const float *p = in + pl * H * W;
    float s = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            int y = 2 * oy - ph + dy, x = 2 * ox - pw + dx;
            if (y >= 0 && y < H && x >= 0 && x < W) s += p[(int64_t)y * W + x];
        }
    out[i] = s * 0.25f;
}

// Running sums of the evaluation statistics of a batch (layout documented at mpc_stats_accumulate).  One wave.
struct HostSums {
    double bits, pixels, bytes, seconds, side_bits;
};

__global__ __launch_bounds__(64) void stats_accumulate_kernel(const double *__restrict__ sse, const double *__restrict__ msssim,
                                                              int n_images, double values_per_image, HostSums hs,
                                                              int reset, double *__restrict__ vec) {
    double sq = 0.0, psnr = 0.0, ms = 0.0, msdb = 0.0;
    for (int i = threadIdx.x; i < n_images; i += 64) {
        const double e = sse[i];
        sq += e;
        const double mse = e / values_per_image;
        psnr += mse > 0.0 ? 10.0 * log10(1.0 / mse) : (double)__builtin_inff();
        if (msssim) {
            const double m = msssim[i];
            ms += m;
            if (m > 0.0 && m < 1.0) msdb += -10.0 * log10(1.0 - m);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sq += __shfl_down(sq, off, 64);
        psnr += __shfl_down(psnr, off, 64);
        ms += __shfl_down(ms, off, 64);
        msdb += __shfl_down(msdb, off, 64);
    }
    // lane i owns vec[i]: no lane reads what another lane has just written
    sq = __shfl(sq, 0, 64); psnr = __shfl(psnr, 0, 64); ms = __shfl(ms, 0, 64); msdb = __shfl(msdb, 0, 64);
    const double add[16] = {hs.bits, hs.pixels, sq, psnr, (double)n_images, hs.bytes, hs.seconds, hs.side_bits, ms, msdb,
                            0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (threadIdx.x < 16) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if ((int)threadIdx.x == k) v = add[k];
        vec[threadIdx.x] = (reset ? 0.0 : vec[threadIdx.x]) + v;
    }
}

}  // namespace

extern "C" {

int mpc_stats_accumulate(mpc_ctx *ctx, const double *sse_dev, const double *msssim_dev, int n_images,
                         int64_t values_per_image, const double host_sums[5], int reset, double *vec_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sse_dev && host_sums && vec_dev && n_images > 0 && values_per_image > 0, "stats_accumulate: bad arguments");
    HostSums hs = {host_sums[0], host_sums[1], host_sums[2], host_sums[3], host_sums[4]};
    stats_accumulate_kernel<<<dim3(1), dim3(64), 0, ctx->stream>>>(sse_dev, msssim_dev, n_images, (double)values_per_image, hs, reset, vec_dev);
    MPC_CHECK_LAUNCH(ctx, "stats_accumulate");
    return MPC_OK;
}

int mpc_ssim_stats(mpc_ctx *ctx, const float *x, const float *y, int planes, int H, int W, const float *window,
                   int win_size, float C1, float C2, double *ssim_sum_dev, double *cs_sum_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, x && y && window && ssim_sum_dev && cs_sum_dev && planes > 0, "ssim_stats: bad arguments");
    MPC_REQUIRE(ctx, win_size >= 1 && win_size <= MAX_WIN && (win_size & 1), "ssim_stats: window size must be odd and <= 15");
    MPC_REQUIRE(ctx, H >= win_size && W >= win_size, "ssim_stats: image smaller than the window");
    MPC_REQUIRE(ctx, planes <= 65535, "ssim_stats: too many planes in one call");
    Window win;
    win.size = win_size;
    for (int i = 0; i < MAX_WIN; ++i) win.w[i] = i < win_size ? window[i] : 0.0f;
    MPC_HIP(ctx, hipMemsetAsync(ssim_sum_dev, 0, sizeof(double) * planes, ctx->stream));
    MPC_HIP(ctx, hipMemsetAsync(cs_sum_dev, 0, sizeof(double) * planes, ctx->stream));
    const int OH = H - win_size + 1, OW = W - win_size + 1;
    const int IH = TH + win_size - 1, IW = TW + win_size - 1;
    size_t lds = sizeof(float) * ((size_t)2 * IH * IW + (size_t)5 * IH * TW);
    dim3 grid((unsigned)mpc_cdiv(OW, TW), (unsigned)mpc_cdiv(OH, TH), (unsigned)planes);
    MPC_REQUIRE(ctx, grid.y <= 65535, "ssim_stats: image too tall");
    if (lds > 48 * 1024)
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)ssim_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ssim_stats_kernel<<<grid, dim3(256), lds, ctx->stream>>>(x, y, H, W, win, C1, C2, ssim_sum_dev, cs_sum_dev);
    MPC_CHECK_LAUNCH(ctx, "ssim_stats");
    return MPC_OK;
}

int mpc_avg_pool2(mpc_ctx *ctx, const float *in, int64_t planes, int H, int W, float *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, in && out && planes > 0 && H > 0 && W > 0, "avg_pool2: bad arguments");
    const int ph = H % 2, pw = W % 2;
    const int Ho = (H + 2 * ph - 2) / 2 + 1, Wo = (W + 2 * pw - 2) / 2 + 1;
    int64_t n = planes * Ho * Wo;
    avg_pool2_kernel<<<dim3((unsigned)mpc_cdiv(n, 256)), dim3(256), 0, ctx->stream>>>(in, planes, H, W, ph, pw, Ho, Wo, out);
    MPC_CHECK_LAUNCH(ctx, "avg_pool2");
    return MPC_OK;
}

}  // extern "C"
