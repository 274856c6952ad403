// Last layer of the synthesis transform on the fp16 matrix pipe: ConvTranspose2d(cin -> cout <= 4, k5, s2,
// p2, output_padding 1) writing the fp32 NCHW image (Balle2018PT_compressor.py:105-127, the deconv to 3
// channels; testolina_compressor.py:91-100).
//
// All four output parities of a base pixel are rows of ONE small GEMM over the 3x3 input neighbourhood:
//   D[row][pixel] = sum_{tap, c} A[row][tap, c] * in[pixel + tap][c],   row = (co*2 + ypar)*2 + xpar  (4*cout <= 16 rows)
// which is the shape of v_mfma_f32_16x16x32_f16: 16 rows x 16 pixels x 32 input channels per instruction
// (12 of 16 rows are real for cout = 3, against 12 of 32 on the 32x32 tile).  The whole A operand
// (9 taps x cin x 16 rows, 2 KB per 64 channels and tap) stays in LDS for the life of a persistent
// workgroup, so the K loop has no barrier; a wave walks over 64-pixel tiles (four 16-pixel MFMA columns
// sharing every A fragment).  The B operand is read straight from the fp16 NHWC activations: lane
// (pixel n, k-group kg) loads 16 bytes, so the four lanes of a pixel cover 64 contiguous bytes per load.
// The x-parity pair of a row pair lands in consecutive accumulator registers of one lane: the output is
// written as float2, 128 contiguous bytes per 16 lanes.
#include "conv_common.h"
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RGB16_WAVES = 8;
constexpr int RGB16_TILE = 64;            // pixels per wave tile

__device__ __forceinline__ half8 as_half8(const float4 &v) { return __builtin_bit_cast(half8, v); }

// A image: [tap 9][chunk cin/64][kstep 2][lane 64][8 halfs]; lane = kg*16 + row holds channels
// chunk*64 + kstep*32 + kg*8 + jj of its row.  Row (co, ypar, xpar) uses weight tap ky = ypar + 2 - 2dy,
// kx = xpar + 2 - 2dx (oy = 2*iy - 2 + ky with oy = 2*by + ypar, iy = by + dy); taps outside the 5x5 kernel are 0.
__global__ void pack_rgb16_kernel(const float *__restrict__ w, int cin, int cout, _Float16 *__restrict__ out, int total) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cb = cin / 64;
    int jj = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) & 1, rest = i >> 10;
    int chunk = rest % cb, tap = rest / cb;
    int row = lane & 15, kg = lane >> 4;
    int c = chunk * 64 + ks * 32 + kg * 8 + jj;
    int xpar = row & 1, ypar = (row >> 1) & 1, co = row >> 2;
    int dy = tap / 3 - 1, dx = tap % 3 - 1;
    int ky = ypar + 2 - 2 * dy, kx = xpar + 2 - 2 * dx;
    float v = 0.0f;
    if (co < cout && ky >= 0 && ky < 5 && kx >= 0 && kx < 5) v = w[(((int64_t)c * cout + co) * 5 + ky) * 5 + kx];
    out[i] = (_Float16)v;
}

__global__ __launch_bounds__(64 * RGB16_WAVES, 2) void deconv_rgb16_kernel(const Params p, const float4 *__restrict__ wimg, int nsteps,
                                                                             int64_t ntiles) {
    extern __shared__ float4 lds[];                        // nsteps * 128 float4
    const int tid = threadIdx.x;
    for (int i = tid; i < nsteps * 128; i += 64 * RGB16_WAVES) lds[i] = wimg[i];
    __syncthreads();

    const int lane = tid & 63, n = lane & 15, kg = lane >> 4, wave = tid >> 6;
    const int cb = p.cin_blocks;                           // cin / 64
    const _Float16 *in = (const _Float16 *)p.in;
    float *out = (float *)p.out;
    const bool clamp01 = (p.flags & MPC_OUT_CLAMP01) != 0;
    const int64_t hw = (int64_t)p.Hb * p.Wb;

    // XCD c (workgroups c, c+8, ...) walks the c-th contiguous eighth of the tiles, a wave at a time
    const int64_t per_xcd = (ntiles + NUM_XCD - 1) / NUM_XCD;
    const int64_t t0 = (int64_t)(blockIdx.x % NUM_XCD) * per_xcd;
    const int64_t t1 = (t0 + per_xcd < ntiles) ? t0 + per_xcd : ntiles;
    const int64_t stride = (int64_t)(gridDim.x / NUM_XCD) * RGB16_WAVES;
    for (int64_t t = t0 + (int64_t)(blockIdx.x / NUM_XCD) * RGB16_WAVES + wave; t < t1; t += stride) {
        bool valid[4];
        int pb[4], py[4], px[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t pix = t * RGB16_TILE + u * 16 + n;
            valid[u] = pix < p.P;
            pb[u] = py[u] = px[u] = 0;
            if (valid[u]) {
                pb[u] = (int)(pix / hw);
                int rem = (int)(pix - (int64_t)pb[u] * hw);
                py[u] = rem / p.Wb;
                px[u] = rem - py[u] * p.Wb;
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto load_b = [&](int step, float4 (&dst)[2][4]) {
            int tap = step / cb, chunk = step - tap * cb;
            int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int iy = py[u] + dy, ix = px[u] + dx;
                bool ok = valid[u] && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
                const _Float16 *src = in + (((int64_t)pb[u] * p.Hi + iy) * p.Wi + ix) * p.Cs + chunk * 64 + kg * 8;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (ok) {
                    v0 = *(const float4 *)src;
                    v1 = *(const float4 *)(src + 32);
                }
                dst[0][u] = v0;
                dst[1][u] = v1;
            }
        };

        float4 bcur[2][4], bnext[2][4];
        load_b(0, bcur);
        for (int s = 0; s < nsteps; ++s) {
            load_b(s + 1 < nsteps ? s + 1 : s, bnext);     // unconditional prefetch
            const half8 a0 = as_half8(lds[(s * 2) * 64 + lane]), a1 = as_half8(lds[(s * 2 + 1) * 64 + lane]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, as_half8(bcur[0][u]), acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, as_half8(bcur[1][u]), acc[u], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int u = 0; u < 4; ++u) bcur[k][u] = bnext[k][u];
        }

        // lane (n, kg) holds rows 4kg..4kg+3 of pixel n: (co, ypar) = (kg, 0) in registers 0,1 and (kg, 1) in 2,3
        if (kg < p.cout) {
            const float bias = p.bias[kg];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!valid[u]) continue;
#pragma unroll
                for (int yp = 0; yp < 2; ++yp) {
                    float v0 = acc[u][2 * yp] + bias, v1 = acc[u][2 * yp + 1] + bias;
                    if (clamp01) {
                        v0 = fminf(fmaxf(v0, 0.0f), 1.0f);
                        v1 = fminf(fmaxf(v1, 0.0f), 1.0f);
                    }
                    float *o = out + (((int64_t)pb[u] * p.cout + kg) * p.Ho + 2 * py[u] + yp) * p.Wo + 2 * px[u];
                    *(float2 *)o = make_float2(v0, v1);
                }
            }
        }
    }
}

// Variant for base grids whose width is a multiple of 32 (every size the codec pads to): a wave owns a strip of 32
// columns.  For an input row and 64-channel chunk it stages the 34 input pixels x0-1 .. x0+32 once, in full
// 128-byte lines, into a wave-private LDS tile and reads the B operands of the three horizontal taps from there at
// shifted positions: a third of the vector-memory traffic of the direct form, and whole cache lines per load
// instruction.  Tile layout: [pixel 34][8 pieces of 16 bytes], piece q of pixel p at column q ^ (p & 7)
// (conflict-free for the 16-byte reads of 16 consecutive pixels).
//
// The wave walks DOWN its strip over a segment of consecutive output rows: an input row is staged once and
// multiplied into the three output rows it belongs to (three accumulator sets, 24 registers); when input row r is
// done, output row r - 1 is complete and is written.  (One output row per tile, each input row then fetched by
// three tiles at three different times, moved 2.7x the input through the fabric: the 4 MB L2 of an XCD does not
// hold what its 512 waves stream in between.  0.39 -> 0.25 ms on 24 x 192 x 256 x 384.)  Input traffic is
// (rows + 2) / rows of the compulsory bytes, LDS operand reads a third per MFMA.  The launch picks the rows per
// segment so that all segments fit ONE round of the persistent grid where the image allows it (a second, partly
// filled round costs more than the longer walk): strip_rows() below.
constexpr int ROWS_WAVES = 16;
constexpr int ROWS_TILE = 32;
constexpr int ROWS_STAGE_F4 = 34 * 8;      // 16-byte slots of one staging tile
constexpr int STRIP_ROWS_MIN = 2, STRIP_ROWS_MAX = 32;

inline int strip_rows(int64_t strips, int Hb) {
    int64_t segs = (int64_t)NUM_CU * ROWS_WAVES / (strips > 0 ? strips : 1);   // segments per strip that fit one round
    if (segs < 1) segs = 1;
    int64_t rows = (Hb + segs - 1) / segs;
    return (int)(rows < STRIP_ROWS_MIN ? STRIP_ROWS_MIN : rows > STRIP_ROWS_MAX ? STRIP_ROWS_MAX : rows);
}

__global__ __launch_bounds__(64 * ROWS_WAVES) void deconv_rgb16_rows_kernel(const Params p, const float4 *__restrict__ wimg,
                                                                               int nsteps, int64_t nunits, int seg_rows) {
    extern __shared__ float4 lds[];                        // nsteps * 128 weights, then ROWS_WAVES staging tiles
    const int tid = threadIdx.x;
    for (int i = tid; i < nsteps * 128; i += 64 * ROWS_WAVES) lds[i] = wimg[i];
    __syncthreads();

    const int lane = tid & 63, n = lane & 15, kg = lane >> 4, wave = tid >> 6;
    float4 *stage = lds + nsteps * 128 + wave * ROWS_STAGE_F4;
    const int cb = p.cin_blocks;                           // cin / 64
    const _Float16 *in = (const _Float16 *)p.in;
    float *out = (float *)p.out;
    const bool clamp01 = (p.flags & MPC_OUT_CLAMP01) != 0;
    const int tw = p.Wb / ROWS_TILE;                       // strips per row
    const int segs = (p.Hb + seg_rows - 1) / seg_rows;     // segments per strip

    // staging role of this lane: pieces j = lane + 64 i of the 34 x 8 tile, i.e. piece sq of pixels sp0 + 8 i
    const int sp0 = lane >> 3, sq = lane & 7;
    const int slot0 = sp0 * 8 + (sq ^ sp0);                // (sp0 + 8 i) & 7 == sp0; slot of pixel sp0 + 8 i is slot0 + 64 i
    const bool fifth = sp0 < 2;                            // pixels 32, 33
    double sq_err = 0.0;                                   // scoring against p.score_target, per image
    int err_image = -1;
    auto flush_err = [&]() {
        if (err_image >= 0) {
            for (int off = 32; off > 0; off >>= 1) sq_err += __shfl_down(sq_err, off, 64);
            if (lane == 0) atomicAdd(p.score_sse + err_image, sq_err);
        }
        sq_err = 0.0;
    };

    const int64_t per_xcd = (nunits + NUM_XCD - 1) / NUM_XCD;
    const int64_t t0 = (int64_t)(blockIdx.x % NUM_XCD) * per_xcd;
    const int64_t t1 = (t0 + per_xcd < nunits) ? t0 + per_xcd : nunits;
    // wave w of the XCD's workgroup k takes units w * (workgroups of the XCD) + k, ...: a grid that is not full leaves
    // waves idle on every CU rather than CUs idle
    const int64_t wgs = gridDim.x / NUM_XCD, stride = wgs * ROWS_WAVES;
    for (int64_t t = t0 + (int64_t)wave * wgs + blockIdx.x / NUM_XCD; t < t1; t += stride) {
        // unit t = (image, segment, strip), strips of one segment next to each other
        const int x0 = (int)(t % tw) * ROWS_TILE;
        const int64_t rest = t / tw;
        const int seg = (int)(rest % segs), b = (int)(rest / segs);
        const int y0 = seg * seg_rows, y1 = (y0 + seg_rows < p.Hb) ? y0 + seg_rows : p.Hb;         // output rows [y0, y1)
        const int r0 = (y0 > 0) ? y0 - 1 : 0, r1 = (y1 < p.Hi) ? y1 : p.Hi - 1;                    // input rows [r0, r1]
        if (p.score_target && b != err_image) {            // (wave-uniform)
            flush_err();
            err_image = b;
        }
        f32x4 acc[3][2];                                   // output row y lives in set y % 3
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto load_stage = [&](int r, int chunk, float4 (&dst)[5]) {
            const _Float16 *lanep = in + (((int64_t)b * p.Hi + r) * p.Wi + x0 - 1 + sp0) * p.Cs + chunk * 64 + sq * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int gx = x0 - 1 + sp0 + 8 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((i < 4 || fifth) && gx >= 0 && gx < p.Wi) v = *(const float4 *)(lanep + (int64_t)(8 * i) * p.Cs);
                dst[i] = v;
            }
        };
        auto store_row = [&](int y, const f32x4 (&a)[2]) {
            // lane (n, kg) holds rows 4kg..4kg+3 of pixel n: (co, ypar) = (kg, 0) in registers 0,1 and (kg, 1) in 2,3
            if (kg < p.cout) {
                const float bias = p.bias[kg];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int yp = 0; yp < 2; ++yp) {
                        float v0 = a[u][2 * yp] + bias, v1 = a[u][2 * yp + 1] + bias;
                        if (clamp01) {
                            v0 = fminf(fmaxf(v0, 0.0f), 1.0f);
                            v1 = fminf(fmaxf(v1, 0.0f), 1.0f);
                        }
                        const int64_t oi = (((int64_t)b * p.cout + kg) * p.Ho + 2 * y + yp) * p.Wo + 2 * (x0 + 16 * u + n);
                        *(float2 *)(out + oi) = make_float2(v0, v1);
                        if (p.score_target) {
                            const float2 tg = *(const float2 *)(p.score_target + oi);
                            const float d0 = v0 - tg.x, d1 = v1 - tg.y;
                            sq_err += (double)(d0 * d0) + (double)(d1 * d1);
                        }
                    }
                }
            }
        };

        float4 sreg[5];
        load_stage(r0, 0, sreg);
        // One input row.  PH = (r - r0) % 3 makes the accumulator sets compile-time: output row y lives in set
        // (y - r0 + 1) % 3, so rows r - 1, r, r + 1 are sets PH, PH + 1, PH + 2 (mod 3).
        auto input_row = [&](auto ph, int r) {
            constexpr int PH = decltype(ph)::value;
            constexpr int S_UP = PH % 3, S_MID = (PH + 1) % 3, S_DOWN = (PH + 2) % 3;
            // which of the three output rows of input row r lie in this segment (wave-uniform)
            const bool up = r - 1 >= y0 && r - 1 < y1, mid = r >= y0 && r < y1, down = r + 1 >= y0 && r + 1 < y1;
            for (int chunk = 0; chunk < cb; ++chunk) {
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < 4 || fifth) stage[slot0 + 64 * i] = sreg[i];
                if (chunk + 1 < cb) load_stage(r, chunk + 1, sreg);
                else if (r < r1) load_stage(r + 1, 0, sreg);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 bv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int px = 16 * u + n + dx;    // = 1 + 16u + n + (dx - 1)
                            bv[u] = as_half8(stage[px * 8 + ((ks * 4 + kg) ^ (px & 7))]);
                        }
                        // output row y = r - dy takes weight tap (dy + 1) * 3 + dx
                        if (up) {                          // dy = +1
                            const half8 a = as_half8(lds[(((6 + dx) * cb + chunk) * 2 + ks) * 64 + lane]);
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[S_UP][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bv[u], acc[S_UP][u], 0, 0, 0);
                        }
                        if (mid) {                         // dy = 0
                            const half8 a = as_half8(lds[(((3 + dx) * cb + chunk) * 2 + ks) * 64 + lane]);
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[S_MID][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bv[u], acc[S_MID][u], 0, 0, 0);
                        }
                        if (down) {                        // dy = -1
                            const half8 a = as_half8(lds[((dx * cb + chunk) * 2 + ks) * 64 + lane]);
#pragma unroll
                            for (int u = 0; u < 2; ++u) acc[S_DOWN][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bv[u], acc[S_DOWN][u], 0, 0, 0);
                        }
                    }
                }
            }
            // output row r - 1 has all its input rows now; its set becomes that of row r + 2
            if (up) {
                store_row(r - 1, acc[S_UP]);
                acc[S_UP][0] = acc[S_UP][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // the image's last row has no input row below it that would complete it
            if (r == r1 && mid) store_row(r, acc[S_MID]);
        };
        for (int r = r0; r <= r1; r += 3) {
            input_row(std::integral_constant<int, 0>{}, r);
            if (r + 1 <= r1) input_row(std::integral_constant<int, 1>{}, r + 1);
            if (r + 2 <= r1) input_row(std::integral_constant<int, 2>{}, r + 2);
        }
    }
    if (p.score_target) flush_err();
}

}  // namespace

// bytes of the packed A image
size_t mpc_conv_rgb16_image_bytes(int cin) { return (size_t)9 * (cin / 64) * 2 * 64 * 16; }

// the row-tiled kernel (the one that can score its output) takes whole 32-pixel runs of input rows
bool mpc_conv_rgb16_rows_supported(int H, int W) { return H > 0 && W > 0 && W % ROWS_TILE == 0; }

bool mpc_conv_rgb16_supported(int cin, int cout) {
    return cout >= 1 && cout <= 4 && cin % 64 == 0 && mpc_conv_rgb16_image_bytes(cin) <= 72 * 1024;
}

hipError_t mpc_conv_rgb16_pack(const float *weight, int cin, int cout, void *image, hipStream_t stream) {
    const int total = (int)(mpc_conv_rgb16_image_bytes(cin) / sizeof(_Float16));
    pack_rgb16_kernel<<<dim3((unsigned)mpc_cdiv(total, 256)), dim3(256), 0, stream>>>(weight, cin, cout, (_Float16 *)image, total);
    return hipGetLastError();
}

int mpc_conv_launch_rgb16(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    const int nsteps = 9 * (c->desc.cin / 64);
    const size_t lds = mpc_conv_rgb16_image_bytes(c->desc.cin);
    static bool attr_set[MPC_MAX_DEVICES] = {};        // per device (the attribute lives in the device's code object)
    bool &attr_done = attr_set[ctx->device];
    if (lds > 48 * 1024 && !attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)deconv_rgb16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    if (p.Wb % ROWS_TILE == 0 && p.Hb == p.Hi && p.Wb == p.Wi) {
        const size_t lds_rows = lds + (size_t)ROWS_WAVES * ROWS_STAGE_F4 * sizeof(float4);
        static bool attr_rows_set[MPC_MAX_DEVICES] = {};
        if (!attr_rows_set[ctx->device]) {
            MPC_HIP(ctx, hipFuncSetAttribute((const void *)deconv_rgb16_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
            attr_rows_set[ctx->device] = true;
        }
        const int seg_rows = strip_rows((int64_t)p.B * (p.Wb / ROWS_TILE), p.Hb);
        const int64_t nu = (int64_t)p.B * mpc_cdiv(p.Hb, seg_rows) * (p.Wb / ROWS_TILE);
        int64_t blocks = nu < NUM_CU ? nu : NUM_CU; // persistent: one 16-wave workgroup per CU
        blocks = xcd_grid(blocks);
        deconv_rgb16_rows_kernel<<<dim3((unsigned)blocks), dim3(64 * ROWS_WAVES), lds_rows, ctx->stream>>>(p, (const float4 *)c->wpk_rgb16, nsteps, nu, seg_rows);
        MPC_CHECK_LAUNCH(ctx, "deconv_rgb16_rows");
        return MPC_OK;
    }
    const int64_t ntiles = mpc_cdiv(p.P, RGB16_TILE);
    // persistent workgroups: two per CU while there is work for them, a multiple of the XCD count
    int64_t blocks = mpc_cdiv(ntiles, RGB16_WAVES);
    if (blocks > 2 * NUM_CU) blocks = 2 * NUM_CU;
    blocks = xcd_grid(blocks);
    deconv_rgb16_kernel<<<dim3((unsigned)blocks), dim3(64 * RGB16_WAVES), lds, ctx->stream>>>(p, (const float4 *)c->wpk_rgb16, nsteps, ntiles);
    MPC_CHECK_LAUNCH(ctx, "deconv_rgb16");
    return MPC_OK;
}
