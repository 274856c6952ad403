// fp16 5x5 stride-2 transposed convolution (the three big layers of the synthesis transform, fused IGDN) with the
// input staged through LDS as a 2-D patch and every global -> LDS transfer done by LDS-DMA.
//
// Why.  A stride-2 transposed convolution is four stride-1 problems on the input grid, one per output parity, with
// 9 / 6 / 6 / 4 taps whose offsets lie in [-1, 1]^2.  In the register-gather kernel (conv_igemm_f16.hip) every lane
// fetches its own pixel's 16 bytes per k-step and tap: one wave-instruction touches 32 cache lines and uses 32 bytes
// of each, and that gather (not the MFMAs, not LDS) paces the K loop -- in-kernel timestamps put the K loop at 43 % of
// the MFMA rate, and the stretch scales with the number of such loads.  Here a workgroup owns a 16 x 8 tile of input
// pixels and keeps the 18 x 10 patch around it, 32 input channels at a time, in LDS: the patch is fetched once per
// channel block in whole 64-byte runs (1.4x the tile's bytes, the halo) and all taps read it with ds_read_b128.
//
// K walk: for each 32-channel block, the taps of this workgroup's phase, two per step.  One tap x 32 channels = one
// k-step of v_mfma_f32_16x16x32_f16 over the wave's 2 NF x 2 output tiles (a wave = 2 tile rows of 16 pixels x all
// output channels in tiles of 16; lane (m, kg) = lane & 15, lane >> 4 holds k-values 8 kg .. 8 kg + 7 of row or
// column m of an operand and rows 4 kg .. 4 kg + 3 of column m of a result).  The 16 x 16 x 32 shape issues the same
// FLOPs per cycle as 32 x 32 x 16 but the chip, power-limited in this loop, holds a higher clock with it: 5-6 % of
// the layers.  The weights are the packed chunks of conv_igemm_f16.hip read in halves (12 KB for 192 output channels),
// streamed by LDS-DMA one step ahead; the next channel block's patch is brought by LDS-DMA into the other patch buffer
// during the steps of the current block.  No transfer passes through registers, so a step is: a few DMA issues,
// 2 x (2 + 2 NF) LDS reads, 8 NF MFMAs, one observed wait for the transfers (dma_wait_last), one s_barrier.
//
// LDS image of a patch (16-byte slots): [row 10][96 slots], pixel c of a row at slots 5c .. 5c+3 (4 x 8 channels) with
// one slot of padding per pixel and six per row: a pixel pitch of 80 bytes spreads the 16 pixels of a tile row over
// all 64 banks, and a row pitch of 1536 bytes (a multiple of the 256-byte bank row) keeps the two tile rows that share
// a ds_read_b128 lane group on distinct slots.  Padding slots are filled too (from a page of zeros): a DMA piece is
// 64 consecutive slots whatever they hold.
//
// Two workgroups of four waves per CU (80 KB of LDS each): their barriers and epilogues fall out of phase, so one's
// MFMAs cover the other's waits and stores.  Same weight format, GDN continuation and output path as
// conv_igemm_f16.hip (gamma comes in this kernel's own k order, the second image that pack_gamma_f16_kernel writes);
// the K axis is summed in another order (channel block major), so results agree with it to fp32 rounding, not bit for
// bit.
#include <type_traits>

#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr float SQ_SCALE = 0.00390625f;      // 2^-8
constexpr float SQ_UNSCALE = 256.0f;
constexpr float HALF_MAX = 65504.0f;

constexpr int TILE_W = 16, TILE_H = 8, NW = 4, NT = 64 * NW;
constexpr int PATCH_COLS = TILE_W + 2, PATCH_ROWS = TILE_H + 2;
constexpr int PIX_F4 = 5;                            // slots per patch pixel (4 data + 1 padding)
constexpr int ROW_F4 = 96;                           // slots per patch row (18 x 5 = 90 + 6 padding)
constexpr int PPW = 4;                               // patch pieces (1 KB of DMA) per wave, at most

__device__ __forceinline__ half8 as_half8(const f4 &v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ _Float16 sat_half(float v) { return (_Float16)fminf(fmaxf(v, -HALF_MAX), HALF_MAX); }

// A step multiplies TWO taps of the 32-channel block (8 NF MFMAs per wave) between one completion fence + barrier and
// the next: with one tap per step the fence, the barrier's skew and the address arithmetic of a step were paid twice
// as often (4 % of the layers).  The weights of a step (two half chunks, 24 KB for 192 output channels) are
// double-buffered: the pair of step s + 1 is issued at the top of step s into the buffer that step s - 1 read.  The
// patch buffers hold exactly the 10 x 96 slots of a patch (15 DMA pieces, the last wave brings 3), which keeps two
// workgroups on a CU: 2 x 15 KB + 2 x 24 KB + 1.5 KB.
constexpr int PATCH2_F4 = PATCH_ROWS * ROW_F4;           // 960 slots = 15 pieces of 1 KB
constexpr int PATCH2_PIECES = PATCH2_F4 / 64;
static_assert(PATCH2_PIECES * 64 == PATCH2_F4 && PATCH2_PIECES > (NW - 1) * PPW && PATCH2_PIECES <= NW * PPW, "patch pieces per wave");

template <int NSTEPS>
__host__ __device__ constexpr int pieces_in_step2(int k) {      // patch pieces a wave issues in step k of a block
    constexpr int per = (PPW + NSTEPS - 1) / NSTEPS;
    const int first = k * per;
    return first >= PPW ? 0 : (first + per > PPW ? PPW - first : per);
}

template <int NF, int EPI>
struct Body {
    static constexpr int COUT = NF * 32;
    static constexpr int WCH_F4 = COUT * 4;              // slots of a half chunk (1 tap x 32 channels): [ks 2][h 2][cout]
    static constexpr int WPW = WCH_F4 / 64 / NW;         // 1 KB pieces of a half chunk that each wave brings
    static_assert(WPW * NW * 64 == WCH_F4, "a half chunk must split evenly into 1 KB pieces per wave");
    static constexpr int NTL = 2 * NF;                   // 16-channel output tiles
    static_assert(NF % 2 == 0, "gamma's half chunks go two per step");

    template <int NTAPS>
    static __device__ __forceinline__ void run(const Params &p, const Phase &ph, f4 *lds) {
        constexpr int NSTEPS = (NTAPS + 1) / 2;          // steps per 32-channel block
        f4 *patch = lds;                                 // 2 x PATCH2_F4
        f4 *wring = lds + 2 * PATCH2_F4;                 // 2 buffers x 2 half chunks
        f4 *par = wring + 4 * WCH_F4;                    // bias [COUT], beta [COUT] (beyond what the transposing store reuses)

        const int tid = threadIdx.x;
        const int lane = tid & 63, m = lane & 15, kg = lane >> 4, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

        const int tiles_x = (p.Wb + TILE_W - 1) / TILE_W, tiles_y = (p.Hb + TILE_H - 1) / TILE_H;
        const int64_t blk = xcd_block();
        if (blk >= (int64_t)p.B * tiles_y * tiles_x) return;     // padding block of the rounded-up grid
        const int b = (int)(blk / (tiles_y * tiles_x));
        const int trem = (int)(blk - (int64_t)b * tiles_y * tiles_x);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        // this lane's two pixels: column m of tile rows 2 wave and 2 wave + 1
        const int by0 = ty * TILE_H + 2 * wave, bx = tx * TILE_W + m;

        // ---- patch staging: slot s of the LDS image <- 16 bytes of the input (or of the zero page) ------------
        const _Float16 *in_b = (const _Float16 *)p.in + (int64_t)b * p.Hi * p.Wi * p.Cs;
        const int iy0 = ty * TILE_H - 1, ix0 = tx * TILE_W - 1;
        int soff[PPW];                                   // element offset inside the image, -1 = zeros
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int s = (wave * PPW + i) * 64 + lane;
            const int r = s / ROW_F4, rem = s - r * ROW_F4;
            const int c = rem / PIX_F4, part = rem - c * PIX_F4;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = r < PATCH_ROWS && c < PATCH_COLS && part < 4 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            soff[i] = ok ? (iy * p.Wi + ix) * p.Cs + part * 8 : -1;
        }
        const _Float16 *zeros = (const _Float16 *)p.zeros;
        auto dma_patch = [&](int i, int cb, int buf) {   // piece i of this wave, channel block cb, into patch buffer buf
            if (wave * PPW + i >= PATCH2_PIECES) return;  // (uniform: the last wave has one piece less)
            const _Float16 *src = soff[i] >= 0 ? in_b + soff[i] + cb * 32 : zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(patch + buf * PATCH2_F4 + (wave * PPW + i) * 64),
                                             16, 0, 0);
        };

        // ---- weights: half chunks (tap, cb) of the packed image, then gamma's, two per step, by LDS-DMA ------------
        const int ncb = p.Cs / 32;                       // 32-channel blocks
        const f4 *wsrc = (const f4 *)((const _Float16 *)p.wpk + ph.w_off);
        const f4 *gsrc = (const f4 *)p.gamma + (int64_t)COUT * COUT / 8;   // (the image in this kernel's k order follows the other one)
        // packed order: chunk (tap, 64-channel block) at (tap * ncb/2 + cb/2) full chunks, its halves 12 KB apart
        auto w_half = [&](int tap, int cb) -> const f4 * {
            return wsrc + ((int64_t)(tap * (ncb >> 1) + (cb >> 1)) * 2 + (cb & 1)) * WCH_F4;
        };
        // A batch of transfers is issued patch pieces first, then the second half chunk, then the first, whose last
        // piece is the one whose arrival is observed (`watched`): its destination is marked (dma_mark) before the batch's
        // first piece goes out, so the marks reach LDS while the other pieces are issued, and the wait for them
        // (dma_marks_stored) in front of the watched piece finds them there.
        auto probe = [&](int buf) { return wring + buf * 2 * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane; };
        auto dma_half = [&](const f4 *src, int buf, int half, bool watched = false) {
#pragma unroll
            for (int q = 0; q < WPW; ++q) {
                const int piece = wave * WPW + q;
                if (watched && q == WPW - 1) dma_marks_stored();
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(wring + (buf * 2 + half) * WCH_F4 + piece * 64), 16, 0, 0);
            }
        };

        // prologue: weights of step 0, patch of block 0; bias and beta go to LDS, where the epilogue finds them
        dma_mark(probe(0));
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_patch(i, 0, 0);
        dma_half(w_half(1, 0), 0, 1);
        dma_half(w_half(0, 0), 0, 0, true);
        if (tid < COUT / 4) par[tid] = ((const f4 *)p.bias)[tid];
        if (EPI == EPI_GDN && tid >= 64 && tid < 64 + COUT / 4) par[COUT / 4 + tid - 64] = ((const f4 *)p.beta)[tid - 64];
        dma_wait_last(probe(0));
        __syncthreads();

        // the accumulators start from the bias: acc[t][u][r] = channel 16 t + 4 kg + r of pixel (row u, column m)
        f32x4 acc[NTL][2];
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[t][0] = acc[t][1] = par[4 * t + kg];

        // the lane's pixel of row 0 in the patch (tap (0,0) is patch row + 1, column + 1), its piece kg of 8 channels
        const int lane_slot = (2 * wave + 1) * ROW_F4 + (m + 1) * PIX_F4 + kg;
        int wbuf = 0;                                    // weight buffer of the current step
        int pbuf = 0;

        auto one_tap = [&](int tap, const f4 *wl) {
            const int2 d = p.tapv[ph.tap_off + tap];     // uniform index: a scalar load from the kernel arguments
            const f4 *bp = patch + pbuf * PATCH2_F4 + lane_slot + d.x * ROW_F4 + d.y * PIX_F4;
            const half8 b0 = as_half8(bp[0]), b1 = as_half8(bp[ROW_F4]);
            half8 a[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) a[t] = as_half8(wl[t * 16]);
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b0, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b1, acc[t][1], 0, 0, 0);
            }
        };

        // one step: taps 2K and 2K + 1 (if the phase has it) of the current 32-channel block
        auto step = [&](auto k_c, int cb, bool more) {
            constexpr int K = decltype(k_c)::value, T0 = 2 * K, T1 = 2 * K + 1;
            // the next block's patch (the last block fetches its own again, into the buffer nobody reads)
            constexpr int first = K * ((PPW + NSTEPS - 1) / NSTEPS);
            dma_mark(probe(wbuf ^ 1));                   // (every wave has passed the barrier behind that buffer's last read)
#pragma unroll
            for (int i = 0; i < pieces_in_step2<NSTEPS>(K); ++i) dma_patch(first + i, more ? cb + 1 : cb, pbuf ^ 1);
            // the next step's weights: in this block, in the next block, or gamma's first two half chunks
            if constexpr (K + 1 < NSTEPS) {
                if constexpr (T0 + 3 < NTAPS) dma_half(w_half(T0 + 3, cb), wbuf ^ 1, 1);
                dma_half(w_half(T0 + 2, cb), wbuf ^ 1, 0, true);
            } else {
                // (the very last step of a layer without GDN has nothing to bring but may carry patch pieces: its first
                // half chunk is fetched once more, as the watched end of the batch)
                const f4 *n0 = more ? w_half(0, cb + 1) : (EPI == EPI_GDN ? gsrc : w_half(0, cb));
                const f4 *n1 = more ? w_half(1, cb + 1) : (EPI == EPI_GDN ? gsrc + WCH_F4 : w_half(1, cb));
                dma_half(n1, wbuf ^ 1, 1);
                dma_half(n0, wbuf ^ 1, 0, true);
            }

            const f4 *wl = wring + wbuf * 2 * WCH_F4 + kg * COUT + m;   // slot [ks][h][cout] with kg = 2 ks + h
            one_tap(T0, wl);
            if constexpr (T1 < NTAPS) one_tap(T1, wl + WCH_F4);
            dma_wait_last(probe(wbuf ^ 1));              // everything this wave has issued is in LDS before the barrier
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            wbuf ^= 1;
        };

        for (int cb = 0; cb < ncb; ++cb) {
            const bool more = cb + 1 < ncb;
            step(std::integral_constant<int, 0>{}, cb, more);
            step(std::integral_constant<int, 1>{}, cb, more);
            if constexpr (NSTEPS > 2) step(std::integral_constant<int, 2>{}, cb, more);
            if constexpr (NSTEPS > 3) {
                step(std::integral_constant<int, 3>{}, cb, more);
                step(std::integral_constant<int, 4>{}, cb, more);
            }
            pbuf ^= 1;
        }

        // ---- epilogue: acc[t][u][r] = channel 16 t + 4 kg + r of pixel (by0 + u, bx) ----------------------------
        const int ox = bx * p.mx + ph.qx;
        if constexpr (EPI == EPI_GDN) {
            const bool inverse = p.act == MPC_ACT_IGDN;
            // norm[c'] = sum_c gamma[c'][c] * y[c]^2 as NF / 2 more steps of two half chunks: half chunk hc holds
            // gamma[:, 32 hc .. 32 hc + 31] in the weight layout, its k-values in the order in which a lane holds the
            // squares: k = 8 kg + e is channel 32 hc + 16 (e >> 2) + 4 kg + (e & 3), registers e & 3 of tiles 2 hc and
            // 2 hc + 1.  y^2 is scaled by 2^-8 and saturated before the conversion to fp16; the scale is undone on the sum.
            f32x4 nrm[NTL][2];                                         // starts from beta (in the scale of the sum: 2^-8)
#pragma unroll
            for (int t = 0; t < NTL; ++t) nrm[t][0] = nrm[t][1] = par[COUT / 4 + 4 * t + kg] * SQ_SCALE;
            auto square = [&](int hc, int u) {
                half8 bv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = acc[2 * hc + (e >> 2)][u][e & 3];
                    bv[e] = (_Float16)fminf(y * y * SQ_SCALE, HALF_MAX);
                }
                return bv;
            };
            half8 bv0 = square(0, 0), bv1 = square(0, 1);
#pragma unroll
            for (int g2 = 0; g2 < NF / 2; ++g2) {
                if (g2 + 1 < NF / 2) {
                    dma_mark(probe(wbuf ^ 1));
                    dma_half(gsrc + (int64_t)(2 * g2 + 3) * WCH_F4, wbuf ^ 1, 1);
                    dma_half(gsrc + (int64_t)(2 * g2 + 2) * WCH_F4, wbuf ^ 1, 0, true);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int hc = 2 * g2 + e;
                    const f4 *wl = wring + (wbuf * 2 + e) * WCH_F4 + kg * COUT + m;
                    // the next half chunk's operands are squared and converted while this one's MFMAs run
                    half8 bn0 = bv0, bn1 = bv1;
                    if (hc + 1 < NF) {
                        bn0 = square(hc + 1, 0);
                        bn1 = square(hc + 1, 1);
                    }
#pragma unroll
                    for (int t = 0; t < NTL; ++t) {
                        const half8 aa = as_half8(wl[t * 16]);
                        nrm[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, bv0, nrm[t][0], 0, 0, 0);
                        nrm[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, bv1, nrm[t][1], 0, 0, 0);
                    }
                    bv0 = bn0;
                    bv1 = bn1;
                }
                if (g2 + 1 < NF / 2) dma_wait_last(probe(wbuf ^ 1));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                wbuf ^= 1;
            }
            __syncthreads();                              // (every transfer was fenced above: the LDS can be reused)
            // y * (beta + norm)^(-1/2) (or ^(+1/2)) through the wave-private transposing store (2 bytes per element:
            // all output channels in one round); v_rsq_f32 is far more accurate than the fp16 output.  Row 16 u + m of
            // the store is pixel (by0 + u, bx); the lanes kg = 0, 1 enter the output index of rows m and 16 + m.
            TransStore ts;
            {
                const int u = kg & 1, by = by0 + u;
                const bool pvalid = by < p.Hb && bx < p.Wb;
                ts.init((char *)lds, wave, lane, 16 * u + m, kg >> 1,
                        pvalid ? (long long)(((int64_t)b * p.Ho + by * p.my + ph.qy) * p.Wo + ox) : -1ll);
            }
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    half4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float a = nrm[t][u][r] * SQ_UNSCALE;
                        float rs = __builtin_amdgcn_rsqf(a);
                        float y = acc[t][u][r];
                        o[r] = sat_half(inverse ? y * (a * rs) : y * rs);
                    }
                    *(half4 *)ts.at(16 * u + m, (16 * t + 4 * kg) * 2) = o;
                }
            ts.template flush<NF * 4>((char *)p.out, (long long)COUT * 2, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                 // (every step has waited for its own transfers)
                const int by = by0 + u;
                if (by >= p.Hb || bx >= p.Wb) continue;
                _Float16 *orow = (_Float16 *)p.out + (((int64_t)b * p.Ho + by * p.my + ph.qy) * p.Wo + ox) * COUT;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    half4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[t][u][r];
                        if (p.act == MPC_ACT_RELU) v = fmaxf(v, 0.0f);
                        o[r] = sat_half(v);
                    }
                    *(half4 *)(orow + 16 * t + 4 * kg) = o;
                }
            }
        }
    }
};


template <int NF, int EPI>
__global__ __launch_bounds__(NT, 2) void deconv5s2_patch_f16_kernel(const Params p) {
    extern __shared__ f4 lds[];
    const Phase &ph = p.ph[blockIdx.y];
    switch (ph.ntaps) {                                  // (uniform: the phase is the workgroup's)
        case 9: Body<NF, EPI>::template run<9>(p, ph, lds); break;
        case 6: Body<NF, EPI>::template run<6>(p, ph, lds); break;
        default: Body<NF, EPI>::template run<4>(p, ph, lds); break;
    }
}

template <int NF, int EPI>
int launch(mpc_ctx *ctx, const Params &p, int nphase) {
    size_t lds = ((size_t)2 * PATCH2_F4 + 4 * (size_t)NF * 32 * 4 + 2 * NF * 8) * sizeof(f4);
    if (EPI == EPI_GDN && lds < (size_t)NW * TS_WAVE_BYTES) lds = (size_t)NW * TS_WAVE_BYTES;
    lds = patch_kernel_lds(lds);
    auto kern = deconv5s2_patch_f16_kernel<NF, EPI>;
    static bool attr_set[MPC_MAX_DEVICES] = {};
    bool &attr_done = attr_set[ctx->device];
    if (!attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int64_t tiles = (int64_t)p.B * mpc_cdiv(p.Hb, TILE_H) * mpc_cdiv(p.Wb, TILE_W);
    const int64_t bx = xcd_grid(tiles);
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many tiles");
#ifdef MPC_STAMPS
    static StampHost stamps;
    stamps.before((size_t)bx * nphase);
#endif
    kern<<<dim3((unsigned)bx, (unsigned)nphase, 1), dim3(NT), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "deconv5s2_patch_f16");
#ifdef MPC_STAMPS
    stamps.after(ctx->stream, (size_t)bx, 4000);
#endif
    return MPC_OK;
}

}  // namespace

// True when the layer / call can run on the patch kernel: a 5x5 stride-2 transposed convolution on the fp16 path with
// four parity phases of 9 / 6 / 6 / 4 taps, input channels a multiple of 64, 128 or 192 output channels.
bool mpc_deconv_patch_f16_supported(const Params &p, const mpc_conv *c) {
    const mpc_conv_desc &d = c->desc;
    if (!c->f16 || c->stacked || d.kind != MPC_DECONV || d.ksize != 5 || d.stride != 2 || d.pad != 2) return false;
    if (d.cin % 64 != 0 || (c->nf != 4 && c->nf != 6) || c->groups != 1 || c->nphase != 4) return false;
    for (int i = 0; i < c->nphase; ++i) {
        const int n = c->ph[i].ntaps;
        if (n != 9 && n != 6 && n != 4) return false;
    }
    for (int t = 0; t < c->ntaps_total; ++t)
        if (abs(c->taps_host[t].x) > 1 || abs(c->taps_host[t].y) > 1) return false;
    const bool gdn = d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN;
    if (!gdn && d.act != MPC_ACT_NONE && d.act != MPC_ACT_RELU) return false;
    if ((int64_t)p.Hi * p.Wi * p.Cs >= (1ll << 31)) return false;        // 32-bit element offsets inside one image
    return true;
}

int mpc_deconv_launch_patch_f16(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    const bool gdn = c->desc.act == MPC_ACT_GDN || c->desc.act == MPC_ACT_IGDN;
    if (gdn && c->nf == 6) return launch<6, EPI_GDN>(ctx, p, c->nphase);
    if (gdn && c->nf == 4) return launch<4, EPI_GDN>(ctx, p, c->nphase);
    if (!gdn && c->nf == 6) return launch<6, EPI_PLAIN>(ctx, p, c->nphase);
    if (!gdn && c->nf == 4) return launch<4, EPI_PLAIN>(ctx, p, c->nphase);
    MPC_FAIL(ctx, MPC_EINVAL, "conv_forward (fp16 patch kernel): unsupported layer shape (cout/32 = %d)", c->nf);
}
