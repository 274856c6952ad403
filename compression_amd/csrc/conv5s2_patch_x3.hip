// Split-precision (f16x3) 5x5 stride-2 convolution with the input staged through LDS as a 2-D patch: the kernel of
// the two big layers of the analysis transform (conv2, conv3: 192 -> 192 with fused GDN) and of conv4.
//
// Why: a 5x5 stride-2 layer reads every input pixel 25/4 times, once per tap.  With activations fetched global ->
// registers per tap (conv_igemm_x3.hip) those re-reads miss the 4 MB L2 of an XCD (5x the compulsory bytes went over
// the fabric) and their latency sits on the K loop.  Here a workgroup owns a 16 x TR tile of output pixels and keeps
// the (2 TR + 3) x 35 input pixels it needs, for 16 input channels at a time, in LDS: the patch is fetched once per
// 16-channel block (1.2x the compulsory bytes, the halo) and all 25 taps read it from LDS.
//
// K walk: for each 16-channel block (one k-step of v_mfma_f32_32x32x16_f16), the 25 taps.  Per tap a wave issues
// 3 MFMAs (hi*lo, lo*hi, hi*hi) for each of its NF output tiles; weights come through a double-buffered LDS chunk
// (1 tap x 16 channels, the packed image of conv_igemm_x3.hip read in halves), one barrier per tap.  The patch of the
// NEXT block is fetched into registers during the first taps of the current block (one 16-byte load per tap and
// thread) and written to LDS between the blocks.
//
// Patch image in LDS (16-byte slots): [row][column parity 2][piece 4][x2 18], piece = (k half h, hi|lo).  A wave's 32
// pixels are two tile rows of 16; for one tap the 16 lanes of a row read 16 consecutive slots (input columns two
// apart share a parity), and two patch rows are 2 x 2304 bytes = a multiple of the 256-byte bank row apart, so the
// 16-lane groups of ds_read_b128 are conflict free.
//
// Same operand formats, scales, GDN continuation and epilogues as conv_igemm_x3.hip; only the summation order of
// the K axis differs (channel block major), so results agree with it to fp32 rounding, not bit for bit.
#include "conv_x3_epilogue.h"

namespace {

constexpr int TILE_W = 16;
constexpr int PATCH_COLS = 2 * TILE_W + 3;          // 35
constexpr int X2 = 18;                              // columns per parity (35 = 18 even + 17 odd, one slot of padding)
constexpr int ROW_F4 = 2 * 4 * X2;                  // 144 slots = 2304 bytes per patch row

// 16 bytes as a native vector: arrays of HIP's float4 struct that are written under masks stay in scratch memory
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 as_half8(const f4 &v) { return __builtin_bit_cast(half8, v); }


template <int NF, int EPI, int TR>
__global__ __launch_bounds__(32 * TR, 2) void conv5s2_patch_x3_kernel(const Params p) {
    constexpr int NW = TR / 2, NT = 64 * NW, COUT = NF * 32;
    constexpr int PR = 2 * TR + 3;                       // patch rows
    constexpr int PATCH_F4 = PR * ROW_F4;
    constexpr int NS = (PATCH_F4 + NT - 1) / NT;         // patch slots staged per thread (one load per tap)
    constexpr int WCH_F4 = COUT * 4;                     // slots of a weight chunk (1 tap x 16 channels): [h][hi|lo][cout]
    constexpr int WPW = WCH_F4 / 64 / NW;                // 1 KB pieces of a chunk that each wave brings
    static_assert(NS <= 24, "one staged patch slot per tap");
    static_assert(WPW * NW * 64 == WCH_F4, "a weight chunk must split evenly into 1 KB pieces per wave");
    extern __shared__ f4 lds[];
    f4 *patch = lds;
    f4 *wring = lds + PATCH_F4;                          // 3 x WCH_F4

    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int cg = blockIdx.z * COUT;

    const int tiles_x = (p.Wo + TILE_W - 1) / TILE_W, tiles_y = (p.Ho + TR - 1) / TR;
    const int64_t blk = xcd_block();
    if (blk >= (int64_t)p.B * tiles_y * tiles_x) return;     // padding block of the rounded-up grid
    const int b = (int)(blk / (tiles_y * tiles_x));
    const int trem = (int)(blk - (int64_t)b * tiles_y * tiles_x);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy = ty * TR + 2 * wave + (j >> 4), ox = tx * TILE_W + (j & 15);
    const bool pvalid = oy < p.Ho && ox < p.Wo;
    const int iy0 = 2 * ty * TR - 2, ix0 = 2 * tx * TILE_W - 2;

    // ---- patch staging: slot s = tid + i*NT of the LDS image <- 16 bytes of the S16 input -------------------
    const _Float16 *in_b = (const _Float16 *)p.in + (int64_t)b * p.Hi * p.Wi * p.Cs * 2;
    int soff[NS];                                         // element (half) offset inside the image, -1 = zero
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = tid + i * NT;
        const int r = s / ROW_F4, rem = s - r * ROW_F4;
        const int par = rem / (4 * X2), rem2 = rem - par * (4 * X2);
        const int q = rem2 / X2, x2 = rem2 - q * X2;
        const int px = 2 * x2 + par, iy = iy0 + r, ix = ix0 + px;
        const bool ok = s < PATCH_F4 && px < PATCH_COLS && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        soff[i] = ok ? (iy * p.Wi + ix) * p.Cs * 2 + q * 8 : -1;
    }
    f4 stage[NS];
    // every load is unconditional (a slot outside the image loads the image's first bytes and is zeroed when it is
    // written to LDS): no branch around a load, no early wait
    auto load_slot = [&](int i, int cb) { stage[i] = *(const f4 *)(in_b + (soff[i] >= 0 ? soff[i] : 0) + cb * 32); };
    auto write_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (tid + i * NT < PATCH_F4) patch[tid + i * NT] = soff[i] >= 0 ? stage[i] : f4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- weights: chunk (tap, cb) of the packed image, through a ring of three chunks in LDS ----------------------
    // In the K loop a chunk travels global -> registers -> LDS: chunk c + 2 is loaded at the top of tap c and written
    // to the ring early in tap c + 1 (behind the MFMAs of its second output tile), so it has a tap and more to arrive,
    // the only waits on the loads are the compiler's counted ones on its registers, and the write is complete in LDS
    // long before the tap's barrier waits for it (barrier_after_lds_writes: a wave may not arrive at the barrier
    // with the write still queued -- the next tap reads the chunk).  LDS-DMA is not used there: the wait for its
    // pieces (vmcnt(0), then dma_wait_last) also waits for the patch loads of the next block, which are meant to stay
    // in flight for many taps (that wait cost 6 % of the layer).  The GDN steps have no other load in flight and keep
    // LDS-DMA with an observed wait per step.
#ifndef MPC_X3_STAGED_W
#define MPC_X3_STAGED_W 1
#endif
#ifndef MPC_X3_W_AT
#define MPC_X3_W_AT 1                                     // output tile after whose MFMAs a tap writes the ring
#endif
    const int ncb = p.Cs / 16;                            // 16-channel blocks
    const f4 *wsrc = (const f4 *)((const _Float16 *)p.wpk + (int64_t)blockIdx.z * p.w_group);
    const bool gdn_mix16 = (EPI == EPI_GDN) && p.y2_scale > 0.0f;
    const f4 *gsrc = (const f4 *)p.gamma_x3;
    auto dma_chunk = [&](const f4 *chunk_src, int slot) {
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int piece = wave * WPW + q;
            if (q == 0) dma_mark(wring + slot * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane);   // the last piece's arrival is observed:
            if (q == WPW - 1) dma_marks_stored();                                               // its marks are in LDS before it goes out
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(chunk_src + piece * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(wring + slot * WCH_F4 + piece * 64), 16, 0, 0);
        }
    };
    const int wofs = wave * WPW * 64 + lane;              // this thread's first slot of a chunk; the others 64 apart
    f4 wo[WPW];                                           // the chunk after the current one, on its way to the ring

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    // prologue: weights of (tap 0, block 0) and (tap 1, block 0), patch of block 0
#if MPC_X3_STAGED_W
#pragma unroll
    for (int q = 0; q < WPW; ++q) wring[wofs + q * 64] = wsrc[wofs + q * 64];
#pragma unroll
    for (int q = 0; q < WPW; ++q) wo[q] = wsrc[(int64_t)ncb * WCH_F4 + wofs + q * 64];
#else
    dma_chunk(wsrc, 0);
    dma_chunk(wsrc + (int64_t)ncb * WCH_F4, 1);
#endif
#pragma unroll
    for (int i = 0; i < NS; ++i) load_slot(i, 0);
    write_patch();
#if !MPC_X3_STAGED_W
    MPC_WAIT_VMCNT(0);
#endif
    __syncthreads();

    // this lane's pixel in the patch: row 2*(row in tile), piece 2h (hi) / 2h+1 (lo), column slot (j & 15)
    const f4 *bbase = patch + (2 * (2 * wave + (j >> 4))) * ROW_F4 + (2 * h) * X2 + (j & 15);
    int slot = 0;                                         // ring slot of the current chunk; the chunk after next goes to slot + 2
    // One tap of the current 16-channel block, expanded 25 times with a literal tap number: the staging registers are
    // addressed by constants.
#define MPC_PATCH_TAP(TAP)                                                                                             \
    {                                                                                                                  \
        constexpr int tap = (TAP), dyp = tap / 5, dxp = tap % 5;                                                      \
        /* chunk after next: (tap + 2, cb), or in the next block, or the first half chunks of gamma */                \
        const f4 *nsrc = (tap < 23) ? wsrc + ((int64_t)(tap + 2) * ncb + cb) * WCH_F4                                  \
                         : more      ? wsrc + ((int64_t)(tap - 23) * ncb + cb + 1) * WCH_F4                            \
                         : gdn_mix16 ? gsrc + (int64_t)(tap - 23) * WCH_F4                                             \
                                     : wsrc;                                                                           \
        MPC_X3_W_TOP(nsrc)                                                                                             \
        if constexpr (tap < NS) load_slot(tap, more ? cb + 1 : cb); /* (the last block re-reads its own: unused) */   \
        __builtin_amdgcn_sched_barrier(0);                  /* the loads stay at the top of the tap */                 \
        const f4 *bp = bbase + dyp * ROW_F4 + (dxp & 1) * (4 * X2) + (dxp >> 1);                                       \
        const half8 bh = as_half8(bp[0]), bl = as_half8(bp[X2]);                                                       \
        const f4 *wl = wring + slot * WCH_F4 + (h * 2) * COUT + j;                                                     \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                               \
            const half8 ah = as_half8(wl[f * 32]);                                                                     \
            const half8 al = as_half8(wl[COUT + f * 32]);                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[f], 0, 0, 0);                                  \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[f], 0, 0, 0);                                  \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[f], 0, 0, 0);                                  \
            if (f == MPC_X3_W_AT) { MPC_X3_W_END() }        /* early in the tap: the write completes under the MFMAs */ \
        }                                                                                                              \
        barrier_after_lds_writes();                         /* the chunk just written is read in the next tap */        \
        slot = slot == 2 ? 0 : slot + 1;                                                                               \
    }
#if MPC_X3_STAGED_W
#define MPC_X3_W_TOP(SRC)                                                                                              \
        f4 wn[WPW];                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < WPW; ++q) wn[q] = (SRC)[wofs + q * 64];
#define MPC_X3_W_END()                                                                                                 \
        {                                                                                                              \
            f4 *wdst = wring + (slot == 2 ? 0 : slot + 1) * WCH_F4 + wofs;                                             \
            _Pragma("unroll") for (int q = 0; q < WPW; ++q) { wdst[q * 64] = wo[q]; wo[q] = wn[q]; }                   \
        }
#else
#define MPC_X3_W_TOP(SRC) dma_chunk(SRC, slot >= 1 ? slot - 1 : 2);
#define MPC_X3_W_END() MPC_WAIT_VMCNT(0);
#endif
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more = cb + 1 < ncb;
        MPC_PATCH_TAP(0) MPC_PATCH_TAP(1) MPC_PATCH_TAP(2) MPC_PATCH_TAP(3) MPC_PATCH_TAP(4)
        MPC_PATCH_TAP(5) MPC_PATCH_TAP(6) MPC_PATCH_TAP(7) MPC_PATCH_TAP(8) MPC_PATCH_TAP(9)
        MPC_PATCH_TAP(10) MPC_PATCH_TAP(11) MPC_PATCH_TAP(12) MPC_PATCH_TAP(13) MPC_PATCH_TAP(14)
        MPC_PATCH_TAP(15) MPC_PATCH_TAP(16) MPC_PATCH_TAP(17) MPC_PATCH_TAP(18) MPC_PATCH_TAP(19)
        MPC_PATCH_TAP(20) MPC_PATCH_TAP(21) MPC_PATCH_TAP(22) MPC_PATCH_TAP(23) MPC_PATCH_TAP(24)
        if (more) {                                       // every wave has read the old patch (barrier above)
            write_patch();                                // (the staged registers' wait also retires the DMA issued before them)
            barrier_after_lds_writes();
        }
    }
#undef MPC_PATCH_TAP
#undef MPC_X3_W_TOP
#undef MPC_X3_W_END
#if MPC_X3_STAGED_W
    if (gdn_mix16) {                                      // the second half chunk of gamma, loaded at the last tap
#pragma unroll
        for (int q = 0; q < WPW; ++q) wring[(slot == 2 ? 0 : slot + 1) * WCH_F4 + wofs + q * 64] = wo[q];
    }
#endif

    // ---- epilogue -------------------------------------------------------------------------------------------
    x3_unscale_bias<NF>(p, acc, cg, h);
    const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
    if constexpr (EPI == EPI_GDN) {
        // norm[c'] = sum_c gamma[c'][c] * y[c]^2 as 2 NF more k-steps: half chunk (kc, s) holds gamma[:, 32kc+16s ..+15]
        // in the weight-chunk layout; the B operand is built from the lane's own accumulators, whose registers
        // 8s..8s+7 of tile kc are the 8 k-values of that k-step (c = 32kc + 16s + 8(jj>>2) + 4h + (jj&3)).
        f32x16 nrm[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[f][r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < NF; ++kc) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int hc = 2 * kc + s;
                dma_chunk(gsrc + (int64_t)(hc + 2 < 2 * NF ? hc + 2 : 0) * WCH_F4, slot >= 1 ? slot - 1 : 2);
                const f4 *wl = wring + slot * WCH_F4 + (h * 2) * COUT + j;
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
                    const half8 ah = as_half8(wl[fo * 32]);
                    const half8 al = as_half8(wl[COUT + fo * 32]);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, nrm[fo], 0, 0, 0);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, nrm[fo], 0, 0, 0);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, nrm[fo], 0, 0, 0);
                }
                dma_wait_last(wring + (slot >= 1 ? slot - 1 : 2) * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                slot = slot == 2 ? 0 : slot + 1;
            }
        }
        lds_dma_fence(wring + (wave * WPW + WPW - 1) * 64 + lane);                                // the surplus DMA pieces have landed before the LDS is reused
        __syncthreads();
        x3_gdn_store<NF>(p, acc, nrm, (float4 *)lds, wave, lane, j, h, pvalid ? (long long)opix : -1ll);
    } else {
        lds_dma_fence(wring + (wave * WPW + WPW - 1) * 64 + lane);                                // no DMA is left in flight when the workgroup ends
        if (!pvalid) return;
        x3_plain_store<NF>(p, acc, cg, h, opix);
    }
}

template <int NF, int EPI, int TR>
int launch_tr(mpc_ctx *ctx, const Params &p, int groups) {
    constexpr int NW = TR / 2;
    size_t lds = ((size_t)(2 * TR + 3) * ROW_F4 + 3 * (size_t)NF * 32 * 4) * sizeof(float4);
    if (EPI == EPI_GDN && lds < (size_t)NW * TS_WAVE_BYTES) lds = (size_t)NW * TS_WAVE_BYTES;
    lds = patch_kernel_lds(lds);
    auto kern = conv5s2_patch_x3_kernel<NF, EPI, TR>;
    static bool attr_set[MPC_MAX_DEVICES] = {};
    bool &attr_done = attr_set[ctx->device];
    if (!attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int64_t tiles = (int64_t)p.B * mpc_cdiv(p.Ho, TR) * mpc_cdiv(p.Wo, TILE_W);
    const int64_t bx = xcd_grid(tiles);
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many tiles");
    kern<<<dim3((unsigned)bx, 1, (unsigned)groups), dim3(64 * NW), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "conv5s2_patch_x3");
    return MPC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// First layer (3 -> COUT, 5x5 stride 2, fused GDN) in the same style.  Its K axis is short (k = tap*4 + c over the
// image padded to 4 channels: 25 taps -> 8 k-steps, then the GDN mix), so a workgroup lives for only 8 + 2 NF steps and
// what matters is that nothing in them waits: the whole 35 x 19 image patch of the 16 x 8 output tile is split into
// fp16 hi/lo ONCE, when it is staged ([hi x4 | lo x4] = one 16-byte slot per pixel, column parities apart so that the
// 16 lanes of a tile row read consecutive slots), the weights and gamma stream through the LDS-DMA ring two steps
// ahead of their use (a chunk is waited for one step after it was issued, never in the step that issued it), and two
// workgroups share a CU so that one's prologue and 768-byte-per-pixel stores overlap the other's MFMAs.  A lane's
// 8 k-values of a k-step are two whole pixels: taps 4S + 2h and 4S + 2h + 1.
//
// The GDN mix runs in TWO PASSES over the channel axis, each for one half of the output channels (NF/2 tiles): a pass
// is NF steps of two k-steps x NF/2 output tiles (the same 3 NF MFMAs and the same 12 KB of gamma per step as one
// k-step x NF tiles would be), after which that half is normalised and stored.  So the wave holds NF/2 norm tiles
// instead of NF: 6 x 16 accumulators + 3 x 16 norms + operands fit 208 registers instead of 248.  For one output tile
// the MFMAs are the ones of the single-pass order, in that order: same bits.  The squares (B operands) are formed once
// per pass.  A pass's chunk is gathered from the k-step-major gamma image by the LDS-DMA's per-lane source addresses.
//
// LDS: [patch | ring slot 0 | slot 1 | slot 2 | beta].  The transposing store of a pass (one tile = 128 bytes per pixel
// at a time, 4 waves x 4 864 bytes) lies over the patch and ring slot 0: the patch is dead after the tap steps, and
// the ring is started so that the last step of the FIRST pass reads slot 0 -- the slot that no transfer targets until
// the next step issues one, which a barrier behind the store holds back.
constexpr int RGB_X2 = 18;                          // slots per column parity
constexpr int RGB_ROW = 2 * RGB_X2;                 // 36 slots per patch row
constexpr int RGB_TR = 8, RGB_PR = 2 * RGB_TR + 3;  // 8 tile rows, 19 patch rows
constexpr int RGB_PATCH = 704;                      // 19 x 36 = 684 slots, rounded up
constexpr int TS1_ROW_BYTES = 128;                  // one 32-channel tile of a pixel (fp32 or the S16 format: 4 bytes per element)
constexpr int TS1_ROW_PAD = 144;                    // LDS row pitch (odd multiple of 16 bytes: conflict-free b128)
constexpr int TS1_WAVE_BYTES = 32 * TS1_ROW_PAD + 32 * 8;

__host__ __device__ constexpr int rgb_tap_slot(int tap) {       // slot offset of a tap inside the patch (taps past 24: weights are zero)
    const int t = tap > 24 ? 24 : tap;
    return (t / 5) * RGB_ROW + ((t % 5) & 1) * RGB_X2 + ((t % 5) >> 1);
}

// Register budget: 208 (the attribute counts the unified 512-entry file in halves); without the cap the compiler spends
// the whole 256 that two workgroups per CU would allow on hoisted loads.
template <int NF>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(104))) void conv1_rgb_patch_x3_kernel(const Params p) {
    constexpr int NW = 4, NT = 64 * NW, COUT = NF * 32;
    constexpr int HF = NF / 2, HOUT = HF * 32;           // output tiles / channels of one GDN pass
    constexpr int WCH_F4 = COUT * 4;                     // slots of a chunk: one k-step [h][hi|lo][cout], or two k-steps x half the outputs
    constexpr int WPW = WCH_F4 / 64 / NW;
    constexpr int KSTEPS = 8;                            // 25 taps x 4 channels = 100 -> 128 = 8 x 16
    constexpr int NSTEPS = KSTEPS + 2 * NF;              // tap steps, then two GDN passes of NF steps
    constexpr int SLOT0 = (3 - (KSTEPS + NF - 1) % 3) % 3;   // ring slot of step 0, such that the first pass ends on slot 0
    static_assert(NF % 2 == 0, "the GDN mix is split into two halves of the output tiles");
    static_assert(WPW * NW * 64 == WCH_F4, "a chunk must split evenly into 1 KB pieces per wave");
    static_assert(NW * TS1_WAVE_BYTES <= (RGB_PATCH + WCH_F4) * 16, "the transposing store must fit in patch + ring slot 0");
    extern __shared__ f4 lds[];
    f4 *patch = lds;
    f4 *wring = lds + RGB_PATCH;                         // 3 x WCH_F4
    float *beta_lds = (float *)(wring + 3 * WCH_F4);     // COUT floats: read per output tile in the epilogues

    const int tid = threadIdx.x;
    const int lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    MPC_STAMP_DECL(blockIdx.x)
    MPC_STAMP(0)
    const int tiles_x = (p.Wo + TILE_W - 1) / TILE_W, tiles_y = (p.Ho + RGB_TR - 1) / RGB_TR;
    const int64_t blk = xcd_block();
    if (blk >= (int64_t)p.B * tiles_y * tiles_x) return;     // padding block of the rounded-up grid
    const int b = (int)(blk / (tiles_y * tiles_x));
    const int trem = (int)(blk - (int64_t)b * tiles_y * tiles_x);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy = ty * RGB_TR + 2 * wave + (j >> 4), ox = tx * TILE_W + (j & 15);
    const bool pvalid = oy < p.Ho && ox < p.Wo;

    const f4 *wsrc = (const f4 *)p.wpk;
    const f4 *gsrc = (const f4 *)p.gamma_x3;
    // Source slot of this lane for each of its pieces.  Weight chunk S: the chunk's own slots, in order.  GDN chunk
    // (pass, kk): LDS slot l = [k-step 2kk + l / (WCH/2)][section (h, hi|lo) = l % (WCH/2) / HOUT][output l % HOUT of the
    // pass's half] <- slot [section][HOUT * pass + output] of that k-step's chunk in the gamma image.
    int goff[WPW];
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int l = (wave * WPW + q) * 64 + lane;
        const int ks = l / (WCH_F4 / 2), rem = l - ks * (WCH_F4 / 2);
        const int sec = rem / HOUT, r = rem - sec * HOUT;
        goff[q] = ks * WCH_F4 + sec * COUT + r;
    }
    auto dma_chunk = [&](int c, int slot) {               // chunk of step c (no step past the last: nothing is issued)
        if (c >= NSTEPS) return;
        const bool gdn = c >= KSTEPS;
        const int g = c - KSTEPS, pass = g >= NF ? 1 : 0, kk = g - pass * NF;
        const f4 *base = gdn ? gsrc + (int64_t)(2 * kk) * WCH_F4 + pass * HOUT : wsrc + (int64_t)c * WCH_F4;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const int piece = wave * WPW + q;
            if (q == 0) dma_mark(wring + slot * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane);   // the last piece's arrival is observed:
            if (q == WPW - 1) dma_marks_stored();                                               // its marks are in LDS before it goes out
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (gdn ? goff[q] : piece * 64 + lane)),
                                             (__attribute__((address_space(3))) void *)(wring + slot * WCH_F4 + piece * 64), 16, 0, 0);
        }
    };
    auto next_slot = [](int s) { return s == 2 ? 0 : s + 1; };
    auto prev_slot = [](int s) { return s == 0 ? 2 : s - 1; };

    dma_chunk(0, SLOT0);
    dma_chunk(1, next_slot(SLOT0));
    if (tid < COUT) beta_lds[tid] = p.beta[tid];
    // the image patch: fp32 pixel (4 channels) -> [hi x4 | lo x4] of value * 2^8
    {
        const f4 *img = (const f4 *)p.in + (int64_t)b * p.Hi * p.Wi;
        const int iy0 = 2 * ty * RGB_TR - 2, ix0 = 2 * tx * TILE_W - 2;
#pragma unroll
        for (int i = 0; i < (RGB_PR * RGB_ROW + NT - 1) / NT; ++i) {
            const int s = tid + i * NT;
            const int r = s / RGB_ROW, rem = s - r * RGB_ROW;
            const int par = rem / RGB_X2, x2 = rem - par * RGB_X2;
            const int px = 2 * x2 + par, iy = iy0 + r, ix = ix0 + px;
            const bool ok = s < RGB_PR * RGB_ROW && px < PATCH_COLS && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = img[(int64_t)iy * p.Wi + ix];
            half4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sv = v[c] * X3_ACT_SCALE;
                const _Float16 hh = (_Float16)sv;
                hi[c] = hh;
                lo[c] = (_Float16)(sv - (float)hh);
            }
            struct { half4 a, b; } two = {hi, lo};
            if (s < RGB_PR * RGB_ROW) patch[s] = __builtin_bit_cast(f4, two);
        }
    }
    dma_wait_last(wring + next_slot(SLOT0) * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane);   // (the second chunk was issued last)
    __syncthreads();
    MPC_STAMP(1)
    MPC_STAMP_KBEGIN()

    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    const f4 *bbase = patch + (2 * (2 * wave + (j >> 4))) * RGB_ROW + (j & 15);
    int slot = SLOT0;
    // End of step t: the chunk of step t + 1 has landed (observed) -- it was issued a whole step earlier, at the top of
    // step t - 1, so its latency lies behind a step's MFMAs; the chunk issued in THIS step (t + 2) stays in flight
    // across the barrier -- and every wave has read the chunk of the step.
    auto step_end = [&](int t) {
        if (t + 1 < NSTEPS) {
            void *watched = wring + next_slot(slot) * WCH_F4 + (wave * WPW + WPW - 1) * 64 + lane;
            if (t + 2 < NSTEPS) dma_wait_last<WPW>(watched); else dma_wait_last<0>(watched);
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = next_slot(slot);
    };
#define MPC_RGB_STEP(S)                                                                                                \
    {                                                                                                                  \
        dma_chunk((S) + 2, prev_slot(slot));                                                                           \
        const f4 s0 = bbase[h ? rgb_tap_slot(4 * (S) + 2) : rgb_tap_slot(4 * (S))];                                     \
        const f4 s1 = bbase[h ? rgb_tap_slot(4 * (S) + 3) : rgb_tap_slot(4 * (S) + 1)];                                 \
        const half8 bh = as_half8(f4{s0[0], s0[1], s1[0], s1[1]}), bl = as_half8(f4{s0[2], s0[3], s1[2], s1[3]});       \
        const f4 *wl = wring + slot * WCH_F4 + (h * 2) * COUT + j;                                                     \
        _Pragma("unroll") for (int f = 0; f < NF; ++f) {                                                               \
            const half8 ah = as_half8(wl[f * 32]);                                                                     \
            const half8 al = as_half8(wl[COUT + f * 32]);                                                              \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[f], 0, 0, 0);                                  \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[f], 0, 0, 0);                                  \
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[f], 0, 0, 0);                                  \
        }                                                                                                              \
        step_end(S);                                                                                                   \
    }
    MPC_RGB_STEP(0) MPC_RGB_STEP(1) MPC_RGB_STEP(2) MPC_RGB_STEP(3)
    MPC_RGB_STEP(4) MPC_RGB_STEP(5) MPC_RGB_STEP(6) MPC_RGB_STEP(7)
#undef MPC_RGB_STEP

    MPC_STAMP(2)
    MPC_STAMP_KEND()
    // ---- epilogue: scales, bias, the GDN mix in two passes, each followed by its half's normalisation and stores ----
    x3_unscale_bias<NF>(p, acc, 0, h);
    const int64_t opix = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
    const bool out_split = (p.flags & MPC_OUT_SPLIT) != 0;
    // wave-private transposing store, one 32-channel tile of the wave's 32 pixels at a time (see conv_common.h: a lane
    // owns one pixel and 4 channels per register quad; the rows are flushed with 16-byte stores, 8 lanes per pixel)
    char *ts_rows = (char *)lds + wave * TS1_WAVE_BYTES;
    long long *ts_opix = (long long *)(ts_rows + 32 * TS1_ROW_PAD);
    // the B operand of k-step (kc, s2): the lane's accumulators 8 s2 .. 8 s2 + 7 of tile kc, squared, scaled and split;
    // built one k-step ahead, so that the conversions run beside the previous k-step's MFMAs instead of in front of
    // this one's
    // (second_pass: the value goes through an empty asm statement, so that the compiler does not recognise the squares
    // of the first pass in it -- it would keep all 12 k-steps' operands, 96 registers, alive across the passes)
    auto square = [&](int kc, int s2, half8 &bh, half8 &bl, bool second_pass) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            float y = acc[kc][8 * s2 + jj];
            if (second_pass) asm volatile("" : "+v"(y));
            float q = fminf(y * y * p.y2_scale, 65504.0f);
            _Float16 hh = (_Float16)q;
            bh[jj] = hh;
            bl[jj] = (_Float16)(q - (float)hh);
        }
    };
    half8 bh, bl;
    square(0, 0, bh, bl, false);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        f32x16 nrm[HF];
#pragma unroll
        for (int f = 0; f < HF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[f][r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < NF; ++kk) {
            dma_chunk(KSTEPS + pass * NF + kk + 2, prev_slot(slot));
            const f4 *wl = wring + slot * WCH_F4 + (h * 2) * HOUT + j;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int hn = 2 * kk + s2 + 1;              // the k-step after this one (the second pass starts over)
                half8 nh = bh, nl = bl;
                if (hn < 2 * NF) square(hn >> 1, hn & 1, nh, nl, pass == 1);
                else if (pass == 0) square(0, 0, nh, nl, true);
#pragma unroll
                for (int fo = 0; fo < HF; ++fo) {
                    const half8 ah = as_half8(wl[s2 * (WCH_F4 / 2) + fo * 32]);
                    const half8 al = as_half8(wl[s2 * (WCH_F4 / 2) + HOUT + fo * 32]);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, nrm[fo], 0, 0, 0);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, nrm[fo], 0, 0, 0);
                    nrm[fo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, nrm[fo], 0, 0, 0);
                }
                bh = nh;
                bl = nl;
                if (s2 == 0) __builtin_amdgcn_sched_barrier(0);   // (the second k-step's operand reads stay behind the first's MFMAs: registers)
            }
            step_end(KSTEPS + pass * NF + kk);
        }
        if (pass == 1) __syncthreads();                  // (no transfer is in flight: every chunk issued has been observed)
        MPC_STAMP(3 + pass)
        // ---- normalise and store this pass's HF tiles: y * (beta + norm)^(-1/2) -------------------------------------
        if (h == 0) ts_opix[j] = pvalid ? (long long)opix : -1ll;   // (in both passes: ring slot 0 has been written over in between)
#pragma unroll
        for (int fl = 0; fl < HF; ++fl) {
            const int f = pass * HF + fl;
            __builtin_amdgcn_sched_barrier(0);           // one tile at a time: nothing of the next tile is hoisted (registers)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 be = *(const float4 *)(beta_lds + 32 * f + 8 * g + 4 * h);
                const float bev[4] = {be.x, be.y, be.z, be.w};
                float o[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = nrm[fl][4 * g + t] * p.gdn_unscale + bev[t];
                    o[t] = acc[f][4 * g + t] * __builtin_amdgcn_rsqf(a);
                }
                char *row = ts_rows + j * TS1_ROW_PAD;
                if (out_split) {
                    half4 hi4, lo4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float sv = fminf(fmaxf(o[t] * X3_ACT_SCALE, -65504.0f), 65504.0f);
                        _Float16 hh = (_Float16)sv;
                        hi4[t] = hh;
                        lo4[t] = (_Float16)(sv - (float)hh);
                    }
                    *(half4 *)(row + g * 32 + 8 * h) = hi4;
                    *(half4 *)(row + g * 32 + 16 + 8 * h) = lo4;
                } else {
                    *(float4 *)(row + (8 * g + 4 * h) * 4) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
            // flush: 32 pixels x 8 pieces of 16 bytes, consecutive lanes on consecutive bytes of a pixel's 128-byte line
#pragma unroll
            for (int i0 = 0; i0 < 32 * 8; i0 += 64) {
                const int i = i0 + lane, px = i >> 3, pc = i & 7;
                const long long o = ts_opix[px];
                if (o >= 0)
                    *(float4 *)((char *)p.out + o * (long long)(COUT * 4) + f * TS1_ROW_BYTES + pc * 16) = *(const float4 *)(ts_rows + px * TS1_ROW_PAD + pc * 16);
            }
        }
        if (pass == 0) barrier_after_lds_writes();       // every wave has left ring slot 0 before the next step's transfer into it
    }
    MPC_STAMP(5)
}

template <int NF>
int launch_rgb(mpc_ctx *ctx, const Params &p) {
    size_t lds = ((size_t)RGB_PATCH + 3 * (size_t)NF * 32 * 4) * sizeof(float4) + (size_t)NF * 32 * sizeof(float);
    lds = patch_kernel_lds(lds);
    auto kern = conv1_rgb_patch_x3_kernel<NF>;
    static bool attr_set[MPC_MAX_DEVICES] = {};
    bool &attr_done = attr_set[ctx->device];
    if (!attr_done) {
        MPC_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int64_t tiles = (int64_t)p.B * mpc_cdiv(p.Ho, RGB_TR) * mpc_cdiv(p.Wo, TILE_W);
    const int64_t bx = xcd_grid(tiles);
    MPC_REQUIRE(ctx, bx < (1ll << 31), "conv: too many tiles");
#ifdef MPC_STAMPS
    static StampHost stamps;
    stamps.before((size_t)bx);
#endif
    kern<<<dim3((unsigned)bx, 1, 1), dim3(256), lds, ctx->stream>>>(p);
    MPC_CHECK_LAUNCH(ctx, "conv1_rgb_patch_x3");
#ifdef MPC_STAMPS
    stamps.after(ctx->stream, (size_t)bx, 4000);
#endif
    return MPC_OK;
}

}  // namespace

// True when the layer / call can run on the patch kernel.
bool mpc_conv_patch_x3_supported(const Params &p, const mpc_conv *c) {
    const mpc_conv_desc &d = c->desc;
    const bool gdn = d.act == MPC_ACT_GDN || d.act == MPC_ACT_IGDN;
    if (!c->x3 || c->cin4 || d.kind != MPC_CONV || d.ksize != 5 || d.stride != 2 || d.pad != 2) return false;
    if (d.cin % 32 != 0 || (c->nf != 4 && c->nf != 6)) return false;
    if (gdn && !(p.gamma_x3 && p.y2_scale > 0.0f)) return false;        // the exact f32 channel mix lives in the other kernel
    if (!gdn && d.act != MPC_ACT_NONE && d.act != MPC_ACT_RELU) return false;
    if ((int64_t)p.Hi * p.Wi * p.Cs * 2 >= (1ll << 31)) return false;    // 32-bit element offsets inside one image
    return true;
}

// 16 x 8 output tiles, 4 waves per workgroup, two workgroups per CU (2 x 80 640 bytes of LDS): their barriers fall
// out of phase, so one's MFMAs cover the other's waits (one 8-wave workgroup of 16 rows measured 10 % slower).
int mpc_conv_launch_patch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    const bool gdn = c->desc.act == MPC_ACT_GDN || c->desc.act == MPC_ACT_IGDN;
    const int ng = c->groups;
    if (gdn && c->nf == 6) return launch_tr<6, EPI_GDN, 8>(ctx, p, ng);
    if (gdn && c->nf == 4) return launch_tr<4, EPI_GDN, 8>(ctx, p, ng);
    if (!gdn && c->nf == 6) return launch_tr<6, EPI_PLAIN, 8>(ctx, p, ng);
    if (!gdn && c->nf == 4) return launch_tr<4, EPI_PLAIN, 8>(ctx, p, ng);
    MPC_FAIL(ctx, MPC_EINVAL, "conv_forward (f16x3 patch kernel): unsupported layer shape (cout/32 = %d)", c->nf);
}

// The 3-channel first layer with a fused GDN mix on the fp16 pipe (5x5 stride 2, one output-channel group).
bool mpc_conv_rgb_patch_x3_supported(const Params &p, const mpc_conv *c) {
    const mpc_conv_desc &d = c->desc;
    if (!c->x3 || !c->cin4 || d.kind != MPC_CONV || d.ksize != 5 || d.stride != 2 || d.pad != 2) return false;
    if (d.act != MPC_ACT_GDN || !(p.gamma_x3 && p.y2_scale > 0.0f)) return false;
    if ((c->nf != 4 && c->nf != 6) || c->groups != 1 || c->nphase != 1 || c->ph[0].ntaps != 25 || c->ph[0].nchunks != 4) return false;
    return true;
}

int mpc_conv_launch_rgb_patch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c) {
    if (c->nf == 6) return launch_rgb<6>(ctx, p);
    return launch_rgb<4>(ctx, p);
}
