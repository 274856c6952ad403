// Definitions shared by the f32 and f16 implicit-GEMM convolution kernels.
#pragma once
#include <stdlib.h>

#include "mpc_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_TAPS = 32;
constexpr int MAX_PHASES = 4;
constexpr int WAVES = 4;
constexpr int PIX_PER_BLOCK = 32 * WAVES;

enum { EPI_PLAIN = 0, EPI_GDN = 1, EPI_STACKED = 2 };

// Workgroups are handed to the 8 XCDs round-robin by linear id, and every XCD has its own L2.  The
// grid's x extent is rounded up to a multiple of 8 and block x is remapped so that XCD c works on
// the c-th contiguous eighth of the pixel blocks: neighbouring pixels, which share most of their
// input taps, then meet in one L2 instead of being fetched by all eight.
constexpr int NUM_XCD = 8;
inline int64_t xcd_grid(int64_t blocks) { return (blocks + NUM_XCD - 1) / NUM_XCD * NUM_XCD; }
// s_waitcnt immediate for "vmcnt <= n" alone (gfx9 encoding: vmcnt bits [3:0] and [15:14], other counters at maximum)
#define MPC_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
#if defined(__HIPCC__)
// Completion of LDS-DMA transfers (global_load_lds), observed in the data.
// A wave's vmcnt drops when the data of an LDS-DMA transfer has come back from memory; the write into LDS is then queued,
// not necessarily performed, and nothing the issuing wave can wait on counts it.  A wave that leaves the following
// s_barrier first can therefore read the old bytes of the destination (seen as single wrong tiles, a few per thousand
// launches, once other kernels kept the CU's LDS busy; delays in front of the barrier only made them rarer).  So the
// arrival is observed.  A wave's transfers are written in the order they were issued, so the LAST piece (64 lanes x 16
// bytes) of a batch is watched: every lane marks the first and the last dword of its own 16 destination bytes with a
// value no operand holds (two fp16 NaNs with every bit set: weights, gamma and activations are finite), the marks are
// complete in LDS (lgkmcnt(0)) before that piece is issued, and the wait for the batch is vmcnt(0) followed by a poll
// of the marked dwords until no lane sees the mark.  The marks can be stored well ahead of the piece (dma_mark at the
// top of a step, the other pieces in between, dma_marks_stored right in front of the watched piece), so that their
// latency is covered by the issue of the other pieces.
//   dma_mark(lane's destination of the last piece);  issue all pieces but the last;  dma_marks_stored();  issue it;  ...
//   dma_wait_last(the same address);  barrier
constexpr unsigned DMA_MARK = 0xFFFFFFFFu;
typedef __attribute__((address_space(3))) volatile unsigned *lds_u32_ptr;

__device__ __forceinline__ void dma_mark(void *lane_dst) {
    lds_u32_ptr q = (lds_u32_ptr)(__attribute__((address_space(3))) void *)lane_dst;
    q[0] = DMA_MARK;
    q[3] = DMA_MARK;
}
__device__ __forceinline__ void dma_marks_stored() {
    __builtin_amdgcn_s_waitcnt((63 & 15) | ((63 >> 4) << 14) | (7 << 4) | (0 << 8));              // lgkmcnt(0): the marks are in LDS
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void dma_mark_last(void *lane_dst) {
    dma_mark(lane_dst);
    dma_marks_stored();
}

// KEEP: vector-memory operations of this wave that may stay in flight (the pieces of a YOUNGER batch, issued behind the
// watched one: the counter retires in order, and the poll below is what proves the arrival anyway).
template <int KEEP = 0>
__device__ __forceinline__ void dma_wait_last(void *lane_dst) {
    __builtin_amdgcn_s_waitcnt((KEEP & 15) | ((KEEP >> 4) << 14) | (7 << 4) | (15 << 8));         // vmcnt(KEEP)
    asm volatile("" ::: "memory");
#ifdef MPC_NO_DMA_FENCE                                    // (measurement builds only: what the wait costs beyond vmcnt(0))
    return;
#endif
    // The poll, as one opaque block (a loop in the source makes the register allocator spill the accumulators that are
    // live across it): read the two marked dwords, leave when no lane sees the mark.  Bounded: a kernel must end even
    // if an operand did hold the mark.
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lane_dst;
    unsigned first, last;
    unsigned long long seen;
    int tries = 1 << 20;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %0, %4\n\t"
        "ds_read_b32 %1, %4 offset:12\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_eq_u32_e32 vcc, -1, %0\n\t"
        "v_cmp_eq_u32_e64 %2, -1, %1\n\t"
        "s_or_b64 vcc, vcc, %2\n\t"
        "s_sub_u32 %3, %3, 1\n\t"
        "s_cmp_eq_u32 %3, 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cbranch_vccnz 1b\n\t"
        "2:\n\t"
        : "=&v"(first), "=&v"(last), "=&s"(seen), "+s"(tries)
        : "v"(addr)
        : "vcc", "scc", "memory");
}

// Workgroup barrier behind LDS writes that other waves read after it.  A raw s_barrier does not wait for the wave's own
// LDS instructions: the hardware lets a wave arrive at the barrier with ds_write (and ds_read) still queued, so a wave
// that leaves the barrier first can read the old bytes.  lgkmcnt(0) is the completion of this wave's LDS writes; with it
// in front of the barrier every write issued before it is in LDS when any wave passes (what __syncthreads() emits, without
// its vmcnt(0), which would also drain the global loads that are meant to stay in flight across the barrier).
__device__ __forceinline__ void barrier_after_lds_writes() {
#ifndef MPC_EXPERIMENT_NO_LDS_WRITE_WAIT
    __builtin_amdgcn_s_waitcnt((63 & 15) | ((63 >> 4) << 14) | (7 << 4) | (0 << 8));              // lgkmcnt(0)
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// vmcnt(0) only: for waves that have issued nothing that needs watching
__device__ __forceinline__ void lds_dma_fence(const void *) {
    __builtin_amdgcn_s_waitcnt((0 & 15) | ((0 >> 4) << 14) | (7 << 4) | (15 << 8));
}

__device__ __forceinline__ int64_t xcd_block() {
    const unsigned per = gridDim.x / NUM_XCD;
    return (int64_t)(blockIdx.x % NUM_XCD) * per + blockIdx.x / NUM_XCD;
}
#endif

// Dynamic LDS of the patch kernels.  A measurement build (-DMPC_EXPERIMENT_ONE_WG_PER_CU) asks for more than half of a
// CU's 160 KB, so that a CU holds one workgroup instead of two: what a layer runs at where another kernel's waves keep
// its second workgroup out.
inline size_t patch_kernel_lds(size_t bytes) {
#ifdef MPC_EXPERIMENT_ONE_WG_PER_CU
    if (bytes < 81 * 1024) bytes = 81 * 1024;
#endif
    return bytes;
}

// Power-of-two scale of activations stored in the split-fp16 ("S16") format.
constexpr float X3_ACT_SCALE = 256.0f;

struct Phase {
    int ntaps;            // real taps of this phase
    int nchunks;          // K chunks (taps*cin/32, or ceil(taps/8) for the 4-channel input path)
    int qy, qx;           // output parity offset
    int64_t w_off;        // float offset of this phase's packed weights
    int tap_off;          // offset into the tap table
};

struct Params {
    const void *in;       // fp32 (or fp16 on the f16 path) NHWC activations
    void *out;            // activation output (may be null when only out_round is wanted)
    float *out_round;     // rint(out) (may be null; f32 path only)
    const void *wpk;      // packed weights (fp32 or fp16)
    const float *bias;
    const float *beta;    // effective GDN parameters
    const void *gamma;    // fp32 [cout][cout], or fp16 in MFMA operand order on the f16 path
    const int2 *taps;     // (dy, dx) per tap, all phases
    int B, Hi, Wi, Cs;    // input NHWC, Cs = channel stride
    int Hb, Wb;           // base grid
    int64_t P;            // B*Hb*Wb
    int sy, sx;           // src = base*s + d
    int Ho, Wo, my, mx;   // dst = base*m + q
    int cin_blocks;       // Cin/32 (f32 path; unused for the 4-channel input) or Cin/64 (f16 path)
    int cout;             // real output channels (all groups)
    int64_t w_group;      // float stride between the packed weights of consecutive cout groups
    int act;              // MPC_ACT_*
    int flags;            // MPC_IN_ABS | MPC_OUT_CLAMP01 | MPC_OUT_NCHW | MPC_OUT_SPLIT
    float x3_unscale;     // f16x3 path: 1 / (weight scale * activation scale)
    const void *gamma_x3; // f16x3 GDN: fp16 hi/lo gamma in MFMA operand order (null -> exact f32 GDN)
    float y2_scale;       // f16x3 GDN: power-of-two scale of y^2 (0 -> exact f32 GDN)
    float gdn_unscale;    // 1 / (gamma scale * y2_scale)
    int16_t *out_sym;     // plain epilogue: int16 symbols of rint(out) (null: none), NHWC like out_round
    float sym_min_feat;   // symbol = rint(out) - sym_min_feat, clamped to [0, sym_max]
    int sym_max;
    int *n_clamped;       // number of symbols that had to be clamped (may be null)
    const void *zeros;    // 256 bytes of zeros (what a tap outside the image reads)
    const float *score_target;   // small-cout fp16 layer: NCHW fp32 image the clamped output is scored against (null: none)
    double *score_sse;           // ... its per-image sums of squared errors, float64 [B], zeroed by the caller
    Phase ph[MAX_PHASES];
    int2 tapv[MAX_TAPS];  // the tap table by value: indexed with a uniform index it is read by scalar loads, which the
                          // vector-memory counter (and an LDS-DMA in flight) does not see
};

#if defined(__HIPCC__)
// Four consecutive channels of one pixel as int16 symbols (8 bytes): ((y_hat - min_feat) / 1).round().short() of the
// reference (src/compression/models/manynets_compressor.py:264-269) with q_intv = 1; out-of-range values are clamped
// and counted so that the caller can widen its tables as the reference does (:212-227).
__device__ __forceinline__ void store_symbols4(const Params &p, int64_t elem, const float (&rounded)[4], int &clamped) {
    typedef short short4_t __attribute__((ext_vector_type(4)));
    short4_t s4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float v = rounded[t] - p.sym_min_feat;
        int s = (int)v;
        if (!(v >= 0.0f)) { s = 0; ++clamped; }
        if (v > (float)p.sym_max) { s = p.sym_max; ++clamped; }
        s4[t] = (short)s;
    }
    *(short4_t *)(p.out_sym + elem) = s4;
}
#endif


// ---------------------------------------------------------------------------------------------
// Wave-private transposing store for the GDN epilogues.  In the accumulator layout a lane owns ONE
// pixel and a few channels, so a direct store writes 8-16 bytes per lane at a whole-pixel stride:
// partial cache lines, the slowest way to write HBM.  Instead each wave collects up to ROW_BYTES
// contiguous output bytes of each of its 32 pixels in LDS (the weight buffers are free once the K loop
// has ended) and flushes them with 16-byte stores in which consecutive lanes cover consecutive bytes
// of a pixel's row: full 128-byte lines.
// ---------------------------------------------------------------------------------------------
constexpr int TS_ROW_BYTES = 384;                 // bytes of one pixel collected per round
constexpr int TS_ROW_PAD = 400;                   // LDS row pitch (odd multiple of 16 B: conflict-free b128)
constexpr int TS_WAVE_BYTES = 32 * TS_ROW_PAD + 32 * 8;   // rows + the 32 output-pixel indices
constexpr int TS_BLOCK_BYTES = WAVES * TS_WAVE_BYTES;

struct TransStore {
    char *rows;          // this wave's [32][TS_ROW_PAD]
    long long *opix;     // this wave's [32] output pixel index (-1 = not stored)
    int lane;

    __device__ __forceinline__ void init(char *lds_base, int wave, int lane_, int j, int h, long long my_opix) {
        rows = lds_base + wave * TS_WAVE_BYTES;
        opix = (long long *)(rows + 32 * TS_ROW_PAD);
        lane = lane_;
        if (h == 0) opix[j] = my_opix;
    }
    // pointer to byte `off` of pixel j's row of the current round
    __device__ __forceinline__ char *at(int j, int off) const { return rows + j * TS_ROW_PAD + off; }
    // write `pieces` 16-byte pieces per pixel to out + opix*pixel_bytes + col_off
    __device__ __forceinline__ void flush(char *out, long long pixel_bytes, int col_off, int pieces) const {
        const int total = 32 * pieces;
        for (int i = lane; i < total; i += 64) {
            int px = i / pieces, pc = i - px * pieces;
            long long o = opix[px];
            if (o >= 0) *(float4 *)(out + o * pixel_bytes + col_off + pc * 16) = *(const float4 *)(rows + px * TS_ROW_PAD + pc * 16);
        }
    }
    template <int PIECES>
    __device__ __forceinline__ void flush(char *out, long long pixel_bytes, int col_off) const {
        if constexpr (PIECES == 0) return;                // (a round with nothing left to write)
#pragma unroll
        for (int i0 = 0; i0 < 32 * PIECES; i0 += 64) {
            const int i = i0 + lane;
            const int px = i / PIECES, pc = i - px * PIECES;
            if (i < 32 * PIECES) {
                long long o = opix[px];
                if (o >= 0) *(float4 *)(out + o * pixel_bytes + col_off + pc * 16) = *(const float4 *)(rows + px * TS_ROW_PAD + pc * 16);
            }
        }
    }
};

// The fp16 / f16x3 kernels come with 4 or 8 waves (128 or 256 pixels) per workgroup.  A workgroup stages
// every weight chunk once for all of its waves, and the weight traffic is most of what goes through the
// CU's vector-memory path, so one 8-wave workgroup per CU moves a third less data than two 4-wave
// workgroups.  Used once there are at least two 256-pixel workgroups per CU (or forced by MPC_WIDE_BLOCK).
constexpr int NUM_CU = 256;
inline bool wide_blocks(const Params &p, int launches_per_pixel_block) {
    static const bool narrow = [] { const char *e = getenv("MPC_NARROW_BLOCKS"); return e && atoi(e) != 0; }();   // development switch
    if (narrow) return false;
    return (p.flags & MPC_WIDE_BLOCK) || (p.P + 255) / 256 * launches_per_pixel_block >= 2 * NUM_CU;
}

// ---------------------------------------------------------------------------------------------
// Diagnostic build (-DMPC_STAMPS): in-kernel timestamps.  Thread 0 of every workgroup records the 100 MHz
// constant-rate counter at up to 6 points (plus the shader-clock ticks of its K loop and its hardware id) into a
// buffer of the kernel's own; with MPC_DUMP=<file> in the environment the launch synchronises and writes one line per
// workgroup (tools/stamps_report.py digests it).  No stamp exists in a normal build.
// ---------------------------------------------------------------------------------------------
#if defined(MPC_STAMPS) && defined(__HIPCC__)
#include <stdio.h>

#include <vector>
static __device__ unsigned long long *g_stamps;
#define MPC_STAMP_DECL(wg_linear) unsigned long long *stamps_ = g_stamps + (size_t)(wg_linear) * 16; unsigned long long kclk_ = 0, seg_[5] = {0, 0, 0, 0, 0}, tprev_ = 0; (void)kclk_; (void)seg_; (void)tprev_;
#define MPC_STAMP(i)                                                                                         \
    if (threadIdx.x == 0) {                                                                                  \
        unsigned long long t_;                                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        stamps_[i] = t_;                                                                                     \
    }
#define MPC_STAMP_KBEGIN() kclk_ = tprev_ = __builtin_amdgcn_s_memtime();
// shader-clock cycles of wave 0 between consecutive MPC_SEG marks, summed per mark index over the K loop
#define MPC_SEG(i)                                                                                           \
    {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long t_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                           \
        seg_[i] += t_ - tprev_;                                                                              \
        tprev_ = t_;                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
#define MPC_STAMP_KEND()                                                                                     \
    if (threadIdx.x == 0) {                                                                                  \
        stamps_[6] = __builtin_amdgcn_s_memtime() - kclk_;                                                   \
        unsigned hw_;                                                                                        \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                    \
        stamps_[7] = hw_;                                                                                    \
        for (int i_ = 0; i_ < 5; ++i_) stamps_[8 + i_] = seg_[i_];                                           \
    }
struct StampHost {
    unsigned long long *dbuf = nullptr;
    size_t nwg = 0;
    void before(size_t n) {
        nwg = n;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * (1 << 17));
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 16 * nwg);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dbuf, sizeof(dbuf));
    }
    void after(hipStream_t stream, size_t blocks_per_phase, size_t min_wg) {
        const char *path = getenv("MPC_DUMP");
        if (!path || nwg < min_wg) return;
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(16 * nwg);
        (void)hipMemcpy(h.data(), dbuf, sizeof(unsigned long long) * 16 * nwg, hipMemcpyDeviceToHost);
        FILE *fp = fopen(path, "w");
        if (!fp) return;
        for (size_t i = 0; i < nwg; ++i) {
            for (int k = 0; k < 8; ++k) fprintf(fp, "%llu ", h[16 * i + k]);
            fprintf(fp, "%d", (int)(i / blocks_per_phase));
            for (int k = 8; k < 13; ++k) fprintf(fp, " %llu", h[16 * i + k]);
            fprintf(fp, "\n");
        }
        fclose(fp);
    }
};
#else
#define MPC_STAMP_DECL(wg_linear)
#define MPC_STAMP(i)
#define MPC_STAMP_KBEGIN()
#define MPC_STAMP_KEND()
#define MPC_SEG(i)
#endif

// Layer object behind the opaque mpc_conv handle.
struct mpc_conv {
    mpc_conv_desc desc;
    int device;
    int nf;               // accumulator tiles per wave
    int groups;           // cout groups (grid.z)
    int64_t w_group;      // packed elements per group
    bool cin4, stacked, f16, x3;
    float w_scale = 1.0f;  // f16x3: power-of-two weight scale
    void *gamma_x3 = nullptr;
    void *wpk_rgb16 = nullptr;   // fp16 last layer: A image of the 16-row kernel (conv_rgb_out_f16.hip)
    float gamma_scale = 0.0f, y2_scale = 0.0f;
    int nphase;
    int sy, sx, my, mx;
    Phase ph[MAX_PHASES];
    void *wpk = nullptr;
    void *wpk_narrow = nullptr;      // exact f32 layers without GDN: the same weights packed as one group per 32 output channels
    int64_t w_group_narrow = 0;      // (for launches whose pixels do not fill the GPU, see conv_igemm.hip)
    float *bias = nullptr, *beta = nullptr;   // bias: nf*groups*32 floats, followed by 64 floats of zeros
    void *gamma = nullptr;
    int2 *taps = nullptr;
    int2 taps_host[MAX_TAPS] = {};   // host copy of the tap table (all phases)
    int ntaps_total = 0;
};

// f16 path (conv_igemm_f16.hip)
int mpc_conv_launch_f16(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
// fp16 last layer, cout <= 4 (conv_rgb_out_f16.hip)
bool mpc_conv_rgb16_supported(int cin, int cout);
bool mpc_conv_rgb16_rows_supported(int H, int W);
size_t mpc_conv_rgb16_image_bytes(int cin);
hipError_t mpc_conv_rgb16_pack(const float *weight, int cin, int cout, void *image, hipStream_t stream);
int mpc_conv_launch_rgb16(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
// fp16 5x5 stride-2 transposed convolutions with the input patch staged in LDS (deconv5s2_patch_f16.hip)
bool mpc_deconv_patch_f16_supported(const Params &p, const mpc_conv *c);
int mpc_deconv_launch_patch_f16(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
// split-fp16 path (conv_igemm_x3.hip)
int mpc_conv_launch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
// split-fp16 5x5 stride-2 layers with the input patch staged in LDS (conv5s2_patch_x3.hip)
bool mpc_conv_patch_x3_supported(const Params &p, const mpc_conv *c);
int mpc_conv_launch_patch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
bool mpc_conv_rgb_patch_x3_supported(const Params &p, const mpc_conv *c);
int mpc_conv_launch_rgb_patch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
