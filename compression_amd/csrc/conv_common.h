// Definitions shared by the f32 and f16 implicit-GEMM convolution kernels.
#pragma once
#include "mpc_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_TAPS = 32;
constexpr int MAX_PHASES = 4;
constexpr int WAVES = 4;
constexpr int PIX_PER_BLOCK = 32 * WAVES;

enum { EPI_PLAIN = 0, EPI_GDN = 1, EPI_STACKED = 2 };

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
    Phase ph[MAX_PHASES];
};


// Layer object behind the opaque mpc_conv handle.
struct mpc_conv {
    mpc_conv_desc desc;
    int device;
    int nf;               // accumulator tiles per wave
    int groups;           // cout groups (grid.z)
    int64_t w_group;      // packed elements per group
    bool cin4, stacked, f16, x3;
    float w_scale = 1.0f;  // f16x3: power-of-two weight scale
    int nphase;
    int sy, sx, my, mx;
    Phase ph[MAX_PHASES];
    void *wpk = nullptr;
    float *bias = nullptr, *beta = nullptr;
    void *gamma = nullptr;
    int2 *taps = nullptr;
};

// f16 path (conv_igemm_f16.hip)
int mpc_conv_launch_f16(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
// split-fp16 path (conv_igemm_x3.hip)
int mpc_conv_launch_x3(mpc_ctx *ctx, const Params &p, const mpc_conv *c);
