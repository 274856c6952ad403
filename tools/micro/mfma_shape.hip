// Microbenchmark: the K-loop skeleton of the split-precision patch kernel (LDS-resident operands, three fp16 MFMAs per
// product, one barrier per step, four waves per workgroup, two workgroups per CU) with v_mfma_f32_32x32x16_f16 against
// v_mfma_f32_16x16x32_f16 at the same output tile per wave (192 x 32), the same LDS bytes per FLOP, random operands.
// Prints TFLOP/s of issued MFMA work for both shapes, interleaved rounds in one process.
// usage: mfma_shape [iterations per workgroup]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 as_half8(const f4 &v) { return __builtin_bit_cast(half8, v); }

constexpr int LDS_F4 = 4096;            // 64 KB of operands per workgroup (two workgroups per CU)

// one step = K 32: 2 x (12 A reads + 2 B reads, 18 MFMAs 32x32x16)
__global__ __launch_bounds__(256, 2) void loop_32x32(const f4 *src, float *out, int iters) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_F4; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x16 acc[6];
    for (int f = 0; f < 6; ++f) for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const f4 *wl = lds + ((it * 7) & 7) * 448 + lane;            // weights: [hi|lo][tile 6][64 lanes] per k-step
        const f4 *bp = lds + 3584 + wave * 128 + ((it * 3) & 1) * 0 + lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 bh = as_half8(bp[ks * 256 % 384]), bl = as_half8(bp[(ks * 256 + 64) % 384]);
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const half8 ah = as_half8(wl[(ks * 12 + f) * 64 % 3520]), al = as_half8(wl[(ks * 12 + 6 + f) * 64 % 3520]);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[f], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int f = 0; f < 6; ++f) for (int r = 0; r < 16; ++r) s += acc[f][r];
    out[blockIdx.x * 256 + tid] = s;
}

// one step = K 32: 24 A reads + 4 B reads, 72 MFMAs 16x16x32 (12 M tiles x 2 N tiles x 3 passes)
__global__ __launch_bounds__(256, 2) void loop_16x16(const f4 *src, float *out, int iters) {
    extern __shared__ f4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_F4; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 acc[12][2];
    for (int f = 0; f < 12; ++f) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) acc[f][n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const f4 *wl = lds + ((it * 7) & 7) * 448 + lane;
        const f4 *bp = lds + 3584 + wave * 128 + lane;
        half8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { bh[n] = as_half8(bp[n * 128 % 384]); bl[n] = as_half8(bp[(n * 128 + 64) % 384]); }
#pragma unroll
        for (int f = 0; f < 12; ++f) {
            const half8 ah = as_half8(wl[f * 64 % 3520]), al = as_half8(wl[(12 + f) * 64 % 3520]);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], acc[f][n], 0, 0, 0);
                acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], acc[f][n], 0, 0, 0);
                acc[f][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], acc[f][n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int f = 0; f < 12; ++f) for (int n = 0; n < 2; ++n) for (int r = 0; r < 4; ++r) s += acc[f][n][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int nwg = 512 * 4;
    std::vector<_Float16> h(LDS_F4 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    f4 *src; float *out;
    CK(hipMalloc(&src, LDS_F4 * 16)); CK(hipMalloc(&out, nwg * 256 * 4));
    CK(hipMemcpy(src, h.data(), LDS_F4 * 16, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)loop_32x32, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F4 * 16 + 14336));
    CK(hipFuncSetAttribute((const void *)loop_16x16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_F4 * 16 + 14336));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = (double)nwg * 4 * iters * 36.0 * 32768.0;        // issued MFMA FLOP per launch (both shapes)
    for (int round = 0; round < 4; ++round) {
        for (int v = 0; v < 2; ++v) {
            // 80 KB per workgroup: two per CU, as in the real kernel
            CK(hipEventRecord(e0));
            if (v == 0) loop_32x32<<<nwg, 256, LDS_F4 * 16 + 14336>>>(src, out, iters);
            else loop_16x16<<<nwg, 256, LDS_F4 * 16 + 14336>>>(src, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d %s: %.3f ms  %.0f TFLOP/s issued\n", round, v == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
        }
    }
    return 0;
}
