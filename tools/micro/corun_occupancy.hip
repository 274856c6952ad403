// Microbenchmark: does a latency-bound "coder-like" kernel that only OCCUPIES the CUs (2 waves and 18 KB of LDS per
// workgroup, ~48 registers, sleeping) slow an MFMA kernel that wants two 4-wave workgroups per CU, and does that depend
// on the MFMA kernel's register count?  The MFMA kernel is built at three register footprints (asm-pinned dummies keep
// the count up): 248, 208 and 160.  Each is timed alone and beside the occupier, which is launched first on another
// stream and outlives it.
// usage: corun_occupancy [occupier workgroups, default 660] [occupier LDS bytes, default 18484] [sleeps, default 3500] [occupier launched second 0|1] [MFMA workgroups, default 2304]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef OCCV
#define OCCV 24                  // live registers of the occupier beyond its bookkeeping (-DOCCV=40: a 48-register kernel)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EXTRA>   // EXTRA: additional live registers beyond the 6 accumulator tiles
__global__ __launch_bounds__(256, 2) void mfma_kernel(float *sink, int iters) {
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 3000; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    f32x16 acc[6];
    for (int f = 0; f < 6; ++f) for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
#ifdef WITH_SCRATCH                                       // a few bytes of private memory per lane, as a kernel with spills has
    volatile float priv[4];
    priv[lane & 3] = 1.0f;
#endif
    float extra[EXTRA];
    for (int i = 0; i < EXTRA; ++i) extra[i] = (float)(lane + i);
    for (int it = 0; it < iters; ++it) {
        const half8 b = __builtin_bit_cast(half8, lds[(it * 7 + lane) % 2048]);
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const half8 a = __builtin_bit_cast(half8, lds[((it + f) * 64 + lane) % 2048]);
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[f], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < EXTRA; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(extra[i]));   // all of them stay live
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int f = 0; f < 6; ++f) for (int r = 0; r < 16; ++r) s += acc[f][r];
    for (int i = 0; i < EXTRA; ++i) s += extra[i];
#ifdef WITH_SCRATCH
    s += priv[(lane + 1) & 3];
#endif
    if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(128) void occupier_kernel(float *sink, int sleeps) {
    extern __shared__ float4 lds[];
    float v[OCCV];
    for (int i = 0; i < OCCV; ++i) v[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = make_float4(v[0], v[1], v[2], v[3]);
    for (int s = 0; s < sleeps; ++s) {
        __builtin_amdgcn_s_sleep(127);
#pragma unroll
        for (int i = 0; i < OCCV; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));
    }
    float t = 0.f;
    for (int i = 0; i < OCCV; ++i) t += v[i];
    if (t == 12345.678f) sink[0] = t + lds[0].x;
}

static int g_mfma_wgs = 18432 / 8;   // workgroups of the MFMA kernel (x iterations each)
static int g_sleeps = 3500;        // occupier lifetime: 3500 sleeps = 12 ms (outlives the MFMA kernel), 370 = 1.3 ms
static bool g_occupier_second = false;   // launch the occupier right BEHIND the MFMA kernel instead of in front of it

template <int EXTRA>
float run(hipStream_t s0, hipStream_t s1, float *sink, bool beside, int occ_wgs, int occ_lds, int iters) {
    auto k = mfma_kernel<EXTRA>;
    const int lds = 48896;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        if (beside && !g_occupier_second) occupier_kernel<<<occ_wgs, 128, occ_lds, s1>>>(sink, g_sleeps);
        CK(hipEventRecord(e0, s0));
        k<<<g_mfma_wgs, 256, lds, s0>>>(sink, iters);
        CK(hipEventRecord(e1, s0));
        if (beside && g_occupier_second) occupier_kernel<<<occ_wgs, 128, occ_lds, s1>>>(sink, g_sleeps);
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const int occ_wgs = argc > 1 ? atoi(argv[1]) : 660, occ_lds = argc > 2 ? atoi(argv[2]) : 18484;
    if (argc > 3) g_sleeps = atoi(argv[3]);
    if (argc > 4) g_occupier_second = atoi(argv[4]) != 0;
    float *sink; CK(hipMalloc(&sink, 64));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipFuncAttributes a;
    CK(hipFuncGetAttributes(&a, (const void *)occupier_kernel)); printf("occupier: %d registers, %d workgroups, %d bytes of LDS\n", a.numRegs, occ_wgs, occ_lds);
    // occupier alone: its duration
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s1)); occupier_kernel<<<occ_wgs, 128, occ_lds, s1>>>(sink, g_sleeps); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
    float oms; CK(hipEventElapsedTime(&oms, e0, e1)); printf("occupier alone: %.3f ms\n", oms);
    int iters = 400;
    if (argc > 5) { g_mfma_wgs = atoi(argv[5]); iters = 400 * (18432 / 8) / g_mfma_wgs; }   // same work in more, shorter workgroups
#define CASE(EXTRA)                                                                                          \
    {                                                                                                        \
        CK(hipFuncGetAttributes(&a, (const void *)mfma_kernel<EXTRA>));                                      \
        float alone = run<EXTRA>(s0, s1, sink, false, occ_wgs, occ_lds, iters);                              \
        float with = run<EXTRA>(s0, s1, sink, true, occ_wgs, occ_lds, iters);                                \
        printf("mfma kernel with %3d registers: alone %.3f ms, beside the occupier %.3f ms (x%.2f)\n", a.numRegs, alone, with, with / alone); \
    }
    CASE(130) CASE(90) CASE(82) CASE(74) CASE(66) CASE(40)
    return 0;
}
