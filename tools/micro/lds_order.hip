// Microbenchmark: which waits make data written to LDS by one wave visible to another wave behind a raw s_barrier.
// Four waves per workgroup, two workgroups per CU, every CU busy.  Per iteration a wave
//   reads a burst of LDS (the operand reads of a K step), multiplies a few MFMAs,
//   deposits 3 KB in its own region of a double-buffered LDS area,
//   waits (the variant under test), passes a raw s_barrier,
//   reads the region of the NEXT wave and compares it with the value that wave must have deposited.
// Variants (deposit / wait):
//   0  ds_write_b128 from registers / nothing
//   1  ds_write_b128 from registers / s_waitcnt lgkmcnt(0)
//   2  global_load_lds_dwordx4      / s_waitcnt vmcnt(0)
//   3  global_load_lds_dwordx4      / s_waitcnt vmcnt(0), one ds_read of the last piece, s_sleep 1
//   4  global_load_lds_dwordx4      / s_waitcnt vmcnt(0), s_waitcnt lgkmcnt(0)
//   5  global_load_lds_dwordx4 issued one iteration ahead / counted wait (vmcnt(3): all but the pieces just issued)
//   6  global_load_lds_dwordx4 issued two iterations ahead / the same counted wait: a deposit is read a whole
//      iteration after the wait that retired it
// Prints the number of wrong 16-byte slots per variant.  A background stream can run a copy kernel beside it (arg 3).
// usage: lds_order [iterations] [rounds] [background 0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4, PIECES = 3;                       // waves per workgroup, 1 KB pieces per wave and deposit
constexpr int REGION = PIECES * 64;                     // slots of one wave's region
constexpr int AREA = NW * REGION;                       // slots of one buffer
constexpr int NBUF = 4;                                 // (variant 6 keeps two deposits ahead of the reader: four buffers)
constexpr int OPER = 1536;                              // slots of read-only operands (24 KB)
constexpr int LDS_SLOTS = 5056;                         // 79 KB per workgroup: two workgroups per CU, as in the layers
constexpr int NPAT = 5;                                 // source patterns (coprime with NBUF and 2)
static_assert(NBUF * AREA + OPER <= LDS_SLOTS, "LDS layout");

__host__ __device__ inline unsigned pattern(int pat, int slot, int c) { return (unsigned)pat * 0x01000193u + (unsigned)(slot * 4 + c) * 2654435761u + 12345u; }

template <int V>
__global__ __launch_bounds__(256, 2) void order_kernel(const u4 *src, unsigned *wrong, float *sink, int iters) {
    extern __shared__ u4 lds[];
    u4 *oper = lds + NBUF * AREA;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < OPER; i += 256) oper[i] = src[(i * 7) % (NPAT * AREA)];
    for (int i = tid; i < NBUF * AREA; i += 256) lds[i] = u4{0u, 0u, 0u, 0u};
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    unsigned bad = 0;
    const int next = (wave + 1) & (NW - 1);

    auto deposit = [&](int it) {                        // this wave's region of buffer it % NBUF <- pattern it % NPAT
        const int pat = it % NPAT, buf = it % NBUF;
        u4 *dst = lds + buf * AREA + wave * REGION;
        if constexpr (V <= 1) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int slot = wave * REGION + q * 64 + lane;
                dst[q * 64 + lane] = u4{pattern(pat, slot, 0), pattern(pat, slot, 1), pattern(pat, slot, 2), pattern(pat, slot, 3)};
            }
        } else {
#pragma unroll
            for (int q = 0; q < PIECES; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + pat * AREA + wave * REGION + q * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(dst + q * 64), 16, 0, 0);
        }
    };
    auto check = [&](int it) {                          // the next wave's region of buffer it % NBUF
        const int pat = it % NPAT, buf = it % NBUF;
        const u4 *p = lds + buf * AREA + next * REGION;
#pragma unroll
        for (int q = 0; q < PIECES; ++q) {
            const int slot = next * REGION + q * 64 + lane;
            const u4 v = p[q * 64 + lane];
            if (v[0] != pattern(pat, slot, 0) || v[1] != pattern(pat, slot, 1) || v[2] != pattern(pat, slot, 2) || v[3] != pattern(pat, slot, 3)) ++bad;
        }
    };

    if constexpr (V == 5) deposit(0);
    if constexpr (V == 6) { deposit(0); deposit(1); }
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 5) deposit(it + 1);          // one iteration ahead
        if constexpr (V == 6) deposit(it + 2);          // two iterations ahead
        // the operand reads and MFMAs of a K step
        const u4 *wl = oper + ((it * 5) & 7) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const half8 a = __builtin_bit_cast(half8, wl[(2 * f) * 64]), b = __builtin_bit_cast(half8, wl[(2 * f + 1) * 64]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if constexpr (V < 5) deposit(it);
        if constexpr (V == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (V == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (V == 3) {
            unsigned probe;
            asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_sleep 1"
                         : "=v"(probe)
                         : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(lds + (it % NBUF) * AREA + wave * REGION + (PIECES - 1) * 64 + lane))
                         : "memory");
        }
        if constexpr (V == 4) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (V == 5 || V == 6) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // all but the deposit just issued (PIECES = 3)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        check(it);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123.456f) sink[0] = s;
    if (bad) atomicAdd(wrong + V, bad);
}

__global__ void background_copy(const u4 *a, u4 *b, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <int V>
void run(const u4 *src, unsigned *wrong, float *sink, int iters, hipStream_t s) {
    auto k = order_kernel<V>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_SLOTS * 16));
    k<<<2048, 256, LDS_SLOTS * 16, s>>>(src, wrong, sink, iters);
    CK(hipGetLastError());
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, rounds = argc > 2 ? atoi(argv[2]) : 3, bg = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<unsigned> h((size_t)NPAT * AREA * 4);
    for (int pat = 0; pat < NPAT; ++pat)
        for (int slot = 0; slot < AREA; ++slot)
            for (int c = 0; c < 4; ++c) h[((size_t)pat * AREA + slot) * 4 + c] = pattern(pat, slot, c);
    u4 *src; unsigned *wrong; float *sink; u4 *ba, *bb;
    const size_t nbg = (size_t)64 << 20;                // 1 GiB each way
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&wrong, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&ba, nbg * 16)); CK(hipMalloc(&bb, nbg * 16)); CK(hipMemset(ba, 1, nbg * 16));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(wrong, 0, 64));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    for (int r = 0; r < rounds; ++r) {
        if (bg) background_copy<<<512, 256, 0, s1>>>(ba, bb, nbg, 8);
        run<0>(src, wrong, sink, iters, s0); run<1>(src, wrong, sink, iters, s0); run<2>(src, wrong, sink, iters, s0);
        run<3>(src, wrong, sink, iters, s0); run<4>(src, wrong, sink, iters, s0); run<5>(src, wrong, sink, iters, s0); run<6>(src, wrong, sink, iters, s0);
        CK(hipDeviceSynchronize());
        unsigned w[8]; CK(hipMemcpy(w, wrong, 32, hipMemcpyDeviceToHost));
        printf("round %d (background %d): wrong slots  ds_write/none %u | ds_write/lgkmcnt %u | dma/vmcnt0 %u | dma/vmcnt0+probe+sleep %u | dma/vmcnt0+lgkmcnt0 %u | dma/ahead-1 counted %u | dma/ahead-2 counted %u   (of %.3g checked per variant)\n",
               r, bg, w[0], w[1], w[2], w[3], w[4], w[5], w[6], (double)(r + 1) * 2048.0 * 256 * PIECES * iters);
        fflush(stdout);
    }
    return 0;
}
