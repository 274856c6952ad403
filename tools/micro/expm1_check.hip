// Walks every float q <= 0 (2^31 bit patterns, NaNs included) and compares (float)mpc_expm1_nonpos((double)q), the
// evaluation the Laplace rows are built from, with (float)expm1((double)q) of the device library: prints how many
// differ, the largest difference in units of the last place, and the first few arguments.  Then the time of a chain
// of dependent evaluations (what the decoder waits for) and of independent ones (throughput) for both.
// usage: expm1_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../compression_amd/csrc/mpc_internal.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Report { unsigned long long differ, nan_differ; unsigned max_ulp; unsigned first[16]; unsigned n_first; };

__global__ __launch_bounds__(256) void walk(Report *rep) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long differ = 0, nan_differ = 0;
    unsigned max_ulp = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {
        const uint32_t bits = 0x80000000u | (uint32_t)i;
        const float q = __uint_as_float(bits);
        const float a = (float)mpc_expm1_nonpos((double)q);
        const float b = (float)expm1((double)q);
        if (q != q) { if (!(a != a)) ++nan_differ; continue; }
        if (a != b) {
            ++differ;
            const unsigned d = (unsigned)abs((int)__float_as_uint(a) - (int)__float_as_uint(b));
            if (d > max_ulp) max_ulp = d;
            const unsigned slot = atomicAdd(&rep->n_first, 1u);
            if (slot < 16) rep->first[slot] = bits;
        }
    }
    if (differ) atomicAdd(&rep->differ, differ);
    if (nan_differ) atomicAdd(&rep->nan_differ, nan_differ);
    if (max_ulp) atomicMax(&rep->max_ulp, max_ulp);
}

// DEPENDENT: every evaluation's argument comes from the one before (one wave per SIMD: the latency of the chain);
// else four independent ones per step and four waves per SIMD (throughput).
template <int WHICH, bool DEPENDENT>
__global__ __launch_bounds__(256) void rate(float *sink, int reps) {
    float q = -1e-3f * (float)(threadIdx.x + 1), acc = 0.0f;
    for (int i = 0; i < reps; ++i) {
        const float v = WHICH ? (float)mpc_expm1_nonpos((double)q) : (float)expm1((double)q);
        acc += v;
        q = DEPENDENT ? -0.5f + 0.25f * v : q * 1.0001f;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    Report *rep;
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMemset(rep, 0, sizeof(Report)));
    walk<<<4096, 256>>>(rep);
    CK(hipDeviceSynchronize());
    Report h;
    CK(hipMemcpy(&h, rep, sizeof(h), hipMemcpyDeviceToHost));
    printf("arguments with a different float result: %llu of 2147483648 (largest difference %u ulp); NaN arguments not mapped to NaN: %llu\n",
           h.differ, h.max_ulp, h.nan_differ);
    for (unsigned i = 0; i < h.n_first && i < 16; ++i) { float q; memcpy(&q, &h.first[i], 4); printf("  q = %.9g (0x%08x)\n", q, h.first[i]); }
    float *sink;
    CK(hipMalloc(&sink, 1024 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 4096;
    for (int v = 0; v < 4; ++v) {
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0));
            if (v == 0) rate<0, true><<<256, 256>>>(sink, reps);
            if (v == 1) rate<1, true><<<256, 256>>>(sink, reps);
            if (v == 2) rate<0, false><<<1024, 256>>>(sink, reps);
            if (v == 3) rate<1, false><<<1024, 256>>>(sink, reps);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-17s %s: %.3f ms for %d evaluations per lane = %.1f ns each\n", (v & 1) ? "mpc_expm1_nonpos" : "library expm1",
               v < 2 ? "dependent chain, one wave per SIMD " : "independent, four waves per SIMD  ", ms, reps, ms * 1e6 / reps);
    }
    return 0;
}
