// Microbenchmark of CDF-row gather variants: out[loc][M*L] = table[idx[loc]][M*L], int16, M=256, L=257.
// usage: gather_bench [n_loc] [skew]     (skew: 0 = uniform prior indices, 1 = a few priors dominate)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int K = 64;
constexpr int64_t VPP = 256 * 257 * 2 / 16;   // 8224 vectors per prior

template <bool NT>
__global__ __launch_bounds__(256) void v_block_per_loc(const u32x4 *__restrict__ table, const int *__restrict__ idx, u32x4 *__restrict__ out) {
    const int64_t loc = blockIdx.x;
    const u32x4 *src = table + (int64_t)idx[loc] * VPP;
    u32x4 *dst = out + loc * VPP;
    int64_t i = threadIdx.x;
    for (; i + 3 * 256 < VPP; i += 4 * 256) {
        u32x4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        if (NT) { __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256); __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768); }
        else { dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d; }
    }
    for (; i < VPP; i += 256) { if (NT) __builtin_nontemporal_store(src[i], dst + i); else dst[i] = src[i]; }
}

// locations sorted by prior; a workgroup of T threads keeps one prior's rows in registers and writes them to each of
// its LPB consecutive sorted locations, reloading only when the prior changes
template <int T, bool NT>
__global__ __launch_bounds__(T) void v_sorted_regs(const u32x4 *__restrict__ table, const int *__restrict__ idx, const int *__restrict__ order,
                                                   int n_loc, int lpb, u32x4 *__restrict__ out) {
    constexpr int R = (VPP + T - 1) / T;
    u32x4 r[R];
    int cur = -1;
    const int first = blockIdx.x * lpb;
    for (int q = first; q < first + lpb && q < n_loc; ++q) {
        const int loc = order[q];
        const int k = idx[loc];
        if (k != cur) {
            cur = k;
            const u32x4 *src = table + (int64_t)k * VPP;
#pragma unroll
            for (int j = 0; j < R; ++j) { int64_t i = (int64_t)j * T + threadIdx.x; if (i < VPP) r[j] = src[i]; }
        }
        u32x4 *dst = out + (int64_t)loc * VPP;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            int64_t i = (int64_t)j * T + threadIdx.x;
            if (i < VPP) { if (NT) __builtin_nontemporal_store(r[j], dst + i); else dst[i] = r[j]; }
        }
    }
}

__global__ void fill_kernel(u32x4 *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    u32x4 v = {1, 2, 3, 4};
    for (; i < n; i += stride) out[i] = v;
}

int main(int argc, char **argv) {
    int n_loc = argc > 1 ? atoi(argv[1]) : 36864;
    int skew = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<int> idx(n_loc), order(n_loc);
    srand(1);
    for (int i = 0; i < n_loc; ++i) { int k = rand() % K; if (skew && (rand() % 10)) k = k % 3; idx[i] = k; order[i] = i; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return idx[a] < idx[b]; });
    u32x4 *table, *out; int *didx, *dorder;
    size_t out_bytes = (size_t)n_loc * VPP * 16;
    CK(hipMalloc(&table, K * VPP * 16)); CK(hipMalloc(&out, out_bytes)); CK(hipMalloc(&didx, n_loc * 4)); CK(hipMalloc(&dorder, n_loc * 4));
    std::vector<unsigned short> ht(K * VPP * 8);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = (unsigned short)(i * 2654435761u >> 16);
    CK(hipMemcpy(table, ht.data(), K * VPP * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(didx, idx.data(), n_loc * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dorder, order.data(), n_loc * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned short> ref(VPP * 8), got(VPP * 8);
    auto run = [&](const char *name, auto launch) {
        CK(hipMemset(out, 0, out_bytes));
        launch(); CK(hipDeviceSynchronize());
        // check three locations
        bool ok = true;
        for (int loc : {0, n_loc / 2 + 1, n_loc - 1}) {
            CK(hipMemcpy(got.data(), out + (int64_t)loc * VPP, VPP * 16, hipMemcpyDeviceToHost));
            ok = ok && memcmp(got.data(), ht.data() + (size_t)idx[loc] * VPP * 8, VPP * 16) == 0;
        }
        float best = 1e9, tot = 0;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); tot += ms;
        }
        printf("%-34s %s  avg %.3f ms  %.0f GB/s   best %.3f ms  %.0f GB/s\n", name, ok ? "ok " : "BAD", tot / 7, out_bytes / (tot / 7) / 1e6, best, out_bytes / best / 1e6);
        fflush(stdout);
    };
    printf("n_loc %d (%.2f GB) skew %d\n", n_loc, out_bytes / 1e9, skew);
    run("fill (no reads)", [&] { fill_kernel<<<4096, 256>>>(out, (int64_t)n_loc * VPP); });
    run("block/loc nt (current)", [&] { v_block_per_loc<true><<<n_loc, 256>>>(table, didx, out); });
    run("block/loc plain", [&] { v_block_per_loc<false><<<n_loc, 256>>>(table, didx, out); });
    for (int lpb : {1, 2, 3, 6}) {
        char nm[64];
        snprintf(nm, 64, "sorted regs T=1024 nt    lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<1024, true><<<(n_loc + lpb - 1) / lpb, 1024>>>(table, didx, dorder, n_loc, lpb, out); });
        snprintf(nm, 64, "sorted regs T=512  nt    lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<512, true><<<(n_loc + lpb - 1) / lpb, 512>>>(table, didx, dorder, n_loc, lpb, out); });
        snprintf(nm, 64, "sorted regs T=256  nt    lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<256, true><<<(n_loc + lpb - 1) / lpb, 256>>>(table, didx, dorder, n_loc, lpb, out); });
    }
    for (int lpb : {4}) {
        char nm[64];
        snprintf(nm, 64, "sorted regs T=1024 plain lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<1024, false><<<(n_loc + lpb - 1) / lpb, 1024>>>(table, didx, dorder, n_loc, lpb, out); });
        snprintf(nm, 64, "sorted regs T=1024 nt    lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<1024, true><<<(n_loc + lpb - 1) / lpb, 1024>>>(table, didx, dorder, n_loc, lpb, out); });
        snprintf(nm, 64, "sorted regs T=512  plain lpb=%d", lpb);
        run(nm, [&] { v_sorted_regs<512, false><<<(n_loc + lpb - 1) / lpb, 512>>>(table, didx, dorder, n_loc, lpb, out); });
    }
    run("hipMemsetAsync", [&] { CK(hipMemsetAsync(out, 1, out_bytes, 0)); });
    return 0;
}
