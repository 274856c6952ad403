// Prints what v_permlane32_swap does to two registers: lane i holds a = i, b = 100 + i before.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *p) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[threadIdx.x] = r[0];
    p[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("first : lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("second: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
