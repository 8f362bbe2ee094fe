// Which XCD runs workgroup i?  (s_getreg HW_REG_XCC_ID; 64-thread workgroups, like k_match_wq)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    unsigned cu;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = cu; }
}
int main() {
    const int n = 1 << 16;
    unsigned* d; hipMalloc(&d, n * 8);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
    unsigned* h = (unsigned*)malloc(n * 8);
    hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
    printf("first 32 workgroups -> XCC_ID:");
    for (int i = 0; i < 32; ++i) printf(" %u", h[2 * i] & 0xF);
    printf("\n");
    int ok8 = 0;
    for (int i = 0; i + 8 < n; ++i) ok8 += ((h[2 * i] & 0xF) == (h[2 * (i + 8)] & 0xF));
    printf("fraction with xcc(i) == xcc(i+8): %.4f\n", (double)ok8 / (n - 8));
    int okmod = 0;
    for (int i = 0; i < n; ++i) okmod += ((h[2 * i] & 0xF) == (unsigned)(i % 8));
    printf("fraction with xcc(i) == i %% 8: %.4f\n", (double)okmod / n);
    return 0;
}
