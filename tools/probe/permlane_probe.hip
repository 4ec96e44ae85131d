#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w) { printf("out%d:", w); for (int i = 0; i < 64; i += 4) printf(" %d", h[w * 64 + i]); printf("\n"); }
    return 0;
}
