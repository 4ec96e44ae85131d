// Does the K = 16 form of the 16 x 16 MFMA cost half of the K = 32 form on gfx950?  (If it did, kernels whose taps do not fill a K = 32 step - c3d: three
// x-offsets x 8 channels + one zero k-group - could pack two taps per K = 16 step instead.)  One wave per SIMD, nine independent accumulators, cycles per MFMA
// from s_memtime.  Measured (MI355X, round 5): 17.22 against 17.18 ticks - the same; only the K = 32 form has the doubled rate.
//   hipcc -O3 --offload-arch=gfx950 mfma_k16_probe.hip -o /tmp/k16 && /tmp/k16
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int mode) {
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (mode == 0) {
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]\n\t"
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]\n\t"
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]"
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","v0","v1","v2","v3");
    } else {
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x16_f16 a[0:3], a[16:17], v[0:1], a[0:3]\n\tv_mfma_f32_16x16x16_f16 a[4:7], a[16:17], v[0:1], a[4:7]\n\tv_mfma_f32_16x16x16_f16 a[8:11], a[16:17], v[0:1], a[8:11]\n\t"
                "v_mfma_f32_16x16x16_f16 a[0:3], a[16:17], v[0:1], a[0:3]\n\tv_mfma_f32_16x16x16_f16 a[4:7], a[16:17], v[0:1], a[4:7]\n\tv_mfma_f32_16x16x16_f16 a[8:11], a[16:17], v[0:1], a[8:11]\n\t"
                "v_mfma_f32_16x16x16_f16 a[0:3], a[16:17], v[0:1], a[0:3]\n\tv_mfma_f32_16x16x16_f16 a[4:7], a[16:17], v[0:1], a[4:7]\n\tv_mfma_f32_16x16x16_f16 a[8:11], a[16:17], v[0:1], a[8:11]"
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","v0","v1","v2","v3");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) out[mode] = t1 - t0;
}
int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, sizeof(h));
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 2; ++m) { hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, m); hipDeviceSynchronize(); }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("16x16x32 f16: %.2f s_memtime ticks per MFMA;  16x16x16 f16: %.2f\n", h[0] / (double)(REP * 9), h[1] / (double)(REP * 9));
    return 0;
}
