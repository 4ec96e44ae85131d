// How long does a dependent 16x16x32 MFMA wait for its accumulator when it does NOT accumulate in place?
//   A: c = mfma(a, b, c) three times in place (vDst == SrcC), next triple on another accumulator
//   B: the register allocator's pattern in c3d.hip: t = mfma(.., X); t = mfma(.., t); X = mfma(.., t)   (vDst != SrcC twice)
//   C: like B but the three triples of a group interleaved (dependent distance 3 MFMAs)
// one wave per SIMD, cycles per MFMA from s_memtime.   hipcc -O3 --offload-arch=gfx950 mfma_chain_probe.hip -o /tmp/mcp && /tmp/mcp
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int mode) {
    unsigned long long t0 = 0, t1 = 0;
    // a[0:3], a[4:7], a[8:11] accumulators, a[12:15] temp, a[16:19] A operand, v[0:3] B operand (contents irrelevant)
    if (mode == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]\n\t"
                "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]\n\t"
                "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]"
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","v0","v1","v2","v3");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    } else if (mode == 1) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]\n\tv_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[12:15]\n\t"
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[4:7]\n\tv_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]\n\tv_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[12:15]\n\t"
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[8:11]\n\tv_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]\n\tv_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[12:15]"
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","v0","v1","v2","v3");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    } else if (mode == 2) {      // B with two vector instructions behind every MFMA (the sweep's 1 : 2 interleave)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int r = 0; r < REP; ++r)
            asm volatile(
#define F "\n\tv_add_f32 v4, v4, v5\n\tv_add_f32 v6, v6, v7\n\t"
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[0:3]" F "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]" F "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[12:15]" F
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[4:7]" F "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]" F "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[12:15]" F
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[8:11]" F "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]" F "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[12:15]" F
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","v0","v1","v2","v3","v4","v5","v6","v7");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    } else if (mode == 3) {      // A (in place) with the same fillers
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]" F "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]" F "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[0:3]" F
                "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]" F "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]" F "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[4:7]" F
                "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]" F "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]" F "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[8:11]" F
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","v0","v1","v2","v3","v4","v5","v6","v7");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    } else if (mode == 5 || mode == 6 || mode == 7) {      // in place, 1.5 fillers per MFMA: spread 2,1,2,1 (5) / clumped 3,0,3,0 (6) / 6,0,0,0 (7)
#define M(acc) "v_mfma_f32_16x16x32_f16 " acc ", a[16:19], v[0:3], " acc "\n\t"
#define V1 "v_add_f32 v4, v4, v5\n\t"
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        if (mode == 5)
            for (int r = 0; r < REP; ++r)
                asm volatile(M("a[0:3]") V1 V1 M("a[0:3]") V1 M("a[0:3]") V1 V1 M("a[4:7]") V1 M("a[4:7]") V1 V1 M("a[4:7]") V1 M("a[8:11]") V1 V1 M("a[8:11]") V1 M("a[8:11]") V1 V1 V1
                             ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","v0","v1","v2","v3","v4","v5");
        else if (mode == 6)
            for (int r = 0; r < REP; ++r)
                asm volatile(M("a[0:3]") V1 V1 V1 M("a[0:3]") M("a[0:3]") V1 V1 V1 M("a[4:7]") M("a[4:7]") V1 V1 V1 M("a[4:7]") M("a[8:11]") V1 V1 V1 M("a[8:11]") M("a[8:11]") V1 V1
                             ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","v0","v1","v2","v3","v4","v5");
        else
            for (int r = 0; r < REP; ++r)
                asm volatile(M("a[0:3]") V1 V1 V1 V1 V1 V1 M("a[0:3]") M("a[0:3]") M("a[4:7]") M("a[4:7]") V1 V1 V1 V1 V1 V1 M("a[4:7]") M("a[8:11]") M("a[8:11]") M("a[8:11]") V1 V1
                             ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","v0","v1","v2","v3","v4","v5");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    } else if (mode == 4) {                     // C: renamed accumulators, interleaved (dependent distance 3), with fillers
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
        for (int r = 0; r < REP; ++r)
            asm volatile(
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[0:3]" F "v_mfma_f32_16x16x32_f16 a[20:23], a[16:19], v[0:3], a[4:7]" F "v_mfma_f32_16x16x32_f16 a[24:27], a[16:19], v[0:3], a[8:11]" F
                "v_mfma_f32_16x16x32_f16 a[12:15], a[16:19], v[0:3], a[12:15]" F "v_mfma_f32_16x16x32_f16 a[20:23], a[16:19], v[0:3], a[20:23]" F "v_mfma_f32_16x16x32_f16 a[24:27], a[16:19], v[0:3], a[24:27]" F
                "v_mfma_f32_16x16x32_f16 a[0:3], a[16:19], v[0:3], a[12:15]" F "v_mfma_f32_16x16x32_f16 a[4:7], a[16:19], v[0:3], a[20:23]" F "v_mfma_f32_16x16x32_f16 a[8:11], a[16:19], v[0:3], a[24:27]" F
                ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","v0","v1","v2","v3","v4","v5","v6","v7");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[mode] = t1 - t0;
}
int main() {
    unsigned long long *d, h[8] = {0};
    hipMalloc(&d, 64); hipMemset(d, 0, 64);
    const char *names[8] = {"A in place", "B renamed (t <- X, t <- t, X <- t)", "B + 2 fillers per MFMA", "A + 2 fillers per MFMA", "C renamed, interleaved, + fillers", "A + 14 fillers per 9 MFMAs spread 2,1,2,1", "A + 14 per 9 clumped 3,0,3,0", "A + 14 per 9 clumped 6,0,0,6,0,0"};
    double best[8];
    for (int m = 0; m < 8; ++m) best[m] = 1e30;
    for (int round = 0; round < 6; ++round)          // (the first launches run on a cold instruction cache and a ramping clock)
        for (int m = 0; m < 8; ++m) {
            hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, m);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            const double c = (double)h[m] / (REP * 9.0);
            if (round > 0 && c < best[m]) best[m] = c;
        }
    for (int m = 0; m < 8; ++m) printf("%-48s %6.2f cycles per MFMA (best of 5)\n", names[m], best[m]);
    return 0;
}
