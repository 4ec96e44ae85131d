// Standalone timing of the plane-sweep kernels (csrc/c3d.hip) on synthetic data, for A/B builds with -D switches (C3_PF, C3_WD,
// C3_SCHED_MASK, C3_ABL ...).  GPU box only:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Iinclude -Inn-active-learning_amd/csrc [-D...] tools/probe/c3d_bench.hip -o /tmp/c3b && /tmp/c3b [N]
#include "../../nn-active-learning_amd/csrc/c3d.hip"

#include <cstdarg>
#include <random>

namespace alq { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
int alq_ctx::prof_begin(int, hipEvent_t *, hipEvent_t *) { return 1; }
void alq_ctx::prof_end(int, hipEvent_t, hipEvent_t, double) {}
int alq_ctx::prof_collect() { return 0; }

__global__ void fill_kernel(float *p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float u = (float)(h & 0xffffff) / 16777216.f - 0.5f;
        p[i] = u * scale;
    }
}
__global__ void fill_bytes(unsigned char *p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (unsigned char)(h & 15u);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    using namespace alq;
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    const size_t F = 32 * 32 * 32 * 8;
    float *inA, *inB, *wd, *bias, *part, *asum, *dB, *sA, *sB;
    unsigned *amax;
    unsigned char *bits, *maskA;
    CK(hipMalloc(&inA, N * F * 4)); CK(hipMalloc(&inB, N * F * 4)); CK(hipMalloc(&wd, F * 4)); CK(hipMalloc(&bias, 64));
    CK(hipMalloc(&part, N * 16)); CK(hipMalloc(&asum, N * 16)); CK(hipMalloc(&amax, N * 4)); CK(hipMalloc(&bits, N * F / 4)); CK(hipMalloc(&maskA, N * F / 4));
    CK(hipMalloc(&dB, N * F * 4)); CK(hipMalloc(&sA, N * F / 2)); CK(hipMalloc(&sB, N * F / 2));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, inA, N * F, 1u, 6.f);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, inB, N * F, 2u, 6.f);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, wd, F, 3u, 0.01f);
    hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, 0, bias, (size_t)16, 4u, 0.1f);
    hipLaunchKernelGGL(fill_bytes, dim3(2048), dim3(256), 0, 0, maskA, N * F / 4, 5u);
    std::vector<unsigned> hm(N);
    const float mx = 3.5f;
    for (int i = 0; i < N; ++i) std::memcpy(&hm[i], &mx, 4);
    CK(hipMemcpy(amax, hm.data(), N * 4, hipMemcpyHostToDevice));
    // weights: He-like
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 0.068f);
    std::vector<float> B((size_t)27 * 16 * 8);
    for (float &w : B) w = nd(rng);
    C3dPlan pf, pb;
    pf.ok = pb.ok = true; pf.D = pb.D = 32; pf.oneacc = pb.oneacc = 1;
    c3d_fwd_pack(&pf, B);
    c3d_bwd_pack(&pb, B);
    CK(hipMalloc(&pf.d_W, pf.h_W.size() * 2)); CK(hipMemcpy(pf.d_W, pf.h_W.data(), pf.h_W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&pb.d_W, pb.h_W.size() * 2)); CK(hipMemcpy(pb.d_W, pb.h_W.data(), pb.h_W.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hwd(F);
    CK(hipMemcpy(hwd.data(), wd, F * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned short> v16;
    c3d_presplit_vec(hwd.data(), (long long)F, 20, &v16);
    void *dv16;
    CK(hipMalloc(&dv16, v16.size() * 2)); CK(hipMemcpy(dv16, v16.data(), v16.size() * 2, hipMemcpyHostToDevice));

    C3FwdArgs a;
    a.inA = inA; a.inB = inB; a.W = pf.d_W; a.bias = bias; a.amaxA = amax; a.amaxB = amax; a.fc_W = wd; a.fc_part = part; a.asum_part = asum;
    a.fc_bits = bits; a.N = N; a.D = 32; a.e_w = pf.w_exp; a.flip_tau = 1e-6f;
    unsigned long long *dclk;
    CK(hipMalloc(&dclk, 256 * 16)); CK(hipMemset(dclk, 0, 256 * 16));
    a.clk = dclk;
    C3BwdArgs b;
    b.bits = bits; b.vec = dv16; b.W = pb.d_W; b.maskA = maskA; b.dB = dB; b.sumA = sA; b.sumB = sB; b.N = N; b.D = 32; b.e_in = 20; b.e_w = pb.w_exp;
    auto kf = c3d_fwd_kernel<true, true>;
#ifndef C3_BRW
#define C3_BRW 4
#endif
    auto kb = c3d_bwd_kernel<true, C3_BRW>;
    const int B3L = 2 * (4 * C3_BRW + 2) * B3_ROW + 2 * B3_ROW;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, B3L));
    const unsigned grid = (unsigned)std::min(N, 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(kf, dim3(grid), dim3(256), C3_LDS, 0, a);
            else hipLaunchKernelGGL(kb, dim3((unsigned)std::min(N * (8 / C3_BRW), 256)), dim3(256), B3L, 0, b);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { best = std::min(best, ms); sum += ms; }
        }
        printf("%s N=%d: avg %.1f us  best %.1f us", which == 0 ? "fwd" : "bwd", N, sum / reps * 1e3, best * 1e3);
#ifdef C3_CLK
        if (which == 0) {
            std::vector<unsigned long long> hc(512);
            CK(hipMemcpy(hc.data(), dclk, 512 * 8, hipMemcpyDeviceToHost));
            std::vector<double> cyc, ghz;
            for (unsigned g = 0; g < grid; ++g) if (hc[2 * g + 1]) { cyc.push_back((double)hc[2 * g]); ghz.push_back((double)hc[2 * g] / ((double)hc[2 * g + 1] * 10.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
            const double steps = 33.0 * ((N + grid - 1) / grid);
            printf("  | median WG cycles %.0f (%.0f per step), clock %.3f GHz", cyc[cyc.size() / 2], cyc[cyc.size() / 2] / steps, ghz[ghz.size() / 2]);
        }
#endif
        printf("\n");
    }
    return 0;
}
