// Direct (VALU) convolution for the first layer of a patch network: 1-4 input channels.
//
// With Ci = 1 a 3x3x3 conv has K = 27: far too little contraction depth to pay for the MFMA
// engine's staging (the general kernel sat at 9 % matrix-pipe use on it), and on gfx950 the fp32
// MFMA shares the vector ALUs anyway.  Here one thread owns one output voxel and CO accumulators;
// the halo'd input block and the whole weight tensor sit in LDS (weights are read as wave-uniform
// broadcasts), the output row of a voxel is CO contiguous floats -> 16-byte stores that are
// contiguous across the lanes of a wave; bias / ReLU / channel sum are fused.
#include "alq_internal.h"

#include <cstring>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CO>
__global__ __launch_bounds__(256) void direct_conv_kernel(const DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int K = a.ntaps * a.Ci;
    float *Wl = lds;                       // [K][CO]
    float *Al = lds + ((K * CO + 3) & ~3); // [PT][HZ][HY][HX][Ci]
    for (int i = tid; i < K * CO; i += 256) Wl[i] = a.W[i];

    int t = blockIdx.x;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int tz = t % a.tilesZ; t /= a.tilesZ;
    const int p0 = t * a.PT;
    const int mz0 = tz * a.TZ, my0 = ty * a.TY, mx0 = tx * a.TX;
    const int nhv = a.PT * a.HZ * a.HY * a.HX;
    for (int i = tid; i < nhv * a.Ci; i += 256) {
        int r = i / a.Ci;
        const int c = i - r * a.Ci;
        const int hx = r % a.HX; r /= a.HX;
        const int hy = r % a.HY; r /= a.HY;
        const int hz = r % a.HZ; r /= a.HZ;
        const int patch = p0 + r;
        const int iz = mz0 + a.minz + hz, iy = my0 + a.miny + hy, ix = mx0 + a.minx + hx;
        float v = 0.f;
        if (patch < a.N && iz >= 0 && iz < a.ID && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
            v = a.in[((((long long)patch * a.ID + iz) * a.IH + iy) * a.IW + ix) * a.in_cs + a.in_c0 + c];
        Al[i] = v;
    }
    __syncthreads();

    // this thread's output voxel
    const int TV = a.TZ * a.TY * a.TX;
    const int v = tid;
    const int pt = v / TV;
    int q = v - pt * TV;
    const int x = q % a.TX; q /= a.TX;
    const int y = q % a.TY;
    const int z = q / a.TY;
    const bool live = v < a.rows && p0 + pt < a.N && mz0 + z < a.MD && my0 + y < a.MH && mx0 + x < a.MW;
    const int base = (((pt * a.HZ + z) * a.HY + y) * a.HX + x) * a.Ci;

    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = a.bias ? a.bias[c] : 0.f;
    int kk = 0;
    for (int dz = 0; dz < a.tnz; ++dz)
        for (int dy = 0; dy < a.tny; ++dy) {
            const int rowoff = base + ((dz * a.HY + dy) * a.HX) * a.Ci;
            for (int dxc = 0; dxc < a.tnx * a.Ci; ++dxc, ++kk) {      // (dx, ci) are contiguous in the halo row
                const float xv = Al[rowoff + dxc];
                const float *w = Wl + kk * CO;
#pragma unroll
                for (int c = 0; c < CO; c += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + c);      // wave-uniform address: LDS broadcast
                    acc[c] = fmaf(xv, wv.x, acc[c]);
                    acc[c + 1] = fmaf(xv, wv.y, acc[c + 1]);
                    acc[c + 2] = fmaf(xv, wv.z, acc[c + 2]);
                    acc[c + 3] = fmaf(xv, wv.w, acc[c + 3]);
                }
            }
        }
    if (!live) return;
    const long long ovox = (((long long)(p0 + pt) * a.OD + mz0 + z) * a.OH + my0 + y) * a.OW + mx0 + x;
    float *orow = a.out + ovox * a.out_cs + a.out_c0;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
        f32x4 o = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
        if (a.relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<f32x4 *>(orow + c) = o;
        sum += (o.x + o.y) + (o.z + o.w);
    }
    if (a.osum) a.osum[ovox] = sum;
}

int direct_build_plan(const IgemmPlan &p1, DirectPlan *dp) {
    dp->ok = false;
    const IgemmArgs &g = p1.a;
    if (!p1.smallc || g.Ci > 4 || g.sm != 1 || g.so != 1 || g.ooffz || g.ooffy || g.ooffx) return ALQ_OK;
    if (!(g.Co == 8 || g.Co == 16 || g.Co == 24 || g.Co == 32)) return ALQ_OK;
    if (g.rows != 256) return ALQ_OK;
    // the taps must be the full box [min, min+n) in x-fastest order (a forward conv's are)
    const int tnx = g.HX - g.TX + 1, tny = g.HY - g.TY + 1, tnz = g.HZ - g.TZ + 1;
    if (tnx * tny * tnz != g.ntaps) return ALQ_OK;
    for (int t = 0; t < g.ntaps; ++t) {
        const int ix = t % tnx, iy = (t / tnx) % tny, iz = t / (tnx * tny);
        if (p1.h_koff[t * g.Ci] != ((iz * g.HY + iy) * g.HX + ix) * g.Ci) return ALQ_OK;
    }
    DirectArgs &a = dp->a;
    std::memset(&a, 0, sizeof(a));
    a.Ci = g.Ci; a.ID = g.ID; a.IH = g.IH; a.IW = g.IW;
    a.Co = g.Co; a.OD = g.OD; a.OH = g.OH; a.OW = g.OW;
    a.MD = g.MD; a.MH = g.MH; a.MW = g.MW;
    a.PT = g.PT; a.TZ = g.TZ; a.TY = g.TY; a.TX = g.TX; a.HZ = g.HZ; a.HY = g.HY; a.HX = g.HX; a.rows = g.rows;
    a.minz = g.minz; a.miny = g.miny; a.minx = g.minx;
    a.ntaps = g.ntaps; a.tnx = tnx; a.tny = tny; a.tnz = tnz;
    a.tilesZ = g.tilesZ; a.tilesY = g.tilesY; a.tilesX = g.tilesX;
    const int K = g.ntaps * g.Ci;
    dp->lds_bytes = ((size_t)((K * g.Co + 3) & ~3) + (size_t)g.PT * g.HZ * g.HY * g.HX * g.Ci) * 4;
    if (dp->lds_bytes > 60 * 1024) return ALQ_OK;
    dp->flops_per_patch = p1.flops_per_patch;
    dp->ok = true;
    return ALQ_OK;
}

int direct_launch(alq_ctx *ctx, const DirectPlan &dp, const View &in, const View &out, const float *bias, int relu,
                  int N, float *osum, int prof_cls) {
    DirectArgs a = dp.a;
    ALQ_REQUIRE(in.C == a.Ci && in.D == a.ID && in.H == a.IH && in.W == a.IW && out.C == a.Co && out.D == a.OD &&
                    out.H == a.OH && out.W == a.OW,
                ALQ_EINVAL, "direct conv: views do not match the plan");
    ALQ_REQUIRE(out.cs % 4 == 0 && out.c0 % 4 == 0 && dp.d_W, ALQ_EINVAL, "direct conv: bad output slice or weights");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.W = dp.d_W; a.bias = bias; a.relu = relu; a.N = N; a.osum = osum;
    const int pgroups = (N + a.PT - 1) / a.PT;
    const dim3 grid((unsigned)(pgroups * a.tilesZ * a.tilesY * a.tilesX));
    ProfScope ps(ctx, prof_cls, dp.flops_per_patch * N);
    switch (a.Co) {
        case 8: hipLaunchKernelGGL(direct_conv_kernel<8>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 16: hipLaunchKernelGGL(direct_conv_kernel<16>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 24: hipLaunchKernelGGL(direct_conv_kernel<24>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        case 32: hipLaunchKernelGGL(direct_conv_kernel<32>, grid, dim3(256), dp.lds_bytes, ctx->stream, a); break;
        default: set_error("direct conv: Co=%d", a.Co); return ALQ_EUNSUPPORTED;
    }
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
