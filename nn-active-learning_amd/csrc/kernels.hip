// Non-GEMM kernels of the scoring path: pooling, ReLU-mask + channel sums, box-dot reductions
// (the factored `shrink_gradient(...,'sum')`), skinny fc, softmax, Fisher finalisation,
// patch gather + normalisation, synthetic patches.  All HBM/L2-bound: coalesced channels-last
// accesses, wave-shuffle + LDS block reductions, fp64 accumulation where it is free.
#include "alq_internal.h"

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BOX_SLAB = 1024;   // voxels per box-dot workgroup

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// 256-thread block sum; result valid in thread 0
__device__ inline double block_sum256(double v, double *sh /*[4]*/) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

#define ALQ_LAUNCH_CHECK() ALQ_HIP(hipGetLastError())

// =========================================================================== pooling
// window == stride (NN.py:1473-1477 is called with [2,2]; NN_extended.py:463-468), TF 'SAME':
// out = ceil(in/s), window origin o*s - lo, positions outside the tensor never win.
__global__ void pool_fwd_kernel(const float *in, int in_cs, int in_c0, int C, int ID, int IH, int IW,
                                float *out, int out_cs, int out_c0, int OD, int OH, int OW,
                                uint8_t *argmax, int wz, int wy, int wx, int lz, int ly, int lx,
                                long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = r % C; r /= C;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH; r /= OH;
        const int oz = r % OD; r /= OD;
        const long long n = r;
        float best = -INFINITY;
        int bidx = 0;
        for (int dz = 0; dz < wz; ++dz) {
            const int iz = oz * wz - lz + dz;
            if (iz < 0 || iz >= ID) continue;
            for (int dy = 0; dy < wy; ++dy) {
                const int iy = oy * wy - ly + dy;
                if (iy < 0 || iy >= IH) continue;
                for (int dx = 0; dx < wx; ++dx) {
                    const int ix = ox * wx - lx + dx;
                    if (ix < 0 || ix >= IW) continue;
                    const float v = in[((((n * ID + iz) * IH + iy) * IW + ix)) * in_cs + in_c0 + c];
                    if (v > best) { best = v; bidx = (dz * wy + dy) * wx + dx; }
                }
            }
        }
        out[(((n * OD + oz) * OH + oy) * OW + ox) * out_cs + out_c0 + c] = best;
        argmax[i] = (uint8_t)bidx;
    }
}

__global__ void pool_bwd_kernel(const float *dout, int do_cs, int do_c0, int C, int OD, int OH, int OW,
                                float *din, int di_cs, int di_c0, int ID, int IH, int IW,
                                const uint8_t *argmax, int wz, int wy, int wx, int lz, int ly, int lx,
                                int accumulate, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = r % C; r /= C;
        const int ix = r % IW; r /= IW;
        const int iy = r % IH; r /= IH;
        const int iz = r % ID; r /= ID;
        const long long n = r;
        const int oz = (iz + lz) / wz, oy = (iy + ly) / wy, ox = (ix + lx) / wx;
        float g = 0.f;
        if (oz < OD && oy < OH && ox < OW) {
            const int widx = (((iz + lz) - oz * wz) * wy + ((iy + ly) - oy * wy)) * wx + ((ix + lx) - ox * wx);
            const long long o = (((n * OD + oz) * OH + oy) * OW + ox);
            if (argmax[o * C + c] == widx) g = dout[o * do_cs + do_c0 + c];
        }
        float *dst = din + ((((n * ID + iz) * IH + iy) * IW + ix)) * di_cs + di_c0 + c;
        *dst = accumulate ? (*dst + g) : g;
    }
}


// float4-over-channels variants (C % 4 == 0, 16-byte aligned slices): one thread per (voxel, 4 channels)
__global__ void pool_fwd_vec_kernel(const float *in, int in_cs, int in_c0, int C4, int ID, int IH, int IW,
                                    float *out, int out_cs, int out_c0, int OD, int OH, int OW,
                                    uint8_t *argmax, int wz, int wy, int wx, int lz, int ly, int lx,
                                    long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c4 = r % C4; r /= C4;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH; r /= OH;
        const int oz = r % OD; r /= OD;
        const long long n = r;
        f32x4 best = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        for (int dz = 0; dz < wz; ++dz) {
            const int iz = oz * wz - lz + dz;
            if (iz < 0 || iz >= ID) continue;
            for (int dy = 0; dy < wy; ++dy) {
                const int iy = oy * wy - ly + dy;
                if (iy < 0 || iy >= IH) continue;
                for (int dx = 0; dx < wx; ++dx) {
                    const int ix = ox * wx - lx + dx;
                    if (ix < 0 || ix >= IW) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(
                        in + ((((n * ID + iz) * IH + iy) * IW + ix)) * in_cs + in_c0 + c4 * 4);
                    const int w = (dz * wy + dy) * wx + dx;
                    if (v.x > best.x) { best.x = v.x; b0 = w; }
                    if (v.y > best.y) { best.y = v.y; b1 = w; }
                    if (v.z > best.z) { best.z = v.z; b2 = w; }
                    if (v.w > best.w) { best.w = v.w; b3 = w; }
                }
            }
        }
        *reinterpret_cast<f32x4 *>(out + (((n * OD + oz) * OH + oy) * OW + ox) * out_cs + out_c0 + c4 * 4) = best;
        reinterpret_cast<unsigned *>(argmax)[i] = (unsigned)b0 | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
    }
}

template <int G>   // G > 0: C4 == G lanes per voxel (power of two): mask + channel-sum fusion possible
__global__ void pool_bwd_vec_kernel(const float *dout, int do_cs, int do_c0, int C4, int OD, int OH, int OW,
                                    float *din, int di_cs, int di_c0, int ID, int IH, int IW,
                                    const uint8_t *argmax, int wz, int wy, int wx, int lz, int ly, int lx,
                                    int accumulate, long long total, const float *act, int a_cs, int a_c0,
                                    float *dsum, const unsigned char *asg) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < ((total + 63) & ~63LL);
         i += (long long)gridDim.x * blockDim.x) {
        if (i >= total) {      // keep whole waves in the shuffles below
            if constexpr (G > 1) {
                float z = 0.f;
#pragma unroll
                for (int o = 1; o < G; o <<= 1) z += __shfl_xor(z, o, 64);
            }
            continue;
        }
        long long r = i;
        const int c4 = r % C4; r /= C4;
        const int ix = r % IW; r /= IW;
        const int iy = r % IH; r /= IH;
        const int iz = r % ID; r /= ID;
        const long long n = r;
        const int oz = (iz + lz) / wz, oy = (iy + ly) / wy, ox = (ix + lx) / wx;
        f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
        if (oz < OD && oy < OH && ox < OW) {
            const unsigned widx = (((iz + lz) - oz * wz) * wy + ((iy + ly) - oy * wy)) * wx + ((ix + lx) - ox * wx);
            const long long o = (((n * OD + oz) * OH + oy) * OW + ox);
            const unsigned am = reinterpret_cast<const unsigned *>(argmax)[o * C4 + c4];
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dout + o * do_cs + do_c0 + c4 * 4);
            g.x = ((am & 255u) == widx) ? d.x : 0.f;
            g.y = (((am >> 8) & 255u) == widx) ? d.y : 0.f;
            g.z = (((am >> 16) & 255u) == widx) ? d.z : 0.f;
            g.w = ((am >> 24) == widx) ? d.w : 0.f;
        }
        const long long ivox = (((n * ID + iz) * IH + iy) * IW + ix);
        f32x4 *dst = reinterpret_cast<f32x4 *>(din + ivox * di_cs + di_c0 + c4 * 4);
        if (accumulate) g += *dst;
        if constexpr (G > 0) {
            if (asg) {        // the activation's sign field (View::sg): one byte instead of 16
                const unsigned nb = asg[(ivox * a_cs + a_c0 + c4 * 4) >> 2];
                g.x = (nb & 1u) ? g.x : 0.f; g.y = (nb & 2u) ? g.y : 0.f;
                g.z = (nb & 4u) ? g.z : 0.f; g.w = (nb & 8u) ? g.w : 0.f;
            } else if (act) {
                const f32x4 m = *reinterpret_cast<const f32x4 *>(act + ivox * a_cs + a_c0 + c4 * 4);
                g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
                g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
            }
        }
        *dst = g;
        if constexpr (G > 0) {
            if (dsum) {
                float sm = (g.x + g.y) + (g.z + g.w);
#pragma unroll
                for (int o = 1; o < G; o <<= 1) sm += __shfl_xor(sm, o, 64);
                if (c4 == 0) dsum[ivox] = sm;
            }
        }
    }
}

// ---- voxel-per-thread pooling (C = 4*C4 <= 32 channels): a thread owns all channels of one voxel, so the
// channel sums (osum forward, dsum backward) are in-thread and every access is a run of 16-byte vectors
template <int C4>
__global__ void pool_fwd_vox_kernel(const float *in, int in_cs, int in_c0, int ID, int IH, int IW, float *out,
                                    int out_cs, int out_c0, int OD, int OH, int OW, uint8_t *argmax, int wz, int wy,
                                    int wx, int lz, int ly, int lx, long long nvox, float *osum) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvox;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH; r /= OH;
        const int oz = r % OD; r /= OD;
        const long long n = r;
        f32x4 best[C4];
        unsigned bidx[C4];
#pragma unroll
        for (int c = 0; c < C4; ++c) { best[c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; bidx[c] = 0; }
        for (int dz = 0; dz < wz; ++dz) {
            const int iz = oz * wz - lz + dz;
            if (iz < 0 || iz >= ID) continue;
            for (int dy = 0; dy < wy; ++dy) {
                const int iy = oy * wy - ly + dy;
                if (iy < 0 || iy >= IH) continue;
                for (int dx = 0; dx < wx; ++dx) {
                    const int ix = ox * wx - lx + dx;
                    if (ix < 0 || ix >= IW) continue;
                    const float *src = in + ((((n * ID + iz) * IH + iy) * IW + ix)) * in_cs + in_c0;
                    const unsigned w = (dz * wy + dy) * wx + dx;
#pragma unroll
                    for (int c = 0; c < C4; ++c) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + c * 4);
                        if (v.x > best[c].x) { best[c].x = v.x; bidx[c] = (bidx[c] & 0xffffff00u) | w; }
                        if (v.y > best[c].y) { best[c].y = v.y; bidx[c] = (bidx[c] & 0xffff00ffu) | (w << 8); }
                        if (v.z > best[c].z) { best[c].z = v.z; bidx[c] = (bidx[c] & 0xff00ffffu) | (w << 16); }
                        if (v.w > best[c].w) { best[c].w = v.w; bidx[c] = (bidx[c] & 0x00ffffffu) | (w << 24); }
                    }
                }
            }
        }
        float *dst = out + i * out_cs + out_c0;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            *reinterpret_cast<f32x4 *>(dst + c * 4) = best[c];
            reinterpret_cast<unsigned *>(argmax)[i * C4 + c] = bidx[c];
            sum += (best[c].x + best[c].y) + (best[c].z + best[c].w);
        }
        if (osum) osum[i] = sum;
    }
}

template <int C4>
__global__ void pool_bwd_vox_kernel(const float *dout, int do_cs, int do_c0, int OD, int OH, int OW, float *din,
                                    int di_cs, int di_c0, int ID, int IH, int IW, const uint8_t *argmax, int wz,
                                    int wy, int wx, int lz, int ly, int lx, int accumulate, long long nvox,
                                    const float *act, int a_cs, int a_c0, float *dsum, int store_din) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvox;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int ix = r % IW; r /= IW;
        const int iy = r % IH; r /= IH;
        const int iz = r % ID; r /= ID;
        const long long n = r;
        const int oz = (iz + lz) / wz, oy = (iy + ly) / wy, ox = (ix + lx) / wx;
        const bool inside = oz < OD && oy < OH && ox < OW;
        const unsigned widx = (((iz + lz) - oz * wz) * wy + ((iy + ly) - oy * wy)) * wx + ((ix + lx) - ox * wx);
        const long long o = (((n * OD + oz) * OH + oy) * OW + ox);
        float *dst = din + i * di_cs + di_c0;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
            if (inside) {
                const unsigned am = reinterpret_cast<const unsigned *>(argmax)[o * C4 + c];
                const f32x4 d = *reinterpret_cast<const f32x4 *>(dout + o * do_cs + do_c0 + c * 4);
                g.x = ((am & 255u) == widx) ? d.x : 0.f;
                g.y = (((am >> 8) & 255u) == widx) ? d.y : 0.f;
                g.z = (((am >> 16) & 255u) == widx) ? d.z : 0.f;
                g.w = ((am >> 24) == widx) ? d.w : 0.f;
            }
            if (accumulate) g += *reinterpret_cast<const f32x4 *>(dst + c * 4);
            if (act) {
                const f32x4 m = *reinterpret_cast<const f32x4 *>(act + i * a_cs + a_c0 + c * 4);
                g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
                g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
            }
            if (store_din) *reinterpret_cast<f32x4 *>(dst + c * 4) = g;
            sum += (g.x + g.y) + (g.z + g.w);
        }
        if (dsum) dsum[i] = sum;
    }
}

// One thread per POOLED voxel (8 channels = C4 x 4): the cotangent of a pooled element belongs to the arg-max
// voxel of its window and survives the producer's ReLU iff the pooled activation is positive.  Only the
// channel sums per input voxel are produced (first parameterised layer: nothing upstream needs more).
template <int C4>
__global__ void pool_bwd_first_kernel(const float *dout, int do_cs, int do_c0, const float *pout, int po_cs, int po_c0,
                                      const uint8_t *argmax, int OD, int OH, int OW, int wz, int ID, int IH, int IW,
                                      long long npool, float *dsum, int accumulate, const unsigned char *psg) {
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < npool;
         o += (long long)gridDim.x * blockDim.x) {
        long long r = o;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH; r /= OH;
        const int oz = r % OD; r /= OD;
        const long long n = r;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            const unsigned am = reinterpret_cast<const unsigned *>(argmax)[o * C4 + c];
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dout + o * do_cs + do_c0 + c * 4);
            float g[4];
            if (psg) {        // the pooled activation's sign field (View::sg)
                const unsigned nb = psg[(o * po_cs + po_c0 + c * 4) >> 2];
                g[0] = (nb & 1u) ? d.x : 0.f; g[1] = (nb & 2u) ? d.y : 0.f; g[2] = (nb & 4u) ? d.z : 0.f; g[3] = (nb & 8u) ? d.w : 0.f;
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(pout + o * po_cs + po_c0 + c * 4);
                g[0] = a.x > 0.f ? d.x : 0.f; g[1] = a.y > 0.f ? d.y : 0.f; g[2] = a.z > 0.f ? d.z : 0.f; g[3] = a.w > 0.f ? d.w : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w = (am >> (8 * j)) & 255u;
#pragma unroll
                for (int p = 0; p < 8; ++p) s[p] += (w == (unsigned)p) ? g[j] : 0.f;
            }
        }
        for (int pz = 0; pz < wz; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const long long i = ((n * ID + oz * wz + pz) * IH + oy * 2 + py) * IW + ox * 2;
                f32x2 v = f32x2{s[(pz * 2 + py) * 2], s[(pz * 2 + py) * 2 + 1]};
                f32x2 *dst = reinterpret_cast<f32x2 *>(dsum + i);
                if (accumulate) v += *dst;
                *dst = v;
            }
    }
}

static unsigned grid_for(long long total, int block = 256, int cap = 256 * 32) {
    long long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

int k_pool_fwd(alq_ctx *ctx, const View &in, const View &out, uint8_t *argmax, const int w[3],
               const int lo[3], int N, float *osum, bool *fused) {
    const long long total = (long long)N * out.vox() * out.C;
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    if (fused) *fused = false;
    if (((in.cs | in.c0 | in.C | out.cs | out.c0) & 3) == 0 && (in.C == 4 || in.C == 8 || in.C == 16 || in.C == 32)) {
        const long long nvox = (long long)N * out.vox();
#define ALQ_PF(CV)                                                                                               \
    hipLaunchKernelGGL(pool_fwd_vox_kernel<CV>, dim3(grid_for(nvox)), dim3(256), 0, ctx->stream, in.p, in.cs, in.c0, \
                       in.D, in.H, in.W, out.p, out.cs, out.c0, out.D, out.H, out.W, argmax, w[0], w[1], w[2], lo[0], \
                       lo[1], lo[2], nvox, osum)
        switch (in.C) { case 4: ALQ_PF(1); break; case 8: ALQ_PF(2); break; case 16: ALQ_PF(4); break; default: ALQ_PF(8); }
#undef ALQ_PF
        ALQ_LAUNCH_CHECK();
        if (fused) *fused = osum != nullptr;
        return ALQ_OK;
    }
    if (((in.cs | in.c0 | in.C | out.cs | out.c0) & 3) == 0) {
        hipLaunchKernelGGL(pool_fwd_vec_kernel, dim3(grid_for(total / 4)), dim3(256), 0, ctx->stream, in.p, in.cs,
                           in.c0, in.C / 4, in.D, in.H, in.W, out.p, out.cs, out.c0, out.D, out.H, out.W, argmax,
                           w[0], w[1], w[2], lo[0], lo[1], lo[2], total / 4);
        ALQ_LAUNCH_CHECK();
        return ALQ_OK;
    }
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, in.p, in.cs,
                       in.c0, in.C, in.D, in.H, in.W, out.p, out.cs, out.c0, out.D, out.H, out.W, argmax,
                       w[0], w[1], w[2], lo[0], lo[1], lo[2], total);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_pool_bwd_first(alq_ctx *ctx, const View &dout, const View &pool_out, const uint8_t *argmax, const int w[3],
                     int ID, int IH, int IW, int N, float *dsum, int accumulate, int use_signs) {
    ALQ_REQUIRE(w[1] == 2 && w[2] == 2 && (w[0] == 1 || w[0] == 2) && ID == dout.D * w[0] && IH == dout.H * 2 && IW == dout.W * 2 &&
                    (dout.C == 4 || dout.C == 8 || dout.C == 16) && ((dout.cs | dout.c0 | pool_out.cs | pool_out.c0) & 3) == 0,
                ALQ_EUNSUPPORTED, "pool_bwd_first: unsupported geometry");
    const long long npool = (long long)N * dout.vox();
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
#define ALQ_PBF(CV)                                                                                               \
    hipLaunchKernelGGL(pool_bwd_first_kernel<CV>, dim3(grid_for(npool)), dim3(256), 0, ctx->stream, dout.p, dout.cs, \
                       dout.c0, pool_out.p, pool_out.cs, pool_out.c0, argmax, dout.D, dout.H, dout.W, w[0], ID, IH, IW, \
                       npool, dsum, accumulate, (use_signs && ((pool_out.cs | pool_out.c0) & 3) == 0) ? pool_out.sg : nullptr)
    switch (dout.C) { case 4: ALQ_PBF(1); break; case 8: ALQ_PBF(2); break; default: ALQ_PBF(4); }
#undef ALQ_PBF
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_pool_bwd(alq_ctx *ctx, const View &dout, const View &din, const uint8_t *argmax, const int w[3],
               const int lo[3], int N, int accumulate, const View *mask_act, float *dsum, bool *fused,
               int store_din, int use_signs) {
    const long long total = (long long)N * din.vox() * din.C;
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    if (fused) *fused = false;
    // voxel-per-thread form only where it measured faster: the finished cotangent is not stored (first
    // layer: sums only); otherwise the lane-per-4-channels form below keeps strided concat slices coalesced
    if (!store_din && dsum && ((din.cs | din.c0 | din.C | dout.cs | dout.c0) & 3) == 0 && (din.C == 4 || din.C == 8) &&
        (!mask_act || ((mask_act->cs | mask_act->c0) & 3) == 0)) {
        const long long nvox = (long long)N * din.vox();
        const float *ap = mask_act ? mask_act->p : nullptr;
        const int acs = mask_act ? mask_act->cs : 0, ac0 = mask_act ? mask_act->c0 : 0;
        const int st = (store_din || !dsum) ? 1 : 0;
#define ALQ_PBV(CV)                                                                                               \
    hipLaunchKernelGGL(pool_bwd_vox_kernel<CV>, dim3(grid_for(nvox)), dim3(256), 0, ctx->stream, dout.p, dout.cs,   \
                       dout.c0, dout.D, dout.H, dout.W, din.p, din.cs, din.c0, din.D, din.H, din.W, argmax, w[0], w[1], \
                       w[2], lo[0], lo[1], lo[2], accumulate, nvox, ap, acs, ac0, dsum, st)
        switch (din.C) { case 4: ALQ_PBV(1); break; case 8: ALQ_PBV(2); break; case 16: ALQ_PBV(4); break; default: ALQ_PBV(8); }
#undef ALQ_PBV
        ALQ_LAUNCH_CHECK();
        if (fused) *fused = dsum != nullptr;
        return ALQ_OK;
    }
    if (((din.cs | din.c0 | din.C | dout.cs | dout.c0) & 3) == 0) {
        const int C4 = din.C / 4;
        const bool can = (dsum != nullptr) && (C4 == 1 || C4 == 2 || C4 == 4 || C4 == 8) &&
                         (!mask_act || ((mask_act->cs | mask_act->c0) & 3) == 0);
        const float *ap = (can && mask_act) ? mask_act->p : nullptr;
        const int acs = mask_act ? mask_act->cs : 0, ac0 = mask_act ? mask_act->c0 : 0;
        float *ds = can ? dsum : nullptr;
        const unsigned char *asg = (ap && use_signs) ? mask_act->sg : nullptr;
#define ALQ_PB(GV)                                                                                              \
    hipLaunchKernelGGL(pool_bwd_vec_kernel<GV>, dim3(grid_for(total / 4)), dim3(256), 0, ctx->stream, dout.p,   \
                       dout.cs, dout.c0, dout.C / 4, dout.D, dout.H, dout.W, din.p, din.cs, din.c0, din.D,      \
                       din.H, din.W, argmax, w[0], w[1], w[2], lo[0], lo[1], lo[2], accumulate, total / 4, ap,   \
                       acs, ac0, ds, asg)
        if (!can) ALQ_PB(0);
        else if (C4 == 1) ALQ_PB(1);
        else if (C4 == 2) ALQ_PB(2);
        else if (C4 == 4) ALQ_PB(4);
        else ALQ_PB(8);
#undef ALQ_PB
        ALQ_LAUNCH_CHECK();
        if (fused) *fused = can;
        return ALQ_OK;
    }
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, dout.p, dout.cs,
                       dout.c0, dout.C, dout.D, dout.H, dout.W, din.p, din.cs, din.c0, din.D, din.H, din.W,
                       argmax, w[0], w[1], w[2], lo[0], lo[1], lo[2], accumulate, total);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== channel sums
// field[n, vox] = sum_c t[n, vox, c]; one thread per voxel, channel slice read as float4 when
// aligned (adjacent threads read adjacent voxel rows -> every fetched line is fully used).
__global__ void chansum_kernel(const float *t, int cs, int c0, int C, float *field, long long nvox) {
    const bool vec = ((cs | c0 | C) & 3) == 0;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nvox;
         v += (long long)gridDim.x * blockDim.x) {
        const float *row = t + v * cs + c0;
        float s = 0.f;
        if (vec) {
            for (int c = 0; c < C; c += 4) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(row + c);
                s += (q.x + q.y) + (q.z + q.w);
            }
        } else {
            for (int c = 0; c < C; ++c) s += row[c];
        }
        field[v] = s;
    }
}

// d <- d * (act > 0) in place (ReLUGrad, only when act != null) and field = channel sum of d
__global__ void mask_chansum_kernel(float *d, int cs, int c0, int C, const float *act, int acs, int ac0,
                                    float *field, long long nvox) {
    const bool vec = ((cs | c0 | C | acs | ac0) & 3) == 0;
    for (long long v = blockIdx.x * (long long)blockDim.x + threadIdx.x; v < nvox;
         v += (long long)gridDim.x * blockDim.x) {
        float *row = d + v * cs + c0;
        const float *arow = act ? act + v * acs + ac0 : nullptr;
        float s = 0.f;
        if (vec) {
            for (int c = 0; c < C; c += 4) {
                f32x4 q = *reinterpret_cast<f32x4 *>(row + c);
                if (arow) {
                    const f32x4 m = *reinterpret_cast<const f32x4 *>(arow + c);
                    q.x = m.x > 0.f ? q.x : 0.f;
                    q.y = m.y > 0.f ? q.y : 0.f;
                    q.z = m.z > 0.f ? q.z : 0.f;
                    q.w = m.w > 0.f ? q.w : 0.f;
                    *reinterpret_cast<f32x4 *>(row + c) = q;
                }
                s += (q.x + q.y) + (q.z + q.w);
            }
        } else {
            for (int c = 0; c < C; ++c) {
                float q = row[c];
                if (arow) {
                    q = arow[c] > 0.f ? q : 0.f;
                    row[c] = q;
                }
                s += q;
            }
        }
        field[v] = s;
    }
}

// Long rows (fc layers: one "voxel" per patch with thousands of channels): one workgroup per
// row, float4 loads, block reduction.
__global__ __launch_bounds__(256) void rowsum_kernel(float *d, int cs, int c0, int C, const float *act,
                                                     int acs, int ac0, float *field) {
    __shared__ double sh[4];
    const long long v = blockIdx.x;
    float *row = d + v * cs + c0;
    const float *arow = act ? act + v * acs + ac0 : nullptr;
    const bool vec = ((cs | c0 | C | acs | ac0) & 3) == 0;
    float s = 0.f;
    if (vec) {
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            f32x4 q = *reinterpret_cast<f32x4 *>(row + c);
            if (arow) {
                const f32x4 m = *reinterpret_cast<const f32x4 *>(arow + c);
                q.x = m.x > 0.f ? q.x : 0.f;
                q.y = m.y > 0.f ? q.y : 0.f;
                q.z = m.z > 0.f ? q.z : 0.f;
                q.w = m.w > 0.f ? q.w : 0.f;
                *reinterpret_cast<f32x4 *>(row + c) = q;
            }
            s += (q.x + q.y) + (q.z + q.w);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float q = row[c];
            if (arow) {
                q = arow[c] > 0.f ? q : 0.f;
                row[c] = q;
            }
            s += q;
        }
    }
    const double tot = block_sum256((double)s, sh);
    if (threadIdx.x == 0) field[v] = (float)tot;
}


// G lanes per voxel, each lane one float4 per step: consecutive lanes read consecutive 16 B, so a
// wave instruction covers whole 128-B lines; the G partial sums meet through xor-shuffles.
template <int G>
__global__ void chansum_grp_kernel(float *d, int cs, int c0, int C, const float *act, int acs, int ac0,
                                   float *field, long long nvox) {
    const long long total = nvox * G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < ((total + 63) & ~63LL);
         i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / G;
        const int sub = (int)(i - v * G);
        float s = 0.f;
        if (v < nvox) {
            float *row = d + v * cs + c0;
            const float *arow = act ? act + v * acs + ac0 : nullptr;
            for (int c = sub * 4; c < C; c += 4 * G) {
                f32x4 q = *reinterpret_cast<f32x4 *>(row + c);
                if (arow) {
                    const f32x4 m = *reinterpret_cast<const f32x4 *>(arow + c);
                    q.x = m.x > 0.f ? q.x : 0.f;
                    q.y = m.y > 0.f ? q.y : 0.f;
                    q.z = m.z > 0.f ? q.z : 0.f;
                    q.w = m.w > 0.f ? q.w : 0.f;
                    *reinterpret_cast<f32x4 *>(row + c) = q;
                }
                s += (q.x + q.y) + (q.z + q.w);
            }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0 && v < nvox) field[v] = s;
    }
}

static int lanes_per_voxel(int C) {
    const int q = C / 4;
    int g = 1;
    while (g < 8 && q % (g * 2) == 0) g *= 2;
    return g;
}

static int launch_chansum_grp(alq_ctx *ctx, float *d, int cs, int c0, int C, const float *act, int acs, int ac0,
                              float *field, long long nvox) {
    const int G = lanes_per_voxel(C);
    const unsigned grid = grid_for(nvox * G);
#define ALQ_CS(GV) \
    case GV: hipLaunchKernelGGL(chansum_grp_kernel<GV>, dim3(grid), dim3(256), 0, ctx->stream, d, cs, c0, C, act, acs, ac0, field, nvox); break
    switch (G) { ALQ_CS(1); ALQ_CS(2); ALQ_CS(4); ALQ_CS(8); }
#undef ALQ_CS
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_rowsum_field(alq_ctx *ctx, const float *field, int64_t len, int N, float *out) {
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    const int len_i = (int)len;
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)N), dim3(256), 0, ctx->stream, const_cast<float *>(field), len_i, 0,
                       len_i, (const float *)nullptr, 0, 0, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

constexpr int ROW_KERNEL_MIN_C = 512;

int k_chansum(alq_ctx *ctx, const View &in, float *field, int N) {
    const long long nvox = (long long)N * in.vox();
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    if (in.C >= ROW_KERNEL_MIN_C) {
        hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)nvox), dim3(256), 0, ctx->stream, in.p, in.cs, in.c0,
                           in.C, (const float *)nullptr, 0, 0, field);
        ALQ_LAUNCH_CHECK();
        return ALQ_OK;
    }
    if (((in.cs | in.c0 | in.C) & 3) == 0)
        return launch_chansum_grp(ctx, in.p, in.cs, in.c0, in.C, nullptr, 0, 0, field, nvox);
    hipLaunchKernelGGL(chansum_kernel, dim3(grid_for(nvox)), dim3(256), 0, ctx->stream, in.p, in.cs, in.c0,
                       in.C, field, nvox);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_mask_chansum(alq_ctx *ctx, const View &dact, const View *act, float *field, int N) {
    const long long nvox = (long long)N * dact.vox();
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    if (dact.C >= ROW_KERNEL_MIN_C) {
        hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)nvox), dim3(256), 0, ctx->stream, dact.p, dact.cs,
                           dact.c0, dact.C, act ? act->p : nullptr, act ? act->cs : 0, act ? act->c0 : 0, field);
        ALQ_LAUNCH_CHECK();
        return ALQ_OK;
    }
    if (((dact.cs | dact.c0 | dact.C | (act ? (act->cs | act->c0) : 0)) & 3) == 0)
        return launch_chansum_grp(ctx, dact.p, dact.cs, dact.c0, dact.C, act ? act->p : nullptr, act ? act->cs : 0,
                                  act ? act->c0 : 0, field, nvox);
    hipLaunchKernelGGL(mask_chansum_kernel, dim3(grid_for(nvox)), dim3(256), 0, ctx->stream, dact.p, dact.cs,
                       dact.c0, dact.C, act ? act->p : nullptr, act ? act->cs : 0, act ? act->c0 : 0, field,
                       nvox);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== box-dot reductions
// conv / fc:  S[n] = sum_x dsum[n,x] * (1 + sum_taps asum[n, x + tap - lo])   (zero outside)
// grid (slabs, N): each workgroup reduces a contiguous slab of BOX_SLAB voxels in fp64 and writes
// one partial; fisher_finalize adds the slab partials in slab order -> deterministic.
__global__ __launch_bounds__(256) void boxdot_conv_kernel(const float *dsum, const float *asum, const float *asum2,
                                                          int D, int H, int W, int kz, int ky, int kx, int lz,
                                                          int ly, int lx, double *Spart, int nslab_max) {
    __shared__ double sh[4];
    const long long n = blockIdx.y;
    const int slab = blockIdx.x;
    const int vox = D * H * W;
    const float *dn = dsum + n * vox;
    const float *an = asum + n * vox;
    const float *an2 = asum2 ? asum2 + n * vox : nullptr;
    double acc = 0;
    const int v1 = min(vox, (slab + 1) * BOX_SLAB);
    for (int v = slab * BOX_SLAB + threadIdx.x; v < v1; v += 256) {
        const float dv = dn[v];
        int r = v;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int z = r / H;
        float box = 0.f;
        for (int dz = 0; dz < kz; ++dz) {
            const int iz = z + dz - lz;
            if (iz < 0 || iz >= D) continue;
            for (int dy = 0; dy < ky; ++dy) {
                const int iy = y + dy - ly;
                if (iy < 0 || iy >= H) continue;
                const float *rowp = an + (iz * H + iy) * W;
                for (int dx = 0; dx < kx; ++dx) {
                    const int ix = x + dx - lx;
                    if (ix >= 0 && ix < W) box += rowp[ix];
                }
                if (an2) {
                    const float *rowq = an2 + (iz * H + iy) * W;
                    for (int dx = 0; dx < kx; ++dx) {
                        const int ix = x + dx - lx;
                        if (ix >= 0 && ix < W) box += rowq[ix];
                    }
                }
            }
        }
        acc += (double)dv * ((double)box + 1.0);
    }
    const double tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) Spart[n * nslab_max + slab] = tot;
}

// conv_transpose: S[n] = sum_q asum[n,q] * sum_t dsum[n, s*q + t - lo] + sum_p dsum[n,p]
// (q on the INPUT grid, p on the output grid = s * input grid); slabs over q
__global__ __launch_bounds__(256) void boxdot_convT_kernel(const float *dsum, const float *asum, const float *asum2,
                                                           int ID, int IH, int IW, int kz, int ky, int kx, int sz,
                                                           int sy, int sx, int lz, int ly, int lx,
                                                           double *Spart, int nslab_max) {
    __shared__ double sh[4];
    const long long n = blockIdx.y;
    const int slab = blockIdx.x;
    const int OD = ID * sz, OH = IH * sy, OW = IW * sx;
    const int ivox = ID * IH * IW;
    const float *dn = dsum + n * (long long)OD * OH * OW;
    const float *an = asum + n * ivox;
    double acc = 0;
    const int v1 = min(ivox, (slab + 1) * BOX_SLAB);
    for (int v = slab * BOX_SLAB + threadIdx.x; v < v1; v += 256) {
        int r = v;
        const int x = r % IW; r /= IW;
        const int y = r % IH;
        const int z = r / IH;
        float box = 0.f;
        for (int dz = 0; dz < kz; ++dz) {
            const int oz = z * sz + dz - lz;
            if (oz < 0 || oz >= OD) continue;
            for (int dy = 0; dy < ky; ++dy) {
                const int oy = y * sy + dy - ly;
                if (oy < 0 || oy >= OH) continue;
                const float *rowp = dn + ((long long)oz * OH + oy) * OW;
                for (int dx = 0; dx < kx; ++dx) {
                    const int ox = x * sx + dx - lx;
                    if (ox >= 0 && ox < OW) box += rowp[ox];
                }
            }
        }
        float own = 0.f;   // bias term: every output point belongs to exactly one input point
        for (int dz = 0; dz < sz; ++dz)
            for (int dy = 0; dy < sy; ++dy) {
                const float *rowp = dn + ((long long)(z * sz + dz) * OH + (y * sy + dy)) * OW + x * sx;
                for (int dx = 0; dx < sx; ++dx) own += rowp[dx];
            }
        const float av = an[v] + (asum2 ? asum2[n * ivox + v] : 0.f);
        acc += (double)av * (double)box + (double)own;
    }
    const double tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) Spart[n * nslab_max + slab] = tot;
}

// LDS-tiled variant: a workgroup owns ZS z-planes; the (zero-padded) planes of asum (+ asum2) it needs
// sit in LDS, so the k^3 window sum is k^3 LDS reads with no bounds logic.
__global__ __launch_bounds__(256) void boxdot_conv_lds_kernel(const float *dsum, const float *asum, const float *asum2,
                                                              int D, int H, int W, int kz, int ky, int kx, int lz,
                                                              int ly, int lx, int ZS, double *Spart, int nslab_max) {
    extern __shared__ float tile[];
    __shared__ double sh[4];
    const long long n = blockIdx.y;
    const int slab = blockIdx.x;
    const int z0 = slab * ZS;
    const int PH = H + ky - 1, PW = W + kx - 1, planes = ZS + kz - 1;
    const int vox = D * H * W;
    const float *an = asum + n * vox;
    const float *an2 = asum2 ? asum2 + n * vox : nullptr;
    const int tot = planes * PH * PW;
    {   // (px, py, pz) of element i = tid + 256 * it, advanced incrementally: a division by a run-time width per
        // element made this loop (and the one below) arithmetic-bound
        int r = threadIdx.x;
        int px = r % PW; r /= PW;
        int py = r % PH;
        int pz = r / PH;
        const int dx = 256 % PW, dy = (256 / PW) % PH, dz = 256 / (PW * PH);
        for (int i = threadIdx.x; i < tot; i += 256) {
            const int iz = z0 + pz - lz, iy = py - ly, ix = px - lx;
            float v = 0.f;
            if (iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                const int o = (iz * H + iy) * W + ix;
                v = an[o];
                if (an2) v += an2[o];
            }
            tile[i] = v;
            px += dx; py += dy; pz += dz;
            if (px >= PW) { px -= PW; ++py; }
            if (py >= PH) { py -= PH; ++pz; }
        }
    }
    __syncthreads();
    const float *dn = dsum + n * vox;
    double acc = 0;
    const int nv = ZS * H * W;
    {
        int r = threadIdx.x;
        int x = r % W; r /= W;
        int y = r % H;
        int zz = r / H;
        const int dx = 256 % W, dy = (256 / W) % H, dz = 256 / (W * H);
        for (int i = threadIdx.x; i < nv; i += 256) {
            if (z0 + zz >= D) break;
            float box = 0.f;
            for (int ez = 0; ez < kz; ++ez)
                for (int ey = 0; ey < ky; ++ey) {
                    const float *rowp = tile + ((zz + ez) * PH + (y + ey)) * PW + x;
                    for (int ex = 0; ex < kx; ++ex) box += rowp[ex];
                }
            acc += (double)dn[((z0 + zz) * H + y) * W + x] * ((double)box + 1.0);
            x += dx; y += dy; zz += dz;
            if (x >= W) { x -= W; ++y; }
            if (y >= H) { y -= H; ++zz; }
        }
    }
    const double tot_s = block_sum256(acc, sh);
    if (threadIdx.x == 0) Spart[n * nslab_max + slab] = tot_s;
}

// Column variant for the usual odd windows on rows that are a multiple of 4 wide: a thread owns 4 x-adjacent voxels
// of one (y) row and walks z; the KY x KX window sums of a plane are KY rows of 64-bit LDS reads (4 + KX - 1 values,
// shared by the 4 voxels) and the KZ planes of a window are a rolling register sum, so a voxel costs
// ~KY * (4 + KX) / 8 LDS reads per plane instead of KZ * KY * KX.  With fewer than 256 columns the threads split the
// z planes of the slab among them.
template <int KZ, int KY, int KX>
__global__ __launch_bounds__(256) void boxdot_conv_col_kernel(const float *dsum, const float *asum, const float *asum2,
                                                              int D, int H, int W, int lz, int ly, int lx, int ZS,
                                                              double *Spart, int nslab_max) {
    extern __shared__ float tile[];
    __shared__ double sh[4];
    const long long n = blockIdx.y;
    const int slab = blockIdx.x;
    const int z0 = slab * ZS;
    const int PH = H + KY - 1, PW = W + KX - 1, planes = ZS + KZ - 1;
    const int vox = D * H * W;
    const float *an = asum + n * vox;
    const float *an2 = asum2 ? asum2 + n * vox : nullptr;
    const int tot = planes * PH * PW;
    for (int i = threadIdx.x; i < tot; i += 256) tile[i] = 0.f;      // the zero padding (and planes outside the volume)
    __syncthreads();
    {   // interior rows as 16-byte loads, 4 (8 with a second source) in flight per thread and trip
        const int W4 = W >> 2, rowq = H * W4, nq = planes * rowq;
        constexpr int U = 4;
        for (int base = threadIdx.x; base < nq; base += 256 * U) {
            f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = base + u * 256;
                dst[u] = -1;
                if (q < nq) {
                    const int pz = q / rowq, rq = q - pz * rowq;
                    const int iy = rq / W4, ix = (rq - iy * W4) * 4;
                    const int iz = z0 + pz - lz;
                    if (iz >= 0 && iz < D) {
                        const int o = (iz * H + iy) * W + ix;
                        v[u] = *reinterpret_cast<const f32x4 *>(an + o);
                        if (an2) {
                            const f32x4 w = *reinterpret_cast<const f32x4 *>(an2 + o);
                            v[u].x += w.x; v[u].y += w.y; v[u].z += w.z; v[u].w += w.w;
                        }
                        dst[u] = (pz * PH + iy + ly) * PW + ix + lx;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) {
                    float *t = tile + dst[u];
                    t[0] = v[u].x; t[1] = v[u].y; t[2] = v[u].z; t[3] = v[u].w;
                }
        }
    }
    __syncthreads();
    constexpr int NR = (4 + KX) & ~1;                 // values read per row: 4 + KX - 1 rounded up to even
    constexpr int ZR = 4;                             // z planes per thread and run
    const float *dn = dsum + n * vox;
    const int cols = (H * W) >> 2;
    const int ceff = cols < 256 ? cols : 256;
    const int zg = 256 / ceff;                        // thread groups along z
    const int nrun = (ZS + ZR - 1) / ZR;
    const int g = threadIdx.x / ceff;
    double acc = 0;
    if (g < zg) {
        for (int c = threadIdx.x - g * ceff; c < cols; c += ceff) {
            const int y = (c * 4) / W, x = (c * 4) - y * W;
            for (int run = g; run < nrun; run += zg) {
                const int zb = run * ZR;
                f32x4 d[ZR];
#pragma unroll
                for (int j = 0; j < ZR; ++j) {
                    const int zz = zb + j;
                    d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (zz < ZS && z0 + zz < D) d[j] = *reinterpret_cast<const f32x4 *>(dn + ((z0 + zz) * H + y) * W + x);
                }
                f32x4 s[KZ];
#pragma unroll
                for (int q = 0; q < KZ; ++q) s[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pp = 0; pp < ZR + KZ - 1; ++pp) {
                    const int p = zb + pp;
                    f32x4 P = {0.f, 0.f, 0.f, 0.f};
                    if (p < planes) {
#pragma unroll
                        for (int ey = 0; ey < KY; ++ey) {
                            const float *rowp = tile + (p * PH + (y + ey)) * PW + x;
                            float r[NR];
#pragma unroll
                            for (int j = 0; j < NR; j += 2) {
                                const f32x2 t = *reinterpret_cast<const f32x2 *>(rowp + j);
                                r[j] = t.x; r[j + 1] = t.y;
                            }
#pragma unroll
                            for (int ex = 0; ex < KX; ++ex) {
                                P.x += r[ex]; P.y += r[1 + ex]; P.z += r[2 + ex]; P.w += r[3 + ex];
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q + 1 < KZ; ++q) s[q] = s[q + 1];
                    s[KZ - 1] = P;
                    if (pp >= KZ - 1) {
                        f32x4 box = s[0];
#pragma unroll
                        for (int q = 1; q < KZ; ++q) { box.x += s[q].x; box.y += s[q].y; box.z += s[q].z; box.w += s[q].w; }
                        const f32x4 dd = d[pp - (KZ - 1)];
                        acc += (double)dd.x * ((double)box.x + 1.0) + (double)dd.y * ((double)box.y + 1.0) +
                               (double)dd.z * ((double)box.z + 1.0) + (double)dd.w * ((double)box.w + 1.0);
                    }
                }
            }
        }
    }
    const double tot_s = block_sum256(acc, sh);
    if (threadIdx.x == 0) Spart[n * nslab_max + slab] = tot_s;
}

static size_t boxdot_tile_bytes(int zs, int H, int W, const int k[3]) {
    return (size_t)(zs + k[0] - 1) * (H + k[1] - 1) * (W + k[2] - 1) * sizeof(float);
}
// z planes per workgroup: ~4096 voxels, fewer if the padded planes would not fit 48 KB of LDS
static int boxdot_zs(int D, int H, int W, const int k[3]) {
    int zs = 4096 / (H * W);
    if (zs < 1) zs = 1;
    if (zs > D) zs = D;
    while (zs > 1 && boxdot_tile_bytes(zs, H, W, k) > 48 * 1024) --zs;
    return zs;
}
static size_t boxdot_lds_bytes(int D, int H, int W, const int k[3]) {
    return boxdot_tile_bytes(boxdot_zs(D, H, W, k), H, W, k);
}
static bool boxdot_use_lds(int D, int H, int W, const int k[3]) {
    return (long long)D * H * W >= 512 && boxdot_lds_bytes(D, H, W, k) <= 48 * 1024;
}

int boxdot_slabs(long long vox) { return (int)((vox + BOX_SLAB - 1) / BOX_SLAB); }
int boxdot_conv_slabs(int D, int H, int W, const int k[3]) {
    if (boxdot_use_lds(D, H, W, k)) { const int zs = boxdot_zs(D, H, W, k); return (D + zs - 1) / zs; }
    return boxdot_slabs((long long)D * H * W);
}

int k_boxdot_conv(alq_ctx *ctx, const float *dsum, const float *asum, const float *asum2, int D, int H, int W,
                  const int k[3], const int lo[3], int N, double *Spart, int nslab_max) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    if (boxdot_use_lds(D, H, W, k)) {
        const int zs = boxdot_zs(D, H, W, k);
        const bool al16 = (((uintptr_t)dsum | (uintptr_t)asum | (uintptr_t)asum2) & 15) == 0;
#define ALQ_BOXCOL(KZ, KY, KX)                                                                                        \
    if (k[0] == KZ && k[1] == KY && k[2] == KX && (W & 3) == 0 && al16) {                                              \
        hipLaunchKernelGGL((boxdot_conv_col_kernel<KZ, KY, KX>), dim3((D + zs - 1) / zs, N), dim3(256),               \
                           boxdot_lds_bytes(D, H, W, k), ctx->stream, dsum, asum, asum2, D, H, W, lo[0], lo[1], lo[2], \
                           zs, Spart, nslab_max);                                                                      \
        ALQ_LAUNCH_CHECK();                                                                                            \
        return ALQ_OK;                                                                                                 \
    }
        ALQ_BOXCOL(3, 3, 3)
        ALQ_BOXCOL(1, 3, 3)
        ALQ_BOXCOL(1, 5, 5)
#undef ALQ_BOXCOL
        hipLaunchKernelGGL(boxdot_conv_lds_kernel, dim3((D + zs - 1) / zs, N), dim3(256), boxdot_lds_bytes(D, H, W, k),
                           ctx->stream, dsum, asum, asum2, D, H, W, k[0], k[1], k[2], lo[0], lo[1], lo[2], zs, Spart,
                           nslab_max);
        ALQ_LAUNCH_CHECK();
        return ALQ_OK;
    }
    hipLaunchKernelGGL(boxdot_conv_kernel, dim3(boxdot_slabs((long long)D * H * W), N), dim3(256), 0, ctx->stream,
                       dsum, asum, asum2, D, H, W, k[0], k[1], k[2], lo[0], lo[1], lo[2], Spart, nslab_max);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// conv_transpose, output-side form: S[n] = sum_p dsum[n,p] * (1 + sum_{q : p in window(q)} asum[n,q]).  The whole
// input-grid field asum (+ asum2) of the patch sits in LDS; the large field (dsum, on the s-times finer output
// grid) is streamed once with 16-byte loads.  Along a dimension the q's of an output point p are
// (p + lo - t) / s for the taps t = (p + lo) mod s, + s, ... < k: k / s of them on average.
template <int K, int S>
__global__ __launch_bounds__(256) void boxdot_convT_out_kernel(const float *dsum, const float *asum, const float *asum2,
                                                               int ID, int IH, int IW, int kz, int lz, int ly, int lx,
                                                               int ZS, double *Spart, int nslab_max) {
    extern __shared__ float tile[];
    __shared__ double sh[4];
    const long long n = blockIdx.y;
    const int slab = blockIdx.x;
    const int sz = kz == 1 ? 1 : S;                               // 2-D layers: depth 1, window 1, stride 1
    const int OD = ID * sz, OH = IH * S, OW = IW * S;
    const int ivox = ID * IH * IW;
    const float *an = asum + n * ivox;
    const float *an2 = asum2 ? asum2 + n * ivox : nullptr;
    const float *dn = dsum + n * (long long)OD * OH * OW;
    const bool al = (((uintptr_t)dn | (uintptr_t)an | (uintptr_t)an2) & 15) == 0;       // 16-byte loads possible
    auto ld4 = [al](const float *p) {
        return al ? *reinterpret_cast<const f32x4 *>(p) : f32x4{p[0], p[1], p[2], p[3]};
    };
    for (int i = threadIdx.x * 4; i < ivox; i += 1024) {          // ivox is a multiple of 4 (checked by the launcher)
        f32x4 v = ld4(an + i);
        if (an2) {
            const f32x4 w = ld4(an2 + i);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<f32x4 *>(tile + i) = v;
    }
    __syncthreads();
    const int z0 = slab * ZS;
    const int z1 = min(OD, z0 + ZS);
    const int OW4 = OW >> 2, ncol = OH * OW4;
    double acc = 0;
    const int ceff = ncol < 256 ? ncol : 256, zg = 256 / ceff, g = threadIdx.x / ceff;     // few columns: split the planes
    for (int c = threadIdx.x - g * ceff; c < ncol && g < zg; c += ceff) {     // a thread owns 4 x-adjacent output points of one row
        const int py = c / OW4, px = (c - py * OW4) * 4;
        for (int pz = z0 + g; pz < z1; pz += zg) {
            const f32x4 d = ld4(dn + ((long long)pz * OH + py) * OW + px);
            float box[4] = {0.f, 0.f, 0.f, 0.f};
            for (int tz = (pz + lz) % sz; tz < kz; tz += sz) {
                const int uz = pz + lz - tz, qz = uz / sz;
                if (uz < 0 || qz >= ID) continue;
#pragma unroll
                for (int iy = 0; iy < (K + S - 1) / S; ++iy) {
                    const int ty = (py + ly) % S + iy * S, uy = py + ly - ty, qy = uy / S;
                    if (ty >= K || uy < 0 || qy >= IH) continue;
                    const float *rowp = tile + (qz * IH + qy) * IW;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ix = 0; ix < (K + S - 1) / S; ++ix) {
                            const int tx = (px + j + lx) % S + ix * S, ux = px + j + lx - tx, qx = ux / S;
                            if (tx < K && ux >= 0 && qx < IW) box[j] += rowp[qx];
                        }
                }
            }
            acc += (double)d.x * ((double)box[0] + 1.0) + (double)d.y * ((double)box[1] + 1.0) +
                   (double)d.z * ((double)box[2] + 1.0) + (double)d.w * ((double)box[3] + 1.0);
        }
    }
    const double tot = block_sum256(acc, sh);
    if (threadIdx.x == 0) Spart[n * nslab_max + slab] = tot;
}

// window 3, stride 2 in y and x, and either the same in z or a flat (2-D) layer
static bool boxdot_convT_out_ok(int ID, int IH, int IW, const int k[3], const int s[3]) {
    const long long ivox = (long long)ID * IH * IW;
    const bool shape = k[1] == 3 && k[2] == 3 && s[1] == 2 && s[2] == 2 &&
                       ((k[0] == 3 && s[0] == 2) || (k[0] == 1 && s[0] == 1 && ID == 1));
    return shape && ivox % 4 == 0 && ivox * 4 <= 48 * 1024;
}
static int boxdot_convT_zs(int ID, int IH, int IW, const int s[3]) {
    int zs = 4096 / (IH * s[1] * IW * s[2]);
    if (zs < 1) zs = 1;
    if (zs > ID * s[0]) zs = ID * s[0];
    return zs;
}
int boxdot_convT_slabs(int ID, int IH, int IW, const int k[3], const int s[3]) {
    if (boxdot_convT_out_ok(ID, IH, IW, k, s)) { const int zs = boxdot_convT_zs(ID, IH, IW, s); return (ID * s[0] + zs - 1) / zs; }
    return boxdot_slabs((long long)ID * IH * IW);
}

int k_boxdot_convT(alq_ctx *ctx, const float *dsum, const float *asum, const float *asum2, int ID, int IH, int IW,
                   const int k[3], const int s[3], const int lo[3], int N, double *Spart, int nslab_max) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    if (boxdot_convT_out_ok(ID, IH, IW, k, s)) {
        const int zs = boxdot_convT_zs(ID, IH, IW, s);
        hipLaunchKernelGGL((boxdot_convT_out_kernel<3, 2>), dim3((ID * s[0] + zs - 1) / zs, N), dim3(256),
                           (size_t)ID * IH * IW * sizeof(float), ctx->stream, dsum, asum, asum2, ID, IH, IW, k[0], lo[0],
                           lo[1], lo[2], zs, Spart, nslab_max);
        ALQ_LAUNCH_CHECK();
        return ALQ_OK;
    }
    hipLaunchKernelGGL(boxdot_convT_kernel, dim3(boxdot_slabs((long long)ID * IH * IW), N), dim3(256), 0,
                       ctx->stream, dsum, asum, asum2, ID, IH, IW, k[0], k[1], k[2], s[0], s[1], s[2], lo[0], lo[1], lo[2],
                       Spart, nslab_max);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== skinny fc (out <= 8)
// logits[n, o] = b[o] + sum_f act[n, f] * Wp[o, f]; Wp rows are in activation-memory order.
// grid (slices, N): each workgroup reduces one contiguous slice of F; the slice partials are
// summed in slice order by fc_small_finish -> deterministic.
constexpr int FC_SLICE = 8192;
int fc_small_slices(int64_t F) { return (int)((F + FC_SLICE - 1) / FC_SLICE); }

// BITS: also emits the signs act > 0, one byte (low nibble) per 4 consecutive elements: the backward pass of a fc head
// on top of a ReLU conv needs nothing else of that tensor
template <int NOUT, bool BITS>
__global__ __launch_bounds__(256) void fc_small_fwd_kernel(const float *act, long long F, const float *Wp,
                                                           float *partials, int nslices, unsigned *maskbits) {
    __shared__ double sh[4];
    const int slice = blockIdx.x;
    const long long n = blockIdx.y;
    const long long f0 = (long long)slice * FC_SLICE;
    const long long f1 = (f0 + FC_SLICE < F) ? f0 + FC_SLICE : F;
    const float *a = act + n * F;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
    if ((F & 3) == 0) {
        for (long long f = f0 + threadIdx.x * 4; f < f1; f += 1024) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(a + f);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wp + o * F + f);
                acc[o] += (av.x * wv.x + av.y * wv.y) + (av.z * wv.z + av.w * wv.w);
            }
            if constexpr (BITS) {
                const unsigned nib = (av.x > 0.f ? 1u : 0u) | (av.y > 0.f ? 2u : 0u) | (av.z > 0.f ? 4u : 0u) | (av.w > 0.f ? 8u : 0u);
                reinterpret_cast<unsigned char *>(maskbits)[(n * F + f) >> 2] = (unsigned char)nib;
            }
        }
    } else {
        for (long long f = f0 + threadIdx.x; f < f1; f += 256) {
            const float av = a[f];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] += av * Wp[o * F + f];
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const double tot = block_sum256((double)acc[o], sh);
        if (threadIdx.x == 0) partials[(n * nslices + slice) * NOUT + o] = (float)tot;
    }
}

__global__ void fc_small_finish_kernel(const float *partials, int nslices, const float *bias, int nout,
                                       int relu, int N, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * nout) return;
    const int n = i / nout, o = i - n * nout;
    double s = 0;
    for (int k = 0; k < nslices; ++k) s += (double)partials[((long long)n * nslices + k) * nout + o];
    float v = (float)s + (bias ? bias[o] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    out[i] = v;
}

// one 64-lane wave per patch: lane l sums partials l, l + 64, ... in fp64, then a fixed butterfly
__global__ __launch_bounds__(64) void fc_small_finish_diff_kernel(const float *partials, int nslices, const float *bias, int N,
                                                                  float *out) {
    const int n = blockIdx.x;
    double s = 0;
    for (int k = threadIdx.x; k < nslices; k += 64) s += (double)partials[(long long)n * nslices + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
        out[2 * n] = (float)s + (bias ? bias[0] - bias[1] : 0.f);
        out[2 * n + 1] = 0.f;
    }
}
int k_fc_small_finish_diff(alq_ctx *ctx, const float *partials, int nslices, const float *bias, int N, float *out) {
    ProfScope ps(ctx, PROF_FC_SMALL, 0);
    hipLaunchKernelGGL(fc_small_finish_diff_kernel, dim3(N), dim3(64), 0, ctx->stream, partials, nslices, bias, N, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_fc_small_fwd(alq_ctx *ctx, const float *act, int64_t F, const float *Wp, int nout, int N,
                   float *partials, int nslices, unsigned *maskbits) {
    ProfScope ps(ctx, PROF_FC_SMALL, 2.0 * F * nout * N);
    dim3 grid(nslices, N);
    ALQ_REQUIRE(!maskbits || F % 1024 == 0, ALQ_EINVAL, "fc_small: mask bits need F %% 1024 == 0");
#define ALQ_FC(NO)                                                                                                     \
    case NO:                                                                                                           \
        if (maskbits)                                                                                                  \
            hipLaunchKernelGGL((fc_small_fwd_kernel<NO, true>), grid, dim3(256), 0, ctx->stream, act, (long long)F, Wp, \
                               partials, nslices, maskbits);                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((fc_small_fwd_kernel<NO, false>), grid, dim3(256), 0, ctx->stream, act, (long long)F,   \
                               Wp, partials, nslices, maskbits);                                                       \
        break
    switch (nout) {
        ALQ_FC(1); ALQ_FC(2); ALQ_FC(3); ALQ_FC(4); ALQ_FC(5); ALQ_FC(6); ALQ_FC(7); ALQ_FC(8);
        default: set_error("fc_small: nout=%d unsupported", nout); return ALQ_EUNSUPPORTED;
    }
#undef ALQ_FC
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_fc_small_finish(alq_ctx *ctx, const float *partials, int nslices, const float *bias, int nout,
                      int relu, int N, float *out) {
    ProfScope ps(ctx, PROF_FC_SMALL, 0);
    hipLaunchKernelGGL(fc_small_finish_kernel, dim3((N * nout + 255) / 256), dim3(256), 0, ctx->stream,
                       partials, nslices, bias, nout, relu, N, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// dact[n, f] = sum_o delta[n, o] * Wp[o, f]; G > 0: rows are [voxel][C = 4G]: ReLU-grad mask by act and
// per-voxel channel sums fused (G lanes per voxel meet through xor-shuffles)
template <int G>
__global__ void fc_small_bwd_kernel(const float *delta, int nout, const float *Wp, long long F, int N,
                                    float *dact, const float *act, float *dsum) {
    const long long total4 = (long long)N * (F >> 2);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < ((total4 + 63) & ~63LL);
         i += (long long)gridDim.x * blockDim.x) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool in = i < total4;
        if (in) {
            const long long n = i / (F >> 2);
            const long long f = (i - n * (F >> 2)) << 2;
            for (int o = 0; o < nout; ++o) {
                const float dv = delta[n * nout + o];
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(Wp + o * F + f);
                s += dv * wv;
            }
            if constexpr (G > 0) {
                if (act) {
                    const f32x4 m = *reinterpret_cast<const f32x4 *>(act + n * F + f);
                    s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f;
                    s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
                }
            }
            *reinterpret_cast<f32x4 *>(dact + n * F + f) = s;
        }
        if constexpr (G > 0) {
            float sm = (s.x + s.y) + (s.z + s.w);
#pragma unroll
            for (int o = 1; o < G; o <<= 1) sm += __shfl_xor(sm, o, 64);
            if (in && dsum && (i & (G - 1)) == 0) dsum[i / G] = sm;
        }
    }
}
__global__ void fc_small_bwd_scalar_kernel(const float *delta, int nout, const float *Wp, long long F, int N,
                                           float *dact) {
    const long long total = (long long)N * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / F, f = i - n * F;
        float s = 0.f;
        for (int o = 0; o < nout; ++o) s += delta[n * nout + o] * Wp[o * F + f];
        dact[i] = s;
    }
}

// The cotangent of a fc head's input when the head's own cotangent is the same for every patch (the unit cotangent of
// the Fisher pass): dact[n, f] = [act[n, f] > 0] * wv[f] with wv = sum_o delta[o] * Wp[o, :].  It is never stored:
// wv (set with the weights) and the sign bytes of the forward pass are all the consuming contraction needs (igemm4 BITSRC),
// and the per-voxel channel sums come from the same two (fc_small_dsum_bits, C = 8 channels = one byte of bits).
// grid (groups of 4 voxels / 256, patch groups): a thread keeps the 32 vector values of its 4 voxels (8 channels each, 8
// sign bytes) in registers for PG patches
constexpr int FC_BITS_PG = 8;
__global__ __launch_bounds__(256) void fc_small_dsum_bits_kernel(const unsigned *maskbits, const float *wv, long long F, int N,
                                                                 float *dsum) {
    const long long words = F >> 5;
    const long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x;      // 32 elements = 4 voxels x 8 channels = 8 bytes
    if (w >= words) return;
    f32x4 a[4], b[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        a[v] = *reinterpret_cast<const f32x4 *>(wv + (w << 5) + v * 8);
        b[v] = *reinterpret_cast<const f32x4 *>(wv + (w << 5) + v * 8 + 4);
    }
    const int n0 = blockIdx.y * FC_BITS_PG, n1 = min(N, n0 + FC_BITS_PG);
    uint2 bits[FC_BITS_PG];
#pragma unroll
    for (int k = 0; k < FC_BITS_PG; ++k)
        bits[k] = n0 + k < n1 ? reinterpret_cast<const uint2 *>(maskbits)[(n0 + k) * words + w] : uint2{0u, 0u};
#pragma unroll
    for (int k = 0; k < FC_BITS_PG; ++k) {
        if (n0 + k >= n1) break;
        float out[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const unsigned two = (v < 2 ? bits[k].x : bits[k].y) >> ((v & 1) * 16);     // bytes 2v (channels 0-3), 2v + 1 (4-7)
            const unsigned m = two, m2 = two >> 8;
            const float sa = ((m & 1u ? a[v].x : 0.f) + (m & 2u ? a[v].y : 0.f)) + ((m & 4u ? a[v].z : 0.f) + (m & 8u ? a[v].w : 0.f));
            const float sb = ((m2 & 1u ? b[v].x : 0.f) + (m2 & 2u ? b[v].y : 0.f)) + ((m2 & 4u ? b[v].z : 0.f) + (m2 & 8u ? b[v].w : 0.f));
            out[v] = sa + sb;
        }
        *reinterpret_cast<f32x4 *>(dsum + ((long long)(n0 + k) * F >> 3) + (w << 2)) = f32x4{out[0], out[1], out[2], out[3]};
    }
}
// out[n] = max_k in[n, k] (unsigned order; one 64-lane wave per row)
__global__ __launch_bounds__(64) void rowmax_u32_kernel(const unsigned *in, int len, unsigned *out) {
    const int n = blockIdx.x;
    unsigned m = 0;
    for (int k = threadIdx.x; k < len; k += 64) m = max(m, in[(long long)n * len + k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if (threadIdx.x == 0) out[n] = m;
}
int k_rowmax_u32(alq_ctx *ctx, const unsigned *in, int len, int N, unsigned *out) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    hipLaunchKernelGGL(rowmax_u32_kernel, dim3(N), dim3(64), 0, ctx->stream, in, len, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// max |x| of every patch's input as float bits (one wave per patch; any element count)
__global__ __launch_bounds__(256) void rowmax_abs_kernel(const float *x, int N, long long K, unsigned *out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float *p = x + (size_t)row * K;
    float m = 0.f;
    if ((K & 3) == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {      // rows start 16-byte aligned: four 16-byte loads in flight per lane
        typedef float v4 __attribute__((ext_vector_type(4)));
        const v4 *q = reinterpret_cast<const v4 *>(p);
        const long long K4 = K >> 2;
        long long k = lane;
        for (; k + 192 < K4; k += 256) {
            const v4 a = q[k], b = q[k + 64], c = q[k + 128], d = q[k + 192];
            const float ma = fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), fmaxf(__builtin_fabsf(a.z), __builtin_fabsf(a.w)));
            const float mb = fmaxf(fmaxf(__builtin_fabsf(b.x), __builtin_fabsf(b.y)), fmaxf(__builtin_fabsf(b.z), __builtin_fabsf(b.w)));
            const float mc = fmaxf(fmaxf(__builtin_fabsf(c.x), __builtin_fabsf(c.y)), fmaxf(__builtin_fabsf(c.z), __builtin_fabsf(c.w)));
            const float md = fmaxf(fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), fmaxf(__builtin_fabsf(d.z), __builtin_fabsf(d.w)));
            m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
        }
        for (; k < K4; k += 64) {
            const v4 a = q[k];
            m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), fmaxf(__builtin_fabsf(a.z), __builtin_fabsf(a.w))));
        }
    } else {
        for (long long k = lane; k < K; k += 64) m = fmaxf(m, __builtin_fabsf(p[k]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) out[row] = __builtin_bit_cast(unsigned, m);
}
int k_rowmax_abs(alq_ctx *ctx, const float *x, int N, long long K, unsigned *out) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    ALQ_REQUIRE(x && out && N >= 1 && K >= 1, ALQ_EINVAL, "rowmax_abs: bad argument");
    hipLaunchKernelGGL(rowmax_abs_kernel, dim3((N + 3) / 4), dim3(256), 0, ctx->stream, x, N, K, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

__global__ void fwd_bounds_kernel(const unsigned *amax0, int N, int stride, FwdBoundsArgs a, unsigned *bound_all) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    float b[16];
    // amax0 = the measured maximum of layer 0's OUTPUT - or (from_input) of the network INPUT, pushed through layer 0 like the others
    b[0] = __builtin_bit_cast(float, amax0[p]);
    if (a.from_input) b[0] = b[0] * a.L[0] + a.B[0];
    bound_all[p] = __builtin_bit_cast(unsigned, b[0]);
    for (int k = 1; k < a.nl && k < 16; ++k) {
        float in = b[k - 1];
        if (a.src2[k] >= 0) in = fmaxf(in, b[a.src2[k]]);
        b[k] = in * a.L[k] + a.B[k];
        bound_all[(size_t)k * stride + p] = __builtin_bit_cast(unsigned, b[k]);
    }
}
int k_fwd_bounds(alq_ctx *ctx, const unsigned *amax0, int N, int stride, const FwdBoundsArgs &a, unsigned *bound_all) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    ALQ_REQUIRE(amax0 && bound_all && a.nl >= 1 && a.nl <= 16, ALQ_EINVAL, "fwd_bounds: bad argument");
    hipLaunchKernelGGL(fwd_bounds_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, amax0, N, stride, a, bound_all);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// Flip-safe fused head, step 1: collect the sign bytes whose bit 4 is set (the fused-head epilogues mark the 4-channel groups
// holding a pre-activation that their fp16x2 contraction left within its error bound of zero; a pre-activation that is EXACTLY
// zero - an all-zero window under a zero bias: the reference's initial weights on a zero-padded volume, PW_AL.py:284-298 - is the
// same +0 in both arithmetics and is not marked).  Streams the bit field once (16 bytes per thread); every scan block keeps its
// own list segment and count: list[block * FLIP_PER_BLOCK + slot] = global byte index, cnt[block] = groups found (ALL of them,
// also beyond the list).
// The segments are cut PER PATCH (FLIP_SEG_PER_PATCH equal parts of a patch's bytes), never across patches.  A segment with
// at most FLIP_PER_BLOCK marked groups is drained from its list (the order the slots were handed out in does not matter: every
// listed group is re-evaluated, each on its own); a segment with more is drained by flip_fix_kernel scanning the segment's bytes
// itself - no group is ever dropped, so the scores do not depend on arrival order or on how the pool was cut into batches.
// *overflow counts the groups beyond the lists (alq_model_engine_info(m, 5)): how often the slower path ran, not a loss.
constexpr int FLIP_SEG_PER_PATCH = 4, FLIP_PER_BLOCK = 128;      // list slots per segment (expected load: ~8)
__device__ inline void flip_segment_range(unsigned seg, long long p16, long long *a, long long *b) {
    const long long patch = seg / FLIP_SEG_PER_PATCH, part = seg % FLIP_SEG_PER_PATCH;
    const long long per = (p16 + FLIP_SEG_PER_PATCH - 1) / FLIP_SEG_PER_PATCH;
    *a = patch * p16 + part * per;
    *b = patch * p16 + ((part + 1) * per < p16 ? (part + 1) * per : p16);
}
__global__ __launch_bounds__(256) void flip_scan_kernel(const uint4 *bits16, long long p16, unsigned *cnt, unsigned *list, unsigned *overflow) {
    __shared__ unsigned lc;
    if (threadIdx.x == 0) lc = 0u;
    __syncthreads();
    long long a, b;       // block = (patch, part): p16 = 16-byte words per patch
    flip_segment_range(blockIdx.x, p16, &a, &b);
    for (long long i = a + threadIdx.x; i < b; i += 256) {
        const uint4 w = bits16[i];
        if (((w.x | w.y | w.z | w.w) & 0x10101010u) == 0u) continue;
        const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
                if ((ws[q] >> (8 * bb + 4)) & 1u) {
                    const unsigned slot = atomicAdd(&lc, 1u);          // LDS: a global counter serialised ~65 k hits at ~10 ns each
                    if (slot < (unsigned)FLIP_PER_BLOCK) list[(long long)blockIdx.x * FLIP_PER_BLOCK + slot] = (unsigned)(i * 16 + q * 4 + bb);
                }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[blockIdx.x] = lc;
        if (lc > (unsigned)FLIP_PER_BLOCK && overflow) atomicAdd(overflow, lc - (unsigned)FLIP_PER_BLOCK);
    }
}
// Step 2: exact re-evaluation of the four pre-activations of every marked group of a stride-1 SAME conv and their sign
// nibble written back (flag cleared).  One wave per group; products in fp64 (exact for fp32 factors), lane partials and the
// wave reduction in a fixed order.  Input = two dense channels-last tensors (the parts of a split concat; CB = 0: one).
// CI / KD / KH / KW > 0: compile-time channel count and kernel box (the index arithmetic of the product loop is divisions by
// them: with run-time divisors the kernel spent its time there, 240 us per 2000 patches); 0 = run-time values.
template <int CI, int KD, int KH, int KW>
__global__ __launch_bounds__(256) void flip_fix_kernel(const unsigned *list, const unsigned *cnt, const float *inA, const float *inB,
                                                       int CA, int CB, int D, int H, int W, int kz_, int ky_, int kx_, int lz, int ly, int lx,
                                                       const float *W32, const float *bias, int Co, unsigned char *bits, long long F) {
    const unsigned seg = blockIdx.x >> 2;                    // four workgroups share the groups scan block `seg` found
    const unsigned n = cnt[seg];
    const int Ci = CI > 0 ? CI : CA + CB, kz = KD > 0 ? KD : kz_, ky = KH > 0 ? KH : ky_, kx = KW > 0 ? KW : kx_;
    const int lane = threadIdx.x & 63, K = kz * ky * kx * Ci;
    const unsigned gpp = (unsigned)(F >> 2);                 // groups (bytes) per patch
    auto fix_group = [&](unsigned b) {                       // the whole wave: group = global byte index b
        const unsigned p = b / gpp, g = b - p * gpp;
        const int co = (int)((g * 4u) % (unsigned)Co);       // first of the group's 4 channels
        int v = (int)((g * 4u) / (unsigned)Co);
        const int x = v % W; v /= W;
        const int y = v % H;
        const int z = v / H;
        const long long pv = (long long)p * D * H * W;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int i = lane; i < K; i += 64) {
            const int ci = i % Ci, t = i / Ci;
            const int tx = t % kx, ty = (t / kx) % ky, tz = t / (kx * ky);
            const int iz = z + tz - lz, iy = y + ty - ly, ix = x + tx - lx;
            if ((unsigned)iz < (unsigned)D && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const long long vox = pv + ((long long)iz * H + iy) * W + ix;
                const double xv = (double)(ci < CA ? inA[vox * CA + ci] : inB[vox * CB + (ci - CA)]);
                const float *wr = W32 + ((long long)t * Ci + ci) * Co + co;
                s0 += xv * (double)wr[0]; s1 += xv * (double)wr[1]; s2 += xv * (double)wr[2]; s3 += xv * (double)wr[3];
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64); s3 += __shfl_xor(s3, off, 64);
        }
        if (lane == 0) {
            const double b0 = bias ? (double)bias[co] : 0.0, b1 = bias ? (double)bias[co + 1] : 0.0;
            const double b2 = bias ? (double)bias[co + 2] : 0.0, b3 = bias ? (double)bias[co + 3] : 0.0;
            bits[b] = (unsigned char)((s0 + b0 > 0.0 ? 1u : 0u) | (s1 + b1 > 0.0 ? 2u : 0u) | (s2 + b2 > 0.0 ? 4u : 0u) | (s3 + b3 > 0.0 ? 8u : 0u));
        }
    };
    const unsigned w16 = (blockIdx.x & 3u) * 4u + (threadIdx.x >> 6);      // this wave among the segment's 16
    if (n <= (unsigned)FLIP_PER_BLOCK) {
        for (unsigned r = w16; r < n; r += 16) fix_group(list[(long long)seg * FLIP_PER_BLOCK + r]);
        return;
    }
    // more marked groups than list slots: the segment's 16 waves sweep its bytes themselves, word i to wave i mod 16 (every
    // lane reads the same word; a wave rewrites only bytes of its own words, which no other wave reads)
    long long a, b;
    flip_segment_range(seg, (F >> 2) / 16, &a, &b);
    const uint4 *bits16 = reinterpret_cast<const uint4 *>(bits);
    for (long long i = a + w16; i < b; i += 16) {
        const uint4 w = bits16[i];
        if (((w.x | w.y | w.z | w.w) & 0x10101010u) == 0u) continue;
        const unsigned ws[4] = {w.x, w.y, w.z, w.w};
        for (int q = 0; q < 4; ++q)
            for (int bb = 0; bb < 4; ++bb)
                if ((ws[q] >> (8 * bb + 4)) & 1u) fix_group((unsigned)(i * 16 + q * 4 + bb));
    }
}
int flip_segments(int N) { return N * FLIP_SEG_PER_PATCH; }
int flip_list_len(int N) { return N * FLIP_SEG_PER_PATCH * FLIP_PER_BLOCK; }

int k_flip_fix(alq_ctx *ctx, unsigned *list, unsigned *cnt, int cap, int N, const float *inA, const float *inB, int CA, int CB,
               int D, int H, int W, int kz, int ky, int kx, int lz, int ly, int lx, const float *W32, const float *bias, int Co,
               unsigned char *bits, long long F, unsigned *overflow) {
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    ALQ_REQUIRE(F % 64 == 0, ALQ_EINVAL, "flip_fix: %lld sign bytes per patch are not whole 16-byte words", (long long)(F >> 2));
    ALQ_REQUIRE(cap >= flip_list_len(N), ALQ_EINVAL, "flip_fix: list of %d slots for %d patches", cap, N);
    if (N <= 0) return ALQ_OK;
    const long long p16 = (F >> 2) / 16;
    const unsigned segs = (unsigned)flip_segments(N);
    hipLaunchKernelGGL(flip_scan_kernel, dim3(segs), dim3(256), 0, ctx->stream, reinterpret_cast<const uint4 *>(bits), p16, cnt, list, overflow);
    if (CA + CB == 16 && kz == 3 && ky == 3 && kx == 3)
        hipLaunchKernelGGL((flip_fix_kernel<16, 3, 3, 3>), dim3(segs * 4), dim3(256), 0, ctx->stream, list, cnt, inA, inB, CA, CB, D, H, W,
                           kz, ky, kx, lz, ly, lx, W32, bias, Co, bits, F);
    else
        hipLaunchKernelGGL((flip_fix_kernel<0, 0, 0, 0>), dim3(segs * 4), dim3(256), 0, ctx->stream, list, cnt, inA, inB, CA, CB, D, H, W,
                           kz, ky, kx, lz, ly, lx, W32, bias, Co, bits, F);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_fc_small_dsum_bits(alq_ctx *ctx, const unsigned *maskbits, const float *wv, int64_t F, int N, float *dsum) {
    ProfScope ps(ctx, PROF_FC_SMALL, 0);
    ALQ_REQUIRE(F % 32 == 0, ALQ_EINVAL, "fc_small_dsum_bits: F %% 32");
    hipLaunchKernelGGL(fc_small_dsum_bits_kernel, dim3((unsigned)((F / 32 + 255) / 256), (unsigned)((N + FC_BITS_PG - 1) / FC_BITS_PG)),
                       dim3(256), 0, ctx->stream, maskbits, wv, (long long)F, N, dsum);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int k_fc_small_bwd(alq_ctx *ctx, const float *delta, int nout, const float *Wp, int64_t F, int N,
                   float *dact, const float *mask_act, float *dsum, int C, bool *fused) {
    ProfScope ps(ctx, PROF_FC_SMALL, 2.0 * F * nout * N);
    if (fused) *fused = false;
    if ((F & 3) == 0) {
        const int G = (dsum && C > 0 && C % 4 == 0) ? C / 4 : 0;
        const dim3 grid(grid_for((long long)N * (F >> 2)));
#define ALQ_FB(GV) \
    hipLaunchKernelGGL(fc_small_bwd_kernel<GV>, grid, dim3(256), 0, ctx->stream, delta, nout, Wp, (long long)F, N, dact, mask_act, dsum)
        if (G == 1) { ALQ_FB(1); if (fused) *fused = true; }
        else if (G == 2) { ALQ_FB(2); if (fused) *fused = true; }
        else if (G == 4) { ALQ_FB(4); if (fused) *fused = true; }
        else if (G == 8) { ALQ_FB(8); if (fused) *fused = true; }
        else hipLaunchKernelGGL(fc_small_bwd_kernel<0>, grid, dim3(256), 0, ctx->stream, delta, nout, Wp, (long long)F, N,
                                dact, (const float *)nullptr, (float *)nullptr);
#undef ALQ_FB
    } else
        hipLaunchKernelGGL(fc_small_bwd_scalar_kernel, dim3(grid_for((long long)N * F)), dim3(256), 0,
                           ctx->stream, delta, nout, Wp, (long long)F, N, dact);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== softmax / scores
// posteriors = softmax over classes (NN.py:185-188): exp(z - max) / sum, fp32; output [c, N].
__global__ void softmax_kernel(const float *logits, int c, int N, float *post, long long *pred) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *z = logits + (long long)n * c;
    float mx = z[0];
    for (int j = 1; j < c; ++j) mx = fmaxf(mx, z[j]);
    float den = 0.f;
    for (int j = 0; j < c; ++j) den += expf(z[j] - mx);
    int best = 0;
    float bp = -1.f;
    for (int j = 0; j < c; ++j) {
        const float p = expf(z[j] - mx) / den;
        post[(long long)j * N + n] = p;
        if (p > bp) { bp = p; best = j; }   // first maximum, like tf.argmax
    }
    if (pred) pred[n] = best;
}

int k_softmax(alq_ctx *ctx, const float *logits, int c, int N, float *post_cN, int64_t *pred) {
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    hipLaunchKernelGGL(softmax_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, logits, c, N, post_cN,
                       reinterpret_cast<long long *>(pred));
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

__global__ void unit_cotangent_kernel(float *d, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    d[2 * n] = 1.f;
    d[2 * n + 1] = -1.f;
}

int k_fill_unit_cotangent(alq_ctx *ctx, float *dlogits, int N) {
    hipLaunchKernelGGL(unit_cotangent_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, dlogits, N);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// |double(p1) - 0.5| is exact in fp64, so the device ordering equals numpy's (PW_NNAL.py:64).
// H = -sum p log p with +10e-8 on exact zeros (NNAL_tools.py:78-83), evaluated in fp32.
__global__ void entropy_kernel(const float *p1, long long n, double *absdev, float *H) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float p = p1[i];
        if (absdev) absdev[i] = fabs((double)p - 0.5);
        if (H) {
            float a = 1.f - p, b = p;
            if (a == 0.f) a += 10e-8f;
            if (b == 0.f) b += 10e-8f;
            H[i] = -(a * logf(a) + b * logf(b));
        }
    }
}

// =========================================================================== Fisher finalisation
// Per patch, from the unit-cotangent layer sums S_t (sum of all entries of d(z0-z1)/d theta_t):
//   class-0 gradient sums = p1 * S, class-1 = -p0 * S  (d log p_j / dz = e_j - p), rounded to
//   fp32 like the reference's np.sum over fp32 gradients, divided by the layer size in fp64
//   (NNAL_tools.py:793-796); three-way saturation branch and A_i of PW_NNAL.py:770-814.
constexpr int FIN_BLOCK = 64;
__global__ __launch_bounds__(FIN_BLOCK) void fisher_finalize_kernel(
    const double *Spart, const int *nslab, int nslab_max, int max_batch, double *S, int L, const double *sizes,
    const float *post, const float *p1_branch, int N, double diag_load, float *p1_out, double *g0o, double *g1o,
    double *Ao, double *tro, double *Apart) {
    const int n = blockIdx.x * FIN_BLOCK + threadIdx.x;
    const bool live = n < N;
    double g0[16], g1[16];
    double p = 0;
    if (live) {
        const float p0f = post[n];
        const float p1f = post[(long long)N + n];
        p = (double)(p1_branch ? p1_branch[n] : p1f);
        if (p1_out) p1_out[n] = p1f;
        bool use0 = true, use1 = true;
        if (p < 1e-6) { p = 0.; use1 = false; }
        else if (p > 1. - 1e-6) { p = 1.; use0 = false; }
        for (int t = 0; t < L; ++t) {
            // slab partials of layer t, added in slab order (fixed -> run-to-run identical)
            const double *sp = Spart + ((long long)t * max_batch + n) * nslab_max;
            double s = 0;
            for (int k = 0; k < nslab[t]; ++k) s += sp[k];
            S[(long long)n * L + t] = s;
            const float s0 = (float)((double)p1f * s);
            const float s1 = (float)(-(double)p0f * s);
            g0[t] = use0 ? (double)s0 / sizes[t] : 0.;
            g1[t] = use1 ? (double)s1 / sizes[t] : 0.;
            if (g0o) g0o[(long long)n * L + t] = g0[t];
            if (g1o) g1o[(long long)n * L + t] = g1[t];
        }
    }
    double tr = 0;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j) {
            double a = 0;
            if (live) {
                a = (1. - p) * (g0[i] * g0[j]) + p * (g1[i] * g1[j]);
                if (i == j) { a += diag_load; tr += a; }
                if (Ao) Ao[((long long)n * L + i) * L + j] = a;
            }
            const double tot = wave_sum(a);   // FIN_BLOCK == one wave
            if (threadIdx.x == 0) Apart[(long long)blockIdx.x * L * L + i * L + j] = tot;
        }
    if (live && tro) tro[n] = tr;
}

__global__ void reduce_Asum_kernel(const double *Apart, int nblocks, int LL, double *Asum) {
    const int e = threadIdx.x;
    if (e >= LL) return;
    double s = 0;
    for (int b = 0; b < nblocks; ++b) s += Apart[(long long)b * LL + e];
    Asum[e] = s;
}

int k_fisher_finalize(alq_ctx *ctx, const double *Spart, const int *nslab, int nslab_max, int max_batch, double *S,
                      int L, const double *sizes, const float *post_cN, const float *p1_branch, int N,
                      double diag_load, float *p1_out, double *g0, double *g1, double *A, double *trace,
                      double *Apart, int *nblocks_out) {
    ALQ_REQUIRE(L <= 16, ALQ_EUNSUPPORTED, "fisher: %d parameterised layers > 16", L);
    const int nb = (N + FIN_BLOCK - 1) / FIN_BLOCK;
    ProfScope ps(ctx, PROF_REDUCE, 0);
    hipLaunchKernelGGL(fisher_finalize_kernel, dim3(nb), dim3(FIN_BLOCK), 0, ctx->stream, Spart, nslab, nslab_max,
                       max_batch, S, L, sizes, post_cN, p1_branch, N, diag_load, p1_out, g0, g1, A, trace, Apart);
    ALQ_LAUNCH_CHECK();
    *nblocks_out = nb;
    return ALQ_OK;
}

int k_reduce_Asum(alq_ctx *ctx, const double *Apart, int nblocks, int LL, double *Asum) {
    ProfScope ps(ctx, PROF_REDUCE, 0);
    hipLaunchKernelGGL(reduce_Asum_kernel, dim3(1), dim3(256), 0, ctx->stream, Apart, nblocks, LL, Asum);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== gather + normalise
template <typename VT, typename OT>
__global__ void gather_norm_kernel(const VT *const *vols, int m, long long P1, long long P2, long long O0,
                                   long long O1, long long O2, const long long *inds, long long n, int d1,
                                   int d2, int d3, const double *stats, int quirk, OT *out) {
    const int C = m * d3;
    const long long total = n * d1 * d2 * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int ch = r % C; r /= C;
        const int a1 = r % d2; r /= d2;
        const int a0 = r % d1; r /= d1;
        const long long b = r;
        const int j = ch / d3, a2 = ch - j * d3;
        long long ind = inds[b];
        const long long i2 = ind % O2; ind /= O2;
        const long long i1 = ind % O1; ind /= O1;
        const long long i0 = ind;
        // window origin in PADDED coordinates is the un-padded index itself (centre = index + radius)
        double v = (double)vols[j][((i0 + a0) * P1 + (i1 + a1)) * P2 + (i2 + a2)];
        if (quirk == 1) {
            if (ch < m) v = (v - stats[2 * ch]) / stats[2 * ch + 1];
        } else if (quirk == 0) {
            v = (v - stats[2 * j]) / stats[2 * j + 1];
        }
        out[i] = (OT)v;
    }
    (void)O0;
}

template <typename VT, typename OT>
static void gather_launch(alq_ctx *ctx, const void *const *vp, int m, const int64_t pad[3], const int64_t orig[3],
                          const int64_t *d_inds, int64_t n, const int32_t ps[3], const double *d_stats, int quirk,
                          void *d_out, long long total) {
    hipLaunchKernelGGL((gather_norm_kernel<VT, OT>), dim3(grid_for(total)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const VT *const *>(vp), m, (long long)pad[1], (long long)pad[2],
                       (long long)orig[0], (long long)orig[1], (long long)orig[2],
                       reinterpret_cast<const long long *>(d_inds), (long long)n, ps[0], ps[1], ps[2], d_stats,
                       quirk, reinterpret_cast<OT *>(d_out));
}

int gather_normalize_impl(alq_ctx *ctx, const void *const *d_vol_ptrs_dev, int m, int is_f64,
                          const int64_t pad[3], const int64_t orig[3], const int64_t *d_inds, int64_t n,
                          const int32_t ps[3], const double *d_stats, int quirk, int out_f64, void *d_out) {
    const long long total = (long long)n * ps[0] * ps[1] * ps[2] * m;
    ProfScope pscope(ctx, PROF_ELEMWISE, 0);
    if (is_f64 && out_f64) gather_launch<double, double>(ctx, d_vol_ptrs_dev, m, pad, orig, d_inds, n, ps, d_stats, quirk, d_out, total);
    else if (is_f64) gather_launch<double, float>(ctx, d_vol_ptrs_dev, m, pad, orig, d_inds, n, ps, d_stats, quirk, d_out, total);
    else if (out_f64) gather_launch<float, double>(ctx, d_vol_ptrs_dev, m, pad, orig, d_inds, n, ps, d_stats, quirk, d_out, total);
    else gather_launch<float, float>(ctx, d_vol_ptrs_dev, m, pad, orig, d_inds, n, ps, d_stats, quirk, d_out, total);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int score_entropy_impl(alq_ctx *ctx, const float *d_p1, int64_t n, double *d_absdev, float *d_H) {
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    hipLaunchKernelGGL(entropy_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, d_p1, (long long)n,
                       d_absdev, d_H);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== row gather (index-list entry points)
// out[i, :] = pool[rows[i], :]  (epp floats per row; 16-byte copies when the row length allows)
__global__ void gather_rows_kernel(const float *pool, const long long *rows, long long n, long long epp, float *out) {
    if ((epp & 3) == 0) {
        const long long q = epp >> 2, total = n * q;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const long long r = i / q, e = i - r * q;
            reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(pool + rows[r] * epp)[e];
        }
    } else {
        const long long total = n * epp;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            const long long r = i / epp, e = i - r * epp;
            out[i] = pool[rows[r] * epp + e];
        }
    }
}

int gather_rows_impl(alq_ctx *ctx, const float *d_pool, const int64_t *d_rows, int64_t n, int64_t epp, float *d_out) {
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * epp / 4 + 1)), dim3(256), 0, ctx->stream, d_pool,
                       reinterpret_cast<const long long *>(d_rows), (long long)n, (long long)epp, d_out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== debug copies
__global__ void view_to_dense_kernel(const float *t, int cs, int c0, int C, long long nvox, float *out) {
    const long long total = nvox * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long v = i / C;
        const int c = (int)(i - v * C);
        out[i] = t[v * cs + c0 + c];
    }
}
__global__ void f64_to_f32_kernel(const double *in, long long n, float *out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
int debug_view_copy(alq_ctx *ctx, const View &v, int N, float *out) {
    const long long nvox = (long long)N * v.vox();
    hipLaunchKernelGGL(view_to_dense_kernel, dim3(grid_for(nvox * v.C)), dim3(256), 0, ctx->stream, v.p, v.cs,
                       v.c0, v.C, nvox, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}
int debug_f64_copy(alq_ctx *ctx, const double *in, long long n, float *out) {
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, in, n, out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// =========================================================================== synthetic patches
// Counter-based generator: element e of patch id -> splitmix64(seed, id, e) -> Box-Muller.
__device__ inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void synth_kernel(unsigned long long seed, long long first_id, long long n, long long epp,
                             float *out) {
    const long long total = n * epp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pid = first_id + i / epp;
        const long long e = i % epp;
        const unsigned long long h = splitmix64(splitmix64(seed ^ (unsigned long long)pid * 0xD1B54A32D192ED03ull) + (unsigned long long)e);
        const float u1 = ((float)(unsigned)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1)
        const float u2 = (float)(unsigned)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
        out[i] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
}

int synth_impl(alq_ctx *ctx, uint64_t seed, int64_t first_id, int64_t n, int64_t epp, float *d_out) {
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(n * epp)), dim3(256), 0, ctx->stream,
                       (unsigned long long)seed, (long long)first_id, (long long)n, (long long)epp, d_out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

}  // namespace alq
