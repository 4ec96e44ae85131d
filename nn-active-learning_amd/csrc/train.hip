// Parameter-gradient side of the path: what `tf.gradients(..., trainable_variables)` hands back in the reference.
//   * full per-sample gradients of the log-posteriors (NN.CNN.get_gradients, NN.py:621-645; consumed by
//     PW_NNAL.gen_A_matrices with shrink methods other than 'sum', by grad_layers subsets and by
//     model_utils.diagonal_Fisher, model_utils.py:294-330);
//   * the mini-batch gradient of the mean softmax cross-entropy (get_optimizer, NN.py:583-588) and the SGD / Adam
//     steps of `train_step` (NN.py:591-615) for the fine-tune between query rounds (PW_AL.finetune_multimg,
//     PW_AL.py:1091-1147).
// Nothing here is on the measured scoring path (the 'sum' shrink never forms a weight gradient): plain fp32 VALU
// kernels, fp32 accumulation inside a voxel slab and a fixed-order fp64 sum across slabs (and samples), so results are
// run-to-run identical and at least as accurate as an fp32 reduction in any order.
#include <algorithm>

#include "alq_internal.h"

namespace alq {

#define ALQ_LAUNCH_CHECK() ALQ_HIP(hipGetLastError())

static inline unsigned grid1(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 65535LL * 16) g = 65535LL * 16;
    return (unsigned)g;
}

// element (voxel row r of the N-patch tensor, channel c) of a View, split concat included
struct DView {
    const float *p;
    int cs, c0, C, split;
    long long delta;
    __device__ inline float at(long long row, int c) const {
        if (split && c >= split) return p[delta + row * cs + (c - split)];
        return p[row * cs + c0 + c];
    }
};
static DView dview(const View &v) {
    DView d;
    d.p = v.p; d.cs = v.cs; d.c0 = v.c0; d.C = v.C; d.split = v.split; d.delta = v.delta;
    return d;
}

__device__ inline unsigned long long tr_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ------------------------------------------------------------------------------------------ dropout
// out[n, e] *= keep(n, e) / keep_prob, keep ~ Bernoulli(keep_prob) from a counter-based generator keyed
// (seed, layer, sample id, element): tf.nn.dropout's scaling (NN.py:169-171), its own reproducible mask (TF's RNG
// stream is not reproducible outside TF; the oracle restates THIS generator, oracle/tfops.py dropout_keep).
// The same kernel is the backward pass (the cotangent takes the same factor).  `e` is the element's index in the
// layer output in MEMORY order ((d, h, w, c) row-major), so the mask does not depend on the batch split.
__global__ void dropout_kernel(float *t, int cs, int c0, int C, long long vox, long long n0, int N, unsigned long long seed,
                               int layer, float keep_prob, float inv_keep) {
    const long long per = vox * C, total = per * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per, e = i - n * per;
        const long long v = e / C;
        const int c = (int)(e - v * C);
        const unsigned long long h = tr_splitmix64(tr_splitmix64(seed ^ ((unsigned long long)(n0 + n) * 0xD1B54A32D192ED03ull) ^
                                                                  ((unsigned long long)layer << 56)) + (unsigned long long)e);
        const float u = (float)(unsigned)(h >> 40) * (1.0f / 16777216.0f);        // [0, 1)
        float *q = t + (n * vox + v) * cs + c0 + c;
        *q = u < keep_prob ? *q * inv_keep : 0.f;
    }
}

int k_dropout(alq_ctx *ctx, const View &t, int N, long long first_sample, unsigned long long seed, int layer, float keep_prob) {
    ALQ_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, ALQ_EINVAL, "dropout keep_prob %g outside (0, 1]", (double)keep_prob);
    ALQ_REQUIRE(!t.split, ALQ_EUNSUPPORTED, "dropout on a split view");
    if (keep_prob == 1.f) return ALQ_OK;
    const long long total = (long long)N * t.vox() * t.C;
    ProfScope ps(ctx, PROF_ELEMWISE, 0);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid1(total)), dim3(256), 0, ctx->stream, t.p, t.cs, t.c0, t.C, (long long)t.vox(),
                       first_sample, N, seed, layer, keep_prob, 1.0f / keep_prob);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// ------------------------------------------------------------------------------------------ cotangents at the logits
// mode 0: d log p_j / dz = e_j - p for class j = cls (NN.py:639-645);
// mode 1: d mean-CE / dz = (p - y) / N, y = one-hot of labels[n] (tf.nn.softmax_cross_entropy_with_logits + reduce_mean,
//         NN.py:583-588); a label outside [0, c) gives a zero row (unlabelled sample).
__global__ void logit_cotangent_kernel(const float *post_cN, int c, int N, int mode, int cls, const int *labels, float scale,
                                       float *dlogits /*[N, c]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * c) return;
    const int n = i / c, j = i - n * c;
    const float p = post_cN[(long long)j * N + n];
    float d;
    if (mode == 0) d = (j == cls ? 1.f : 0.f) - p;
    else {
        const int y = labels[n];
        d = (y >= 0 && y < c) ? (p - (j == y ? 1.f : 0.f)) * scale : 0.f;
    }
    dlogits[i] = d;
}

int k_logit_cotangent(alq_ctx *ctx, const float *post_cN, int c, int N, int mode, int cls, const int *labels, float scale,
                      float *dlogits) {
    hipLaunchKernelGGL(logit_cotangent_kernel, dim3((N * c + 255) / 256), dim3(256), 0, ctx->stream, post_cN, c, N, mode, cls,
                       labels, scale, dlogits);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// mean cross-entropy of a batch (reported by the training step): -log p[y_n, n] summed in fp64 by one workgroup
__global__ __launch_bounds__(256) void ce_loss_kernel(const float *post_cN, int c, int N, const int *labels, double *out) {
    __shared__ double sh[256];
    double s = 0;
    int cnt = 0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const int y = labels[n];
        if (y >= 0 && y < c) { s -= log((double)fmaxf(post_cN[(long long)y * N + n], 1e-38f)); ++cnt; }
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] / (double)(N > 0 ? N : 1);
    (void)cnt;
}

int k_ce_loss(alq_ctx *ctx, const float *post_cN, int c, int N, const int *labels, double *d_out) {
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, ctx->stream, post_cN, c, N, labels, d_out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// ------------------------------------------------------------------------------------------ weight gradients
// G[n][t][v][u] = sum_q U[n, q, u] * V[n, s*q + t - lo, v]     (zero outside V's grid)
//   conv            U = masked pre-activation cotangent (output grid, u = co), V = layer input (v = ci), s = 1:
//                   G = dW in TF layout [t][ci][co];
//   conv_transpose  U = layer input (coarse grid, u = ci), V = cotangent (fine grid, v = co), s = stride:
//                   G = dW in TF layout [t][co][ci].
// grid (slab of the q grid, sample): the slab's U rows sit in LDS (broadcast reads), a thread owns one (t, v) pair and
// UC accumulators and walks the slab; partial sums per slab, added in slab order by wgrad_reduce_kernel.
constexpr int WG_SLAB = 256;      // q points per workgroup
template <int UC>
__global__ __launch_bounds__(256) void wgrad_kernel(DView U, DView V, int QD, int QH, int QW, int VD, int VH, int VW, int kz, int ky,
                                                    int kx, int sz, int sy, int sx, int lz, int ly, int lx, int nslab, int u0,
                                                    float *partial /*[N][nslab][T*VC][UC]*/) {
    __shared__ float Us[WG_SLAB * UC];
    const int slab = blockIdx.x, n = blockIdx.y;
    const long long qvox = (long long)QD * QH * QW, vvox = (long long)VD * VH * VW;
    const int q0 = slab * WG_SLAB;
    const int nq = (int)min((long long)WG_SLAB, qvox - q0);
    for (int i = threadIdx.x; i < nq * UC; i += 256) {
        const int q = i / UC, u = i - q * UC;
        Us[i] = u0 + u < U.C ? U.at((long long)n * qvox + q0 + q, u0 + u) : 0.f;
    }
    __syncthreads();
    const int T = kz * ky * kx, M = T * V.C;
    for (int m = threadIdx.x; m < M; m += 256) {
        const int t = m / V.C, v = m - t * V.C;
        const int tz = t / (ky * kx), ty = (t / kx) % ky, tx = t % kx;
        float acc[UC];
#pragma unroll
        for (int u = 0; u < UC; ++u) acc[u] = 0.f;
        int qx = q0 % QW, qy = (q0 / QW) % QH, qz = q0 / (QW * QH);
        for (int q = 0; q < nq; ++q) {
            const int pz = sz * qz + tz - lz, py = sy * qy + ty - ly, px = sx * qx + tx - lx;
            if ((unsigned)pz < (unsigned)VD && (unsigned)py < (unsigned)VH && (unsigned)px < (unsigned)VW) {
                const float val = V.at((long long)n * vvox + ((long long)pz * VH + py) * VW + px, v);
                const float *us = Us + q * UC;
#pragma unroll
                for (int u = 0; u < UC; ++u) acc[u] = fmaf(val, us[u], acc[u]);
            }
            if (++qx == QW) { qx = 0; if (++qy == QH) { qy = 0; ++qz; } }
        }
        float *dst = partial + (((long long)n * nslab + slab) * M + m) * UC;
#pragma unroll
        for (int u = 0; u < UC; ++u) dst[u] = acc[u];
    }
}

// out[(n or 0)][m][u0 + u] (row length Utot) = sum_{slab} (and over n when `sum_n`) partial[..][m][u < Ucnt], fp64, fixed order
__global__ void wgrad_reduce_kernel(const float *partial, int N, int nslab, int M, int UC, int Ucnt, int u0, int Utot, int sum_n,
                                    float *out, long long out_stride) {
    const long long per = (long long)M * Ucnt;
    const long long total = per * (sum_n ? 1 : N);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per, e = i - n * per;
        const int m = (int)(e / Ucnt), u = (int)(e - (long long)m * Ucnt);
        double s = 0;
        const int n_lo = sum_n ? 0 : (int)n, n_hi = sum_n ? N : (int)n + 1;
        for (int nn = n_lo; nn < n_hi; ++nn)
            for (int sl = 0; sl < nslab; ++sl) s += (double)partial[(((long long)nn * nslab + sl) * M + m) * UC + u];
        out[n * out_stride + (long long)m * Utot + u0 + u] = (float)s;
    }
}

long long wgrad_partial_floats(const View &U, const View &V, const int k[3]) {       // per sample
    const int UC = U.C <= 8 ? 8 : (U.C <= 16 ? 16 : 32);
    const long long nslab = (U.vox() + WG_SLAB - 1) / WG_SLAB;
    return nslab * (long long)k[0] * k[1] * k[2] * V.C * UC;
}

int k_wgrad(alq_ctx *ctx, const View &U, const View &V, const int k[3], const int s[3], const int lo[3], int N, int sum_n,
            float *partial, float *d_out, long long out_stride) {
    const int UC = U.C <= 8 ? 8 : (U.C <= 16 ? 16 : 32);
    const int nslab = (int)((U.vox() + WG_SLAB - 1) / WG_SLAB);
    const int M = k[0] * k[1] * k[2] * V.C;
    const DView du = dview(U), dv = dview(V);
    ProfScope ps(ctx, PROF_REDUCE, 2.0 * (double)N * U.vox() * M * U.C);
    dim3 grid((unsigned)nslab, (unsigned)N);
    for (int u0 = 0; u0 < U.C; u0 += UC) {       // more than 32 channels on the accumulator side: blocks of 32
#define ALQ_WG(UCV)                                                                                                        \
    hipLaunchKernelGGL(wgrad_kernel<UCV>, grid, dim3(256), 0, ctx->stream, du, dv, U.D, U.H, U.W, V.D, V.H, V.W, k[0], k[1], k[2], \
                       s[0], s[1], s[2], lo[0], lo[1], lo[2], nslab, u0, partial)
        if (UC == 8) ALQ_WG(8); else if (UC == 16) ALQ_WG(16); else ALQ_WG(32);
#undef ALQ_WG
        ALQ_LAUNCH_CHECK();
        const int ucnt = std::min(UC, U.C - u0);
        const long long total = (long long)M * ucnt * (sum_n ? 1 : N);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid1(total)), dim3(256), 0, ctx->stream, partial, N, nslab, M, UC, ucnt, u0,
                           U.C, sum_n, d_out, out_stride);
        ALQ_LAUNCH_CHECK();
    }
    return ALQ_OK;
}

// bias gradient: db[n][c] = sum_x delta[n, x, c]  (fp64, one workgroup per (sample or whole batch, channel block))
__global__ __launch_bounds__(256) void bgrad_kernel(DView D, long long vox, int N, int sum_n, float *out, long long out_stride) {
    __shared__ double sh[256];
    const int c = blockIdx.x, n = blockIdx.y;
    const long long rows = sum_n ? vox * N : vox, r0 = sum_n ? 0 : (long long)n * vox;
    double s = 0;
    for (long long r = threadIdx.x; r < rows; r += 256) s += (double)D.at(r0 + r, c);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long long)n * out_stride + c] = (float)sh[0];
}

int k_bgrad(alq_ctx *ctx, const View &delta, int N, int sum_n, float *d_out, long long out_stride) {
    const DView d = dview(delta);
    hipLaunchKernelGGL(bgrad_kernel, dim3((unsigned)delta.C, (unsigned)(sum_n ? 1 : N)), dim3(256), 0, ctx->stream, d,
                       (long long)delta.vox(), N, sum_n, d_out, out_stride);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// fc: dW[n][o][f_tf] = delta[n, o] * a[n, f_mem(f_tf)]; a = layer input with geometry (D, H, W, C) (1,1,1,F after an fc);
// f_tf = ((c*W + w)*H + h)*D + d is the reference's flatten order (NN.py:296-301).  sum_n: summed over the batch (fp64).
__global__ void fc_wgrad_kernel(const float *delta /*[N, nout]*/, DView A, int D, int H, int W, int C, int nout, int N, int sum_n,
                                float *out, long long out_stride) {
    const long long F = (long long)D * H * W * C;
    const long long per = F * nout, total = per * (sum_n ? 1 : N);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per, e = i - n * per;
        const int o = (int)(e / F);
        long long ft = e - (long long)o * F;
        const int d = (int)(ft % D); ft /= D;
        const int h = (int)(ft % H); ft /= H;
        const int w = (int)(ft % W); ft /= W;
        const int c = (int)ft;
        const long long vrow = ((long long)d * H + h) * W + w;
        const long long vox = (long long)D * H * W;
        if (sum_n) {
            double s = 0;
            for (int nn = 0; nn < N; ++nn) s += (double)delta[(long long)nn * nout + o] * (double)A.at((long long)nn * vox + vrow, c);
            out[e] = (float)s;
        } else {
            out[n * out_stride + e] = delta[n * nout + o] * A.at(n * vox + vrow, c);
        }
    }
}

int k_fc_wgrad(alq_ctx *ctx, const float *delta, const View &a, int nout, int N, int sum_n, float *d_out, long long out_stride) {
    const DView da = dview(a);
    const long long total = (long long)a.vox() * a.C * nout * (sum_n ? 1 : N);
    ProfScope ps(ctx, PROF_REDUCE, 0);
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(grid1(total)), dim3(256), 0, ctx->stream, delta, da, a.D, a.H, a.W, a.C, nout, N, sum_n,
                       d_out, out_stride);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// ------------------------------------------------------------------------------------------ optimiser steps
// tf.train.GradientDescentOptimizer: theta -= lr * g
__global__ void sgd_kernel(float *theta, const float *g, long long n, float lr) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        theta[i] -= lr * g[i];
}
// tf.train.AdamOptimizer (TF 1.x, beta1 .9, beta2 .999, eps 1e-8): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t);
// m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; theta -= lr_t m / (sqrt(v) + eps)
__global__ void adam_kernel(float *theta, const float *g, float *m, float *v, long long n, float lr_t, float b1, float b2, float eps) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

int k_sgd(alq_ctx *ctx, float *theta, const float *g, long long n, float lr) {
    hipLaunchKernelGGL(sgd_kernel, dim3(grid1(n)), dim3(256), 0, ctx->stream, theta, g, n, lr);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}
int k_adam(alq_ctx *ctx, float *theta, const float *g, float *m, float *v, long long n, float lr_t, float b1, float b2, float eps) {
    hipLaunchKernelGGL(adam_kernel, dim3(grid1(n)), dim3(256), 0, ctx->stream, theta, g, m, v, n, lr_t, b1, b2, eps);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// sum over samples of squared per-sample gradients (model_utils.diagonal_Fisher, model_utils.py:294-330): acc += g^2
__global__ void sq_accum_kernel(const float *g, long long per, int N, double *acc) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        double s = 0;
        for (int n = 0; n < N; ++n) { const double x = (double)g[(long long)n * per + i]; s += x * x; }
        acc[i] += s;
    }
}
int k_sq_accum(alq_ctx *ctx, const float *g, long long per, int N, double *acc) {
    hipLaunchKernelGGL(sq_accum_kernel, dim3(grid1(per)), dim3(256), 0, ctx->stream, g, per, N, acc);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// shrink_gradient(grads, 'sum') of MATERIALISED per-sample gradients (NNAL_tools.py:784-796): out[n, t] = (sum of the entries of
// layer t's weight and bias gradient) / (|W_t| + |b_t|).  One workgroup per (layer, sample); every thread walks a fixed stride,
// the partials fold in a fixed tree: fp64, run-to-run identical.
struct ShrinkOffsets { long long off[65]; };
__global__ void shrink_sum_kernel(const float *g, long long P, ShrinkOffsets o, double *out, int L) {
    __shared__ double red[256];
    const int t = blockIdx.x, n = blockIdx.y;
    const long long a = o.off[t], b = o.off[t + 1];
    const float *row = g + (long long)n * P;
    double s = 0;
    for (long long i = a + threadIdx.x; i < b; i += 256) s += (double)row[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long long)n * L + t] = red[0] / (double)(b - a);
}
int k_shrink_sum(alq_ctx *ctx, const float *g, int N, long long P, const long long *off, int L, double *out) {
    ShrinkOffsets o;
    for (int t = 0; t <= L; ++t) o.off[t] = off[t];
    hipLaunchKernelGGL(shrink_sum_kernel, dim3(L, N), dim3(256), 0, ctx->stream, g, P, o, out, L);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

// multi-class conditional Fisher matrices (NNAL.py:399-409): A_i = sum_j w[i, j] g[i, j, :] g[i, j, :]^T + diag[i] I, classes in
// ascending order, fp64.  One workgroup per sample, one thread per entry.
__global__ void fisher_classes_kernel(const double *g, const double *w, const double *diag, int c, int L, double *A) {
    const int i = blockIdx.x;
    for (int e = threadIdx.x; e < L * L; e += blockDim.x) {
        const int r = e / L, s = e % L;
        double acc = 0;
        for (int j = 0; j < c; ++j) {
            const double wj = w[(long long)i * c + j];
            if (wj != 0.0) acc += g[((long long)i * c + j) * L + r] * g[((long long)i * c + j) * L + s] * wj;
        }
        A[(long long)i * L * L + e] = acc + (r == s ? diag[i] : 0.0);
    }
}
int k_fisher_classes(alq_ctx *ctx, const double *g, const double *w, const double *diag, int N, int c, int L, double *A) {
    hipLaunchKernelGGL(fisher_classes_kernel, dim3(N), dim3(64), 0, ctx->stream, g, w, diag, c, L, A);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

}  // namespace alq
