// fp64 evaluation of the scored path on the device: the ACCURACY REFERENCE of the build (not a product engine, never on the
// scoring path): plain one-thread-per-output kernels in double precision for the layers the model supports (conv /
// conv_transpose / max-pool / fc, 'con' skips, ReLU flags; NN.py:258-340, NN_extended.py:366-601), ONE backward pass with the
// unit cotangent (+1, -1) on the two logits and the factored layer sums of DESIGN.md 3 (what PW_NNAL.gen_A_matrices :757-814
// + NNAL_tools.shrink_gradient 'sum' :784-796 reduce a per-sample gradient to), all in fp64.
//
// Two things make it the arbiter of "fp32-level engines disagree on a patch" (DESIGN.md 2):
//  * it lists a sample's FRAGILE decisions - ReLU inputs within eps x the layer's rms pre-activation of zero, max-pool windows
//    whose two largest inputs lie within eps x rms of each other - the decisions fp32 rounding may legitimately take either way;
//  * it evaluates a sample with a given set of such decisions INVERTED (alq_flip_t): an engine's scores must equal the plain
//    fp64 value or one of those evaluations (nn-active-learning_amd/ref64.py does the search).  A ReLU unit inverted to
//    'passes' (pre-activation s <= 0, within eps x rms of zero) outputs |s|, the value mirrored across the boundary: an engine
//    that let it pass computed a POSITIVE value there, and a max-pool behind the layer compares it with the zeros of its window
//    (with s itself the pool would route the window's cotangent elsewhere - NET-B's conv4 -> max2, round 6).
// bench.py reports, for the shipped engines and for the exact-fp32 engine alike, how many patches differ from this evaluation
// and how many inverted decisions explain each difference.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace alq {
namespace {

struct T64 {              // a dense channels-last fp64 tensor [N, D, H, W, C]
    double *p = nullptr;
    int D = 1, H = 1, W = 1, C = 0;
    long long vox() const { return (long long)D * H * W; }
    long long elems() const { return vox() * C; }
};

struct R64Layer {
    alq_layer_t spec;
    int pidx = -1;
    int src_main = -1, src_skip = -1;      // producing layers of the input (-1 = the patch); concat order [skip | main]
    T64 pre, act, dact;                    // pre-activation, output, cotangent of the output (accumulated by the consumers)
    unsigned char *keep = nullptr;         // ReLU decision per output element (after flips)
    int *argmax = nullptr;                 // pool: input voxel of every output element
    int lo[3] = {0, 0, 0};
    bool flat_in = false;                  // fc
    long long F = 0;
    const double *W = nullptr, *b = nullptr;
    int Cin = 0, Ca = 0;                   // input channels in all, of which the first Ca come from the skip source
};

constexpr int TPB = 256;
inline unsigned nblk(long long n) { return (unsigned)std::min<long long>((n + TPB - 1) / TPB, 1 << 20); }

__device__ __forceinline__ bool flipped(const alq_flip_t *fl, int nf, long long n, int layer, long long idx, double *delta) {
    if (!fl) return false;
    const alq_flip_t *f = fl + n * nf;
    for (int i = 0; i < nf; ++i)
        if (f[i].layer == layer && f[i].idx == idx) { if (delta) *delta = f[i].delta; return true; }
    return false;
}

// x fp32 [rows] -> fp64 [N, epp]
__global__ void r64_load(const float *x, const long long *rows, long long N, long long epp, double *out) {
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < N * epp; i += (long long)gridDim.x * TPB) {
        const long long n = i / epp, e = i - n * epp;
        out[i] = (double)x[(rows ? rows[n] : n) * epp + e];
    }
}

// ---- forward --------------------------------------------------------------------------------------------------------------
// conv, SAME, stride 1: W [kd, kh, kw, Ci, Co].  Input = channels [0, Ca) of A followed by the channels of B.
__global__ void r64_conv_fwd(const double *A, int Ca, const double *B, int Cb, int D, int H, int Wd, const double *Wt, const double *bias,
                             int kd, int kh, int kw, int lz, int ly, int lx, int Co, long long N, int layer, int relu,
                             const alq_flip_t *fl, int nf, double *pre, double *act, unsigned char *keep) {
    const long long vox = (long long)D * H * Wd, total = N * vox * Co;
    const int Ci = Ca + Cb;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int co = (int)(i % Co);
        const long long v = (i / Co) % vox, n = i / (Co * vox);
        const int x = (int)(v % Wd), y = (int)((v / Wd) % H), z = (int)(v / ((long long)Wd * H));
        double s = bias[co];
        for (int tz = 0; tz < kd; ++tz) {
            const int iz = z + tz - lz;
            if (iz < 0 || iz >= D) continue;
            for (int ty = 0; ty < kh; ++ty) {
                const int iy = y + ty - ly;
                if (iy < 0 || iy >= H) continue;
                for (int tx = 0; tx < kw; ++tx) {
                    const int ix = x + tx - lx;
                    if (ix < 0 || ix >= Wd) continue;
                    const long long iv = n * vox + ((long long)iz * H + iy) * Wd + ix;
                    const double *w = Wt + ((long long)((tz * kh + ty) * kw + tx) * Ci) * Co + co;
                    const double *a = A + iv * Ca;
                    for (int c = 0; c < Ca; ++c) s = fma(a[c], w[(long long)c * Co], s);
                    const double *b = B + iv * Cb;
                    w += (long long)Ca * Co;
                    for (int c = 0; c < Cb; ++c) s = fma(b[c], w[(long long)c * Co], s);
                }
            }
        }
        pre[i] = s;
        bool k = true;
        if (relu) k = (s > 0.0) != flipped(fl, nf, n, layer, v * Co + co, nullptr);
        keep[i] = k ? 1 : 0;
        act[i] = k ? (relu ? fabs(s) : s) : 0.0;      // (a ReLU unit inverted to 'passes' takes the mirrored value: see the header)
    }
}

// conv_transpose, output = s * input (NN_extended.py:574-587): out[o] = b + sum_t [(o + lo - t) = s i] in[i] W[t][co][ci]
__global__ void r64_convt_fwd(const double *A, int Ci, int ID, int IH, int IW, const double *Wt, const double *bias, int kd, int kh, int kw,
                              int sz, int sy, int sx, int lz, int ly, int lx, int Co, long long N, int layer, int relu,
                              const alq_flip_t *fl, int nf, double *pre, double *act, unsigned char *keep) {
    const int OD = ID * sz, OH = IH * sy, OW = IW * sx;
    const long long ovox = (long long)OD * OH * OW, ivox = (long long)ID * IH * IW, total = N * ovox * Co;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int co = (int)(i % Co);
        const long long v = (i / Co) % ovox, n = i / (Co * ovox);
        const int x = (int)(v % OW), y = (int)((v / OW) % OH), z = (int)(v / ((long long)OW * OH));
        double s = bias[co];
        for (int tz = 0; tz < kd; ++tz) {
            const int az = z + lz - tz;
            if (az < 0 || az % sz) continue;
            const int iz = az / sz;
            if (iz >= ID) continue;
            for (int ty = 0; ty < kh; ++ty) {
                const int ay = y + ly - ty;
                if (ay < 0 || ay % sy) continue;
                const int iy = ay / sy;
                if (iy >= IH) continue;
                for (int tx = 0; tx < kw; ++tx) {
                    const int ax = x + lx - tx;
                    if (ax < 0 || ax % sx) continue;
                    const int ix = ax / sx;
                    if (ix >= IW) continue;
                    const double *a = A + (n * ivox + ((long long)iz * IH + iy) * IW + ix) * Ci;
                    const double *w = Wt + ((long long)((tz * kh + ty) * kw + tx) * Co + co) * Ci;
                    for (int c = 0; c < Ci; ++c) s = fma(a[c], w[c], s);
                }
            }
        }
        pre[i] = s;
        bool k = true;
        if (relu) k = (s > 0.0) != flipped(fl, nf, n, layer, v * Co + co, nullptr);
        keep[i] = k ? 1 : 0;
        act[i] = k ? (relu ? fabs(s) : s) : 0.0;      // (a ReLU unit inverted to 'passes' takes the mirrored value: see the header)
    }
}

// max-pool, window = stride, SAME (-inf beyond the end): the first maximum in scan order wins.  A flip lifts ONE input element.
// Also lists near-ties (two largest inputs within thr of each other, maximum > 0) as fragile decisions.
__global__ void r64_pool_fwd(const double *A, int C, int ID, int IH, int IW, int kd, int kh, int kw, int lz, int ly, int lx, int OD, int OH, int OW,
                             long long N, int layer, const alq_flip_t *fl, int nf, double *out, int *argmax,
                             const double *rms, double eps, int cap, alq_flip_t *cand, double *ckey, int *ccount) {
    const long long ovox = (long long)OD * OH * OW, ivox = (long long)ID * IH * IW, total = N * ovox * C;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int c = (int)(i % C);
        const long long v = (i / C) % ovox, n = i / (C * ovox);
        const int x = (int)(v % OW), y = (int)((v / OW) % OH), z = (int)(v / ((long long)OW * OH));
        double best = -INFINITY, second = -INFINITY;
        long long bi = -1, si = -1;
        for (int tz = 0; tz < kd; ++tz) {
            const int iz = z * kd + tz - lz;
            if (iz < 0 || iz >= ID) continue;
            for (int ty = 0; ty < kh; ++ty) {
                const int iy = y * kh + ty - ly;
                if (iy < 0 || iy >= IH) continue;
                for (int tx = 0; tx < kw; ++tx) {
                    const int ix = x * kw + tx - lx;
                    if (ix < 0 || ix >= IW) continue;
                    const long long iv = ((long long)iz * IH + iy) * IW + ix;
                    double a = A[(n * ivox + iv) * C + c], d = 0.0;
                    if (flipped(fl, nf, n, layer, iv * C + c, &d)) a += d;
                    if (a > best) { second = best; si = bi; best = a; bi = iv; }
                    else if (a > second) { second = a; si = iv; }
                }
            }
        }
        out[i] = best;
        argmax[i] = (int)bi;
        if (cand && si >= 0 && best > 0.0) {
            const double r = rms[n] > 0.0 ? rms[n] : 1.0;
            const double gap = (best - second) / r;
            if (gap <= eps) {
                const int k = atomicAdd(&ccount[n], 1);
                if (k < cap) {
                    alq_flip_t f;
                    f.layer = layer; f.pad = 1; f.idx = si * C + c; f.delta = 2.0 * (best - second) + 1e-12 * r;
                    cand[n * cap + k] = f;
                    ckey[n * cap + k] = gap;
                }
            }
        }
    }
}

// fc: W [out][F] with F in ACTIVATION-MEMORY order of the input (the host permutes the reference's flatten order once)
__global__ void r64_fc_fwd(const double *A, long long F, const double *Wt, const double *bias, int nout, long long N, int layer, int relu,
                           const alq_flip_t *fl, int nf, double *pre, double *act, unsigned char *keep) {
    __shared__ double red[TPB];
    const long long job = blockIdx.x;           // (n, o)
    const long long n = job / nout;
    const int o = (int)(job % nout);
    const double *a = A + n * F, *w = Wt + (long long)o * F;
    double s = 0.0;
    for (long long j = threadIdx.x; j < F; j += TPB) s = fma(a[j], w[j], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = TPB / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = red[0] + bias[o];
        pre[job] = v;
        bool k = true;
        if (relu) k = (v > 0.0) != flipped(fl, nf, n, layer, o, nullptr);
        keep[job] = k ? 1 : 0;
        act[job] = k ? (relu ? fabs(v) : v) : 0.0;
    }
}

// ---- backward -------------------------------------------------------------------------------------------------------------
__global__ void r64_mask(const double *dact, const unsigned char *keep, long long total, double *dpre) {
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) dpre[i] = keep[i] ? dact[i] : 0.0;
}

// conv backward-data: d in[i][ci] += sum_t sum_co dpre[i - t + lo][co] W[t][ci][co]; channels [0, Ca) go to dA, the rest to dB
__global__ void r64_conv_bwd(const double *dpre, int Co, int D, int H, int Wd, const double *Wt, int kd, int kh, int kw, int lz, int ly, int lx,
                             int Ca, int Cb, long long N, double *dA, double *dB) {
    const long long vox = (long long)D * H * Wd;
    const int Ci = Ca + Cb;
    const long long total = N * vox * Ci;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int ci = (int)(i % Ci);
        const long long v = (i / Ci) % vox, n = i / (Ci * vox);
        const int x = (int)(v % Wd), y = (int)((v / Wd) % H), z = (int)(v / ((long long)Wd * H));
        double s = 0.0;
        for (int tz = 0; tz < kd; ++tz) {
            const int oz = z - tz + lz;
            if (oz < 0 || oz >= D) continue;
            for (int ty = 0; ty < kh; ++ty) {
                const int oy = y - ty + ly;
                if (oy < 0 || oy >= H) continue;
                for (int tx = 0; tx < kw; ++tx) {
                    const int ox = x - tx + lx;
                    if (ox < 0 || ox >= Wd) continue;
                    const double *d = dpre + (n * vox + ((long long)oz * H + oy) * Wd + ox) * Co;
                    const double *w = Wt + ((long long)((tz * kh + ty) * kw + tx) * Ci + ci) * Co;
                    for (int c = 0; c < Co; ++c) s = fma(d[c], w[c], s);
                }
            }
        }
        if (ci < Ca) { if (dA) dA[(n * vox + v) * Ca + ci] += s; }
        else if (dB) dB[(n * vox + v) * Cb + (ci - Ca)] += s;
    }
}

// conv_transpose backward-data: d in[i][ci] += sum_t sum_co dpre[s i + t - lo][co] W[t][co][ci]
__global__ void r64_convt_bwd(const double *dpre, int Co, int ID, int IH, int IW, const double *Wt, int kd, int kh, int kw, int sz, int sy, int sx,
                              int lz, int ly, int lx, int Ci, long long N, double *dA) {
    const int OD = ID * sz, OH = IH * sy, OW = IW * sx;
    const long long ovox = (long long)OD * OH * OW, ivox = (long long)ID * IH * IW, total = N * ivox * Ci;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int ci = (int)(i % Ci);
        const long long v = (i / Ci) % ivox, n = i / (Ci * ivox);
        const int x = (int)(v % IW), y = (int)((v / IW) % IH), z = (int)(v / ((long long)IW * IH));
        double s = 0.0;
        for (int tz = 0; tz < kd; ++tz) {
            const int oz = z * sz + tz - lz;
            if (oz < 0 || oz >= OD) continue;
            for (int ty = 0; ty < kh; ++ty) {
                const int oy = y * sy + ty - ly;
                if (oy < 0 || oy >= OH) continue;
                for (int tx = 0; tx < kw; ++tx) {
                    const int ox = x * sx + tx - lx;
                    if (ox < 0 || ox >= OW) continue;
                    const double *d = dpre + (n * ovox + ((long long)oz * OH + oy) * OW + ox) * Co;
                    const double *w = Wt + ((long long)((tz * kh + ty) * kw + tx) * Co) * Ci + ci;
                    for (int c = 0; c < Co; ++c) s = fma(d[c], w[(long long)c * Ci], s);
                }
            }
        }
        dA[i] += s;
    }
}

__global__ void r64_pool_bwd(const double *dout, const int *argmax, int C, long long ovox, long long ivox, long long N, double *din) {
    const long long total = N * ovox * C;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int c = (int)(i % C);
        const long long n = i / (C * ovox);
        din[(n * ivox + argmax[i]) * C + c] += dout[i];       // windows are disjoint (window = stride): one writer per element
    }
}

__global__ void r64_fc_bwd(const double *dpre, int nout, const double *Wt, long long F, long long N, double *din) {
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < N * F; i += (long long)gridDim.x * TPB) {
        const long long n = i / F, j = i - n * F;
        double s = 0.0;
        for (int o = 0; o < nout; ++o) s = fma(dpre[n * nout + o], Wt[(long long)o * F + j], s);
        din[i] += s;
    }
}

// ---- layer sums -----------------------------------------------------------------------------------------------------------
// out[n][v] = sum_c A[n][v][c] (+ sum_c B[n][v][c])
__global__ void r64_chansum(const double *A, int Ca, const double *B, int Cb, long long rows, double *out) {
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < rows; i += (long long)gridDim.x * TPB) {
        double s = 0.0;
        for (int c = 0; c < Ca; ++c) s += A[i * Ca + c];
        for (int c = 0; c < Cb; ++c) s += B[i * Cb + c];
        out[i] = s;
    }
}

// one block per sample: out[n] = sum over the sample of f(...)
//   mode 0 (conv):  sum_x dsum[x] (box_k(asum)[x] + 1), SAME zero padding, same grid for both fields
//   mode 1 (convT): sum_q asum[q] sum_t dsum[s q + t - lo] + sum_p dsum[p]     (asum on the input grid, dsum on the output grid)
//   mode 2 (fc):    (sum dsum) (sum asum + 1)
//   mode 3:         sum of squares of asum (rms of a tensor)
__global__ void r64_score(int mode, const double *asum, const double *dsum, int D, int H, int Wd, int kd, int kh, int kw, int sz, int sy, int sx,
                          int lz, int ly, int lx, long long na, long long nd, double *out, int stride, int col) {
    __shared__ double red[TPB], red2[TPB];
    const long long n = blockIdx.x;
    const double *a = asum + n * na, *d = dsum ? dsum + n * nd : nullptr;
    double s = 0.0, s2 = 0.0;
    if (mode == 0) {
        for (long long v = threadIdx.x; v < nd; v += TPB) {
            const int x = (int)(v % Wd), y = (int)((v / Wd) % H), z = (int)(v / ((long long)Wd * H));
            double b = 1.0;
            for (int tz = 0; tz < kd; ++tz) {
                const int iz = z + tz - lz;
                if (iz < 0 || iz >= D) continue;
                for (int ty = 0; ty < kh; ++ty) {
                    const int iy = y + ty - ly;
                    if (iy < 0 || iy >= H) continue;
                    for (int tx = 0; tx < kw; ++tx) {
                        const int ix = x + tx - lx;
                        if (ix < 0 || ix >= Wd) continue;
                        b += a[((long long)iz * H + iy) * Wd + ix];
                    }
                }
            }
            s = fma(d[v], b, s);
        }
    } else if (mode == 1) {      // D, H, Wd = the INPUT grid
        const int OD = D * sz, OH = H * sy, OW = Wd * sx;
        for (long long v = threadIdx.x; v < na; v += TPB) {
            const int x = (int)(v % Wd), y = (int)((v / Wd) % H), z = (int)(v / ((long long)Wd * H));
            double b = 0.0;
            for (int tz = 0; tz < kd; ++tz) {
                const int oz = z * sz + tz - lz;
                if (oz < 0 || oz >= OD) continue;
                for (int ty = 0; ty < kh; ++ty) {
                    const int oy = y * sy + ty - ly;
                    if (oy < 0 || oy >= OH) continue;
                    for (int tx = 0; tx < kw; ++tx) {
                        const int ox = x * sx + tx - lx;
                        if (ox < 0 || ox >= OW) continue;
                        b += d[((long long)oz * OH + oy) * OW + ox];
                    }
                }
            }
            s = fma(a[v], b, s);
        }
        for (long long v = threadIdx.x; v < nd; v += TPB) s += d[v];
    } else if (mode == 2) {
        for (long long v = threadIdx.x; v < na; v += TPB) s += a[v];
        for (long long v = threadIdx.x; v < nd; v += TPB) s2 += d[v];
    } else {
        for (long long v = threadIdx.x; v < na; v += TPB) s = fma(a[v], a[v], s);
    }
    red[threadIdx.x] = s; red2[threadIdx.x] = s2;
    __syncthreads();
    for (int h = TPB / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { red[threadIdx.x] += red[threadIdx.x + h]; red2[threadIdx.x] += red2[threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double r = red[0];
        if (mode == 2) r = red2[0] * (red[0] + 1.0);
        if (mode == 3) r = sqrt(red[0] / (double)na);
        out[n * stride + col] = r;
    }
}

// fragile ReLU inputs: |pre| <= eps * rms[n]
__global__ void r64_fragile(const double *pre, long long per, long long N, int layer, const double *rms, double eps, int cap,
                            alq_flip_t *cand, double *ckey, int *ccount) {
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < N * per; i += (long long)gridDim.x * TPB) {
        const long long n = i / per;
        const double r = rms[n] > 0.0 ? rms[n] : 1.0;
        const double k = fabs(pre[i]) / r;
        if (k <= eps) {
            const int s = atomicAdd(&ccount[n], 1);
            if (s < cap) {
                alq_flip_t f;
                f.layer = layer; f.pad = 0; f.idx = i - n * per; f.delta = 0.0;
                cand[n * cap + s] = f;
                ckey[n * cap + s] = k;
            }
        }
    }
}

int same_lo(int in, int k, int s) {
    const int out = (in + s - 1) / s;
    const int total = std::max((out - 1) * s + k - in, 0);
    return total / 2;
}

}  // namespace
}  // namespace alq

using namespace alq;

extern "C" int alq_ref64_scores(alq_ctx *ctx, const alq_layer_t *specs, int n_layers, const int32_t in_dims[4],
                                const double *const *h_dW, const double *const *h_db, const float *d_x, const int64_t *d_rows, int N,
                                const alq_flip_t *d_flips, int flips_per_sample, double eps, int cand_cap,
                                double *d_logits, double *d_S, double *d_rms, alq_flip_t *d_cand, double *d_cand_key, int32_t *d_cand_count) {
    ALQ_REQUIRE(ctx && specs && in_dims && h_dW && h_db && d_x && d_logits && d_S, ALQ_EINVAL, "alq_ref64_scores: null argument");
    ALQ_REQUIRE(n_layers > 0 && N > 0, ALQ_EINVAL, "alq_ref64_scores: empty model or batch");
    ALQ_REQUIRE(cand_cap == 0 || (d_cand && d_cand_key && d_cand_count), ALQ_EINVAL, "alq_ref64_scores: candidate buffers missing");
    hipStream_t st = ctx->stream;
    std::vector<void *> owned;
    auto fail = [&](int rc) { (void)hipStreamSynchronize(st); for (void *p : owned) (void)hipFree(p); return rc; };
    auto dal = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        owned.push_back(p);
        return p;
    };
#define R64_ALLOC(ptr, type, count)                                                                                          \
    do {                                                                                                                      \
        ptr = static_cast<type *>(dal((size_t)(count) * sizeof(type)));                                                       \
        if (!ptr) { set_error("alq_ref64_scores: hipMalloc of %zu bytes failed", (size_t)(count) * sizeof(type)); return fail(ALQ_ENOMEM); } \
    } while (0)
#define R64_CHECK()                                                                                                           \
    do {                                                                                                                      \
        hipError_t e_ = hipGetLastError();                                                                                    \
        if (e_ != hipSuccess) { set_error("alq_ref64_scores: launch failed: %s", hipGetErrorString(e_)); return fail(ALQ_EHIP); } \
    } while (0)

    // ---- shapes ---------------------------------------------------------------------------------------------------------
    std::vector<R64Layer> L(n_layers);
    T64 x0;
    x0.D = in_dims[0]; x0.H = in_dims[1]; x0.W = in_dims[2]; x0.C = in_dims[3];
    const long long epp = x0.elems();
    R64_ALLOC(x0.p, double, (long long)N * epp);
    int np = 0;
    {
        T64 cur = x0;
        bool flat = false;
        long long curF = 0;
        for (int i = 0; i < n_layers; ++i) {
            R64Layer &ly = L[i];
            ly.spec = specs[i];
            const alq_layer_t &sp = ly.spec;
            ly.src_main = i - 1;
            ly.src_skip = sp.skip_src;
            T64 in = cur;
            ly.Ca = 0;
            if (sp.skip_src >= 0) {
                if (!(sp.skip_src < i - 1) || flat) { set_error("alq_ref64_scores: layer %d: bad skip source", i); return fail(ALQ_EUNSUPPORTED); }
                const T64 &s = L[sp.skip_src].act;
                if (!(s.D == cur.D && s.H == cur.H && s.W == cur.W)) { set_error("alq_ref64_scores: layer %d: concat of maps of different size", i); return fail(ALQ_EUNSUPPORTED); }
                ly.Ca = s.C;
                in.C = cur.C + s.C;
            }
            ly.Cin = in.C;
            T64 out = in;
            if (sp.type == ALQ_CONV) {
                if (flat || sp.s[0] != 1 || sp.s[1] != 1 || sp.s[2] != 1) { set_error("alq_ref64_scores: layer %d: strided conv", i); return fail(ALQ_EUNSUPPORTED); }
                out.C = sp.cout;
                ly.lo[0] = same_lo(in.D, sp.k[0], 1); ly.lo[1] = same_lo(in.H, sp.k[1], 1); ly.lo[2] = same_lo(in.W, sp.k[2], 1);
                ly.pidx = np++;
            } else if (sp.type == ALQ_CONVT) {
                if (flat || sp.skip_src >= 0) { set_error("alq_ref64_scores: layer %d: conv_transpose on a concat / flat input", i); return fail(ALQ_EUNSUPPORTED); }
                out.D = in.D * sp.s[0]; out.H = in.H * sp.s[1]; out.W = in.W * sp.s[2]; out.C = sp.cout;
                ly.lo[0] = same_lo(out.D, sp.k[0], sp.s[0]); ly.lo[1] = same_lo(out.H, sp.k[1], sp.s[1]); ly.lo[2] = same_lo(out.W, sp.k[2], sp.s[2]);
                ly.pidx = np++;
            } else if (sp.type == ALQ_POOL) {
                if (flat || sp.skip_src >= 0 || sp.k[0] != sp.s[0] || sp.k[1] != sp.s[1] || sp.k[2] != sp.s[2]) { set_error("alq_ref64_scores: layer %d: pool window != stride", i); return fail(ALQ_EUNSUPPORTED); }
                out.D = (in.D + sp.s[0] - 1) / sp.s[0]; out.H = (in.H + sp.s[1] - 1) / sp.s[1]; out.W = (in.W + sp.s[2] - 1) / sp.s[2];
                ly.lo[0] = same_lo(in.D, sp.k[0], sp.s[0]); ly.lo[1] = same_lo(in.H, sp.k[1], sp.s[1]); ly.lo[2] = same_lo(in.W, sp.k[2], sp.s[2]);
            } else if (sp.type == ALQ_FC) {
                if (sp.skip_src >= 0) { set_error("alq_ref64_scores: layer %d: fc on a concat", i); return fail(ALQ_EUNSUPPORTED); }
                ly.flat_in = true;
                ly.F = flat ? curF : in.elems();
                out.D = out.H = out.W = 1; out.C = sp.cout;
                ly.pidx = np++;
                flat = true;
                curF = sp.cout;
            } else { set_error("alq_ref64_scores: layer %d: unknown type", i); return fail(ALQ_EINVAL); }
            const long long oe = (long long)N * out.elems();
            ly.act = out; ly.pre = out; ly.dact = out;
            R64_ALLOC(ly.act.p, double, oe);
            R64_ALLOC(ly.dact.p, double, oe);
            if (sp.type == ALQ_POOL) R64_ALLOC(ly.argmax, int, oe);
            else { R64_ALLOC(ly.pre.p, double, oe); R64_ALLOC(ly.keep, unsigned char, oe); }
            if (ly.pidx >= 0) { ly.W = h_dW[ly.pidx]; ly.b = h_db[ly.pidx]; if (!ly.W || !ly.b) { set_error("alq_ref64_scores: weights of layer %d missing", i); return fail(ALQ_EINVAL); } }
            cur = out;
        }
        if (!(L.back().spec.type == ALQ_FC)) { set_error("alq_ref64_scores: the last layer must be the fc head"); return fail(ALQ_EUNSUPPORTED); }
    }
    const int nclass = L.back().spec.cout;
    if (nclass != 2) { set_error("alq_ref64_scores: two-class head only (the unit cotangent of the binary score path)"); return fail(ALQ_EUNSUPPORTED); }
    double *rms = nullptr, *tmpA = nullptr, *tmpD = nullptr, *dpre = nullptr;
    long long max_elems = epp, max_vox = x0.vox();
    for (auto &ly : L) { max_elems = std::max(max_elems, ly.act.elems()); max_vox = std::max(max_vox, ly.act.vox()); }
    R64_ALLOC(rms, double, (long long)N * n_layers);
    R64_ALLOC(tmpA, double, (long long)N * std::max(max_vox, max_elems));
    R64_ALLOC(tmpD, double, (long long)N * std::max(max_vox, (long long)1));
    R64_ALLOC(dpre, double, (long long)N * max_elems);
    if (hipMemsetAsync(rms, 0, (size_t)N * n_layers * sizeof(double), st) != hipSuccess) return fail(ALQ_EHIP);
    if (cand_cap > 0 && hipMemsetAsync(d_cand_count, 0, (size_t)N * sizeof(int32_t), st) != hipSuccess) return fail(ALQ_EHIP);

    hipLaunchKernelGGL(r64_load, dim3(nblk((long long)N * epp)), dim3(TPB), 0, st, d_x, (const long long *)d_rows, (long long)N, epp, x0.p);
    R64_CHECK();
    auto act_of = [&](int idx) -> const T64 & { return idx < 0 ? x0 : L[idx].act; };

    // ---- forward --------------------------------------------------------------------------------------------------------
    for (int i = 0; i < n_layers; ++i) {
        R64Layer &ly = L[i];
        const alq_layer_t &sp = ly.spec;
        const T64 &mainT = act_of(ly.src_main);
        const long long oe = (long long)N * ly.act.elems();
        if (sp.type == ALQ_CONV) {
            const T64 *sk = ly.src_skip >= 0 ? &L[ly.src_skip].act : nullptr;
            hipLaunchKernelGGL(r64_conv_fwd, dim3(nblk(oe)), dim3(TPB), 0, st, sk ? sk->p : mainT.p, sk ? sk->C : 0, mainT.p, mainT.C, mainT.D, mainT.H, mainT.W,
                               ly.W, ly.b, sp.k[0], sp.k[1], sp.k[2], ly.lo[0], ly.lo[1], ly.lo[2], sp.cout, (long long)N, i, sp.relu, d_flips, flips_per_sample,
                               ly.pre.p, ly.act.p, ly.keep);
        } else if (sp.type == ALQ_CONVT) {
            hipLaunchKernelGGL(r64_convt_fwd, dim3(nblk(oe)), dim3(TPB), 0, st, mainT.p, mainT.C, mainT.D, mainT.H, mainT.W, ly.W, ly.b, sp.k[0], sp.k[1], sp.k[2],
                               sp.s[0], sp.s[1], sp.s[2], ly.lo[0], ly.lo[1], ly.lo[2], sp.cout, (long long)N, i, sp.relu, d_flips, flips_per_sample,
                               ly.pre.p, ly.act.p, ly.keep);
        } else if (sp.type == ALQ_POOL) {
            // rms of the pool's input (for the near-tie listing)
            hipLaunchKernelGGL(r64_score, dim3(N), dim3(TPB), 0, st, 3, mainT.p, (const double *)nullptr, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, mainT.elems(), 0LL, rms + (long long)i * N, 1, 0);
            R64_CHECK();
            hipLaunchKernelGGL(r64_pool_fwd, dim3(nblk(oe)), dim3(TPB), 0, st, mainT.p, mainT.C, mainT.D, mainT.H, mainT.W, sp.k[0], sp.k[1], sp.k[2], ly.lo[0], ly.lo[1], ly.lo[2],
                               ly.act.D, ly.act.H, ly.act.W, (long long)N, i, d_flips, flips_per_sample, ly.act.p, ly.argmax,
                               (const double *)(rms + (long long)i * N), eps, cand_cap, cand_cap > 0 ? d_cand : nullptr, d_cand_key, d_cand_count);
        } else {
            hipLaunchKernelGGL(r64_fc_fwd, dim3((unsigned)((long long)N * sp.cout)), dim3(TPB), 0, st, mainT.p, ly.F, ly.W, ly.b, sp.cout, (long long)N, i, sp.relu,
                               d_flips, flips_per_sample, ly.pre.p, ly.act.p, ly.keep);
        }
        R64_CHECK();
        if (sp.type != ALQ_POOL && sp.relu) {      // rms pre-activation per sample, then the fragile units of this layer
            hipLaunchKernelGGL(r64_score, dim3(N), dim3(TPB), 0, st, 3, ly.pre.p, (const double *)nullptr, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, ly.pre.elems(), 0LL, rms + (long long)i * N, 1, 0);
            R64_CHECK();
            if (cand_cap > 0) {
                hipLaunchKernelGGL(r64_fragile, dim3(nblk(oe)), dim3(TPB), 0, st, ly.pre.p, ly.pre.elems(), (long long)N, i, (const double *)(rms + (long long)i * N), eps, cand_cap,
                                   d_cand, d_cand_key, d_cand_count);
                R64_CHECK();
            }
        }
        if (hipMemsetAsync(ly.dact.p, 0, (size_t)oe * sizeof(double), st) != hipSuccess) return fail(ALQ_EHIP);
    }
    if (hipMemcpyAsync(d_logits, L.back().pre.p, (size_t)N * nclass * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(ALQ_EHIP);
    if (d_rms && hipMemcpyAsync(d_rms, rms, (size_t)N * n_layers * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(ALQ_EHIP);

    // ---- backward: unit cotangent (+1, -1) on the logits (d(z0 - z1)) ----------------------------------------------------
    {
        std::vector<double> unit((size_t)N * 2);
        for (int n = 0; n < N; ++n) { unit[2 * n] = 1.0; unit[2 * n + 1] = -1.0; }
        if (hipMemcpyAsync(L.back().dact.p, unit.data(), unit.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return fail(ALQ_EHIP);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(ALQ_EHIP);
    }
    for (int i = n_layers - 1; i >= 0; --i) {
        R64Layer &ly = L[i];
        const alq_layer_t &sp = ly.spec;
        const long long oe = (long long)N * ly.act.elems();
        const T64 &mainT = act_of(ly.src_main);
        double *dmain = ly.src_main >= 0 ? L[ly.src_main].dact.p : nullptr;
        if (sp.type == ALQ_POOL) {
            if (dmain)
                hipLaunchKernelGGL(r64_pool_bwd, dim3(nblk(oe)), dim3(TPB), 0, st, ly.dact.p, ly.argmax, ly.act.C, ly.act.vox(), mainT.vox(), (long long)N, dmain);
            R64_CHECK();
            continue;
        }
        hipLaunchKernelGGL(r64_mask, dim3(nblk(oe)), dim3(TPB), 0, st, ly.dact.p, ly.keep, oe, dpre);
        R64_CHECK();
        const T64 *sk = ly.src_skip >= 0 ? &L[ly.src_skip].act : nullptr;
        // layer sum S[n][t]
        if (sp.type == ALQ_FC) {
            hipLaunchKernelGGL(r64_score, dim3(N), dim3(TPB), 0, st, 2, mainT.p, dpre, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, ly.F, (long long)sp.cout, d_S, np, ly.pidx);
        } else {
            const long long rows_in = (long long)N * mainT.vox(), rows_out = (long long)N * ly.act.vox();
            hipLaunchKernelGGL(r64_chansum, dim3(nblk(rows_in)), dim3(TPB), 0, st, sk ? sk->p : mainT.p, sk ? sk->C : 0, mainT.p, mainT.C, rows_in, tmpA);
            hipLaunchKernelGGL(r64_chansum, dim3(nblk(rows_out)), dim3(TPB), 0, st, dpre, sp.cout, dpre, 0, rows_out, tmpD);
            R64_CHECK();
            if (sp.type == ALQ_CONV)
                hipLaunchKernelGGL(r64_score, dim3(N), dim3(TPB), 0, st, 0, tmpA, tmpD, mainT.D, mainT.H, mainT.W, sp.k[0], sp.k[1], sp.k[2], 1, 1, 1,
                                   ly.lo[0], ly.lo[1], ly.lo[2], mainT.vox(), ly.act.vox(), d_S, np, ly.pidx);
            else
                hipLaunchKernelGGL(r64_score, dim3(N), dim3(TPB), 0, st, 1, tmpA, tmpD, mainT.D, mainT.H, mainT.W, sp.k[0], sp.k[1], sp.k[2], sp.s[0], sp.s[1], sp.s[2],
                                   ly.lo[0], ly.lo[1], ly.lo[2], mainT.vox(), ly.act.vox(), d_S, np, ly.pidx);
        }
        R64_CHECK();
        if (ly.pidx == 0) break;      // nothing below the first parameterised layer needs a cotangent
        double *dskip = ly.src_skip >= 0 ? L[ly.src_skip].dact.p : nullptr;
        if (sp.type == ALQ_FC) {
            if (dmain) hipLaunchKernelGGL(r64_fc_bwd, dim3(nblk((long long)N * ly.F)), dim3(TPB), 0, st, dpre, sp.cout, ly.W, ly.F, (long long)N, dmain);
        } else if (sp.type == ALQ_CONV) {
            const int Ca = sk ? sk->C : 0;
            hipLaunchKernelGGL(r64_conv_bwd, dim3(nblk((long long)N * mainT.vox() * ly.Cin)), dim3(TPB), 0, st, dpre, sp.cout, mainT.D, mainT.H, mainT.W, ly.W,
                               sp.k[0], sp.k[1], sp.k[2], ly.lo[0], ly.lo[1], ly.lo[2], Ca, mainT.C, (long long)N, dskip, dmain);
        } else {
            if (dmain)
                hipLaunchKernelGGL(r64_convt_bwd, dim3(nblk((long long)N * mainT.elems())), dim3(TPB), 0, st, dpre, sp.cout, mainT.D, mainT.H, mainT.W, ly.W,
                                   sp.k[0], sp.k[1], sp.k[2], sp.s[0], sp.s[1], sp.s[2], ly.lo[0], ly.lo[1], ly.lo[2], mainT.C, (long long)N, dmain);
        }
        R64_CHECK();
    }
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("alq_ref64_scores: %s", hipGetErrorString(hipGetLastError())); return fail(ALQ_EHIP); }
    for (void *p : owned) (void)hipFree(p);
    return ALQ_OK;
#undef R64_ALLOC
#undef R64_CHECK
}
