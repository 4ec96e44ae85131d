// Cosine-similarity side of the representativeness strategies (query_multimg 'rep-entropy', PW_NNAL.py:284-351, and
// 'core-set', :353-451): the reference forms  dots = F.T @ G  on float64 feature matrices with NumPy, divides by the
// outer product of the column norms and then runs greedy selections over the similarity matrix in Python loops.
// Here: features stay on the device as fp32 rows [samples, features] (what alq_forward returns), products are
// accumulated in fp64 (the reference's float64 arithmetic on fp32-valued features: every product is exact, only the
// summation order differs), and the greedy steps are streaming reductions over the fp64 similarity matrix.
#include "alq_internal.h"

namespace alq {

#define ALQ_LAUNCH_CHECK() ALQ_HIP(hipGetLastError())

// norms[i] = sqrt(sum_f A[i, f]^2), one wave per row
__global__ __launch_bounds__(256) void row_norm_kernel(const float *A, long long n, int f, double *norms) {
    const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0;
    for (int k = lane; k < f; k += 64) { const double v = (double)A[row * f + k]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) norms[row] = sqrt(s);
}

// C[i, j] = (sum_f A[i, f] B[j, f]) / (na[i] nb[j])   (na / nb null: plain dot products)
// 64 x 64 output tile per workgroup, 16-deep K slices through LDS, 4 x 4 fp64 accumulators per thread.
constexpr int SG_T = 64, SG_K = 16;
__global__ __launch_bounds__(256) void cos_gemm_kernel(const float *A, long long n, const float *B, int b, int f, const double *na,
                                                       const double *nb, double *C) {
    __shared__ float As[SG_K][SG_T + 1], Bs[SG_K][SG_T + 1];
    const long long i0 = (long long)blockIdx.x * SG_T;
    const int j0 = blockIdx.y * SG_T;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0;
    for (int k0 = 0; k0 < f; k0 += SG_K) {
        for (int e = threadIdx.x; e < SG_T * SG_K; e += 256) {
            const int r = e / SG_K, k = e - r * SG_K;
            As[k][r] = (i0 + r < n && k0 + k < f) ? A[(i0 + r) * f + k0 + k] : 0.f;
            Bs[k][r] = (j0 + r < b && k0 + k < f) ? B[(long long)(j0 + r) * f + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SG_K; ++k) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = (double)As[k][ty * 4 + a];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = (double)Bs[k][tx * 4 + c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = fma(av[a], bv[c], acc[a][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const long long i = i0 + ty * 4 + a;
        if (i >= n) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            if (j >= b) continue;
            double v = acc[a][c];
            if (na) v = v / (na[i] * nb[j]);
            C[i * b + j] = v;
        }
    }
}

// part[chunk][j] = sum_{i in chunk} max(cmax[i], S[i, j])   (cmax null: S[i, j]); rows with skip[i] != 0 are left out
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_max_kernel(const double *S, long long n, int b, const double *cmax, const unsigned char *skip,
                                                         double *part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const long long r0 = (long long)blockIdx.y * CS_ROWS;
    if (j >= b) return;
    double s = 0;
    const long long r1 = min(n, r0 + CS_ROWS);
    for (long long i = r0; i < r1; ++i) {
        if (skip && skip[i]) continue;
        const double v = S[i * b + j];
        s += cmax ? fmax(cmax[i], v) : v;
    }
    part[(long long)blockIdx.y * b + j] = s;
}
__global__ void colsum_finish_kernel(const double *part, int nchunk, int b, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= b) return;
    double s = 0;
    for (int c = 0; c < nchunk; ++c) s += part[(long long)c * b + j];
    out[j] = s;
}
// v[i] = max(v[i], S[i, j])  (first != 0: v[i] = S[i, j])
__global__ void take_colmax_kernel(const double *S, long long n, int b, int j, int first, double *v) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double s = S[i * b + j];
        v[i] = first ? s : fmax(v[i], s);
    }
}
// v[j] = max(v[j], max_i S[i, j])   (column maxima of a [t, n] block folded into a running vector)
__global__ void fold_rowmax_kernel(const double *S, int t, long long n, double *v) {
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < n; j += (long long)gridDim.x * blockDim.x) {
        double m = v[j];
        for (int i = 0; i < t; ++i) m = fmax(m, S[(long long)i * n + j]);
        v[j] = m;
    }
}

static unsigned sgrid(long long n) { long long g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 1048576 ? 1048576 : g)); }

}  // namespace alq

using namespace alq;

extern "C" {

int alq_row_norms(alq_ctx *ctx, const float *d_A, int64_t n, int f, double *d_norms) {
    ALQ_REQUIRE(ctx && (n == 0 || (d_A && d_norms)) && f >= 1 && n >= 0, ALQ_EINVAL, "alq_row_norms: bad argument");
    if (n == 0) return ALQ_OK;
    ALQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_A, (long long)n, f, d_norms);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int alq_cosine_sims(alq_ctx *ctx, const float *d_A, int64_t n, const float *d_B, int b, int f, const double *d_na, const double *d_nb,
                    double *d_C) {
    ALQ_REQUIRE(ctx && n >= 0 && b >= 0 && f >= 1 && ((d_na == nullptr) == (d_nb == nullptr)), ALQ_EINVAL, "alq_cosine_sims: bad argument");
    if (n == 0 || b == 0) return ALQ_OK;
    ALQ_REQUIRE(d_A && d_B && d_C, ALQ_EINVAL, "alq_cosine_sims: null tensor");
    ALQ_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, PROF_REDUCE, 2.0 * (double)n * b * f);
    hipLaunchKernelGGL(cos_gemm_kernel, dim3((unsigned)((n + SG_T - 1) / SG_T), (unsigned)((b + SG_T - 1) / SG_T)), dim3(256), 0, ctx->stream,
                       d_A, (long long)n, d_B, b, f, d_na, d_nb, d_C);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

size_t alq_colsum_work_bytes(int64_t n, int b) { return (size_t)((n + CS_ROWS - 1) / CS_ROWS) * (size_t)b * sizeof(double); }

int alq_colsum_max(alq_ctx *ctx, const double *d_S, int64_t n, int b, const double *d_cmax, const unsigned char *d_skip, double *d_out,
                   void *d_work) {
    ALQ_REQUIRE(ctx && d_S && d_out && d_work && n >= 1 && b >= 1, ALQ_EINVAL, "alq_colsum_max: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    const int nchunk = (int)((n + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(colsum_max_kernel, dim3((unsigned)((b + 255) / 256), (unsigned)nchunk), dim3(256), 0, ctx->stream, d_S, (long long)n, b,
                       d_cmax, d_skip, reinterpret_cast<double *>(d_work));
    ALQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const double *>(d_work), nchunk, b, d_out);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int alq_take_colmax(alq_ctx *ctx, const double *d_S, int64_t n, int b, int j, int first, double *d_v) {
    ALQ_REQUIRE(ctx && d_S && d_v && n >= 1 && j >= 0 && j < b, ALQ_EINVAL, "alq_take_colmax: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(take_colmax_kernel, dim3(sgrid(n)), dim3(256), 0, ctx->stream, d_S, (long long)n, b, j, first, d_v);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

int alq_fold_rowmax(alq_ctx *ctx, const double *d_S, int t, int64_t n, double *d_v) {
    ALQ_REQUIRE(ctx && d_S && d_v && n >= 1 && t >= 1, ALQ_EINVAL, "alq_fold_rowmax: bad argument");
    ALQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fold_rowmax_kernel, dim3(sgrid(n)), dim3(256), 0, ctx->stream, d_S, t, (long long)n, d_v);
    ALQ_LAUNCH_CHECK();
    return ALQ_OK;
}

}  // extern "C"
