// Top-B "most uncertain" selection = full ascending sort of (|p-0.5| as fp64 bits, index) pairs
// with a bitonic network (replaces np.argsort(np.abs(posts-.5))[:B], PW_NNAL.py:64,109,730).
// The index is part of the sort key, so equal scores come out in ascending index order
// whatever the network does (the tie rule this build defines; numpy's default sort is unstable).
// Steps with partner distance < 2048 run in LDS (one 2048-pair chunk per workgroup, 32 KiB);
// only the wider steps stream the array through HBM.
#include <algorithm>

#include "alq_internal.h"

namespace alq {

struct KeyIdx {
    unsigned long long key;
    unsigned long long idx;
};

constexpr int CH = 2048;   // pairs per workgroup chunk (256 threads x 8)

__device__ inline bool kless(const KeyIdx &a, const KeyIdx &b) {
    return a.key < b.key || (a.key == b.key && a.idx < b.idx);
}

__device__ inline void cswap(KeyIdx &a, KeyIdx &b, bool ascending) {
    const bool sw = ascending ? kless(b, a) : kless(a, b);
    if (sw) { const KeyIdx t = a; a = b; b = t; }
}

__global__ void topk_init_kernel(const double *keys, long long n, long long P, KeyIdx *w) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P;
         i += (long long)gridDim.x * blockDim.x) {
        KeyIdx e;
        if (i < n) {
            e.key = (unsigned long long)__double_as_longlong(keys[i]);   // keys are >= 0: bits are monotone
            e.idx = (unsigned long long)i;
        } else {
            e.key = ~0ull;
            e.idx = ~0ull;
        }
        w[i] = e;
    }
}

// all steps with j < CH for k in [kfirst, klast] (k = subsequence length), chunk-local
__global__ __launch_bounds__(256) void bitonic_local_kernel(KeyIdx *w, long long kfirst, long long klast) {
    __shared__ KeyIdx s[CH];
    const long long base = (long long)blockIdx.x * CH;
    for (int t = threadIdx.x; t < CH; t += 256) s[t] = w[base + t];
    __syncthreads();
    for (long long k = kfirst; k <= klast; k <<= 1) {
        long long jstart = k >> 1;
        if (jstart >= CH) jstart = CH >> 1;
        for (int j = (int)jstart; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < CH / 2; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const int l = i | j;
                const bool asc = (((base + i) & k) == 0);
                cswap(s[i], s[l], asc);
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < CH; t += 256) w[base + t] = s[t];
}

__global__ void bitonic_global_kernel(KeyIdx *w, long long P, long long k, long long j) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < P / 2;
         t += (long long)gridDim.x * blockDim.x) {
        const long long i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const long long l = i | j;
        KeyIdx a = w[i], b = w[l];
        const bool asc = ((i & k) == 0);
        const bool sw = asc ? kless(b, a) : kless(a, b);
        if (sw) { w[i] = b; w[l] = a; }
    }
}

__global__ void topk_emit_kernel(const KeyIdx *w, long long B, long long *out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < B;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = (long long)w[i].idx;
}

static long long pow2ceil_ll(long long v) {
    long long p = CH;   // at least one chunk
    while (p < v) p <<= 1;
    return p;
}

size_t topk_work_bytes_impl(int64_t n) { return (size_t)pow2ceil_ll(n) * sizeof(KeyIdx); }

int topk_impl(alq_ctx *ctx, const double *d_keys, int64_t n, int64_t B, int64_t *d_out, void *d_work) {
    ALQ_REQUIRE(n >= 0 && B >= 0 && B <= n, ALQ_EINVAL, "topk: need 0 <= B <= n (B=%lld n=%lld)", (long long)B,
                (long long)n);
    if (B == 0) return ALQ_OK;
    KeyIdx *w = reinterpret_cast<KeyIdx *>(d_work);
    const long long P = pow2ceil_ll(n);
    const unsigned gblocks = (unsigned)std::min<long long>((P + 255) / 256, 256 * 16);
    ProfScope ps(ctx, PROF_REDUCE, 0);
    hipLaunchKernelGGL(topk_init_kernel, dim3(gblocks), dim3(256), 0, ctx->stream, d_keys, (long long)n, P, w);
    const unsigned chunks = (unsigned)(P / CH);
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(chunks), dim3(256), 0, ctx->stream, w, 2LL, (long long)CH);
    for (long long k = 2LL * CH; k <= P; k <<= 1) {
        for (long long j = k >> 1; j >= CH; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_kernel, dim3(gblocks), dim3(256), 0, ctx->stream, w, P, k, j);
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(chunks), dim3(256), 0, ctx->stream, w, k, k);
    }
    hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)std::min<long long>((B + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, w, (long long)B, reinterpret_cast<long long *>(d_out));
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
