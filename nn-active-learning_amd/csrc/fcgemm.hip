// Streaming GEMM for wide fully connected layers (NN.create_PW1's 6144 -> 4096 -> 4096): C = A * B + bias.
//
//   A [M x K] fp32 activations (M = patches of the batch), B [K x N] weights, C [M x N] fp32.
//
// Same arithmetic as the conv engines: every fp32 operand = hi + mid + lo bf16 (exact), six piece products on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  Unlike a convolution nothing is reused across taps, so both
// operands stream: a workgroup owns a 128 (patches) x 64 (features) tile of C and walks K in steps of 32;
// the weights are split and tiled on the host once (one contiguous 12 KB block per (feature tile, k-step), copied
// to LDS as it is), the activations are split on the way from registers to LDS; LDS is double buffered and the
// global loads of step k+1 are issued before the MFMAs of step k.
// LDS layout of either operand: [piece][k-group q = 0..3][row] x 16 bytes (8 bf16 along k): the 16 lanes of a
// ds_read_b128 lane group (8 rows of group q, 8 rows of group q+1) hit 16 different 16-byte bank granules.
// The weights are the MFMA A operand (rows = features), so a lane ends up with 4 consecutive features of one
// patch: 16-byte stores.
// F16 forward (round 6): fc1 / fc2 of a pass measure the per-patch maximum of their input (fc_rowmax_kernel: one read of [M x K]) and
// contract with fp16 pairs scaled per patch; bias and ReLU ride in the epilogue.  ALQ_NO_FC_F16_FWD=1 at model creation: bf16 triples.
// F16 (round 5): backward launches of a Fisher pass know a static bound on their input (the cotangent under the unit cotangent,
// chained down from the head like the conv launches' - model.hip, run_backward_main): they contract with fp16 PAIRS at their true
// scale, x 2^e = h + l, three products in one accumulator instead of six (c3d.hip's one-accumulator form; needs fp16 subnormals
// in the matrix cores: c3d_subnormals_ok), two LDS pieces per operand instead of three.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int FC_BM = 128, FC_BN = 64, FC_BK = 32;
constexpr int FC_XBYTES = 3 * 4 * FC_BM * 16;      // 24 KB
constexpr int FC_WBYTES = 3 * 4 * FC_BN * 16;      // 12 KB

__device__ inline unsigned fc_split2(float &a, float &b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    a -= __builtin_bit_cast(float, hb << 16);
    b -= __builtin_bit_cast(float, hb & 0xffff0000u);
    return hb;
}
__device__ inline unsigned fc_pack2(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}

struct FcGemmArgs {
    const float *A;
    const void *Bp;
    float *C;
    const float *bias;
    int M, N, K, lda, ldc, relu;
    float scale, inv;      // F16: 2^e_in and 2^-(e_in + e_w)
    // F16 forward launches (round 6): per-row (patch) maxima |A| (float bits, fc_rowmax_kernel) instead of one static bound: row m is
    // scaled by 2^(14 - ex_m), its results by the inverse; e_w = the weights' exponent
    const unsigned *row_amax;
    int e_w;
};

// max |A[m][:]| per row as float bits (one wave per row; rows beyond M are not written)
__global__ __launch_bounds__(256) void fc_rowmax_kernel(const float *A, int M, int K, int lda, unsigned *out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float *p = A + (size_t)row * lda;
    float m = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + k);
        m = fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) out[row] = __builtin_bit_cast(unsigned, m);
}

// scale exponent of a row with maximum bits fm: max |x| < 2^ex -> 2^(14 - ex) (c3d.hip's rule; an all-zero row: 0)
__device__ inline int fc_row_exp(unsigned fm) {
    const int ex = (int)((fm >> 23) & 255u) - 126;
    const int ce = 14 - ex;
    return fm ? (ce < 96 ? ce : 96) : 0;
}

// BN = features per workgroup tile: 64 (the packed weight tile), or 128 = two packed tiles side by side (fp16 pairs only, round 6):
// a wave then owns 64 x 64 = 4 x 4 MFMA tiles - 16 fragment reads per 48 MFMAs instead of 12 per 24 (the 64-wide kernel is
// LDS-read bound: 56 % LDS active at 28 % MFMA busy) - and the activation tile is staged and split for half as many workgroups.
// TALL (128-wide tiles only): a wave owns all 128 patches x 32 features (8 x 2 MFMA tiles) instead of 64 x 64: every weight
// fragment is then loaded from global memory by ONE wave (the two 64 x 64 tiles of a column load the same 8 KB per k-step: the
// vector-memory path of a CU ran at ~62 of its 64 B/clk), at the price of twice the activation fragment reads from LDS
// (NET-B, 2048 patches, same box: fc launches 1.05 -> 0.93 ms, pass 3.70 -> 3.57 ms; a 256-wide workgroup tile with 128 x 64 wave
// tiles - half the loads and reads per MFMA again, but one wave per SIMD - scored like the 64 x 64 tiles: 3.69 ms).  Same sums in
// the same order per output element: bit-identical to the 64 x 64 tiles (ALQ_FC_SQUARE=1 = that arm).
template <bool F16, int BN, bool TALL = false>
__global__ __launch_bounds__(256, 2) void fcgemm_kernel(const FcGemmArgs a) {
    static_assert(BN == FC_BN || (F16 && BN == 2 * FC_BN), "128-wide tiles for the fp16-pair form only");
    static_assert(!TALL || BN == 2 * FC_BN, "tall wave tiles: 128-wide workgroup tiles");
    // WD: the weight fragments come straight from global memory (the packed layout IS the MFMA lane layout: 256 contiguous bytes per
    // 16 lanes), one k-step ahead in registers, instead of through LDS: the 64-wide kernel's LDS was 56 % active - 32 KB of writes
    // at ~80 B/clk and 64 KB of fragment reads per k-step - of which the weights were half
    constexpr bool WD = BN == 2 * FC_BN;
    constexpr int NP = F16 ? 2 : 3;                     // pieces per operand
    constexpr int NC = BN / FC_BN;                      // packed 64-feature weight tiles per workgroup
    constexpr int NI = TALL ? 2 : BN / 32;              // MFMA column tiles per wave
    constexpr int MI = TALL ? 8 : 4;                    // MFMA row tiles per wave
    constexpr int XB = NP * 4 * FC_BM * 16, WB = WD ? 0 : NP * 4 * BN * 16, WB64 = NP * 4 * FC_BN * 16;
    extern __shared__ __attribute__((aligned(16))) char fc_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lq = lane >> 4;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int m0 = mt * FC_BM, n0 = nt * BN;
    const int nks = a.K / FC_BK;
    auto Xbuf = [&](int buf) { return fc_lds + buf * (XB + WB); };
    auto Wbuf = [&](int buf) { return fc_lds + buf * (XB + WB) + XB; };

    // staging roles: activations: 2 tasks per thread, task = (row, k-group): 8 floats; weights: 3 x 16 B per thread
    const int xr0 = tid >> 2, xq = tid & 3;          // rows xr0 and xr0 + 64, k-group xq
    const char *wsrc = reinterpret_cast<const char *>(a.Bp) + (size_t)(nt * NC) * nks * WB64 + tid * 16;
    f32x4 xa[2][2];
    i32x4 wr[NC][NP];
    // WD: this lane's fragment of MFMA column tile ni = feature row wn + 16 ni + lrow, k-group lq, of packed tile (row >> 6)
    const int wm = TALL ? 0 : (wave & 1) * 64, wn = TALL ? wave * 32 : (wave >> 1) * (BN / 2);
    i32x4 Wn[NP][NI];
    const char *wfrag[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int rowf = wn + ni * 16 + lrow;
        wfrag[ni] = reinterpret_cast<const char *>(a.Bp) + (size_t)(nt * NC + (rowf >> 6)) * nks * WB64 + (lq * FC_BN + (rowf & 63)) * 16;
    }
    float rsc[2] = {a.scale, a.scale};          // F16: scale of this thread's two staging rows
    if constexpr (F16) {
        if (a.row_amax) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = m0 + xr0 + t * 64;
                rsc[t] = __builtin_ldexpf(1.f, fc_row_exp(row < a.M ? a.row_amax[row] : 0u));
            }
        }
    }
    auto fetch = [&](int ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = m0 + xr0 + t * 64;
            const float *src = a.A + (size_t)row * a.lda + ks * FC_BK + xq * 8;
            if (row < a.M) {
                xa[t][0] = *reinterpret_cast<const f32x4 *>(src);
                xa[t][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            } else {
                xa[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                xa[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (WD) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Wn[p][ni] = *reinterpret_cast<const i32x4 *>(wfrag[ni] + (size_t)ks * WB64 + p * 4096);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const char *ws = wsrc + ((size_t)c * nks + ks) * WB64;
#pragma unroll
                for (int j = 0; j < NP; ++j) wr[c][j] = *reinterpret_cast<const i32x4 *>(ws + j * 4096);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[8] = {xa[t][0].x, xa[t][0].y, xa[t][0].z, xa[t][0].w, xa[t][1].x, xa[t][1].y, xa[t][1].z, xa[t][1].w};
            i32x4 pc[NP];
            if constexpr (F16) {
                int hh[4], ll[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = v[2 * j] * rsc[t], x1 = v[2 * j + 1] * rsc[t];
                    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
                    const f16x2 l = __builtin_convertvector(f32x2{x0 - (float)h.x, x1 - (float)h.y}, f16x2);
                    hh[j] = __builtin_bit_cast(int, h);
                    ll[j] = __builtin_bit_cast(int, l);
                }
                pc[0] = i32x4{hh[0], hh[1], hh[2], hh[3]};
                pc[1] = i32x4{ll[0], ll[1], ll[2], ll[3]};
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    pc[p].x = (int)fc_split2(v[0], v[1]); pc[p].y = (int)fc_split2(v[2], v[3]);
                    pc[p].z = (int)fc_split2(v[4], v[5]); pc[p].w = (int)fc_split2(v[6], v[7]);
                }
                pc[NP - 1].x = (int)fc_pack2(v[0], v[1]); pc[NP - 1].y = (int)fc_pack2(v[2], v[3]);
                pc[NP - 1].z = (int)fc_pack2(v[4], v[5]); pc[NP - 1].w = (int)fc_pack2(v[6], v[7]);
            }
            const int row = xr0 + t * 64;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<i32x4 *>(Xbuf(buf) + ((p * 4 + xq) * FC_BM + row) * 16) = pc[p];
        }
        // LDS weights: [piece][k-group q][BN feature rows] x 16 bytes; thread = (q = tid >> 6, row = tid & 63) of its 64-feature tile
        if constexpr (!WD) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    *reinterpret_cast<i32x4 *>(Wbuf(buf) + ((j * 4 + (tid >> 6)) * BN + c * FC_BN + (tid & 63)) * 16) = wr[c][j];
        }
    };

    // wave tile: 64 patches x BN / 2 features = 4 x NI MFMA tiles
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + wn + ni * 16 + lq * 4;
            if (!F16 && a.bias) acc[mi][ni] = *reinterpret_cast<const f32x4 *>(a.bias + n);      // (F16 launches carry no bias: the host checks)
        }

    fetch(0);
    stash(0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        i32x4 Wf[NP][NI], Xf[NP][MI];
        if constexpr (WD) {      // the fragments fetched one k-step ago (fetch(0) in front of the loop)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) Wf[p][ni] = Wn[p][ni];
        }
        if (ks + 1 < nks) fetch(ks + 1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                if constexpr (!WD) Wf[p][ni] = *reinterpret_cast<const i32x4 *>(Wbuf(buf) + ((p * 4 + lq) * BN + wn + ni * 16 + lrow) * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                Xf[p][mi] = *reinterpret_cast<const i32x4 *>(Xbuf(buf) + ((p * 4 + lq) * FC_BM + wm + mi * 16 + lrow) * 16);
        }
        // the piece products, smallest first
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f32x4 c = acc[mi][ni];
                if constexpr (F16) {
                    auto H = [](const i32x4 &v) { return __builtin_bit_cast(f16x8, v); };
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(Wf[1][ni]), H(Xf[0][mi]), c, 0, 0, 0);      // (l, h) (h, l) (h, h)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(Wf[0][ni]), H(Xf[1][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(H(Wf[0][ni]), H(Xf[0][mi]), c, 0, 0, 0);
                } else {
                    auto B = [](const i32x4 &v) { return __builtin_bit_cast(bf16x8, v); };
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[1][ni]), B(Xf[1][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[NP - 1][ni]), B(Xf[0][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[0][ni]), B(Xf[NP - 1][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[1][ni]), B(Xf[0][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[0][ni]), B(Xf[1][mi]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B(Wf[0][ni]), B(Xf[0][mi]), c, 0, 0, 0);
                }
                acc[mi][ni] = c;
            }
        if (ks + 1 < nks) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm + mi * 16 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            f32x4 v = acc[mi][ni];
            if constexpr (F16) {
                const float iv = a.row_amax ? __builtin_ldexpf(1.f, -(fc_row_exp(a.row_amax[m]) + a.e_w)) : a.inv;
                v.x *= iv; v.y *= iv; v.z *= iv; v.w *= iv;
                if (a.bias) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + n0 + wn + ni * 16 + lq * 4);
                    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                }
            }
            if (a.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<f32x4 *>(a.C + (size_t)m * a.ldc + n0 + wn + ni * 16 + lq * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
int fcgemm_build_plan(int K, int N, FcGemmPlan *plan) {
    plan->ok = false;
    if (K % FC_BK != 0 || N % FC_BN != 0 || K < 256 || N < 256) return ALQ_OK;
    plan->K = K; plan->N = N;
    plan->form = getenv("ALQ_FC_BN64") ? 2 : (getenv("ALQ_FC_SQUARE") ? 1 : 0);
    plan->ok = true;
    return ALQ_OK;
}

static unsigned short fc_bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float fc_bf16_to_f(unsigned short hb) {
    const unsigned u = (unsigned)hb << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// packed: [feature tile][k-step][piece][k-group q][feature row][8 bf16 along k]
void fcgemm_pack_weights(FcGemmPlan *plan, const std::vector<float> &Bmat /* [K][N] */) {
    const int K = plan->K, N = plan->N;
    const int nks = K / FC_BK, ntl = N / FC_BN;
    plan->h_W.assign((size_t)K * N * 3, 0);
    for (int nt = 0; nt < ntl; ++nt)
        for (int ks = 0; ks < nks; ++ks) {
            unsigned short *blk = &plan->h_W[((size_t)nt * nks + ks) * (FC_WBYTES / 2)];
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < FC_BN; ++r)
                    for (int j = 0; j < 8; ++j) {
                        float w = Bmat[(size_t)(ks * FC_BK + q * 8 + j) * N + nt * FC_BN + r];
                        for (int p = 0; p < 3; ++p) {
                            const unsigned short hb = fc_bf16_rne(w);
                            w -= fc_bf16_to_f(hb);
                            blk[((size_t)(p * 4 + q) * FC_BN + r) * 8 + j] = hb;
                        }
                    }
        }
}

// fp16 pairs of w 2^e_w at their true scale: [feature tile][k-step][piece (h, l)][k-group q][feature row][8 fp16 along k]
void fcgemm_pack_weights_f16(FcGemmPlan *plan, const std::vector<float> &Bmat /* [K][N] */) {
    const int K = plan->K, N = plan->N;
    const int nks = K / FC_BK, ntl = N / FC_BN;
    float amax = 0.f;
    for (float w : Bmat) amax = std::max(amax, std::fabs(w));
    int ex = 0;
    if (amax > 0.f) (void)std::frexp(amax, &ex);
    plan->w_exp = 14 - ex;
    const size_t blk_elems = (size_t)2 * 4 * FC_BN * 8;
    plan->h_W16.assign((size_t)K * N * 2, 0);
    for (int nt = 0; nt < ntl; ++nt)
        for (int ks = 0; ks < nks; ++ks) {
            unsigned short *blk = &plan->h_W16[((size_t)nt * nks + ks) * blk_elems];
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < FC_BN; ++r)
                    for (int j = 0; j < 8; ++j) {
                        const float ws = std::ldexp(Bmat[(size_t)(ks * FC_BK + q * 8 + j) * N + nt * FC_BN + r], plan->w_exp);
                        const _Float16 h = (_Float16)ws;
                        const _Float16 l = (_Float16)(ws - (float)h);
                        unsigned short hb, lb;
                        std::memcpy(&hb, &h, 2);
                        std::memcpy(&lb, &l, 2);
                        blk[((size_t)(0 * 4 + q) * FC_BN + r) * 8 + j] = hb;
                        blk[((size_t)(1 * 4 + q) * FC_BN + r) * 8 + j] = lb;
                    }
        }
}

// the fp16-pair launch: 128-feature tiles where the layer's width allows (ALQ_FC_BN64=1: the 64-wide kernel, A/B)
static int fc16_go(alq_ctx *ctx, const FcGemmPlan &plan, const FcGemmArgs &a, int M) {
    const bool bn64 = plan.form == 2, tall = plan.form == 0;
    ProfScope ps16(ctx, PROF_IGEMM_F16, 2.0 * M * (double)plan.K * plan.N);
    if (plan.N % (2 * FC_BN) == 0 && !bn64 && tall) {
        const size_t lds = 2 * (2 * 4 * FC_BM * 16);
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fcgemm_kernel<true, 2 * FC_BN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((fcgemm_kernel<true, 2 * FC_BN, true>), dim3(plan.N / (2 * FC_BN), (M + FC_BM - 1) / FC_BM), dim3(256), lds, ctx->stream, a);
    } else if (plan.N % (2 * FC_BN) == 0 && !bn64) {
        const size_t lds = 2 * (2 * 4 * FC_BM * 16);          // activations only: the weight fragments come from global memory
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fcgemm_kernel<true, 2 * FC_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((fcgemm_kernel<true, 2 * FC_BN>), dim3(plan.N / (2 * FC_BN), (M + FC_BM - 1) / FC_BM), dim3(256), lds, ctx->stream, a);
    } else {
        const size_t lds = 2 * (2 * 4 * FC_BM * 16 + 2 * 4 * FC_BN * 16);
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fcgemm_kernel<true, FC_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((fcgemm_kernel<true, FC_BN>), dim3(plan.N / FC_BN, (M + FC_BM - 1) / FC_BM), dim3(256), lds, ctx->stream, a);
    }
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

int fcgemm_launch(alq_ctx *ctx, const FcGemmPlan &plan, const View &in, const View &out, const float *bias, int relu,
                  int M, int prof_cls, float in_bound, unsigned *row_amax) {
    ALQ_REQUIRE(plan.d_W && in.C == plan.K && out.C == plan.N && in.c0 == 0 && out.c0 == 0 && in.cs % 4 == 0 && out.cs % 4 == 0 &&
                    in.vox() == 1 && out.vox() == 1,
                ALQ_EINVAL, "fcgemm: views do not match the plan");
    FcGemmArgs a;
    a.A = in.p; a.Bp = plan.d_W; a.C = out.p; a.bias = bias;
    a.M = M; a.N = plan.N; a.K = plan.K; a.lda = in.cs; a.ldc = out.cs; a.relu = relu;
    a.scale = 1.f; a.inv = 1.f; a.row_amax = nullptr; a.e_w = plan.w_exp;
    const dim3 grid(plan.N / FC_BN, (M + FC_BM - 1) / FC_BM);
    if (row_amax && plan.d_W16) {      // fp16 pairs under the MEASURED per-patch maximum of the input (a forward launch; bias + ReLU in the epilogue)
        hipLaunchKernelGGL(fc_rowmax_kernel, dim3((M + 3) / 4), dim3(256), 0, ctx->stream, in.p, M, plan.K, in.cs, row_amax);
        a.Bp = plan.d_W16;
        a.row_amax = row_amax;
        ALQ_TRY(fc16_go(ctx, plan, a, M));
        return ALQ_OK;
    }
    if (in_bound > 0.f && plan.d_W16 && !bias) {      // fp16 pairs under a static bound on the input (a backward launch of a Fisher pass)
        int ex = 0;
        (void)std::frexp(in_bound, &ex);
        const int e_in = 14 - ex;
        a.Bp = plan.d_W16;
        a.scale = std::ldexp(1.f, e_in); a.inv = std::ldexp(1.f, -(e_in + plan.w_exp));
        ALQ_TRY(fc16_go(ctx, plan, a, M));
        return ALQ_OK;
    }
    const size_t lds = 2 * (FC_XBYTES + FC_WBYTES);
    ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fcgemm_kernel<false, FC_BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(ctx, prof_cls, 2.0 * M * (double)plan.K * plan.N);
    hipLaunchKernelGGL((fcgemm_kernel<false, FC_BN>), grid, dim3(256), lds, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

}  // namespace alq
