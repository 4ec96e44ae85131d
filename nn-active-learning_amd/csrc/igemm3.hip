// Implicit-GEMM engine on the bf16 matrix cores with fp32-equivalent accuracy (v3).
//
// gfx950's fp32 MFMA runs at the fp32 VECTOR rate and, as measured on this path, does not overlap
// with vector work; the bf16 MFMA is 16x faster per clock and runs beside the VALU.  Every fp32
// operand is split exactly into three bf16 pieces  x = hi + mid + lo  (each the round-to-nearest bf16
// of the remainder: 3 x 8 = 24 significand bits), and a product is assembled from the six piece
// products whose weight is >= 2^-16:  hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid  (bf16 x bf16 is
// exact in fp32, the MFMA accumulates in fp32, the dropped terms are <= 2^-24 relative: fp32 rounding
// level).  One v_mfma_f32_16x16x32_bf16 contracts 4 taps x 8 channels; six of them replace eight
// v_mfma_f32_16x16x4_f32 at a sixteenth of the per-MAC cost.
//
// Everything else is the v2 structure (igemm2.hip): persistent workgroups, host-built tile / slot
// tables, register prefetch of the next halo block through buffer loads, transposed output tile
// with 16-byte stores, deferred epilogue with fused ReLU / mask / channel sums.  Activations are
// split once per staged element when the prefetched registers are written to LDS
// (v_cvt_pk_bf16_f32); weights are split on the host.  LDS row of a halo voxel: [hi8 | mid8 | lo8].
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
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// In-kernel phase stamps (diagnostic build only, -DALQ_STAMPS; see igemm2.hip)
#ifdef ALQ_STAMPS
#define STAMP3(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define PHASE3_END(idx)                        \
    do {                                       \
        unsigned long long t_now_;             \
        STAMP3(t_now_);                        \
        ph[idx] += t_now_ - t_last;            \
        t_last = t_now_;                       \
    } while (0)
#else
#define PHASE3_END(idx) do {} while (0)
#endif

constexpr int I3_ROWB = 48;       // bytes per halo voxel row in LDS: 3 pieces x 8 channels x 2 B
constexpr int I3_MAXSLOT = 8;
constexpr int I3_MAXS = 7;        // k-steps of 4 taps (<= 28 taps)
constexpr int I3_WRES_MAX = 44 * 1024;

// x -> (hi, rem): hi = bf16(x) packed pairwise, rem = x - hi (exact)
__device__ inline unsigned split2(float &a, float &b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    a -= __builtin_bit_cast(float, hb << 16);
    b -= __builtin_bit_cast(float, hb & 0xffff0000u);
    return hb;
}

// Tile descriptors are read with an explicit scalar load: the compiler will not scalarise a load that
// follows stores through a possibly aliasing pointer, and the vector load it emits instead is followed by
// s_waitcnt vmcnt(0), which drains every store of the epilogue once per tile.
__device__ inline i32x4 sload4(const int *p) {
    i32x4 v;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
}

__device__ inline int sload1(const unsigned *p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
}

// F16 (round 6): the fp16-pair form for launches whose input has a host-known bound (the backward launches of a Fisher pass):
// x 2^e = h + l 2^-11 (the lo pieces scaled up: the bound is loose by orders of magnitude and true-scale lo pieces of typical
// values would be fp16 subnormals), weights likewise; three products instead of six - (h, h) in one accumulator, (l, h) + (h, l)
// at 2^11 in a second one, joined in the epilogue.  LDS rows keep their 48-byte pitch ([h8 | l8 | -]).
template <int NTW, bool WRES, bool SUMS, bool F16>
__global__ __launch_bounds__(256, 2) void igemm3_kernel(const Igemm2Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds3[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 15;
    const int lq = lane >> 4;

    const int S = (a.ntaps + 3) >> 2;
    const int halo = a.HZ * a.HY * a.HX;
    const int nhv = a.PT * halo;
    constexpr int NP = F16 ? 2 : 3;                        // operand pieces
    const int Wchunk = S * NP * NTW * 1024;                // bytes of one 8-channel weight chunk
    const int Wbytes = WRES ? a.nchunks * Wchunk : Wchunk;
    char *Wl = lds3;
    char *Al = lds3 + Wbytes;
    constexpr int WREGS = WRES ? 1 : (I3_MAXS * NP * NTW + 3) / 4;    // 16-byte registers per thread per chunk
    constexpr int NSLOT = I3_MAXSLOT;
    const char *Wg = reinterpret_cast<const char *>(a.W);

    if constexpr (WRES) {       // eight loads in flight per thread (a rolled copy pays one L2 latency per 4 KB)
        for (int i0 = tid * 16; i0 < Wbytes; i0 += 8 * 4096) {
            i32x4 w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 4096;
                w8[u] = (i < Wbytes) ? *reinterpret_cast<const i32x4 *>(Wg + i) : i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 4096;
                if (i < Wbytes) *reinterpret_cast<i32x4 *>(Wl + i) = w8[u];
            }
        }
    }

    // ---- per-thread A staging slots (halo voxel, half) from the host-built slot table ------------
    const int nslots = nhv * 2;
    const int nit = (nslots + 255) >> 8;
    const int half4 = (tid & 1) * 4;
    int s_rel[NSLOT];
    unsigned s_mlo[NSLOT], s_mhi[NSLOT];
#pragma unroll
    for (int it = 0; it < NSLOT; ++it) {
        const int slot = tid + it * 256;
        int rel = 0;
        unsigned mlo = 0, mhi = 0;
        if (slot < nslots) {
            const int4 sd = *reinterpret_cast<const int4 *>(a.sdesc + (slot >> 1) * 4);
            rel = sd.x * a.in_cs + a.in_c0 + half4;
            mlo = (unsigned)sd.y;
            mhi = (unsigned)sd.z;
        }
        s_rel[it] = rel;
        s_mlo[it] = mlo;
        s_mhi[it] = mhi;
    }

    // ---- per-lane LDS byte offsets: row blocks and the taps of this lane group --------------------
    const int TV = a.TZ * a.TY * a.TX;
    int vbase[4];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
        int v = wave * 64 + ms * 16 + lrow;
        if (v >= a.rows) v = 0;
        const int pt = v / TV;
        int r = v - pt * TV;
        const int x = r % a.TX; r /= a.TX;
        const int y = r % a.TY;
        const int z = r / a.TY;
        vbase[ms] = (((pt * a.HZ + z * a.sm) * a.HY + y * a.sm) * a.HX + x * a.sm) * I3_ROWB;
    }
    // k-step s contracts taps 4s..4s+3; lane group lq supplies tap 4s+lq (a tap past the end reads any
    // valid row: its weights are zero)
    int toff[I3_MAXS];
#pragma unroll
    for (int s = 0; s < I3_MAXS; ++s) {
        int t = 4 * s + lq;
        if (t >= a.ntaps) t = 0;
        const int ix = t % a.tnx, iy = (t / a.tnx) % a.tny, iz = t / (a.tnx * a.tny);
        toff[s] = (a.t0 + iz * a.tsz + iy * a.tsy + ix * a.tsx) * 4;      // t0/ts* are in floats of a 48-byte row
    }

    // ---- epilogue geometry (transposed tile: lane = (channel group lq, voxel lrow)) ----------------
    int eoff[4], evox[4];
    bool erow_ok = true;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
        const int v = wave * 64 + ms * 16 + lrow;
        const int vv = v < a.rows ? v : 0;
        const int pt = vv / TV;
        int q = vv - pt * TV;
        const int x = q % a.TX; q /= a.TX;
        const int y = q % a.TY;
        const int z = q / a.TY;
        evox[ms] = ((pt * a.OD + z * a.so) * a.OH + y * a.so) * a.OW + x * a.so;
        eoff[ms] = evox[ms] * a.out_cs + a.out_c0 + lq * 4;
        erow_ok = erow_ok && v < a.rows;
    }
    f32x4 bias4[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = nt * 16 + lq * 4;
        if (a.bias && c < a.Co) bias4[nt] = *reinterpret_cast<const f32x4 *>(a.bias + c);
    }

    // ---- tile walk (see igemm2.hip) ------------------------------------------------------------------
    const int pgroups = (a.N + a.PT - 1) / a.PT;
    const int gp = gridDim.x / a.tpg, gl = gridDim.x % a.tpg;
    int fpg = blockIdx.x / a.tpg, fl = blockIdx.x % a.tpg;
    auto advance_cursor = [&]() {
        fl += gl;
        const int c = fl >= a.tpg;
        fl -= c ? a.tpg : 0;
        fpg += gp + c;
    };
    int f_out = 0, c_out = 0, p_out = 0;
    int f_cls = 0, c_cls = 0, p_cls = 0;
    // F16: the scales of the tile being staged / contracted (c_) and of the tile whose epilogue is pending (p_)
    int f_t = 0;
    float c_sc = a.f16_sc, c_bsc = a.f16_bias_sc, p_inv = a.f16_inv, c_inv = a.f16_inv;
    int f_l = 0, c_l = 0, p_l = 0, f_g = 0, c_g = 0, p_g = 0;

    int goff[NSLOT];
    auto locate = [&]() {
        const i32x4 td = sload4(a.tdesc + fl * 8);
        const int in_org = (td.x + fpg * a.in_pstride) * a.in_cs;
        f_out = td.y + fpg * a.out_pstride;
        f_cls = td.z;
        f_l = fl; f_g = fpg;
        if constexpr (F16) {
            if (a.f16_bound) {      // bound = f 2^ex, f in [.5, 1): x 2^(14 - ex) < 2^14; the exponent field E = ex + 126 (clamped: no inf / denormal scale)
                int E = (sload1(a.f16_bound + (fpg < a.N ? fpg : 0)) >> 23) & 255;
                E = E < 40 ? 40 : (E > 200 ? 200 : E);
                f_t = 140 - E;
            }
        }
        const int cls = f_cls & 63;
        const unsigned bit = 1u << (cls & 31);
        const bool hi = cls >= 32;
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            const unsigned m = hi ? s_mhi[it] : s_mlo[it];           // slots past the end carry empty masks
            goff[it] = (m & bit) ? (in_org + s_rel[it]) * 4 : 0x7fffff00;
        }
        if ((fpg + 1) * a.PT > a.N) {                                  // last, partly filled patch group (rare)
#pragma unroll 1
            for (int it = 0; it < nit; ++it) {
                const int pt = a.sdesc[((tid + it * 256) >> 1) * 4 + 3] >> 24;
                if (fpg * a.PT + pt >= a.N) {
#pragma unroll
                    for (int j = 0; j < NSLOT; ++j)
                        if (j == it) goff[j] = 0x7fffff00;
                }
            }
        }
    };

    f32x4 R[NSLOT];
    i32x4 Wr[WREGS];
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    // ONE prefetch site, unconditional loads (a slot without work points past the buffer and reads zeros
    // without touching memory): a second site or a branch around the loads makes the compiler merge the
    // two register sets with copies, and the copies wait for the loads on the spot.
    auto fetch = [&](int chunk) {
        const int soff = chunk * 32;
#pragma unroll
        for (int it = 0; it < NSLOT; ++it)
            R[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[it], soff, 0));
        if constexpr (!WRES) {
            const char *src = Wg + (long long)chunk * Wchunk;
#pragma unroll
            for (int w = 0; w < WREGS; ++w) {
                const int i = (tid + w * 256) * 16;
                Wr[w] = (i < Wchunk) ? *reinterpret_cast<const i32x4 *>(src + i) : i32x4{0, 0, 0, 0};
            }
        }
    };
    // slot (hv, half) -> LDS byte offset of its 4 channels inside the hi piece; mid / lo at +16 / +32
    const int sbase = (tid >> 1) * I3_ROWB + (tid & 1) * 8;
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            if (it < nit && tid + it * 256 < nslots) {
                float v0 = R[it].x, v1 = R[it].y, v2 = R[it].z, v3 = R[it].w;
                char *dst = Al + sbase + it * (128 * I3_ROWB);
                if constexpr (F16) {
                    const float sc = c_sc, sc11 = c_sc * 2048.f;
                    const f16x2 h01 = __builtin_convertvector(f32x2{v0 * sc, v1 * sc}, f16x2), h23 = __builtin_convertvector(f32x2{v2 * sc, v3 * sc}, f16x2);
                    // (x 2^e - h) 2^11: exact in fp32 (the remainder of a rounding to 11 bits)
                    const f16x2 l01 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h01.x, -2048.f, v0 * sc11), __builtin_fmaf((float)h01.y, -2048.f, v1 * sc11)}, f16x2);
                    const f16x2 l23 = __builtin_convertvector(f32x2{__builtin_fmaf((float)h23.x, -2048.f, v2 * sc11), __builtin_fmaf((float)h23.y, -2048.f, v3 * sc11)}, f16x2);
                    *reinterpret_cast<uint2 *>(dst) = uint2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                    *reinterpret_cast<uint2 *>(dst + 16) = uint2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                } else {
                    uint2 hi, mid, lo;
                    hi.x = split2(v0, v1);  hi.y = split2(v2, v3);
                    mid.x = split2(v0, v1); mid.y = split2(v2, v3);
                    lo.x = split2(v0, v1);  lo.y = split2(v2, v3);
                    *reinterpret_cast<uint2 *>(dst) = hi;
                    *reinterpret_cast<uint2 *>(dst + 16) = mid;
                    *reinterpret_cast<uint2 *>(dst + 32) = lo;
                }
            }
        }
        if constexpr (!WRES) {
#pragma unroll
            for (int w = 0; w < WREGS; ++w) {
                const int i = (tid + w * 256) * 16;
                if (i < Wchunk) *reinterpret_cast<i32x4 *>(Wl + i) = Wr[w];
            }
        }
    };

    // ---------------- deferred epilogue (identical to igemm2.hip) ---------------------------------------
    f32x4 acc[4][NTW];
    f32x4 accl[F16 ? 4 : 1][F16 ? NTW : 1];      // F16: the (l, h) + (h, l) products at 2^11
    bool have_pend = false;
    char *outb = reinterpret_cast<char *>(a.out);
    const char *maskb = reinterpret_cast<const char *>(a.mask);
    auto flush = [&]() {
        const int obase_e = p_out * a.out_cs;
        const bool p_full = (p_cls >> 8) & 1 && (p_g + 1) * a.PT <= a.N;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            bool live = erow_ok;
            if (!p_full) {
                const i32x4 t0 = sload4(a.tdesc + p_l * 8), t1 = sload4(a.tdesc + p_l * 8 + 4);
                const i32x4 tq = i32x4{t0.w, t1.x, t1.y, 0};        // (z0, y0, x0) of the tile
                const int v = wave * 64 + ms * 16 + lrow;
                const int vv = v < a.rows ? v : 0;
                const int pt = vv / TV;
                int q = vv - pt * TV;
                const int x = q % a.TX; q /= a.TX;
                const int y = q % a.TY;
                const int z = q / a.TY;
                live = v < a.rows && p_g * a.PT + pt < a.N && tq.x + z < a.MD && tq.y + y < a.MH && tq.z + x < a.MW;
            }
            f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int c = nt * 16 + lq * 4;
                f32x4 val = acc[ms][nt];
                if constexpr (F16) {
                    const f32x4 cl = accl[ms][nt];
                    val.x = __builtin_fmaf(cl.x, 0x1p-11f, val.x) * p_inv; val.y = __builtin_fmaf(cl.y, 0x1p-11f, val.y) * p_inv;
                    val.z = __builtin_fmaf(cl.z, 0x1p-11f, val.z) * p_inv; val.w = __builtin_fmaf(cl.w, 0x1p-11f, val.w) * p_inv;
                }
                const bool on = live && c < a.Co;
                if (on) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(outb + (unsigned)((obase_e + eoff[ms] + nt * 16) * 4));
                    if (a.accumulate) val += *dst;
                    if (a.relu) {
                        val.x = __builtin_amdgcn_fmed3f(val.x, 0.f, __builtin_inff());
                        val.y = __builtin_amdgcn_fmed3f(val.y, 0.f, __builtin_inff());
                        val.z = __builtin_amdgcn_fmed3f(val.z, 0.f, __builtin_inff());
                        val.w = __builtin_amdgcn_fmed3f(val.w, 0.f, __builtin_inff());
                    }
                    if (a.mask && c >= a.mask_from) {
                        const int mo = (p_out + evox[ms]) * a.mask_cs + a.mask_c0 + (c - a.mask_from);
                        const f32x4 mk = *reinterpret_cast<const f32x4 *>(maskb + (unsigned)(mo * 4));
                        val.x = mk.x > 0.f ? val.x : 0.f; val.y = mk.y > 0.f ? val.y : 0.f;
                        val.z = mk.z > 0.f ? val.z : 0.f; val.w = mk.w > 0.f ? val.w : 0.f;
                    }
                    *dst = val;
                }
                if constexpr (SUMS) {
                    if (!on) val = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float sel = (lrow == 0) ? (c < a.split ? 1.f : 0.f) : ((lrow == 1) ? (c < a.split ? 0.f : 1.f) : 0.f);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, (val.x + val.y) + (val.z + val.w), sacc, 0, 0, 0);
                }
            }
            if constexpr (SUMS) {
                if (live && lq == 0) {
                    if (a.osumA) a.osumA[p_out + evox[ms]] = sacc.x;
                    if (a.osumB) a.osumB[p_out + evox[ms]] = sacc.y;
                }
            }
        }
    };

#ifdef ALQ_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    STAMP3(t_last);
#endif
    auto park = [&]() {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) goff[it] = 0x7fffff00;
    };
    if (fpg < pgroups) {          // flat loop over (tile, chunk) steps
        locate();
        fetch(0);
        PHASE3_END(0);
        int chunk = 0;
        bool first = true;
        for (;;) {
            if (chunk == 0) {
                c_out = f_out; c_cls = f_cls; c_l = f_l; c_g = f_g;
                if constexpr (F16) {
                    if (a.f16_bound) {
                        c_sc = __builtin_bit_cast(float, (unsigned)(127 + f_t) << 23);
                        c_bsc = __builtin_bit_cast(float, (unsigned)(127 + f_t + a.f16_wexp) << 23);
                        c_inv = __builtin_bit_cast(float, (unsigned)(127 - f_t - a.f16_wexp) << 23);
                    }
                }
            }
            if (!first) __syncthreads();
            first = false;
            PHASE3_END(1);
            stash();
            PHASE3_END(2);
            __syncthreads();
            PHASE3_END(3);
            if (chunk == 0 && have_pend) {
                flush();
                have_pend = false;
            }
            PHASE3_END(4);
            int nchunk = chunk + 1;
            const bool tile_done = nchunk == a.nchunks;
            bool next_valid = true;
            if (tile_done) {
                nchunk = 0;
                advance_cursor();
                next_valid = fpg < pgroups;
                if (next_valid) locate(); else park();
            }
            fetch(nchunk);
            PHASE3_END(5);
            if (chunk == 0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        acc[ms][nt] = bias4[nt];
                        if constexpr (F16) {
                            acc[ms][nt] = f32x4{bias4[nt].x * c_bsc, bias4[nt].y * c_bsc, bias4[nt].z * c_bsc, bias4[nt].w * c_bsc};
                            accl[ms][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
            }
            const char *Wc = Wl + (WRES ? chunk * Wchunk : 0) + lane * 16;
            for (int rep = 0; rep <= a.dbg_repeat; ++rep)
#pragma unroll
            for (int s = 0; s < I3_MAXS; ++s) {
                if (s < S) {
                    if constexpr (F16) {
                        f16x8 Wf[2][NTW];
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int nt = 0; nt < NTW; ++nt)
                                Wf[p][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(Wc + ((s * 2 + p) * NTW + nt) * 1024));
#pragma unroll
                        for (int mh = 0; mh < 4; mh += 2) {
                            f16x8 X[2][2];
#pragma unroll
                            for (int m2 = 0; m2 < 2; ++m2) {
                                const char *row = Al + vbase[mh + m2] + toff[s];
#pragma unroll
                                for (int p = 0; p < 2; ++p)
                                    X[p][m2] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4 *>(row + 16 * p));
                            }
#pragma unroll
                            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                                for (int m2 = 0; m2 < 2; ++m2) {
                                    f32x4 cl = accl[mh + m2][nt];
                                    cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][nt], X[0][m2], cl, 0, 0, 0);
                                    cl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[1][m2], cl, 0, 0, 0);
                                    accl[mh + m2][nt] = cl;
                                    acc[mh + m2][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][nt], X[0][m2], acc[mh + m2][nt], 0, 0, 0);
                                }
                        }
                    } else {
                    bf16x8 Wf[3][NTW];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            Wf[p][nt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(Wc + ((s * 3 + p) * NTW + nt) * 1024));
                    // two row blocks at a time (register budget); six piece products, smallest weights first
#pragma unroll
                    for (int mh = 0; mh < 4; mh += 2) {
                        bf16x8 X[3][2];
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            const char *row = Al + vbase[mh + m2] + toff[s];
#pragma unroll
                            for (int p = 0; p < 3; ++p)
                                X[p][m2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4 *>(row + 16 * p));
                        }
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                            for (int m2 = 0; m2 < 2; ++m2) {
                                f32x4 c = acc[mh + m2][nt];
#ifdef ALQ_V3_EIGHT
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][nt], X[1][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[2][m2], c, 0, 0, 0);
#endif
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[1][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][nt], X[0][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[2][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][nt], X[0][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[1][m2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][nt], X[0][m2], c, 0, 0, 0);
                                acc[mh + m2][nt] = c;
                            }
                    }
                    }
                }
            }
            PHASE3_END(6);
            if (tile_done) {
                have_pend = true;
                p_out = c_out; p_cls = c_cls; p_l = c_l; p_g = c_g; p_inv = c_inv;
                if (!next_valid) break;
            }
            chunk = nchunk;
        }
    }
    if (have_pend) flush();
#ifdef ALQ_STAMPS
    PHASE3_END(7);
    if (a.dbg && tid == 0)
        for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 8 + i] = ph[i];
#endif
}

// ------------------------------------------------------------------------------------------------------
static unsigned short bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int igemm3_build_plan(const Igemm2Plan &p2, Igemm3Plan *p3) {
    p3->ok = false;
    if (!p2.ok) return ALQ_OK;
    const Igemm2Args &a = p2.a;
    const int S = (a.ntaps + 3) / 4;
    if (S > I3_MAXS) return ALQ_OK;
    const int nhv = a.PT * a.HZ * a.HY * a.HX;
    if (nhv * 2 > 256 * I3_MAXSLOT) return ALQ_OK;
    const size_t wchunk = (size_t)S * 3 * p2.NTW * 1024;
    p3->wres = wchunk * a.nchunks <= (size_t)I3_WRES_MAX;
    const size_t lds = (p3->wres ? wchunk * a.nchunks : wchunk) + (size_t)nhv * I3_ROWB;
    if (lds > 156 * 1024) return ALQ_OK;
    p3->lds_bytes = lds;
    p3->NTW = p2.NTW;
    p3->wgs_per_cu = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / (lds + 512))));
    p3->ok = true;
    return ALQ_OK;
}

// packed layout: [chunk][s][piece][nt][lane][8] bf16, lane = (g = lane>>4 -> tap 4s+g, co = lane&15), j -> channel
void igemm3_pack_weights(const Igemm2Plan &p2, Igemm3Plan *p3, const std::vector<float> &Bmat) {
    const Igemm2Args &a = p2.a;
    const int NTW = p3->NTW, Ci = a.Ci, Co = a.Co, S = (a.ntaps + 3) / 4;
    p3->h_W.assign((size_t)a.nchunks * S * 3 * NTW * 64 * 8, 0);
    for (int ch = 0; ch < a.nchunks; ++ch)
        for (int s = 0; s < S; ++s)
            for (int nt = 0; nt < NTW; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int tp = 4 * s + (lane >> 4);
                        const int co = nt * 16 + (lane & 15);
                        float w = 0.f;
                        if (tp < a.ntaps && co < Co) w = Bmat[((size_t)tp * Ci + ch * 8 + j) * Co + co];
                        for (int p = 0; p < 3; ++p) {
                            const unsigned short h = bf16_rne(w);
                            w -= bf16_to_f(h);
                            p3->h_W[((((((size_t)ch * S + s) * 3 + p) * NTW + nt) * 64 + lane) * 8) + j] = h;
                        }
                    }
}

// fp16 bits of x (round to nearest even, host side; |x| < 65504 by the callers' scaling)
static unsigned short f16_rne(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

// The fp16-pair twin of the packed weights: W 2^e = h + l 2^-11 with max |W| 2^e in [2^13, 2^14); same layout, two pieces
void igemm3_pack_weights_f16(const Igemm2Plan &p2, Igemm3Plan *p3, const std::vector<float> &Bmat) {
    const Igemm2Args &a = p2.a;
    const int NTW = p3->NTW, Ci = a.Ci, Co = a.Co, S = (a.ntaps + 3) / 4;
    float amax = 0.f;
    for (float w : Bmat) amax = std::max(amax, std::fabs(w));
    int ex = 0;
    (void)std::frexp(amax > 0.f ? amax : 1.f, &ex);       // amax = f 2^ex, f in [.5, 1)
    p3->w16_exp = 14 - ex;
    const float sw = std::ldexp(1.f, p3->w16_exp);
    p3->h_W16.assign((size_t)a.nchunks * S * 2 * NTW * 64 * 8, 0);
    for (int ch = 0; ch < a.nchunks; ++ch)
        for (int s = 0; s < S; ++s)
            for (int nt = 0; nt < NTW; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int tp = 4 * s + (lane >> 4);
                        const int co = nt * 16 + (lane & 15);
                        float w = 0.f;
                        if (tp < a.ntaps && co < Co && ch * 8 + j < Ci) w = Bmat[((size_t)tp * Ci + ch * 8 + j) * Co + co] * sw;
                        const _Float16 h = (_Float16)w;
                        const float l = (w - (float)h) * 2048.f;
                        const size_t base = ((((size_t)ch * S + s) * 2) * NTW + nt) * 64 * 8 + (size_t)lane * 8 + j;
                        p3->h_W16[base] = f16_rne((float)h);
                        p3->h_W16[base + (size_t)NTW * 64 * 8] = f16_rne(l);
                    }
}

template <int NTW, bool WRES, bool SUMS, bool F16 = false>
static int launch3_s(alq_ctx *ctx, const Igemm3Plan &plan, const Igemm2Args &a, unsigned grid) {
    auto kfn = igemm3_kernel<NTW, WRES, SUMS, F16>;
    if (plan.lds_bytes > 64 * 1024)
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)plan.lds_bytes));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), plan.lds_bytes, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

template <int NTW, bool WRES>
static int launch3_t(alq_ctx *ctx, const Igemm3Plan &plan, const Igemm2Args &a, unsigned grid) {
    if (a.f16_sc > 0.f)
        return (a.osumA || a.osumB) ? launch3_s<NTW, WRES, true, true>(ctx, plan, a, grid)
                                    : launch3_s<NTW, WRES, false, true>(ctx, plan, a, grid);
    return (a.osumA || a.osumB) ? launch3_s<NTW, WRES, true>(ctx, plan, a, grid)
                                : launch3_s<NTW, WRES, false>(ctx, plan, a, grid);
}

int igemm3_launch(alq_ctx *ctx, const Igemm2Plan &p2, const Igemm3Plan &plan, const View &in, const View &out,
                  const float *bias, int relu, int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse) {
    Igemm2Args a = p2.a;
    ALQ_REQUIRE(in.C == a.Ci && in.D == a.ID && in.H == a.IH && in.W == a.IW, ALQ_EINVAL, "igemm3: input view mismatch");
    ALQ_REQUIRE(out.C == a.Co && out.D == a.OD && out.H == a.OH && out.W == a.OW, ALQ_EINVAL, "igemm3: output view mismatch");
    ALQ_REQUIRE(plan.d_W != nullptr, ALQ_EINVAL, "igemm3: weights not set");
    ALQ_REQUIRE((long long)N * in.vox() * in.cs < (1LL << 29) && (long long)N * out.vox() * out.cs < (1LL << 29),
                ALQ_EUNSUPPORTED, "igemm3: tensor exceeds the 32-bit byte-offset range (lower the batch)");
    ALQ_REQUIRE(in.cs % 4 == 0 && in.c0 % 4 == 0 && out.cs % 4 == 0 && out.c0 % 4 == 0, ALQ_EUNSUPPORTED,
                "igemm3: channel slice not 16-byte aligned");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.W = reinterpret_cast<const float *>(plan.d_W);
    a.bias = bias; a.relu = relu; a.accumulate = accumulate; a.N = N;
    a.in_bytes = (int)((long long)N * in.vox() * in.cs * 4);
    a.dbg = nullptr;
    if (g_igemm2_dbg) {   // diagnostic: stamp only the v3 launch whose ordinal (since the buffer was set) is ALQ_STAMP_ONLY
        static int want = -2;
        if (want == -2) { const char *e = getenv("ALQ_STAMP_ONLY"); want = e ? atoi(e) : -1; }
        static int ordinal = 0;
        static unsigned long long *last = nullptr;
        if (last != g_igemm2_dbg) { last = g_igemm2_dbg; ordinal = 0; }
        if (want < 0 || ordinal == want) a.dbg = g_igemm2_dbg;
        ++ordinal;
    }
    a.dbg_repeat = g_dbg_knobs[0];
    a.dbg_flags = g_dbg_knobs[1];
    a.split = 1 << 30;
    if (fuse) {
        ALQ_REQUIRE(fuse->split % 4 == 0 && fuse->mask_cs % 4 == 0 && fuse->mask_c0 % 4 == 0 && fuse->mask_from % 4 == 0,
                    ALQ_EUNSUPPORTED, "igemm3: fused epilogue needs 4-channel aligned slices");
        a.mask = fuse->mask; a.mask_cs = fuse->mask_cs; a.mask_c0 = fuse->mask_c0; a.mask_from = fuse->mask_from;
        a.osumA = fuse->osumA; a.osumB = fuse->osumB;
        a.split = fuse->split > 0 ? fuse->split : (1 << 30);
    }
    // the fp16-pair instantiation: a host-known bound on the input (Igemm2Fuse::in_bound - the cotangent bound of a Fisher pass),
    // the twin weights packed, nothing accumulated into (those launches keep the exact split)
    a.f16_sc = a.f16_inv = a.f16_bias_sc = 0.f;
    a.f16_bound = nullptr; a.f16_wexp = 0;
    // ... or one scale per patch (forward launches): Igemm2Fuse::in_amax = a bound on max |input| of every patch; tiles of one patch
    const bool f16p = fuse && fuse->in_amax && !fuse->in_amax2 && plan.d_W16 && a.PT == 1 && !accumulate && !g_no_f16x2;
    const bool f16 = f16p || (fuse && fuse->in_bound > 0.f && plan.d_W16 && !accumulate && !g_no_f16x2 && !g_dbg_knobs[2]);
    if (f16p) {
        a.W = reinterpret_cast<const float *>(plan.d_W16);
        a.f16_bound = fuse->in_amax;
        a.f16_wexp = plan.w16_exp;
        a.f16_sc = a.f16_inv = a.f16_bias_sc = 1.f;       // (selects the instantiation; the kernel derives the scales per tile)
    } else if (f16) {
        int ex = 0;
        (void)std::frexp(fuse->in_bound, &ex);            // bound = f 2^ex, f in [.5, 1): |x| 2^(14 - ex) < 2^14
        a.W = reinterpret_cast<const float *>(plan.d_W16);
        a.f16_sc = std::ldexp(1.f, 14 - ex);
        a.f16_inv = std::ldexp(1.f, -(14 - ex) - plan.w16_exp);
        a.f16_bias_sc = std::ldexp(1.f, (14 - ex) + plan.w16_exp);
    }
    const int pgroups = (N + a.PT - 1) / a.PT;
    const long long total = (long long)pgroups * a.tpg;
    const unsigned grid = (unsigned)std::min<long long>(total, 256LL * plan.wgs_per_cu);
    ProfScope ps(ctx, f16 ? (int)PROF_IGEMM_F16 : prof_cls, p2.flops_per_patch * N);
#define ALQ_L3(NT) \
    case NT: return plan.wres ? launch3_t<NT, true>(ctx, plan, a, grid) : launch3_t<NT, false>(ctx, plan, a, grid)
    switch (plan.NTW) { ALQ_L3(1); ALQ_L3(2); ALQ_L3(3); }
#undef ALQ_L3
    set_error("igemm3: NTW=%d unsupported", plan.NTW);
    return ALQ_EUNSUPPORTED;
}

}  // namespace alq
