// Implicit-GEMM engine, pipelined form (v2) for the layers that dominate the scoring path:
// input channels a multiple of 8, one N block.
//
// Differences from igemm.hip (which remains the general fallback):
//   * persistent workgroups walk tiles (tile = 256 GEMM rows) with stride gridDim.x; when the whole
//     layer's weights fit 16 KiB per 16-column tile they are copied to LDS ONCE per workgroup
//     (WRES), otherwise one 8-channel chunk of weights is staged per step;
//   * the halo'd input block (and weight chunk) of the NEXT step is fetched from HBM/L2 into
//     registers while the MFMAs of the current step run and is written to LDS after them, so the
//     global latency is hidden behind matrix work inside one workgroup; LDS stays small
//     (one A buffer) so that 2-3 workgroups share a CU and cover each other's barrier phases;
//   * A fragments are read with ds_read_b64: K is permuted inside a chunk (k-step s, lane group q
//     -> channel 2q+s) so that one 8-byte read feeds two MFMAs; rows are padded to 12 floats, which
//     makes the 32-lane read groups hit 64 distinct banks; B fragments are packed on the host as
//     [chunk][tap][tile][lane][2] and read linearly with ds_read_b64; fragment reads of tap t+1
//     are issued before the MFMAs of tap t;
//   * fused epilogue: bias / ReLU (forward) or ReLU-grad mask by the destination layer's activation
//     (backward), accumulate, and the channel sums of the produced rows (the `asum` / `dsum` fields
//     of the factored shrink_gradient), optionally split at a concat boundary.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// In-kernel phase stamps (diagnostic build only: -DALQ_STAMPS; wave 0 adds up shader-clock ticks per
// phase and lane 0 writes them to a.dbg[blockIdx.x*8 + phase] at the end; never in the product build).
#ifdef ALQ_STAMPS
#define STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define PHASE_END(idx)                         \
    do {                                       \
        unsigned long long t_now_;             \
        STAMP(t_now_);                         \
        ph[idx] += t_now_ - t_last;            \
        t_last = t_now_;                       \
    } while (0)
#else
#define PHASE_END(idx) do {} while (0)
#endif

constexpr int I2_CBP = 12;        // floats per halo voxel row in LDS (8 channels + 4 pad)
constexpr int I2_MAXSLOT = 8;     // float4 A staging slots per thread (nhv*2 <= 256*I2_MAXSLOT)
constexpr int I2_WRES_MAX = 28 * 1024;   // resident weights: at most this many bytes per 16-col tile (tools/gpu_ab.py)

// GEO: 1 / 2 -> 3x3x3 taps over a halo block with HY = 6, HX = 18 (the 4x4x16 tile every 16^3 / 32^3 layer
// uses), walked in ascending (forward conv) / descending (backward-data) order: every LDS fragment
// address is then `register + immediate`, so the MFMA loop carries no address arithmetic at all;
// 0 -> box extents and strides read at run time.
template <int NTW, bool WRES, int GEO, bool SUMS>
__global__ __launch_bounds__(256, 2) void igemm2_kernel(const Igemm2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 15;
    const int lq = lane >> 4;

    const int halo = a.HZ * a.HY * a.HX;
    const int nhv = a.PT * halo;
    const int Wchunk = a.ntaps * NTW * 128;              // floats of one 8-channel weight chunk
    const int Wfloats = WRES ? a.nchunks * Wchunk : Wchunk;
    float *Wl = lds;
    float *Al = lds + Wfloats;
    constexpr int WREGS = WRES ? 1 : (27 * NTW * 32 + 255) / 256;   // f32x4 per thread for one weight chunk
    constexpr int NSLOT = GEO > 0 ? 6 : I2_MAXSLOT;                 // GEO: 6*6*18 halo voxels * 2 halves / 256

    if constexpr (WRES) {
        for (int i = tid * 4; i < Wfloats; i += 1024)
            *reinterpret_cast<f32x4 *>(Wl + i) = *reinterpret_cast<const f32x4 *>(a.W + i);
    }

    // ---- per-thread A staging slots: (halo voxel, half) from the host-built slot table ------------
    // s_rel = element offset relative to the tile's halo origin voxel; s_mlo/s_mhi = 64-bit mask "is
    // this halo voxel inside the tensor" indexed by the tile's border class (4 classes per dimension:
    // first / middle / last / first-and-last)
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

    // ---- per-lane LDS row base of the 4 row blocks of this wave ---------------------------
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
        vbase[ms] = (((pt * a.HZ + z * a.sm) * a.HY + y * a.sm) * a.HX + x * a.sm) * I2_CBP + 2 * lq;
    }

    // ---- epilogue geometry (tile-independent).  The MFMAs are issued with the WEIGHT fragment as the
    // A operand and the activation fragment as B, i.e. they compute the transposed tile: D column =
    // lane&15 = GEMM row (voxel), D row = (lane>>4)*4 + reg = output channel.  A lane therefore ends
    // with 4 consecutive channels of ONE voxel per accumulator: one 16-byte store, no shuffles.
    int eoff[4], evox[4];       // element offset of (voxel, channel 4*lq) / voxel offset, relative to the tile's first output voxel
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
    f32x4 bias4[NTW];     // the accumulators START at the bias (channels 4*lq..4*lq+3 of each 16-channel tile)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = nt * 16 + lq * 4;
        if (a.bias && c < a.Co) bias4[nt] = *reinterpret_cast<const f32x4 *>(a.bias + c);   // Co % 4 == 0 (plan)
    }

    // ---- tile walk without divisions: cursor (patch group, tile inside the group) advanced by gridDim.x;
    // everything tile-dependent comes from the host-built descriptor of the tile inside its group
    const int pgroups = (a.N + a.PT - 1) / a.PT;
    const int gp = gridDim.x / a.tpg, gl = gridDim.x % a.tpg;
    int fpg = blockIdx.x / a.tpg, fl = blockIdx.x % a.tpg;    // cursor of the tile being FETCHED
    auto advance_cursor = [&]() {
        fl += gl;
        const int c = fl >= a.tpg;
        fl -= c ? a.tpg : 0;
        fpg += gp + c;
    };

    // uniform per-tile quantities, carried through the pipeline: fetch -> compute -> deferred epilogue
    int f_out = 0, c_out = 0, p_out = 0;          // output voxel index of the tile's first row
    int f_cls = 0, c_cls = 0, p_cls = 0;          // border class | full << 8
    int f_l = 0, c_l = 0, p_l = 0, f_g = 0, c_g = 0, p_g = 0;

    int goff[NSLOT];     // byte offset of every slot of the fetch tile (past num_records where the halo leaves the tensor)
    auto locate = [&]() {
        const int *td = a.tdesc + fl * 8;
        const int in_org = (td[0] + fpg * a.in_pstride) * a.in_cs;
        f_out = td[1] + fpg * a.out_pstride;
        f_cls = td[2];
        f_l = fl; f_g = fpg;
        const int cls = f_cls & 63;
        const bool partial = (fpg + 1) * a.PT > a.N;      // last patch group with fewer than PT patches
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            int g = 0x7fffff00;
            if (it < nit) {
                const unsigned m = cls < 32 ? s_mlo[it] : s_mhi[it];
                bool ok = (m >> (cls & 31)) & 1u;
                if (partial) {     // uniform, rare: re-check the patch index of the slot
                    const int pt = a.sdesc[((tid + it * 256) >> 1) * 4 + 3] >> 24;
                    ok = ok && fpg * a.PT + pt < a.N;
                }
                g = ok ? (in_org + s_rel[it]) * 4 : 0x7fffff00;     // byte offset, or past the end -> reads 0
            }
            goff[it] = g;
        }
    };

    f32x4 R[NSLOT];
    f32x4 Wr[WREGS];
    // buffer descriptor over the whole input tensor: an offset past num_records reads as zero, which
    // is exactly the zero padding of the halo (no branch, no zero-fill instructions)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
    auto fetch = [&](int chunk) {
        const int soff = (a.dbg_flags & 2) ? 0x7ffffff0 : chunk * 32;     // uniform byte offset of the chunk
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            if (it < nit)
                R[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[it], soff, 0));
        }
        if constexpr (!WRES) {
            const float *src = a.W + (long long)chunk * Wchunk;
#pragma unroll
            for (int w = 0; w < WREGS; ++w) {
                const int i = (tid + w * 256) * 4;
                Wr[w] = (i < Wchunk) ? *reinterpret_cast<const f32x4 *>(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    const int sbase = (tid >> 1) * I2_CBP + half4;      // slot -> LDS float offset: + it*128*CBP
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < NSLOT; ++it) {
            if (it < nit && tid + it * 256 < nslots)
                *reinterpret_cast<f32x4 *>(Al + sbase + it * (128 * I2_CBP)) = R[it];
        }
        if constexpr (!WRES) {
#pragma unroll
            for (int w = 0; w < WREGS; ++w) {
                const int i = (tid + w * 256) * 4;
                if (i < Wchunk) *reinterpret_cast<f32x4 *>(Wl + i) = Wr[w];
            }
        }
    };

    // ---------------- deferred epilogue: ReLU / mask, stores, channel sums -------------------------
    // The results stay in the accumulators; their stores are issued one step later, right after the
    // barrier and BEFORE the next prefetch (loads and stores retire in issue order on one counter, so
    // a prefetch wait must never sit behind fresh stores), and before the accumulators are re-armed.
    f32x4 acc[4][NTW];
    bool have_pend = false;
    char *outb = reinterpret_cast<char *>(a.out);
    const char *maskb = reinterpret_cast<const char *>(a.mask);
    auto flush = [&]() {
        const int obase_e = p_out * a.out_cs;         // uniform
        const bool p_full = (p_cls >> 8) & 1 && (p_g + 1) * a.PT <= a.N;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            bool live = erow_ok;
            if (!p_full) {      // border / partial tiles only: per-row bounds
                const int *td = a.tdesc + p_l * 8;
                const int v = wave * 64 + ms * 16 + lrow;
                const int vv = v < a.rows ? v : 0;
                const int pt = vv / TV;
                int q = vv - pt * TV;
                const int x = q % a.TX; q /= a.TX;
                const int y = q % a.TY;
                const int z = q / a.TY;
                live = v < a.rows && p_g * a.PT + pt < a.N && td[3] + z < a.MD && td[4] + y < a.MH && td[5] + x < a.MW;
            }
            // channel sums of the finished rows: one more tiny contraction on the matrix pipe instead
            // of cross-lane shuffles.  The accumulator register r of lane (lq, voxel) is exactly the B
            // operand element B[k = lq][j = voxel] of a 16x16x4 MFMA, so D = S * val_r summed over r
            // with S[0][k] = [channel group < split], S[1][k] = [>= split] leaves sumA / sumB of the
            // voxel in rows 0 / 1 of D, i.e. in registers x / y of the lanes lq == 0.
            f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int c = nt * 16 + lq * 4;
                f32x4 val = acc[ms][nt];
                const bool on = live && c < a.Co;
                if (on) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(outb + (unsigned)((obase_e + eoff[ms] + nt * 16) * 4));
                    if (a.accumulate) val += *dst;      // a skip destination wrote this slice first
                    if (a.relu) {       // one v_med3_f32 per element (clamp to [0, +inf])
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
                    if (!(a.dbg_flags & 1)) *dst = val;
                }
                if constexpr (SUMS) {
                    if (!on) val = f32x4{0.f, 0.f, 0.f, 0.f};
                    // selector S[i = lrow][k = lq] for this 16-channel tile
                    const float sel = (lrow == 0) ? (c < a.split ? 1.f : 0.f) : ((lrow == 1) ? (c < a.split ? 0.f : 1.f) : 0.f);
                    if (!(a.dbg_flags & 4)) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, val.x, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, val.y, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, val.z, sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, val.w, sacc, 0, 0, 0);
                    }
                }
            }
            if constexpr (SUMS) {
                if (live && lq == 0 && !(a.dbg_flags & 8)) {
                    if (a.osumA) a.osumA[p_out + evox[ms]] = sacc.x;
                    if (a.osumB) a.osumB[p_out + evox[ms]] = sacc.y;
                }
            }
        }
    };

#ifdef ALQ_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last;
    STAMP(t_last);
#endif

    bool more = fpg < pgroups;        // does the fetch cursor point at a real tile?
    if (more) {
        locate();
        fetch(0);
    }
    PHASE_END(0);
    bool first = true;
    while (more) {
        // the fetched tile becomes the compute tile
        c_out = f_out; c_cls = f_cls; c_l = f_l; c_g = f_g;
        bool next_more = false;
        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            if (!first) __syncthreads();      // every wave has finished reading the previous step's LDS
            first = false;
            PHASE_END(1);                     // barrier A
            stash();
            PHASE_END(2);                     // wait for the prefetch + LDS writes
            __syncthreads();
            PHASE_END(3);                     // barrier B
            if (chunk == 0 && have_pend) {
                flush();
                have_pend = false;
            }
            PHASE_END(4);                     // deferred epilogue
            // prefetch the next step while this one computes
            if (chunk + 1 < a.nchunks) {
                fetch(chunk + 1);
            } else {
                advance_cursor();
                next_more = fpg < pgroups;
                if (next_more) {
                    locate();
                    fetch(0);
                }
            }
            PHASE_END(5);                     // locate + prefetch issue
            if (chunk == 0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[ms][nt] = bias4[nt];
            }
            const float *Wc = Wl + (WRES ? chunk * Wchunk : 0) + lane * 2;
            // Software-pipelined fragment reads with two register sets: the LDS reads of tap t+1 are
            // issued BEFORE the MFMAs of tap t (sched_barrier pins that order; left alone, hipcc
            // rotates the loop back into read -> wait -> MFMA).  The taps of every contraction form a
            // box: offset = t0 + iz*tsz + iy*tsy + ix*tsx.
            f32x2 a0[4], b0[NTW], a1[4], b1[NTW];
            auto rd = [&](f32x2 *av, f32x2 *bv, int toff, int tap) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) av[ms] = *reinterpret_cast<const f32x2 *>(Al + vbase[ms] + toff);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[nt] = *reinterpret_cast<const f32x2 *>(Wc + (tap * NTW + nt) * 128);
            };
            auto mm = [&](const f32x2 *av, const f32x2 *bv) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int ms = 0; ms < 4; ++ms)
                        acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt].x, av[ms].x, acc[ms][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int ms = 0; ms < 4; ++ms)
                        acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt].y, av[ms].y, acc[ms][nt], 0, 0, 0);
            };
            for (int rep = 0; rep <= a.dbg_repeat; ++rep)
            if constexpr (GEO > 0) {
                constexpr int BOX = 27;
                constexpr int SX = I2_CBP, SY = 18 * I2_CBP, SZ = 6 * 18 * I2_CBP;
                auto off = [&](int t) {
                    const int o = (t / 9) * SZ + ((t / 3) % 3) * SY + (t % 3) * SX;
                    return GEO == 1 ? o : (2 * SZ + 2 * SY + 2 * SX) - o;
                };
                rd(a0, b0, off(0), 0);
#pragma unroll
                for (int tap = 0; tap < BOX; tap += 2) {
                    if (tap + 1 < BOX) rd(a1, b1, off(tap + 1), tap + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap + 1 < BOX) {
                        if (tap + 2 < BOX) rd(a0, b0, off(tap + 2), tap + 2);
                        __builtin_amdgcn_sched_barrier(0);
                        mm(a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                int iz = 0, iy = 0, ix = 0;
                auto advance = [&]() {
                    if (++ix == a.tnx) { ix = 0; if (++iy == a.tny) { iy = 0; ++iz; } }
                    return a.t0 + iz * a.tsz + iy * a.tsy + ix * a.tsx;
                };
                rd(a0, b0, a.t0, 0);
                for (int tap = 0; tap < a.ntaps; tap += 2) {
                    const bool has1 = tap + 1 < a.ntaps;
                    {
                        const int toff = has1 ? advance() : a.t0;
                        rd(a1, b1, toff, has1 ? tap + 1 : 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (has1) {
                        const bool has2 = tap + 2 < a.ntaps;
                        const int toff = has2 ? advance() : a.t0;
                        rd(a0, b0, toff, has2 ? tap + 2 : 0);
                        __builtin_amdgcn_sched_barrier(0);
                        mm(a1, b1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            PHASE_END(6);
        }
        // results wait in registers: their stores are issued one step later, right after the barrier
        // and BEFORE the next prefetch, so that the `vmcnt` wait that retires a prefetch never has
        // to wait for fresh stores (loads and stores retire in issue order on one counter)
        have_pend = true;
        p_out = c_out; p_cls = c_cls; p_l = c_l; p_g = c_g;
        more = next_more;
    }
    if (have_pend) flush();
#ifdef ALQ_STAMPS
    PHASE_END(7);
    if (a.dbg && tid == 0)
        for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 8 + i] = ph[i];
#endif
}

// ------------------------------------------------------------------------------------------
int igemm2_build_plan(const IgemmPlan &p1, Igemm2Plan *p2) {
    // eligibility: channel-chunked path of v1 with CB = 8, one N block, at most 27 taps
    p2->ok = false;
    const IgemmArgs &a1 = p1.a;
    if (p1.smallc || p1.CB != 8 || a1.NB != 1 || p1.NTW > 3 || a1.ntaps > 27 || a1.Co % 4) return ALQ_OK;
    const int halo = a1.HZ * a1.HY * a1.HX;
    const int nhv = a1.PT * halo;
    if (nhv * 2 > 256 * I2_MAXSLOT) return ALQ_OK;
    if (a1.HZ > 255 || a1.HY > 255 || a1.HX > 255 || a1.PT > 127) return ALQ_OK;
    const size_t wchunk = (size_t)a1.ntaps * p1.NTW * 128 * 4;
    size_t wres_max = I2_WRES_MAX;
    if (const char *e = getenv("ALQ_WRES_KB")) wres_max = (size_t)atoi(e) * 1024;     // tuning experiment
    const bool wres = wchunk * a1.nchunks <= wres_max * p1.NTW;
    const size_t wbytes = wres ? wchunk * a1.nchunks : wchunk;
    const size_t lds = wbytes + (size_t)nhv * I2_CBP * 4;
    if (lds > 156 * 1024) return ALQ_OK;
    Igemm2Args &a = p2->a;
    std::memset(&a, 0, sizeof(a));
    a.Ci = a1.Ci; a.ID = a1.ID; a.IH = a1.IH; a.IW = a1.IW;
    a.Co = a1.Co; a.OD = a1.OD; a.OH = a1.OH; a.OW = a1.OW;
    a.MD = a1.MD; a.MH = a1.MH; a.MW = a1.MW;
    a.sm = a1.sm; a.so = a1.so; a.ooffz = a1.ooffz; a.ooffy = a1.ooffy; a.ooffx = a1.ooffx;
    a.PT = a1.PT; a.TZ = a1.TZ; a.TY = a1.TY; a.TX = a1.TX; a.HZ = a1.HZ; a.HY = a1.HY; a.HX = a1.HX;
    a.rows = a1.rows; a.minz = a1.minz; a.miny = a1.miny; a.minx = a1.minx;
    a.ntaps = a1.ntaps; a.nchunks = a1.nchunks;
    a.tilesZ = a1.tilesZ; a.tilesY = a1.tilesY; a.tilesX = a1.tilesX;
    // the tap set must be a box walked x-fastest: offset = t0 + iz*tsz + iy*tsy + ix*tsx
    {
        std::vector<int> off(a1.ntaps);
        for (int t = 0; t < a1.ntaps; ++t) off[t] = a1.tapoff[t] / 10 * I2_CBP;   // v1 rows are CB+2 = 10 floats
        bool found = false;
        for (int nx = 1; nx <= a1.ntaps && !found; ++nx) {
            if (a1.ntaps % nx) continue;
            for (int ny = 1; nx * ny <= a1.ntaps && !found; ++ny) {
                if ((a1.ntaps / nx) % ny) continue;
                const int nz = a1.ntaps / (nx * ny);
                const int sx = nx > 1 ? off[1] - off[0] : 0;
                const int sy = ny > 1 ? off[nx] - off[0] : 0;
                const int sz = nz > 1 ? off[nx * ny] - off[0] : 0;
                bool okb = true;
                for (int t = 0; t < a1.ntaps && okb; ++t) {
                    const int ix = t % nx, iy = (t / nx) % ny, iz = t / (nx * ny);
                    okb = off[t] == off[0] + iz * sz + iy * sy + ix * sx;
                }
                if (okb) {
                    a.t0 = off[0]; a.tsx = sx; a.tsy = sy; a.tsz = sz; a.tnx = nx; a.tny = ny;
                    found = true;
                }
            }
        }
        if (!found) return ALQ_OK;   // not a box: the general kernel handles it
    }
    a.split = 1 << 30;
    // ---- host tables: tile descriptors (one patch group) and halo-slot descriptors ----------------
    {
        const int dimsI[3] = {a.ID, a.IH, a.IW}, dimsM[3] = {a.MD, a.MH, a.MW};
        const int T[3] = {a.TZ, a.TY, a.TX}, tiles[3] = {a.tilesZ, a.tilesY, a.tilesX};
        const int mins[3] = {a.minz, a.miny, a.minx}, H[3] = {a.HZ, a.HY, a.HX};
        auto cls_of = [&](int d, int t) { return (t == 0 ? 1 : 0) | (t == tiles[d] - 1 ? 2 : 0); };   // 0 mid, 1 first, 2 last, 3 both
        auto valid1 = [&](int d, int t, int h) {
            const int c = t * T[d] * a.sm + mins[d] + h;
            return c >= 0 && c < dimsI[d];
        };
        // every "middle" tile must see the same (fully valid) halo, else the class scheme does not apply
        for (int d = 0; d < 3; ++d)
            for (int t = 1; t + 1 < tiles[d]; ++t)
                for (int h = 0; h < H[d]; ++h)
                    if (!valid1(d, t, h)) return ALQ_OK;
        a.tpg = a.tilesZ * a.tilesY * a.tilesX;
        a.in_pstride = a.PT * a.ID * a.IH * a.IW;
        a.out_pstride = a.PT * a.OD * a.OH * a.OW;
        p2->h_tdesc.assign((size_t)a.tpg * 8, 0);
        for (int tz = 0; tz < a.tilesZ; ++tz)
            for (int ty = 0; ty < a.tilesY; ++ty)
                for (int tx = 0; tx < a.tilesX; ++tx) {
                    int *td = &p2->h_tdesc[(((size_t)tz * a.tilesY + ty) * a.tilesX + tx) * 8];
                    const int mz0 = tz * a.TZ, my0 = ty * a.TY, mx0 = tx * a.TX;
                    const int z0 = mz0 * a.sm + a.minz, y0 = my0 * a.sm + a.miny, x0 = mx0 * a.sm + a.minx;
                    td[0] = (z0 * a.IH + y0) * a.IW + x0;
                    td[1] = ((mz0 * a.so + a.ooffz) * a.OH + my0 * a.so + a.ooffy) * a.OW + mx0 * a.so + a.ooffx;
                    const int cls = cls_of(0, tz) | (cls_of(1, ty) << 2) | (cls_of(2, tx) << 4);
                    const bool full = a.rows == 256 && mz0 + a.TZ <= a.MD && my0 + a.TY <= a.MH && mx0 + a.TX <= a.MW;
                    td[2] = cls | (full ? 256 : 0);
                    td[3] = mz0; td[4] = my0; td[5] = mx0;
                }
        (void)dimsM;
        // representative tile index of a class in dimension d
        auto rep = [&](int d, int c) { return (c & 1) ? 0 : ((c & 2) ? tiles[d] - 1 : (tiles[d] > 2 ? 1 : 0)); };
        p2->h_sdesc.assign((size_t)nhv * 4, 0);
        for (int pt = 0; pt < a.PT; ++pt)
            for (int hz = 0; hz < a.HZ; ++hz)
                for (int hy = 0; hy < a.HY; ++hy)
                    for (int hx = 0; hx < a.HX; ++hx) {
                        const int hv = ((pt * a.HZ + hz) * a.HY + hy) * a.HX + hx;
                        int *sd = &p2->h_sdesc[(size_t)hv * 4];
                        sd[0] = ((pt * a.ID + hz) * a.IH + hy) * a.IW + hx;
                        unsigned long long m = 0;
                        for (int c = 0; c < 64; ++c) {
                            const int cz = c & 3, cy = (c >> 2) & 3, cx = (c >> 4) & 3;
                            if (valid1(0, rep(0, cz), hz) && valid1(1, rep(1, cy), hy) && valid1(2, rep(2, cx), hx))
                                m |= 1ull << c;
                        }
                        sd[1] = (int)(unsigned)(m & 0xffffffffull);
                        sd[2] = (int)(unsigned)(m >> 32);
                        sd[3] = (pt << 24) | (hz << 16) | (hy << 8) | hx;
                    }
    }
    p2->NTW = p1.NTW;
    p2->wres = wres;
    p2->lds_bytes = lds;
    int max_wgs = 2;      // measured: 2 persistent workgroups per CU beat 1, 3 and 4 (tools/gpu_ab.py)
    if (const char *e = getenv("ALQ_MAX_WGS")) max_wgs = atoi(e);                      // tuning experiment
    p2->wgs_per_cu = std::max<int>(1, std::min<int>(max_wgs, (int)((160 * 1024) / (lds + 512))));
    p2->flops_per_patch = p1.flops_per_patch;
    p2->ok = true;
    return ALQ_OK;
}

void igemm2_pack_weights(Igemm2Plan *p2, const std::vector<float> &Bmat) {
    const Igemm2Args &a = p2->a;
    const int NTW = p2->NTW, Ci = a.Ci, Co = a.Co;
    p2->h_W.assign((size_t)a.nchunks * a.ntaps * NTW * 128, 0.f);
    for (int ch = 0; ch < a.nchunks; ++ch)
        for (int tp = 0; tp < a.ntaps; ++tp)
            for (int nt = 0; nt < NTW; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 2; ++s) {
                        const int c = ch * 8 + 2 * (lane >> 4) + s;    // k-step s, lane group q -> channel 2q+s
                        const int co = nt * 16 + (lane & 15);
                        if (co < Co)
                            p2->h_W[((((size_t)ch * a.ntaps + tp) * NTW + nt) * 64 + lane) * 2 + s] =
                                Bmat[((size_t)tp * Ci + c) * Co + co];
                    }
}

unsigned long long *g_igemm2_dbg = nullptr;
int g_no_f16x2 = 0;
int g_no_xcd_order = 0;
int g_no_fixed = 0;
int g_dbg_knobs[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0 repeat, 1 flags, 2 no-bwd-fuse, 3 no-fwd-fuse   // set by alq_debug_set_stamp_buffer (diagnostic build)

template <int NTW, bool WRES, int GEO, bool SUMS>
static int launch2_s(alq_ctx *ctx, const Igemm2Plan &plan, const Igemm2Args &a, unsigned grid) {
    auto kfn = igemm2_kernel<NTW, WRES, GEO, SUMS>;
    if (plan.lds_bytes > 64 * 1024)
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)plan.lds_bytes));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), plan.lds_bytes, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

template <int NTW, bool WRES, int GEO>
static int launch2_t(alq_ctx *ctx, const Igemm2Plan &plan, const Igemm2Args &a, unsigned grid) {
    return (a.osumA || a.osumB) ? launch2_s<NTW, WRES, GEO, true>(ctx, plan, a, grid)
                                : launch2_s<NTW, WRES, GEO, false>(ctx, plan, a, grid);
}

int igemm2_launch(alq_ctx *ctx, const Igemm2Plan &plan, const View &in, const View &out, const float *bias, int relu,
                  int accumulate, int N, int prof_cls, const Igemm2Fuse *fuse) {
    Igemm2Args a = plan.a;
    ALQ_REQUIRE(in.C == a.Ci && in.D == a.ID && in.H == a.IH && in.W == a.IW, ALQ_EINVAL,
                "igemm2: input view does not match plan");
    ALQ_REQUIRE(out.C == a.Co && out.D == a.OD && out.H == a.OH && out.W == a.OW, ALQ_EINVAL,
                "igemm2: output view does not match plan");
    ALQ_REQUIRE(plan.d_W != nullptr, ALQ_EINVAL, "igemm2: weights not set");
    ALQ_REQUIRE((long long)N * in.vox() * in.cs < (1LL << 29) && (long long)N * out.vox() * out.cs < (1LL << 29),
                ALQ_EUNSUPPORTED, "igemm2: tensor exceeds the 32-bit byte-offset range (lower the batch)");
    ALQ_REQUIRE(in.cs % 4 == 0 && in.c0 % 4 == 0 && out.cs % 4 == 0 && out.c0 % 4 == 0, ALQ_EUNSUPPORTED,
                "igemm2: channel slice not 16-byte aligned");
    if (fuse) ALQ_REQUIRE(fuse->split % 4 == 0 && fuse->mask_cs % 4 == 0 && fuse->mask_c0 % 4 == 0 && fuse->mask_from % 4 == 0, ALQ_EUNSUPPORTED,
                          "igemm2: fused epilogue needs 4-channel aligned slices");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.W = plan.d_W; a.bias = bias; a.relu = relu; a.accumulate = accumulate; a.N = N;
    a.in_bytes = (int)((long long)N * in.vox() * in.cs * 4);
    a.dbg = nullptr;
    if (g_igemm2_dbg) {   // diagnostic: stamp only the launch whose ordinal (since the buffer was set) is ALQ_STAMP_ONLY
        static int want = -2;
        if (want == -2) { const char *e = getenv("ALQ_STAMP_ONLY"); want = e ? atoi(e) : -1; }
        static int ordinal = 0;
        static unsigned long long *last = nullptr;
        if (last != g_igemm2_dbg) { last = g_igemm2_dbg; ordinal = 0; }
        if (want < 0 || ordinal == want) a.dbg = g_igemm2_dbg;
        ++ordinal;
    }
    // timing experiments only (alq_debug_set / env): repeat the MFMA phase, drop stores / loads / sums
    a.dbg_repeat = g_dbg_knobs[0];
    a.dbg_flags = g_dbg_knobs[1];
    if (fuse) {
        a.mask = fuse->mask; a.mask_cs = fuse->mask_cs; a.mask_c0 = fuse->mask_c0; a.mask_from = fuse->mask_from;
        a.osumA = fuse->osumA; a.osumB = fuse->osumB;
        a.split = fuse->split > 0 ? fuse->split : (1 << 30);
    }
    const int pgroups = (N + a.PT - 1) / a.PT;
    const long long total = (long long)pgroups * a.tilesZ * a.tilesY * a.tilesX;
    const unsigned grid = (unsigned)std::min<long long>(total, 256LL * plan.wgs_per_cu);
    ProfScope ps(ctx, prof_cls, plan.flops_per_patch * N);
    int geo = 0;
    if (a.ntaps == 27 && a.tnx == 3 && a.tny == 3 && a.HY == 6 && a.HX == 18) {
        const int SX = I2_CBP, SY = 18 * I2_CBP, SZ = 6 * 18 * I2_CBP;
        if (a.t0 == 0 && a.tsx == SX && a.tsy == SY && a.tsz == SZ) geo = 1;
        if (a.t0 == 2 * (SX + SY + SZ) && a.tsx == -SX && a.tsy == -SY && a.tsz == -SZ) geo = 2;
    }
#define ALQ_L2(NT)                                                                                       \
    case NT:                                                                                             \
        if (geo == 1) return plan.wres ? launch2_t<NT, true, 1>(ctx, plan, a, grid) : launch2_t<NT, false, 1>(ctx, plan, a, grid); \
        if (geo == 2) return plan.wres ? launch2_t<NT, true, 2>(ctx, plan, a, grid) : launch2_t<NT, false, 2>(ctx, plan, a, grid); \
        return plan.wres ? launch2_t<NT, true, 0>(ctx, plan, a, grid) : launch2_t<NT, false, 0>(ctx, plan, a, grid)
    switch (plan.NTW) { ALQ_L2(1); ALQ_L2(2); ALQ_L2(3); }
#undef ALQ_L2
    set_error("igemm2: NTW=%d unsupported", plan.NTW);
    return ALQ_EUNSUPPORTED;
}

}  // namespace alq
