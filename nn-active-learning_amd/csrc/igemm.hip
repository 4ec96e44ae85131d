// Implicit-GEMM engine on v_mfma_f32_16x16x4_f32 (exact fp32, gfx950).
//
// One kernel serves conv fwd / conv bwd-data / conv_transpose fwd (per parity class) /
// conv_transpose bwd-data / fc fwd / fc bwd-data.  See alq_internal.h for the GEMM it computes.
//
// Work decomposition
//   workgroup = 256 threads = 4 waves; M tile = 256 GEMM rows = PT patches x (TZ x TY x TX)
//   points of the M grid; every wave owns 64 rows (4 MFMA row-blocks of 16) and all NTW
//   16-column blocks of this workgroup's N block.
//   K loop = channel chunks of CB input channels; per chunk the halo'd input block
//   [PT][HZ][HY][HX][CB(+2 pad)] and the weight chunk [NTW][ntaps*CB][16] are staged in LDS
//   once and re-used by all taps.
// LDS reads
//   A fragment: lane l holds A[row l&15][k l>>4]; the 16 rows are 16 consecutive x (TX >= 16),
//   row stride CB+2 floats -> banks (10*r + k) mod 32 are distinct for the 32 lanes of a
//   ds_read_b32 group (CB = 8), likewise 34*r + k for CB = 32.
//   B fragment: lane l reads weight[k0*16 + l]: linear, conflict-free.
#include "alq_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace alq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CB, int NTW, bool SMALLC>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 15;
    const int lk = lane >> 4;

    int t = blockIdx.x;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY; t /= a.tilesY;
    const int tz = t % a.tilesZ; t /= a.tilesZ;
    const int p0 = t * a.PT;
    const int nb = blockIdx.y;
    const int mz0 = tz * a.TZ, my0 = ty * a.TY, mx0 = tx * a.TX;

    const int cbp = SMALLC ? a.Ci : (CB + 2);
    const int halo = a.HZ * a.HY * a.HX;
    const int nhv = a.PT * halo;                 // halo voxels of the block
    const int A_elems = (nhv * cbp + 3) & ~3;
    const int KC = SMALLC ? ((a.K + 3) & ~3) : a.ntaps * CB;
    const int B_elems = NTW * KC * 16;
    float *Alds = lds;
    float *Blds = lds + A_elems;
    int *gofs = reinterpret_cast<int *>(Blds + B_elems);   // [nhv] global voxel index or -1
    int *kofftab = gofs + nhv;                              // SMALLC: [KC] LDS offset of (tap, ci)
    if constexpr (SMALLC) {
        for (int i = tid; i < KC; i += 256) kofftab[i] = a.koff[i];
    }

    // ---- per-tile table: global offset of every halo voxel (channel 0 of our slice) --------
    for (int hv = tid; hv < nhv; hv += 256) {
        int r = hv;
        const int hx = r % a.HX; r /= a.HX;
        const int hy = r % a.HY; r /= a.HY;
        const int hz = r % a.HZ; r /= a.HZ;
        const int pt = r;
        const int iz = mz0 * a.sm + a.minz + hz;
        const int iy = my0 * a.sm + a.miny + hy;
        const int ix = mx0 * a.sm + a.minx + hx;
        const int patch = p0 + pt;
        int g = -1;
        if (patch < a.N && iz >= 0 && iz < a.ID && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) {
            // offsets are in units of 4 floats when the channel count allows it (keeps int range)
            const long long vox = (((long long)patch * a.ID + iz) * a.IH + iy) * a.IW + ix;
            g = (int)vox;   // voxel index; multiplied by in_cs at use (fits: checked on host)
        }
        gofs[hv] = g;
    }

    // ---- per-lane LDS base of the 4 row-blocks this wave owns ------------------------------
    const int TV = a.TZ * a.TY * a.TX;
    int vbase[4];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
        int v = wave * 64 + ms * 16 + lrow;
        if (v >= a.rows) v = 0;           // under-filled tile: read a valid row, discard in the epilogue
        const int pt = v / TV;
        int r = v - pt * TV;
        const int x = r % a.TX; r /= a.TX;
        const int y = r % a.TY;
        const int z = r / a.TY;
        vbase[ms] = (((pt * a.HZ + z * a.sm) * a.HY + y * a.sm) * a.HX + x * a.sm) * cbp;
    }

    f32x4 acc[4][NTW];
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[ms][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        // ---------------- stage A --------------------------------------------------------
        if constexpr (SMALLC) {
            const int tot = nhv * a.Ci;
            for (int i = tid; i < tot; i += 256) {
                const int hv = i / a.Ci;
                const int c = i - hv * a.Ci;
                const int g = gofs[hv];
                float v = 0.f;
                if (g >= 0) v = a.in[(long long)g * a.in_cs + a.in_c0 + c];
                Alds[i] = v;
            }
        } else {
            constexpr int Q = CB / 4;
            const int tot = nhv * Q;
            const int cbase = a.in_c0 + chunk * CB;
            for (int i = tid; i < tot; i += 256) {
                const int hv = i / Q;
                const int q = i - hv * Q;
                const int g = gofs[hv];
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g >= 0)
                    v = *reinterpret_cast<const f32x4 *>(a.in + (long long)g * a.in_cs + cbase + q * 4);
                float *dst = Alds + hv * cbp + q * 4;
                *reinterpret_cast<float2 *>(dst) = float2{v.x, v.y};
                *reinterpret_cast<float2 *>(dst + 2) = float2{v.z, v.w};
            }
        }
        // ---------------- stage B (contiguous copy) --------------------------------------
        {
            const float *src = a.W + ((long long)chunk * a.NB + nb) * B_elems;
            for (int i = tid * 4; i < B_elems; i += 1024)
                *reinterpret_cast<f32x4 *>(Blds + i) = *reinterpret_cast<const f32x4 *>(src + i);
        }
        __syncthreads();

        // ---------------- MFMA ------------------------------------------------------------
        if constexpr (SMALLC) {
            const int K4 = KC >> 2;
            for (int k4 = 0; k4 < K4; ++k4) {
                const int off = kofftab[k4 * 4 + lk];
                float av[4];
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) av[ms] = Alds[vbase[ms] + off];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const float bv = Blds[(nt * KC + k4 * 4) * 16 + lane];
#pragma unroll
                    for (int ms = 0; ms < 4; ++ms)
                        acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ms], bv, acc[ms][nt], 0, 0, 0);
                }
            }
        } else {
            for (int tap = 0; tap < a.ntaps; ++tap) {
                const int toff = a.tapoff[tap] + lk;
#pragma unroll
                for (int k4 = 0; k4 < CB / 4; ++k4) {
                    float av[4];
#pragma unroll
                    for (int ms = 0; ms < 4; ++ms) av[ms] = Alds[vbase[ms] + toff + k4 * 4];
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const float bv = Blds[(nt * KC + tap * CB + k4 * 4) * 16 + lane];
#pragma unroll
                        for (int ms = 0; ms < 4; ++ms)
                            acc[ms][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ms], bv, acc[ms][nt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---------------- epilogue: C/D map col = lane&15, row = (lane>>4)*4 + reg ---------------
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = wave * 64 + ms * 16 + lk * 4 + r;
            if (v >= a.rows) continue;
            const int pt = v / TV;
            int q = v - pt * TV;
            const int x = q % a.TX; q /= a.TX;
            const int y = q % a.TY;
            const int z = q / a.TY;
            const int mz = mz0 + z, my = my0 + y, mx = mx0 + x;
            const int patch = p0 + pt;
            if (patch >= a.N || mz >= a.MD || my >= a.MH || mx >= a.MW) continue;
            const int oz = mz * a.so + a.ooffz, oy = my * a.so + a.ooffy, ox = mx * a.so + a.ooffx;
            if (oz >= a.OD || oy >= a.OH || ox >= a.OW) continue;
            float *orow = a.out + ((((long long)patch * a.OD + oz) * a.OH + oy) * a.OW + ox) * a.out_cs + a.out_c0;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int co = (nb * NTW + nt) * 16 + lrow;
                if (co < a.Co) {
                    float val = acc[ms][nt][r];
                    if (a.bias) val += a.bias[co];
                    if (a.relu) val = fmaxf(val, 0.f);
                    if (a.accumulate) val += orow[co];
                    orow[co] = val;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
static int pow2ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int igemm_build_plan(const ConvDesc &d, int max_batch, IgemmPlan *plan) {
    IgemmArgs &a = plan->a;
    std::memset(&a, 0, sizeof(a));
    const int ntaps = (int)d.tz.size();
    ALQ_REQUIRE(ntaps >= 1 && ntaps <= IG_MAXTAPS, ALQ_EUNSUPPORTED, "igemm: %d taps unsupported", ntaps);
    a.Ci = d.Ci; a.ID = d.ID; a.IH = d.IH; a.IW = d.IW;
    a.Co = d.Co; a.OD = d.OD; a.OH = d.OH; a.OW = d.OW;
    a.MD = d.MD; a.MH = d.MH; a.MW = d.MW;
    a.sm = d.sm; a.so = d.so;
    a.ooffz = d.ooff[0]; a.ooffy = d.ooff[1]; a.ooffx = d.ooff[2];
    a.ntaps = ntaps;
    int minz = 1 << 30, miny = 1 << 30, minx = 1 << 30, maxz = -(1 << 30), maxy = maxz, maxx = maxz;
    for (int i = 0; i < ntaps; ++i) {
        minz = std::min(minz, d.tz[i]); maxz = std::max(maxz, d.tz[i]);
        miny = std::min(miny, d.ty[i]); maxy = std::max(maxy, d.ty[i]);
        minx = std::min(minx, d.tx[i]); maxx = std::max(maxx, d.tx[i]);
    }
    a.minz = minz; a.miny = miny; a.minx = minx;

    plan->smallc = (d.Ci % 8 != 0);
    if (plan->smallc) {
        ALQ_REQUIRE(ntaps * d.Ci <= IG_MAXK_SMALL, ALQ_EUNSUPPORTED,
                    "igemm: K = %d taps x %d channels exceeds the single-chunk limit %d", ntaps, d.Ci,
                    IG_MAXK_SMALL);
        plan->CB = d.Ci;
    } else {
        // deep K with a single tap (fc): larger channel chunk = fewer barriers per MFMA
        plan->CB = (ntaps == 1 && d.Ci % 32 == 0) ? 32 : 8;
    }
    const int NT = (d.Co + 15) / 16;
    int NTW;
    if (NT <= 3) NTW = NT;
    else if (NT % 6 == 0 && NT <= 6) NTW = 6;
    else if (NT % 4 == 0) NTW = 4;
    else if (NT % 3 == 0) NTW = 3;
    else if (NT % 2 == 0) NTW = 2;
    else NTW = 1;
    if (plan->smallc && NTW > 2) NTW = (NT % 2 == 0) ? 2 : 1;
    plan->NTW = NTW;
    a.NB = NT / NTW;

    // ---- tile search: PT*TZ*TY*TX = 256, minimise tiles * (mfma + staging) ------------------
    const int cbp = plan->smallc ? d.Ci : plan->CB + 2;
    double best = 1e300;
    int bt[4] = {256, 1, 1, 1};
    for (int TX = 1; TX <= 256; TX <<= 1) {
        if (TX > pow2ceil(d.MW)) break;
        for (int TY = 1; TX * TY <= 256; TY <<= 1) {
            if (TY > pow2ceil(d.MH)) break;
            for (int TZ = 1; TX * TY * TZ <= 256; TZ <<= 1) {
                if (TZ > pow2ceil(d.MD)) break;
              for (int PT = 256 / (TX * TY * TZ); PT >= 1; PT >>= 1) {
                if (PT > 1 && (TX < pow2ceil(d.MW) || TY < pow2ceil(d.MH) || TZ < pow2ceil(d.MD)))
                    continue;   // several patches per tile only when a tile covers a whole patch
                const int HZ = (TZ - 1) * d.sm + (maxz - minz) + 1;
                const int HY = (TY - 1) * d.sm + (maxy - miny) + 1;
                const int HX = (TX - 1) * d.sm + (maxx - minx) + 1;
                const double halo = (double)PT * HZ * HY * HX;
                const size_t ldsb = ((size_t)halo * cbp + 4 + (size_t)NTW * ntaps * std::max(plan->CB, 4) * 16) * 4 + (size_t)halo * 4;
                if (ldsb > 150 * 1024) continue;
                const double tiles = std::ceil((double)max_batch / PT) * std::ceil((double)d.MD / TZ) *
                                     std::ceil((double)d.MH / TY) * std::ceil((double)d.MW / TX);
                double cost = tiles * (256.0 * ntaps * 8 * NTW + 6.0 * halo * 8);
                if (TX < 16 && d.MW >= 16) cost *= 1.5;    // bank conflicts on the A fragment
                if (cost < best) { best = cost; bt[0] = PT; bt[1] = TZ; bt[2] = TY; bt[3] = TX; }
              }
            }
        }
    }
    ALQ_REQUIRE(best < 1e299, ALQ_EUNSUPPORTED, "igemm: no tile fits LDS");
    a.PT = bt[0]; a.TZ = bt[1]; a.TY = bt[2]; a.TX = bt[3];
    a.rows = a.PT * a.TZ * a.TY * a.TX;   // < 256 only for maps too small to fill a tile within LDS
    a.HZ = (a.TZ - 1) * d.sm + (maxz - minz) + 1;
    a.HY = (a.TY - 1) * d.sm + (maxy - miny) + 1;
    a.HX = (a.TX - 1) * d.sm + (maxx - minx) + 1;
    a.tilesZ = (d.MD + a.TZ - 1) / a.TZ;
    a.tilesY = (d.MH + a.TY - 1) / a.TY;
    a.tilesX = (d.MW + a.TX - 1) / a.TX;

    const int halo = a.HZ * a.HY * a.HX;
    const int nhv = a.PT * halo;
    int KC;
    if (plan->smallc) {
        a.K = ntaps * d.Ci;
        KC = (a.K + 3) & ~3;
        a.nchunks = 1;
        plan->h_koff.assign(KC, 0);
        for (int tp = 0; tp < ntaps; ++tp)
            for (int c = 0; c < d.Ci; ++c)
                plan->h_koff[tp * d.Ci + c] =
                    (((d.tz[tp] - minz) * a.HY + (d.ty[tp] - miny)) * a.HX + (d.tx[tp] - minx)) * cbp + c;
    } else {
        KC = ntaps * plan->CB;
        a.nchunks = d.Ci / plan->CB;
        for (int tp = 0; tp < ntaps; ++tp)
            a.tapoff[tp] = (((d.tz[tp] - minz) * a.HY + (d.ty[tp] - miny)) * a.HX + (d.tx[tp] - minx)) * cbp;
    }
    const size_t A_elems = ((size_t)nhv * cbp + 3) & ~(size_t)3;
    plan->lds_bytes = (A_elems + (size_t)NTW * KC * 16 + nhv + (plan->smallc ? KC : 0)) * 4;
    ALQ_REQUIRE(plan->lds_bytes <= 160 * 1024, ALQ_EUNSUPPORTED, "igemm: LDS %zu too large", plan->lds_bytes);
    ALQ_REQUIRE((int64_t)max_batch * d.ID * d.IH * d.IW < (1LL << 31), ALQ_EUNSUPPORTED,
                "igemm: batch*voxels exceeds int range");
    plan->flops_per_patch = 2.0 * d.MD * d.MH * d.MW * ntaps * d.Ci * d.Co;
    return ALQ_OK;
}

void igemm_pack_weights(IgemmPlan *plan, const std::vector<float> &Bmat) {
    const IgemmArgs &a = plan->a;
    const int NTW = plan->NTW, NB = a.NB, Co = a.Co, Ci = a.Ci;
    if (plan->smallc) {
        const int KC = (a.K + 3) & ~3;
        plan->h_W.assign((size_t)NB * NTW * KC * 16, 0.f);
        for (int nb = 0; nb < NB; ++nb)
            for (int nt = 0; nt < NTW; ++nt)
                for (int kk = 0; kk < a.K; ++kk)
                    for (int c = 0; c < 16; ++c) {
                        const int co = (nb * NTW + nt) * 16 + c;
                        if (co < Co)
                            plan->h_W[(((size_t)nb * NTW + nt) * KC + kk) * 16 + c] = Bmat[(size_t)kk * Co + co];
                    }
        return;
    }
    const int CB = plan->CB, KC = a.ntaps * CB;
    plan->h_W.assign((size_t)a.nchunks * NB * NTW * KC * 16, 0.f);
    for (int ch = 0; ch < a.nchunks; ++ch)
        for (int nb = 0; nb < NB; ++nb)
            for (int nt = 0; nt < NTW; ++nt)
                for (int tp = 0; tp < a.ntaps; ++tp)
                    for (int cb = 0; cb < CB; ++cb) {
                        const size_t krow = (size_t)tp * Ci + ch * CB + cb;
                        float *dst = &plan->h_W[(((((size_t)ch * NB + nb) * NTW + nt) * KC) + tp * CB + cb) * 16];
                        for (int c = 0; c < 16; ++c) {
                            const int co = (nb * NTW + nt) * 16 + c;
                            if (co < Co) dst[c] = Bmat[krow * Co + co];
                        }
                    }
}

template <int CB, int NTW, bool SMALLC>
static int launch_t(alq_ctx *ctx, const IgemmPlan &plan, const IgemmArgs &a, dim3 grid) {
    auto kfn = igemm_kernel<CB, NTW, SMALLC>;
    if (plan.lds_bytes > 64 * 1024)
        ALQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes));
    hipLaunchKernelGGL(kfn, grid, dim3(256), plan.lds_bytes, ctx->stream, a);
    ALQ_HIP(hipGetLastError());
    return ALQ_OK;
}

int igemm_launch(alq_ctx *ctx, const IgemmPlan &plan, const View &in, const View &out,
                 const float *bias, int relu, int accumulate, int N, int prof_cls) {
    IgemmArgs a = plan.a;
    ALQ_REQUIRE(in.C == a.Ci && in.D == a.ID && in.H == a.IH && in.W == a.IW, ALQ_EINVAL,
                "igemm: input view %dx%dx%dx%d does not match plan %dx%dx%dx%d", in.D, in.H, in.W, in.C,
                a.ID, a.IH, a.IW, a.Ci);
    ALQ_REQUIRE(out.C == a.Co && out.D == a.OD && out.H == a.OH && out.W == a.OW, ALQ_EINVAL,
                "igemm: output view %dx%dx%dx%d does not match plan %dx%dx%dx%d", out.D, out.H, out.W,
                out.C, a.OD, a.OH, a.OW, a.Co);
    ALQ_REQUIRE(plan.d_W != nullptr, ALQ_EINVAL, "igemm: weights not set");
    if (!plan.smallc)
        ALQ_REQUIRE(in.cs % 4 == 0 && in.c0 % 4 == 0, ALQ_EUNSUPPORTED, "igemm: channel slice not 16-byte aligned");
    a.in = in.p; a.in_cs = in.cs; a.in_c0 = in.c0;
    a.out = out.p; a.out_cs = out.cs; a.out_c0 = out.c0;
    a.koff = plan.d_koff;
    a.W = plan.d_W; a.bias = bias; a.relu = relu; a.accumulate = accumulate; a.N = N;
    const int pgroups = (N + a.PT - 1) / a.PT;
    dim3 grid((unsigned)(pgroups * a.tilesZ * a.tilesY * a.tilesX), (unsigned)a.NB);
    ProfScope ps(ctx, prof_cls, plan.flops_per_patch * N);
#define ALQ_IG(CBv, NTWv, SC) \
    if (plan.CB == CBv && plan.NTW == NTWv && plan.smallc == SC) return launch_t<CBv, NTWv, SC>(ctx, plan, a, grid)
    if (plan.smallc) {
        if (plan.NTW == 1) return launch_t<4, 1, true>(ctx, plan, a, grid);
        if (plan.NTW == 2) return launch_t<4, 2, true>(ctx, plan, a, grid);
    } else {
        ALQ_IG(8, 1, false); ALQ_IG(8, 2, false); ALQ_IG(8, 3, false); ALQ_IG(8, 4, false); ALQ_IG(8, 6, false);
        ALQ_IG(32, 1, false); ALQ_IG(32, 2, false); ALQ_IG(32, 3, false); ALQ_IG(32, 4, false); ALQ_IG(32, 6, false);
    }
#undef ALQ_IG
    set_error("igemm: no kernel instance for CB=%d NTW=%d smallc=%d", plan.CB, plan.NTW, (int)plan.smallc);
    return ALQ_EUNSUPPORTED;
}

}  // namespace alq
