// fp32 matrix-core GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// One kernel family serves every dense contraction of the HyperPocket step:
//   encoder 1x1-conv stacks      /root/reference/model/encoder.py:14-28      (X W^T + b, ReLU)
//   encoder fc / mu / std heads  model/encoder.py:30-36
//   hypernetwork trunk + heads   model/hyper_network.py:16-43
//   target-network layers        model/target_network.py:31-38 (batched: one weight set per cloud)
//   and all their backward contractions (dX = dZ W, dW = dZ^T X), which the reference gets from
//   autograd over cuBLAS.
//
//   C[z](i,j) = epi( sum_k A[z](i,k) * B[z](k,j) ),  z = 0..batch-1
//   A(i,k) at A + z*sAz + i*sAi + k*sAk ; B(k,j) at B + z*sBz + k*sBk + j*sBj ; C row-major, ldc.
//   Either stride of an operand may be the unit one: a "K-contiguous" operand is fetched with
//   16-byte loads along k (4-byte ones when its rows are not 16-byte aligned), an "i/j-contiguous"
//   one with lane-coalesced loads along i/j.  All land
//   in the same LDS image ([row][k], 80-byte rows: conflict-free ds_read_b128 fragment reads), so
//   the MFMA loop is layout-agnostic.
//   epi: + bias[j] -> ReLU -> * (mask(i,j) > 0)    (each optional; mask = stored post-ReLU
//        activation, i.e. the ReLU backward fused into the producing GEMM).
//   Split-K (ksplit > 1): partial slabs in `ws`, combined in split order by a second kernel that
//   applies the epilogue — ordered and atomic-free, so results are run-to-run identical.
//
// Workgroup = 4 waves (8 for the 128x128 tile: 32x64 per wave, 80 VGPRs, 6 waves/SIMD — measured 2 % of the whole step
// better than 4 waves of 64x64 at 4 waves/SIMD); a wave owns TMxTN tiles of 32x32 (f32x16 accumulators).
// Tile ids are remapped so that the 8 XCDs each get a contiguous run of tiles (neighbouring
// tiles share an A or B panel in that XCD's L2).
#include "hp_common.h"
#include "hp_gemm.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// BK (k-tile depth) is a template parameter: 16 for the 8-wave 128x128, the 64x128 and the 128x32 tiles (the
// instantiations at the bottom of this file; BK=32/64 on the 128x128 tile measured no faster), 32 for the 64x64 tile
// (half the barriers per MFMA, full 128-byte lines per staged row).  LDS rows are padded by 4 floats (BK+4): 16-byte
// aligned and conflict-free for the ds_read_b128 fragment reads at both depths.
constexpr int kMaxBK = 32;

struct KParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* mask;
    const float* add;
    float* ws;
    long sAz, sBz, sCz, sBiasz, sMaskz, sAddz;
    long sAi, sAk, sBk, sBj;
    int ldc, ldmask, ldadd;
    int M, N, K;
    int ksplit, kchunk;
    int flags;
    int vecA, vecB;
    int tiles_m, tiles_n;
    float* cmax;
    int* cidx;
    int group_rows;
    float* rsum;
    long sRsumz;
    long ws_rsum_off;   // offset (floats) of the row-sum partials inside ws when ksplit > 1
    const int* dyn;     // device-side row count (see HpGemmDesc::dyn_count), or NULL
    int dyn_kind;       // 1: it bounds M, 2: it bounds K
};

__device__ __forceinline__ float4 ld4(const float* __restrict__ base, long rowoff, int k, long s_k, bool row_ok, int kend, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    const float* p = base + rowoff + (long)k * s_k;
    if (vec && k + 3 < kend) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (k < kend) v.x = p[0];
        if (k + 1 < kend) v.y = p[s_k];
        if (k + 2 < kend) v.z = p[2 * s_k];
        if (k + 3 < kend) v.w = p[3 * s_k];
    }
    return v;
}

// branch-free staging load of 4 consecutive k for one (clamped) row: a 16-byte load when the operand is K-contiguous,
// 4 lane-coalesced strided loads when it is i/j-contiguous
template <bool KC>
__device__ __forceinline__ float4 ld4_fast(const float* __restrict__ p, long s_k) {
    if (KC) return *reinterpret_cast<const float4*>(p);
    return make_float4(p[0], p[s_k], p[2 * s_k], p[3 * s_k]);
}

// MODE = am + 3*bm: the staging loader of each operand.
//   0  i/j-contiguous: lanes run along the rows, 4 lane-coalesced loads at stride s_k
//   1  K-contiguous and 16-byte loadable: lanes run along k, one 16-byte load
//   2  K-contiguous, any alignment (the per-cloud weight slices of theta, row stride 19011; d theta): lanes run
//      along k, 4 scalar loads on 4 rows — coalesced like 1 (loader 0 on such an operand touches one cache line
//      per lane and is bound by the texture-address path), tails predicated in place
// Loaders 0/1 clamp rows/cols instead of predicating them (the epilogue never stores them) on whole k-tiles —
// the k-range is a multiple of BK and K-contiguous operands are 16-byte aligned (checked on the host) — and fall
// back to the generic predicated loaders on a K tail.  The generic loaders' divergent-branch scaffolding costs
// ~30 % of the MFMA rate (tools/exp_gemm.py).
template <int BM, int BN, int WGM, int WGN, int BK, int MODE>
__global__ __launch_bounds__(WGM* WGN * 64, (BM >= 128 && BN >= 128) ? ((BM / WGM / 32) * (BN / WGN / 32) == 2 ? 6 : 4) : 1) void gemm_kernel(const KParams p) {
    constexpr int NT = WGM * WGN * 64;
    constexpr int LDK = BK + 4, KQ = BK / 4;   // KQ float4 groups per staged row
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NA = (BM * KQ + NT - 1) / NT, NB = (BN * KQ + NT - 1) / NT;
    constexpr int AM = MODE % 3, BMD = MODE / 3;
    constexpr int RSTEP = NT / BK;             // loader 2: rows between a thread's consecutive elements
    // The 128x128 tile runs at the 128-VGPR limit of 4 waves/SIMD: its loaders 0/1 carry no K-tail path (the host
    // sends problems whose k-range is not whole k-tiles to the smaller tiles), which keeps it free of scratch spills.
    constexpr bool TAILS = !(BM >= 128 && BN >= 128);
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(NT % BK == 0, "loader 2 keeps one k per thread");
    __shared__ __attribute__((aligned(16))) float As_[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs_[BN * LDK];

    // XCD-aware bijective remap of the tile id (cdna_hip_programming.md T1)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (!(p.dyn && p.dyn_kind == 1)) {   // (with a device-side row count the live tiles are the first ones: a contiguous
                                         //  run per XCD would leave most XCDs idle — keep the round-robin order there)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int z = blockIdx.y / p.ksplit, split = blockIdx.y - z * p.ksplit;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    // sizes known only on the device (the encoder backward's compacted critical rows): M or K shrinks to *dyn; the grid
    // was sized for the static bound, surplus row tiles leave at once and the split ranges re-partition the real K
    int M = p.M, K = p.K, kchunk = p.kchunk;
    if (p.dyn) {
        const int v = *p.dyn;
        if (p.dyn_kind == 1) {
            M = min(M, v);
        } else {
            K = min(K, v);
            kchunk = (((K + p.ksplit - 1) / p.ksplit + kMaxBK - 1) / kMaxBK) * kMaxBK;
            if (kchunk == 0) kchunk = kMaxBK;
        }
        if (row0 >= M) return;
    }
    const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);

    const float* A = p.A + (long)z * p.sAz;
    const float* B = p.B + (long)z * p.sBz;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int r = lane & 31, h = lane >> 5;
    constexpr bool a_kc = AM != 0, b_kc = BMD != 0;

    // staging assignment: each thread moves NA (NB) groups of 4 consecutive k of one row
    int a_row[NA], a_kq[NA], b_row[NB], b_kq[NB];
#pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * NT;
        a_row[e] = a_kc ? (idx / KQ) : (idx % BM);
        a_kq[e] = a_kc ? (idx % KQ) : (idx / BM);
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int idx = tid + e * NT;
        b_row[e] = b_kc ? (idx / KQ) : (idx % BN);
        b_kq[e] = b_kc ? (idx % KQ) : (idx / BN);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float rowsum = 0.f;
    float4 ra[NA], rb[NB];
    // per-thread base pointers (row clamped into range), advanced by BK*s_k per k-tile
    const float* pa[NA];
    const float* pb[NB];
    // loader 2: element (e, u) of a thread is (row = tid/BK + (4e+u)*RSTEP, k = tid%BK); bit 4e+u of the mask = in range
    const int k2 = tid % BK, r2 = tid / BK;
    unsigned a_ok = 0, b_ok = 0;
    const long a_ustep = (long)RSTEP * p.sAi, b_ustep = (long)RSTEP * p.sBj;
    if (AM == 2) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            pa[e] = A + (long)min(row0 + r2 + 4 * e * RSTEP, M - 1) * p.sAi + (kbeg + k2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r2 + (4 * e + u) * RSTEP;
                if (row < BM && row0 + row < M) a_ok |= 1u << (4 * e + u);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NA; ++e)
            pa[e] = A + (long)min(row0 + a_row[e], M - 1) * p.sAi + (long)(kbeg + a_kq[e] * 4) * p.sAk;
    }
    if (BMD == 2) {
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            pb[e] = B + (long)min(col0 + r2 + 4 * e * RSTEP, p.N - 1) * p.sBj + (kbeg + k2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int col = r2 + (4 * e + u) * RSTEP;
                if (col < BN && col0 + col < p.N) b_ok |= 1u << (4 * e + u);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NB; ++e)
            pb[e] = B + (long)min(col0 + b_row[e], p.N - 1) * p.sBj + (long)(kbeg + b_kq[e] * 4) * p.sBk;
    }
    // loader 2: out-of-range elements read the operand's first element (always valid) and are zeroed
    auto ld_sel = [&](const float* q, const float* safe, bool ok) -> float {
        const float v = *(ok ? q : safe);
        return ok ? v : 0.f;
    };
    auto fetch = [&](int k0) {
        const bool whole = k0 + BK <= kend;
        if (AM == 2) {
            const bool kin = k0 + k2 < kend;
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                ra[e].x = ld_sel(pa[e], A, kin && (a_ok >> (4 * e) & 1));
                ra[e].y = ld_sel(pa[e] + a_ustep, A, kin && (a_ok >> (4 * e + 1) & 1));
                ra[e].z = ld_sel(pa[e] + 2 * a_ustep, A, kin && (a_ok >> (4 * e + 2) & 1));
                ra[e].w = ld_sel(pa[e] + 3 * a_ustep, A, kin && (a_ok >> (4 * e + 3) & 1));
                pa[e] += BK;
            }
        } else if (whole || !TAILS) {
#pragma unroll
            for (int e = 0; e < NA; ++e)
                if (BM * KQ % NT == 0 || tid + e * NT < BM * KQ) {
                    ra[e] = ld4_fast<AM == 1>(pa[e], p.sAk);
                    pa[e] += (long)BK * p.sAk;
                }
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int row = row0 + a_row[e];
                const bool ok = (tid + e * NT < BM * KQ) && row < M;
                ra[e] = ld4(A, (long)row * p.sAi, k0 + a_kq[e] * 4, p.sAk, ok, kend, p.vecA);
            }
        }
        if (BMD == 2) {
            const bool kin = k0 + k2 < kend;
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                rb[e].x = ld_sel(pb[e], B, kin && (b_ok >> (4 * e) & 1));
                rb[e].y = ld_sel(pb[e] + b_ustep, B, kin && (b_ok >> (4 * e + 1) & 1));
                rb[e].z = ld_sel(pb[e] + 2 * b_ustep, B, kin && (b_ok >> (4 * e + 2) & 1));
                rb[e].w = ld_sel(pb[e] + 3 * b_ustep, B, kin && (b_ok >> (4 * e + 3) & 1));
                pb[e] += BK;
            }
        } else if (whole || !TAILS) {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                if (BN * KQ % NT == 0 || tid + e * NT < BN * KQ) {
                    rb[e] = ld4_fast<BMD == 1>(pb[e], p.sBk);
                    pb[e] += (long)BK * p.sBk;
                }
        } else {
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int col = col0 + b_row[e];
                const bool ok = (tid + e * NT < BN * KQ) && col < p.N;
                rb[e] = ld4(B, (long)col * p.sBj, k0 + b_kq[e] * 4, p.sBk, ok, kend, p.vecB);
            }
        }
    };

    auto stage = [&](int buf) {   // staged registers -> LDS image `buf`
        float* As = As_ + buf * BM * LDK;
        float* Bs = Bs_ + buf * BN * LDK;
        if (AM == 2) {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int row = r2 + 4 * e * RSTEP;
                if (row < BM) As[row * LDK + k2] = ra[e].x;
                if (row + RSTEP < BM) As[(row + RSTEP) * LDK + k2] = ra[e].y;
                if (row + 2 * RSTEP < BM) As[(row + 2 * RSTEP) * LDK + k2] = ra[e].z;
                if (row + 3 * RSTEP < BM) As[(row + 3 * RSTEP) * LDK + k2] = ra[e].w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e)
                if (BM * KQ % NT == 0 || tid + e * NT < BM * KQ)
                    *reinterpret_cast<float4*>(&As[a_row[e] * LDK + a_kq[e] * 4]) = ra[e];
        }
        if (BMD == 2) {
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int col = r2 + 4 * e * RSTEP;
                if (col < BN) Bs[col * LDK + k2] = rb[e].x;
                if (col + RSTEP < BN) Bs[(col + RSTEP) * LDK + k2] = rb[e].y;
                if (col + 2 * RSTEP < BN) Bs[(col + 2 * RSTEP) * LDK + k2] = rb[e].z;
                if (col + 3 * RSTEP < BN) Bs[(col + 3 * RSTEP) * LDK + k2] = rb[e].w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NB; ++e)
                if (BN * KQ % NT == 0 || tid + e * NT < BN * KQ)
                    *reinterpret_cast<float4*>(&Bs[b_row[e] * LDK + b_kq[e] * 4]) = rb[e];
        }
    };
    auto compute = [&](int buf) {   // one k-tile of MFMAs out of LDS image `buf`
        const float* As = As_ + buf * BM * LDK;
        const float* Bs = Bs_ + buf * BN * LDK;
        if ((p.flags & HP_GEMM_ROWSUM) && tile_n == 0 && tid < BM) {   // bias gradient: row sums of the staged A tile
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&As[tid * LDK + q * 4]);
                rowsum += (v.x + v.y) + (v.z + v.w);
            }
        }
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + r) * LDK + 8 * t + 4 * h]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + r) * LDK + 8 * t + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? a[i].x : (s == 1 ? a[i].y : (s == 2 ? a[i].z : a[i].w));
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? b[j].x : (s == 1 ? b[j].y : (s == 2 ? b[j].z : b[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // (Measured without gain on the 8-wave 128x128 tile: a double-buffered LDS image with one barrier per k-tile —
    // 98.9 vs 102.3 TFLOP/s on conv5, same binary, same device.)
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage(0);
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);  // next tile's global loads fly under this tile's MFMAs
        compute(0);
        __syncthreads();
    }

    if ((p.flags & HP_GEMM_ROWSUM) && tile_n == 0 && tid < BM && row0 + tid < M) {
        if (p.ksplit > 1) p.ws[p.ws_rsum_off + (long)blockIdx.y * p.M + row0 + tid] = rowsum;
        else p.rsum[(long)z * p.sRsumz + row0 + tid] = rowsum;
    }
    // epilogue.  C/D map of the 32x32 f32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (p.flags & HP_GEMM_COLMAX) {
        // fused max-pool over this tile's rows (model/encoder.py:45): first row attaining the max wins
        __shared__ float smax[WGM][BN];
        __shared__ int sidx[WGM][BN];
        const float* bias = (p.flags & HP_GEMM_BIAS) ? p.bias + (long)z * p.sBiasz : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = wn * WN + j * 32 + r;
            const int col = col0 + cl;
            const float bv = (bias && col < p.N) ? bias[col] : 0.f;
            float best = -__builtin_inff();
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {   // rows ascend with (i, e) for a fixed lane half
                    const int row = row0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = acc[i][j][e] + bv;
                    if (row < p.M && v > best) {
                        best = v;
                        bi = row;
                    }
                }
            const float ov = __shfl_xor(best, 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
            if (h == 0) {
                smax[wm][cl] = best;
                sidx[wm][cl] = bi;
            }
        }
        __syncthreads();
        if (tid < BN && col0 + tid < p.N) {
            float best = smax[0][tid];
            int bi = sidx[0][tid];
#pragma unroll
            for (int q = 1; q < WGM; ++q)
                if (smax[q][tid] > best) {   // later wave rows are larger: strict > keeps the first row
                    best = smax[q][tid];
                    bi = sidx[q][tid];
                }
            // batched (z > 0: the second encoder of a pair): the partial arrays of batch z lie z*sCz elements further on
            p.cmax[(long)z * p.sCz + (long)tile_m * p.N + col0 + tid] = best;
            p.cidx[(long)z * p.sCz + (long)tile_m * p.N + col0 + tid] = bi % p.group_rows;
        }
        return;
    }
    const bool partial = p.ksplit > 1;
    float* C = partial ? p.ws + ((long)blockIdx.y) * p.M * p.N : p.C + (long)z * p.sCz;
    const int ldc = partial ? p.N : p.ldc;
    const float* bias = (p.flags & HP_GEMM_BIAS) ? p.bias + (long)z * p.sBiasz : nullptr;
    const float* mask = (p.flags & HP_GEMM_MASK) ? p.mask + (long)z * p.sMaskz : nullptr;
    const float* add = (p.flags & HP_GEMM_ADD) ? p.add + (long)z * p.sAddz : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * WN + j * 32 + r;
            if (col >= p.N) continue;
            const float bv = (!partial && bias) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = acc[i][j][e];
                if (!partial) {
                    v += bv;
                    if (add) v += add[(long)row * p.ldadd + col];
                    if (p.flags & HP_GEMM_RELU) v = fmaxf(v, 0.f);
                    if (mask) v = (mask[(long)row * p.ldmask + col] > 0.f) ? v : 0.f;
                }
                C[(long)row * ldc + col] = v;
            }
        }
    }
}

// sum_s w[s * stride] in slab order; the loads of 16 slabs are issued together (a serial chain of 64 dependent
// round trips to HBM costs 30 us on its own).
__device__ __forceinline__ float slab_sum(const float* __restrict__ w, long stride, int n) {
    float v = 0.f;
    int s = 0;
    for (; s + 16 <= n; s += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = w[(long)(s + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) v += t[u];
    }
    for (; s + 4 <= n; s += 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = w[(long)(s + u) * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += t[u];
    }
    for (; s < n; ++s) v += w[(long)s * stride];
    return v;
}

// C = epi(sum_s ws[z][s]) in split order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const KParams p) {
    const long mn = (long)p.M * p.N;
    const int z = blockIdx.y;
    const float* bias = (p.flags & HP_GEMM_BIAS) ? p.bias + (long)z * p.sBiasz : nullptr;
    const float* mask = (p.flags & HP_GEMM_MASK) ? p.mask + (long)z * p.sMaskz : nullptr;
    const float* add = (p.flags & HP_GEMM_ADD) ? p.add + (long)z * p.sAddz : nullptr;
    float* C = p.C + (long)z * p.sCz;
    // dense output (ldc == N), no addend / mask: 16-byte loads and stores on the flat index (the heads' theta: 3 slabs of
    // 64 x 19011 — 14 us with 4-byte accesses, the largest reduce of the step)
    if (p.ldc == p.N && !add && !mask && (mn & 3) == 0 && ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(p.ws)) & 15) == 0 &&
        (((long)z * p.ksplit * mn) & 3) == 0 && p.ksplit <= 8) {
        const long q4 = mn >> 2;
        for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < q4; t += (long)gridDim.x * 256) {
            const float* w = p.ws + (long)z * p.ksplit * mn + 4 * t;
            float4 v[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
                if (s2 < p.ksplit) v[s2] = *reinterpret_cast<const float4*>(w + (long)s2 * mn);
            float4 a = v[0];
#pragma unroll
            for (int s2 = 1; s2 < 8; ++s2)
                if (s2 < p.ksplit) a = make_float4(a.x + v[s2].x, a.y + v[s2].y, a.z + v[s2].z, a.w + v[s2].w);
            float o[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bias) o[u] += bias[(int)((4 * t + u) % p.N)];
                if (p.flags & HP_GEMM_RELU) o[u] = fmaxf(o[u], 0.f);
            }
            *reinterpret_cast<float4*>(C + 4 * t) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < mn; t += (long)gridDim.x * 256) {
        const float* w = p.ws + (long)z * p.ksplit * mn + t;
        const float v0 = slab_sum(w, mn, p.ksplit);
        float v = v0;
        const int row = (int)(t / p.N), col = (int)(t - (long)row * p.N);
        if (bias) v += bias[col];
        if (add) v += add[(long)row * p.ldadd + col];
        if (p.flags & HP_GEMM_RELU) v = fmaxf(v, 0.f);
        if (mask) v = (mask[(long)row * p.ldmask + col] > 0.f) ? v : 0.f;
        C[(long)row * p.ldc + col] = v;
    }
    if ((p.flags & HP_GEMM_ROWSUM) && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < p.M; i += 256) {
            const float* w = p.ws + p.ws_rsum_off + (long)z * p.ksplit * p.M + i;
            p.rsum[(long)z * p.sRsumz + i] = slab_sum(w, p.M, p.ksplit);
        }
    }
}

// Column sums (bias gradients): out[z][j] = sum_i X[z](i,j).
// Stage 1: a workgroup owns 64 columns x one row slab; its 4 waves stride the slab's rows (each wave reads
// 256 contiguous bytes per row) and combine through LDS -> partial[z][slab][j].  Stage 2 adds the slabs in slab
// order.  Ordered and atomic-free.  With one slab stage 1 writes `out` directly.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long sXz, int ldx, int M, int N,
                                                     const float* __restrict__ mask, long sMaskz, int ldmask,
                                                     float* __restrict__ out, long sOz, int slabs, int rows_per_slab) {
    __shared__ float red[4][64];
    const int z = blockIdx.z, slab = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const float* x = X + (long)z * sXz;
    const float* mk = mask ? mask + (long)z * sMaskz : nullptr;
    const int r0 = slab * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int i = r0 + w;
        if (!mk) {
            for (; i + 12 < r1; i += 16) {
                s0 += x[(long)i * ldx + c];
                s1 += x[(long)(i + 4) * ldx + c];
                s2 += x[(long)(i + 8) * ldx + c];
                s3 += x[(long)(i + 12) * ldx + c];
            }
        }
        for (; i < r1; i += 4) {
            float v = x[(long)i * ldx + c];
            if (mk) v = (mk[(long)i * ldmask + c] > 0.f) ? v : 0.f;
            s0 += v;
        }
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < N) {
        const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (slabs == 1) out[(long)z * sOz + c] = t;
        else out[((long)z * slabs + slab) * N + c] = t;   // out = partial workspace
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int slabs, int N,
                                                            float* __restrict__ out, long sOz) {
    const int z = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const float* p = partial + (long)z * slabs * N + c;
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += p[(long)k * N];
    out[(long)z * sOz + c] = s;
}

template <int BM, int BN, int WGM, int WGN, int BK>
int launch_cfg(KParams& p, int batch, hipStream_t stream) {
    p.kchunk = (p.kchunk + BK - 1) / BK * BK;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    dim3 grid(p.tiles_m * p.tiles_n, batch * p.ksplit), block(WGM * WGN * 64);
    // loader per operand (see gemm_kernel): 1 = K-contiguous and 16-byte loadable, 2 = K-contiguous otherwise,
    // 0 = i/j-contiguous.  The pairs the step produces are instantiated; any other falls back to loader 0, which is
    // correct for every layout.
    const int am = p.sAk == 1 ? (p.vecA ? 1 : 2) : 0, bm = p.sBk == 1 ? (p.vecB ? 1 : 2) : 0;
    static const int pad_lds = [] { const char* e = getenv("HP_GEMM_PAD_LDS"); return e ? atoi(e) : 0; }();   // experiment: occupancy cap
#define HP_GEMM_LAUNCH(MODE_) hipLaunchKernelGGL((gemm_kernel<BM, BN, WGM, WGN, BK, MODE_>), grid, block, (BM >= 128 && BN >= 128) ? pad_lds : 0, stream, p)
    if (am == 1 && bm == 1) HP_GEMM_LAUNCH(4);
    else if (am == 1 && bm == 2) HP_GEMM_LAUNCH(7);
    else if (am == 2 && bm == 0) HP_GEMM_LAUNCH(2);
    else if (am == 2 && bm == 2) HP_GEMM_LAUNCH(8);
    else if (am == 1) HP_GEMM_LAUNCH(1);
    else if (bm == 1) HP_GEMM_LAUNCH(3);
    else HP_GEMM_LAUNCH(0);
#undef HP_GEMM_LAUNCH
    return (int)hipGetLastError();
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Tile choice: the largest tile that still yields >= ~2 workgroups per CU; skinny problems (M <= 64, the
// hypernetwork's B x 19011 heads) and small ones fall through to smaller tiles instead of idling CUs.
// 0: 128x32, 1: 128x128, 2: 64x128, 3: 64x64
constexpr int kCfgRows[4] = {128, 128, 64, 64};
int choose_cfg(const HpGemmDesc* d, int ksplit) {
    const long zs = (long)d->batch * ksplit;
    // a device-side row count is expected to be about a third of its static bound (distinct critical points of an
    // encoder: ~170 of 512 per cloud): tiles are chosen for that many rows, the grid still covers the bound
    const int Meff = (d->dyn_count && d->dyn_kind == 1) ? std::max(64, d->M / 3) : d->M;
    auto wgs = [&](int bm, int bn) { return (long)((Meff + bm - 1) / bm) * ((d->N + bn - 1) / bn) * zs; };
    if (d->N <= 32) return 0;
    // the 128x128 kernel has no K-tail path: whole 16-deep k-tiles per split only
    // (split ranges start at multiples of 32, so K % 16 == 0 makes every range a whole number of k-tiles)
    const bool whole_tiles = d->K >= 16 && d->K % 16 == 0 && !(d->dyn_count && d->dyn_kind == 2);
    if (whole_tiles && Meff > 64 && d->N > 64 && wgs(128, 128) >= 384) return 1;
    if (d->N > 64 && wgs(64, 128) >= 512) return 2;
    return 3;
}

}  // namespace

HP_API long hp_gemm_workspace_floats(const HpGemmDesc* d) {
    if (!d || d->ksplit <= 1) return 0;
    return (long)d->batch * d->ksplit * d->M * d->N + ((d->flags & HP_GEMM_ROWSUM) ? (long)d->batch * d->ksplit * d->M : 0);
}

// rows per output tile hp_gemm_f32 will use for this problem (HP_GEMM_COLMAX partial layout)
HP_API int hp_gemm_tile_rows(const HpGemmDesc* d) {
    if (!d) return -1;
    return kCfgRows[choose_cfg(d, d->ksplit > 1 ? d->ksplit : 1)];
}

HP_API int hp_gemm_f32(const HpGemmDesc* d, hipStream_t stream) {
    HP_CHECK_ARG(d && d->M >= 0 && d->N >= 0 && d->K >= 0 && d->batch >= 0);
    if (d->M == 0 || d->N == 0 || d->batch == 0) return 0;
    HP_CHECK_ARG(d->A && d->B && (d->C || (d->flags & HP_GEMM_COLMAX)));
    if (d->flags & HP_GEMM_COLMAX) {
        HP_CHECK_ARG(d->cmax && d->cidx && d->group_rows > 0 && d->batch >= 1 && d->ksplit <= 1);
        HP_CHECK_ARG(d->group_rows % hp_gemm_tile_rows(d) == 0 && d->M % d->group_rows == 0);
    }
    HP_CHECK_ARG(d->sAi == 1 || d->sAk == 1 || d->M == 1 || d->K == 1);
    HP_CHECK_ARG(d->sBk == 1 || d->sBj == 1 || d->N == 1 || d->K == 1);
    HP_CHECK_ARG(!(d->flags & HP_GEMM_BIAS) || d->bias);
    HP_CHECK_ARG(!(d->flags & HP_GEMM_MASK) || d->mask);
    HP_CHECK_ARG(!(d->flags & HP_GEMM_ADD) || d->add);
    HP_CHECK_ARG(!(d->flags & HP_GEMM_ROWSUM) || d->rsum);
    HP_CHECK_ARG(!d->dyn_count || ((d->dyn_kind == 1 && d->ksplit <= 1 && !(d->flags & HP_GEMM_COLMAX)) || d->dyn_kind == 2));
    HP_CHECK_ARG(d->batch * (long)(d->ksplit > 1 ? d->ksplit : 1) <= 65535);
    KParams p;
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.mask = d->mask; p.add = d->add; p.ws = d->ws;
    p.sAz = d->sAz; p.sBz = d->sBz; p.sCz = d->sCz; p.sBiasz = d->sBiasz; p.sMaskz = d->sMaskz; p.sAddz = d->sAddz;
    p.ldadd = d->ldadd;
    p.cmax = d->cmax; p.cidx = d->cidx; p.group_rows = d->group_rows;
    p.rsum = d->rsum; p.sRsumz = d->sRsumz;
    p.dyn = d->dyn_count; p.dyn_kind = d->dyn_kind;
    p.ws_rsum_off = (long)d->batch * (d->ksplit > 1 ? d->ksplit : 1) * d->M * d->N;
    p.sAi = d->sAi; p.sAk = d->sAk; p.sBk = d->sBk; p.sBj = d->sBj;
    p.ldc = d->ldc; p.ldmask = d->ldmask; p.M = d->M; p.N = d->N; p.K = d->K; p.flags = d->flags;
    p.ksplit = d->ksplit > 1 ? d->ksplit : 1;
    p.kchunk = ((d->K + p.ksplit - 1) / p.ksplit + kMaxBK - 1) / kMaxBK * kMaxBK;   // multiple of every BK in use
    if (p.kchunk == 0) p.kchunk = kMaxBK;
    // a split whose range is empty still writes its (zero) slab, so every slab is initialised
    HP_CHECK_ARG(p.ksplit == 1 || d->ws);
    p.vecA = (d->sAk == 1) && (d->sAi % 4 == 0) && (d->sAz % 4 == 0) && aligned16(d->A);
    p.vecB = (d->sBk == 1) && (d->sBj % 4 == 0) && (d->sBz % 4 == 0) && aligned16(d->B);

    int rc;
    switch (choose_cfg(d, p.ksplit)) {
        case 0: rc = launch_cfg<128, 32, 4, 1, 16>(p, d->batch, stream); break;
        case 1: rc = launch_cfg<128, 128, 4, 2, 16>(p, d->batch, stream); break;
        case 2: rc = launch_cfg<64, 128, 2, 2, 16>(p, d->batch, stream); break;
        default: rc = launch_cfg<64, 64, 2, 2, 32>(p, d->batch, stream); break;
    }
    if (rc) return rc;
    if (p.ksplit > 1) {
        const long mn = (long)d->M * d->N;
        const int blocks = (int)std::min<long>((mn + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, d->batch), dim3(256), 0, stream, p);
    }
    HP_RETURN_LAST_ERROR();
}

// Column sums (bias gradients): out[z][j] = sum_i (mask(i,j)>0 ? X(i,j) : 0).  `ws` (may be NULL) holds
// hp_colsum_workspace_floats floats; without it the rows are not split across workgroups.
HP_API long hp_colsum_workspace_floats(int batch, int M, int N) {
    (void)M;
    return (long)batch * 32 * N;
}

HP_API int hp_colsum_f32(int batch, int M, int N, const float* X, long sXz, int ldx, const float* mask, long sMaskz,
                         int ldmask, float* out, long sOz, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(batch >= 0 && M >= 0 && N >= 0);
    if (batch == 0 || N == 0) return 0;
    HP_CHECK_ARG(batch <= 65535);
    const int colblocks = (N + 63) / 64;
    int slabs = 1;
    if (ws && M >= 512) {
        const long base = (long)colblocks * batch;
        slabs = (int)std::min<long>(32, std::max<long>(1, std::min<long>((768 + base - 1) / base, M / 128)));
    }
    const int rows_per_slab = (M + slabs - 1) / slabs;
    hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, slabs, batch), dim3(256), 0, stream, X, sXz, ldx, M, N, mask, sMaskz,
                       ldmask, slabs == 1 ? out : ws, sOz, slabs, rows_per_slab);
    if (slabs > 1)
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 255) / 256, batch), dim3(256), 0, stream, ws, slabs, N, out, sOz);
    HP_RETURN_LAST_ERROR();
}
