// Strided fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: bit-exact fp32 fmaf
// chains, no reduced-precision path), with decomposed index maps on every operand.
//
// Serves the dense pieces of the layer, forward and backward, without materialising transposes
// or outer products:
//   radial FullyConnectedNet  (eigenn/nn/utils.py:247-252,262)
//   o3.Linear                 (eigenn/nn/point_conv.py:56-61,79 and :130-135,180)
//   FullyConnectedTensorProduct self connection (eigenn/nn/point_conv.py:159-164,185-188)
//
//   C[cR(r) + cJ(j)] (+)= epi( alpha * sum_k A1[a1R(r) + a1K(k)] * A2[a2R(r) + a2K(k)] * B[bK(k) + bJ(j)] )
//   with  f(t) = (t / (d1*d2))*s0 + ((t / d2) % d1)*s1 + (t % d2)*s2  for every map.
//
// Block = 256 threads = 4 waves (2 x 2); block tile (64*TM) x (64*TN), K tile 16; each wave owns
// TM x TN accumulators of 32 x 32.  Operand tiles are staged through LDS as [k][x + 2 pad]
// (k-major so that the MFMA operand read is lane-contiguous and conflict free).  The next K tile's
// global loads are issued before the MFMAs of the current one.
#include "tfc_common.h"

#include <algorithm>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

struct IMapD {
    int d1, d2;
    long s0, s1, s2;
    // constants for advancing the index by BK (mixed-radix add)
    int b2, B1;
    long K0, K1, K2;
};

struct GemmArgs {
    long m, n, k;
    const float* a1;
    const float* a2;
    const float* b;
    float* c;
    const float* aux;
    IMapD a1_r, a1_k, a2_r, a2_k, b_k, b_j, c_r, c_j;
    float alpha, act_cst;
    int accumulate, epilogue;
    int splits;      // > 1: K is cut into `splits` ranges, raw partial tiles go to `slab`
    float* slab;     // [splits][m][n]
};

__device__ __forceinline__ long imap_eval(const IMapD& m, long t) {
    if (m.d1 == 1 && m.d2 == 1) return t * m.s0;
    const long q2 = t / m.d2, r2 = t - q2 * m.d2;
    const long q1 = q2 / m.d1, r1 = q2 - q1 * m.d1;
    return q1 * m.s0 + r1 * m.s1 + r2 * m.s2;
}

struct KTrack {
    int r1, r2;
    long off;
    __device__ __forceinline__ void init(const IMapD& m, long t) {
        const long q2 = t / m.d2;
        r2 = (int)(t - q2 * m.d2);
        const long q1 = q2 / m.d1;
        r1 = (int)(q2 - q1 * m.d1);
        off = q1 * m.s0 + r1 * m.s1 + r2 * m.s2;
    }
    __device__ __forceinline__ void step(const IMapD& m) {
        r2 += m.b2;
        const bool c2 = r2 >= m.d2;
        r2 -= c2 ? m.d2 : 0;
        r1 += m.B1 + (c2 ? 1 : 0);
        const bool c1 = r1 >= m.d1;
        r1 -= c1 ? m.d1 : 0;
        off += m.K0 + (c2 ? m.K2 : 0) + (c1 ? m.K1 : 0);
    }
};

// One operand tile [BK][BX] loaded by 256 threads.  KFAST: consecutive threads walk k (operand
// contiguous along k); otherwise they walk x (contiguous along the row / column index).
template <int BX, bool KFAST, bool HAS2>
struct Operand {
    static constexpr int NE = BX * BK / 256;     // elements per thread
    static constexpr int NT = KFAST ? 1 : NE;    // k tracks per thread
    static constexpr int NX = KFAST ? NE : 1;    // x offsets per thread
    static constexpr int LDS_STRIDE = BX + 2;

    KTrack kt[NT], kt2[HAS2 ? NT : 1];
    long xo[NX], xo2[HAS2 ? NX : 1];
    int kk[NT];  // k index inside the tile of each track
    float v[NE];

    __device__ __forceinline__ void init(const IMapD& mx, const IMapD& mk, const IMapD& mx2, const IMapD& mk2,
                                         long x0, long xmax, int t, long kstart) {
        if (KFAST) {
            kk[0] = t % BK;
            kt[0].init(mk, kstart + kk[0]);
            if (HAS2) kt2[0].init(mk2, kstart + kk[0]);
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                long x = x0 + t / BK + (256 / BK) * q;
                x = x < xmax ? x : xmax - 1;
                xo[q] = imap_eval(mx, x);
                if (HAS2) xo2[q] = imap_eval(mx2, x);
            }
        } else {
            long x = x0 + t % BX;
            x = x < xmax ? x : xmax - 1;
            xo[0] = imap_eval(mx, x);
            if (HAS2) xo2[0] = imap_eval(mx2, x);
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                kk[q] = t / BX + (256 / BX) * q;
                kt[q].init(mk, kstart + kk[q]);
                if (HAS2) kt2[q].init(mk2, kstart + kk[q]);
            }
        }
    }

    // issue the global loads of the tile starting at k0 (values with k >= K read as 0)
    __device__ __forceinline__ void load(const float* __restrict__ p, const float* __restrict__ p2, long k0,
                                         long K) {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int ti = KFAST ? 0 : q, xi = KFAST ? q : 0;
            const bool ok = k0 + kk[ti] < K;
            float a = ok ? p[xo[xi] + kt[ti].off] : 0.f;
            if (HAS2) {
                const float b = ok ? p2[xo2[xi] + kt2[ti].off] : 0.f;
                a *= b;
            }
            v[q] = a;
        }
    }

    __device__ __forceinline__ void advance(const IMapD& mk, const IMapD& mk2) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            kt[q].step(mk);
            if (HAS2) kt2[q].step(mk2);
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int t) const {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            int k, x;
            if (KFAST) {
                k = t % BK;
                x = t / BK + (256 / BK) * q;
            } else {
                k = t / BX + (256 / BX) * q;
                x = t % BX;
            }
            lds[k * LDS_STRIDE + x] = v[q];
        }
    }
};

__device__ __forceinline__ float ssp_act(float v, float cst) {
    const float sp = v > 20.f ? v : log1pf(expf(v));
    return cst * (sp - 0.69314718055994530942f);
}

// d act / d pre from the activation output h = cst * (softplus(pre) - ln 2)
__device__ __forceinline__ float ssp_grad_from_out(float h, float cst) {
    const float sp = h / cst + 0.69314718055994530942f;
    return cst * (1.f - expf(-sp));
}

template <int TM, int TN, bool A_KFAST, bool B_KFAST, bool HAS2>
__global__ __launch_bounds__(256) void tfc_gemm_kernel(const GemmArgs g) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ float lds[BK * (BM + 2) + BK * (BN + 2)];
    float* As = lds;
    float* Bs = lds + BK * (BM + 2);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long nbm = (g.m + BM - 1) / BM, nbn = (g.n + BN - 1) / BN;
    const unsigned vb = tfc_xcd_remap(blockIdx.x, gridDim.x);
    const long tile = vb % (nbm * nbn);
    const int z = (int)(vb / (nbm * nbn));      // K range of this block
    const long bm = tile % nbm, bn = tile / nbm;  // consecutive blocks share the B tile
    const long r0 = bm * BM, j0 = bn * BN;
    const long nkt_all = (g.k + BK - 1) / BK;
    const long per = (nkt_all + g.splits - 1) / g.splits;
    const long kt_begin = z * per;
    const long kt_end = kt_begin + per < nkt_all ? kt_begin + per : nkt_all;

    Operand<BM, A_KFAST, HAS2> A;
    Operand<BN, B_KFAST, false> B;
    A.init(g.a1_r, g.a1_k, g.a2_r, g.a2_k, r0, g.m, t, kt_begin * BK);
    B.init(g.b_j, g.b_k, g.b_j, g.b_k, j0, g.n, t, kt_begin * BK);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long nkt = kt_end;
    A.load(g.a1, g.a2, kt_begin * BK, g.k);
    B.load(g.b, nullptr, kt_begin * BK, g.k);
    for (long kt = kt_begin; kt < nkt; ++kt) {
        A.store(As, t);
        B.store(Bs, t);
        __syncthreads();
        if (kt + 1 < nkt) {
            A.advance(g.a1_k, g.a2_k);
            B.advance(g.b_k, g.b_k);
            A.load(g.a1, g.a2, (kt + 1) * BK, g.k);
            B.load(g.b, nullptr, (kt + 1) * BK, g.k);
        }
        const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[(2 * s + half) * (BM + 2) + wm * 32 * TM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * s + half) * (BN + 2) + wn * 32 * TN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: acc[i][j][reg] is C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long col = j0 + wn * 32 * TN + j * 32 + l31;
        if (col >= g.n) continue;
        const long co = imap_eval(g.c_j, col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const long row = r0 + wm * 32 * TM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (row >= g.m) continue;
                if (g.splits > 1) {
                    g.slab[((long)z * g.m + row) * g.n + col] = acc[i][j][reg];
                    continue;
                }
                const long o = imap_eval(g.c_r, row) + co;
                float v = g.alpha * acc[i][j][reg];
                if (g.epilogue == TFC_EPI_SSP) v = ssp_act(v, g.act_cst);
                if (g.epilogue == TFC_EPI_SSP_GRAD) v *= ssp_grad_from_out(g.aux[o], g.act_cst);
                if (g.accumulate) v += g.c[o];
                g.c[o] = v;
            }
        }
    }
}

// second stage of split-K: ordered sum of the partial slabs, then the usual epilogue
__global__ __launch_bounds__(256) void tfc_gemm_reduce_kernel(const GemmArgs g) {
    const long total = g.m * g.n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long row = i / g.n, col = i - row * g.n;
        float s = 0.f;
        for (int z = 0; z < g.splits; ++z) s += g.slab[(long)z * total + i];
        const long o = imap_eval(g.c_r, row) + imap_eval(g.c_j, col);
        float v = g.alpha * s;
        if (g.epilogue == TFC_EPI_SSP) v = ssp_act(v, g.act_cst);
        if (g.epilogue == TFC_EPI_SSP_GRAD) v *= ssp_grad_from_out(g.aux[o], g.act_cst);
        if (g.accumulate) v += g.c[o];
        g.c[o] = v;
    }
}

int convert_map(const tfc_imap& in, IMapD* out, const char* name) {
    TFC_REQUIRE(in.d1 >= 1 && in.d2 >= 1, TFC_EINVAL, "tfc_gemm: map %s has d1=%d d2=%d", name, in.d1, in.d2);
    out->d1 = in.d1;
    out->d2 = in.d2;
    out->s0 = in.s0;
    out->s1 = in.s1;
    out->s2 = in.s2;
    const int a2 = BK / in.d2;
    out->b2 = BK % in.d2;
    const int A1 = a2 / in.d1;
    out->B1 = a2 % in.d1;
    out->K0 = (long)out->b2 * in.s2 + (long)out->B1 * in.s1 + (long)A1 * in.s0;
    out->K2 = in.s1 - (long)in.d2 * in.s2;
    out->K1 = in.s0 - (long)in.d1 * in.s1;
    return TFC_OK;
}

long map_unit_step(const tfc_imap& m) {  // f(1) - f(0)
    if (m.d2 > 1) return m.s2;
    if (m.d1 > 1) return m.s1;
    return m.s0;
}

template <int TM, int TN>
int launch_tile(const GemmArgs& g, bool a_kfast, bool b_kfast, bool has2, const char* tag, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    const long nb = ((g.m + BM - 1) / BM) * ((g.n + BN - 1) / BN) * g.splits;
    TFC_REQUIRE(nb < (1LL << 31), TFC_EUNSUPPORTED, "tfc_gemm: grid too large");
    const dim3 grid((unsigned)nb), block(256);
    TfcProfScope prof(tag, stream);
#define TFC_GEMM_LAUNCH(AK, BKF, H2) tfc_gemm_kernel<TM, TN, AK, BKF, H2><<<grid, block, 0, stream>>>(g)
    if (has2) {
        if (a_kfast) {
            if (b_kfast) TFC_GEMM_LAUNCH(true, true, true); else TFC_GEMM_LAUNCH(true, false, true);
        } else {
            if (b_kfast) TFC_GEMM_LAUNCH(false, true, true); else TFC_GEMM_LAUNCH(false, false, true);
        }
    } else {
        if (a_kfast) {
            if (b_kfast) TFC_GEMM_LAUNCH(true, true, false); else TFC_GEMM_LAUNCH(true, false, false);
        } else {
            if (b_kfast) TFC_GEMM_LAUNCH(false, true, false); else TFC_GEMM_LAUNCH(false, false, false);
        }
    }
#undef TFC_GEMM_LAUNCH
    TFC_LAUNCH_CHECK("tfc_gemm_kernel");
    if (g.splits > 1) {
        const long total = g.m * g.n;
        tfc_gemm_reduce_kernel<<<(unsigned)std::min<long>((total + 255) / 256, 4096), 256, 0, stream>>>(g);
        TFC_LAUNCH_CHECK("tfc_gemm_reduce_kernel");
    }
    return TFC_OK;
}

}  // namespace

extern "C" int tfc_gemm(const tfc_gemm_desc* d, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(d, TFC_EINVAL, "tfc_gemm: null descriptor");
    TFC_REQUIRE(d->m >= 0 && d->n >= 0 && d->k >= 0, TFC_EINVAL, "tfc_gemm: negative size");
    TFC_REQUIRE(d->m < (1LL << 31) && d->n < (1LL << 31) && d->k < (1LL << 31), TFC_EUNSUPPORTED,
                "tfc_gemm: dimension >= 2^31");
    if (d->m == 0 || d->n == 0) return TFC_OK;
    TFC_REQUIRE(d->c, TFC_EINVAL, "tfc_gemm: null C");
    TFC_REQUIRE(d->k == 0 || (d->a1 && d->b), TFC_EINVAL, "tfc_gemm: null operand");
    TFC_REQUIRE(d->epilogue >= TFC_EPI_NONE && d->epilogue <= TFC_EPI_SSP_GRAD, TFC_EINVAL,
                "tfc_gemm: unknown epilogue %d", d->epilogue);
    TFC_REQUIRE(d->epilogue != TFC_EPI_SSP_GRAD || d->aux, TFC_EINVAL, "tfc_gemm: SSP_GRAD needs aux");
    GemmArgs g;
    g.m = d->m;
    g.n = d->n;
    g.k = d->k;
    g.a1 = d->a1;
    g.a2 = d->a2;
    g.b = d->b;
    g.c = d->c;
    g.aux = d->aux;
    g.alpha = d->alpha;
    g.act_cst = d->act_cst;
    g.accumulate = d->accumulate;
    g.epilogue = d->epilogue;
    int rc;
    if ((rc = convert_map(d->a1_r, &g.a1_r, "a1_r"))) return rc;
    if ((rc = convert_map(d->a1_k, &g.a1_k, "a1_k"))) return rc;
    if (d->a2) {
        if ((rc = convert_map(d->a2_r, &g.a2_r, "a2_r"))) return rc;
        if ((rc = convert_map(d->a2_k, &g.a2_k, "a2_k"))) return rc;
    } else {
        g.a2_r = g.a1_r;
        g.a2_k = g.a1_k;
    }
    if ((rc = convert_map(d->b_k, &g.b_k, "b_k"))) return rc;
    if ((rc = convert_map(d->b_j, &g.b_j, "b_j"))) return rc;
    if ((rc = convert_map(d->c_r, &g.c_r, "c_r"))) return rc;
    if ((rc = convert_map(d->c_j, &g.c_j, "c_j"))) return rc;

    // walk each operand along the index that is contiguous in memory
    const bool a_kfast = !(map_unit_step(d->a1_r) == 1 && map_unit_step(d->a1_k) != 1);
    const bool b_kfast = (map_unit_step(d->b_k) == 1 && map_unit_step(d->b_j) != 1);
    const bool has2 = d->a2 != nullptr;

    const char* tag = d->tag ? d->tag : "tfc_gemm_kernel";
    g.splits = 1;
    g.slab = nullptr;
    // tile: large where the problem fills the chip, small otherwise
    const long b22 = ((d->m + 127) / 128) * ((d->n + 127) / 128);
    if (d->n > 64 && d->m > 64 && b22 >= 256) return launch_tile<2, 2>(g, a_kfast, b_kfast, has2, tag, stream);
    if (d->n <= 64 && d->m > 64 && ((d->m + 127) / 128) >= 256)
        return launch_tile<2, 1>(g, a_kfast, b_kfast, has2, tag, stream);
    // small grids with a long K: cut K so that the chip is filled (deterministic: partial tiles
    // go to the caller's workspace and are summed in a fixed order)
    const long b11 = ((d->m + 63) / 64) * ((d->n + 63) / 64);
    const long nkt = (d->k + BK - 1) / BK;
    if (b11 < 256 && nkt >= 16 && d->workspace) {
        long splits = std::min<long>((512 + b11 - 1) / b11, nkt / 8);
        splits = std::min<long>(splits, d->workspace_floats / std::max<long>(d->m * d->n, 1));
        if (splits > 1) {
            const long per = (nkt + splits - 1) / splits;
            splits = (nkt + per - 1) / per;  // no empty range
            g.splits = (int)splits;
            g.slab = d->workspace;
        }
    }
    return launch_tile<1, 1>(g, a_kfast, b_kfast, has2, tag, stream);
}

extern "C" int64_t tfc_gemm_workspace_floats(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const long b11 = ((m + 63) / 64) * ((n + 63) / 64);
    const long nkt = (k + BK - 1) / BK;
    if (b11 >= 256 || nkt < 16) return 0;
    const long splits = std::min<long>((512 + b11 - 1) / b11, nkt / 8);
    return splits > 1 ? splits * m * n : 0;
}
