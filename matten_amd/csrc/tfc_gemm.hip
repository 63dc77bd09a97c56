// Strided fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: bit-exact fp32 fmaf
// chains, no reduced-precision path), with decomposed index maps on every operand, launched
// one problem at a time (tfc_gemm) or as a group of independent problems (tfc_gemm_group).
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
// TM x TN accumulators of 32 x 32.  Operand tiles are staged through LDS as [k][x + 4 pad]
// (k-major: the MFMA operand read is lane-contiguous and conflict free), double buffered, one
// barrier per K tile, the next tile's global loads in flight during the MFMAs of the current one.
//
// Every operand address is rowoff(x) + koff(k).  Both are evaluated ONCE per block (rows) or
// once per K tile by 16 threads (k) into small LDS tables, so that no thread carries per-element
// index state in registers: the accumulators and the staging registers are all a wave holds
// (the first version of this kernel tracked a mixed-radix index per element in VGPRs and
// spilled at the 128 x 128 tile).
#include "tfc_common.h"

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <vector>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;   // K tile (and the unit the host counts K ranges in)
// (The staging code and the kernel take the K tile as a template parameter, BKT.  32 was measured in round 3, all
// instantiations correct: with every 64 x 64 / 128 x 64 tile at 32 the radial network's weight gradient gains 10 %
// and o3.Linear loses 20 % (config 4: 0.29 -> 0.23 of the matrix peak; 90 - 150 VGPRs and 43 KB of LDS instead of
// 40 - 44 and 25 KB, three workgroups per CU instead of four), step unchanged; with only the weight gradient of the
// radial network at 32 the step is unchanged too (config 3 / 4 / 5: 2.71 / 39.0 / 3.84 against 2.70 / 39.2 / 3.81 ms).
// Every instantiation therefore runs at 16.)
constexpr int KOFF = 128;  // k offsets evaluated together (KOFF / BKT tiles at a time)
constexpr int MAXP = 8;  // problems per launch (the group is passed by value as kernel argument)

struct Map {
    int d1, d2;
    long s0, s1, s2;
};

struct Problem {
    const float* a1;
    const float* a2;
    const float* b;
    float* c;
    const float* aux;
    float* slab;  // [splits][m][n] raw partial tiles when splits > 1
    Map a1_r, a1_k, a2_r, a2_k, b_k, b_j, c_r, c_j;
    int m, n, k;
    int splits;       // > 1: K is cut into `splits` ranges
    int block_begin;  // first virtual block of this problem in the launch
    int accumulate, epilogue;
    int a_rep;         // A element = sum over a_rep repetitions of the two factors, strides below
    long a1_rs, a2_rs;
    float alpha, act_cst;
};

struct Group {
    int n;
    Problem p[MAXP];
};

__device__ __forceinline__ long map_eval(const Map& m, int t) {
    const int q2 = t / m.d2, r2 = t - q2 * m.d2;
    const int q1 = q2 / m.d1, r1 = q2 - q1 * m.d1;
    return (long)q1 * m.s0 + (long)r1 * m.s1 + (long)r2 * m.s2;
}

__device__ __forceinline__ float ssp_act(float v, float cst) {
    const float sp = v > 20.f ? v : log1pf(expf(v));
    return cst * (sp - 0.69314718055994530942f);
}

// d act / d pre from the activation output h = cst * (softplus(pre) - ln 2)
__device__ __forceinline__ float ssp_grad_from_out(float h, float cst) {
    const float sp = h / cst + 0.69314718055994530942f;
    return cst * (1.f - expf(-sp));
}

__device__ __forceinline__ float epilogue_apply(const Problem& P, float acc, long o) {
    float v = P.alpha * acc;
    if (P.epilogue == TFC_EPI_SSP) v = ssp_act(v, P.act_cst);
    if (P.epilogue == TFC_EPI_SSP_GRAD) v *= ssp_grad_from_out(P.aux[o], P.act_cst);
    if (P.accumulate) v += P.c[o];
    return v;
}

// One operand tile [BK][BX] loaded by 256 threads.  KFAST: consecutive threads walk k (operand
// contiguous along k); otherwise they walk x (contiguous along the row / column index).
// load() only ISSUES the global loads; the product of the two factors and the zeroing of the K
// tail happen in store(), after the MFMAs of the current tile, so that nothing waits on the loads
// while they are in flight.
template <int BX, bool KFAST, bool HAS2, bool VEC, int REP = 1, int BK = 16>
struct Operand {
    static constexpr int NE = BX * BK / 256;  // elements per thread
    static constexpr int S = BX + 4;          // LDS row stride
    float v[NE], v2[HAS2 ? NE : 1];

    __device__ __forceinline__ void load(const float* __restrict__ p, const float* __restrict__ p2,
                                         const long* __restrict__ roff, const long* __restrict__ roff2,
                                         const long* __restrict__ ko, const long* __restrict__ ko2, int t, long,
                                         long) {
        if (KFAST) {
            const int kk = t % BK;
            const long o = ko[kk];
            const long o2 = HAS2 ? ko2[kk] : 0;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int x = t / BK + (256 / BK) * q;
                v[q] = p[roff[x] + o];
                if (HAS2) v2[q] = p2[roff2[x] + o2];
            }
        } else {
            const int x = t % BX;
            const long r = roff[x];
            const long r2 = HAS2 ? roff2[x] : 0;
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int kk = t / BX + (256 / BX) * q;
                v[q] = p[r + ko[kk]];
                if (HAS2) v2[q] = p2[r2 + ko2[kk]];
            }
        }
    }

    // k_left = K - (first k of the tile): elements with kk >= k_left are stored as 0
    __device__ __forceinline__ void store(float* __restrict__ lds, int t, int k_left) const {
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
            float a = v[q];
            if (HAS2) a *= v2[q];
            lds[k * S + x] = k < k_left ? a : 0.f;
        }
    }
};

// 16-byte variant: every thread moves groups of 4 elements that are consecutive along the walked
// index.  A single factor must be contiguous there (one global_load_dwordx4 per group).  With two
// factors, the FIRST is constant over a group (one dword, e.g. x[n,u,m] over the attribute index
// v of the self connection) and the SECOND contiguous; the host swaps the factors into that order
// and selects this variant only when the maps, the base pointers and the extents guarantee
// aligned, in-bounds groups (flags_of()).
template <int BX, bool KFAST, bool HAS2, int REP, int BK>
struct Operand<BX, KFAST, HAS2, true, REP, BK> {
    static constexpr int NV = BX * BK / 1024;  // groups per thread
    static constexpr int S = BX + 4;
    static constexpr int KQ = BK / 4;          // 16-byte groups along k per row
    float4 v[NV][REP];
    float s[HAS2 ? NV : 1][REP];

    // REP > 1 (two factors only): the element is sum_{m < REP} f1(. + m rs1) * f2(. + m rs2), summed
    // at store time - a contraction index that is too short for the matrix cores (the 2l+1
    // components of an irrep) folded into the staging instead of lengthening K by 2l+1
    __device__ __forceinline__ void load(const float* __restrict__ p, const float* __restrict__ p2,
                                         const long* __restrict__ roff, const long* __restrict__ roff2,
                                         const long* __restrict__ ko, const long* __restrict__ ko2, int t,
                                         long rs1, long rs2) {
        // (p, roff, ko) address the broadcast factor when HAS2, the only factor otherwise
        if (KFAST) {
            const int kq = t % KQ;  // k = 4 kq .. 4 kq + 3
            const long o = ko[4 * kq];
            const long o2 = HAS2 ? ko2[4 * kq] : 0;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int x = t / KQ + (256 / KQ) * q;
                if (HAS2) {
                    const float* b1 = p + roff[x] + o;
                    const float* b2 = p2 + roff2[x] + o2;
#pragma unroll
                    for (int m = 0; m < REP; ++m) {
                        s[q][m] = b1[m * rs1];
                        v[q][m] = *reinterpret_cast<const float4*>(b2 + m * rs2);
                    }
                } else {
                    v[q][0] = *reinterpret_cast<const float4*>(p + roff[x] + o);
                }
            }
        } else {
            const int xq = t % (BX / 4);  // x = 4 xq .. 4 xq + 3
            const long r = roff[4 * xq];
            const long r2 = HAS2 ? roff2[4 * xq] : 0;
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int kk = t / (BX / 4) + (1024 / BX) * q;
                if (HAS2) {
                    const float* b1 = p + r + ko[kk];
                    const float* b2 = p2 + r2 + ko2[kk];
#pragma unroll
                    for (int m = 0; m < REP; ++m) {
                        s[q][m] = b1[m * rs1];
                        v[q][m] = *reinterpret_cast<const float4*>(b2 + m * rs2);
                    }
                } else {
                    v[q][0] = *reinterpret_cast<const float4*>(p + r + ko[kk]);
                }
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int t, int k_left) const {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float4 a = v[q][0];
            if (HAS2) {
                a.x *= s[q][0];
                a.y *= s[q][0];
                a.z *= s[q][0];
                a.w *= s[q][0];
#pragma unroll
                for (int m = 1; m < REP; ++m) {
                    a.x = fmaf(s[q][m], v[q][m].x, a.x);
                    a.y = fmaf(s[q][m], v[q][m].y, a.y);
                    a.z = fmaf(s[q][m], v[q][m].z, a.z);
                    a.w = fmaf(s[q][m], v[q][m].w, a.w);
                }
            }
            if (KFAST) {
                const int k = 4 * (t % KQ), x = t / KQ + (256 / KQ) * q;
                if (k_left < BK) {  // K tail (last tile only)
                    a.x = k + 0 < k_left ? a.x : 0.f;
                    a.y = k + 1 < k_left ? a.y : 0.f;
                    a.z = k + 2 < k_left ? a.z : 0.f;
                    a.w = k + 3 < k_left ? a.w : 0.f;
                }
                lds[(k + 0) * S + x] = a.x;
                lds[(k + 1) * S + x] = a.y;
                lds[(k + 2) * S + x] = a.z;
                lds[(k + 3) * S + x] = a.w;
            } else {
                const int k = t / (BX / 4) + (1024 / BX) * q, x = 4 * (t % (BX / 4));
                if (k >= k_left) a = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&lds[k * S + x]) = a;
            }
        }
    }
};

template <int TM, int TN, int REP, bool A_KFAST, bool B_KFAST, bool HAS2, bool AV, bool BV, int BKT = 16>
__global__ __launch_bounds__(256, (TM * TN == 4 ? 4 : 1)) void tfc_gemm_kernel(const Group g) {
    constexpr int BK = BKT, KT = KOFF / BK;  // (KT: K tiles whose k offsets are evaluated together)
    constexpr int BM = 64 * TM, BN = 64 * TN, SA = BM + 4, SB = BN + 4;
    constexpr int TILE = BK * SA + BK * SB;
    __shared__ __attribute__((aligned(16))) float tiles[2 * TILE];
    __shared__ long roff[3 * BM + BN];  // rows of A1, rows of A2, rows of C, columns of B
    __shared__ long koff[2][3][KT * BK];  // k offsets of KT tiles at a time (double buffered): A1, A2, B
    long* roffA = roff;
    long* roffA2 = roff + BM;
    long* roffC = roff + 2 * BM;
    long* roffB = roff + 3 * BM;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int vb = (int)tfc_xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
    for (int i = 1; i < g.n; ++i)
        if (vb >= g.p[i].block_begin) pi = i;
    const Problem& P = g.p[pi];
    // loop invariants out of the kernel-argument segment (a scalar load inside the K loop would
    // put an lgkmcnt(0) wait, which also drains the LDS queue, into every tile)
    const float* __restrict__ pa1 = P.a1;
    const float* __restrict__ pa2 = HAS2 ? P.a2 : P.a1;
    const float* __restrict__ pb = P.b;
    const int K = P.k;
    const Map ka1 = P.a1_k, ka2 = P.a2_k, kb = P.b_k;

    const int local = vb - P.block_begin;
    const int nbm = (P.m + BM - 1) / BM, nbn = (P.n + BN - 1) / BN;
    const int ntile = nbm * nbn;
    const int z = local / ntile;  // K range of this block
    const int tile = local - z * ntile;
    const int bm = tile % nbm, bn = tile / nbm;  // consecutive blocks share the B tile
    const int r0 = bm * BM, j0 = bn * BN;
    const int nkt_all = (P.k + BK - 1) / BK;
    const int per = (nkt_all + P.splits - 1) / P.splits;
    const int kt_begin = z * per;
    const int kt_end = kt_begin + per < nkt_all ? kt_begin + per : nkt_all;
    const int n_tiles = kt_end - kt_begin;

    // offset tables
    if (t < BM) {
        // rows past the end re-read valid ones (never stored); a 16-byte group stays a group
        const int row = r0 + t < P.m ? r0 + t : ((AV && !A_KFAST) ? P.m - 4 + (t & 3) : P.m - 1);
        roffA[t] = map_eval(P.a1_r, row);
        if (HAS2) roffA2[t] = map_eval(P.a2_r, row);
        roffC[t] = map_eval(P.c_r, r0 + t < P.m ? r0 + t : P.m - 1);
    }
    if (t >= 128 && t < 128 + BN) {
        const int col = j0 + (t - 128) < P.n ? j0 + (t - 128) : ((BV && !B_KFAST) ? P.n - 4 + (t & 3) : P.n - 1);
        roffB[t - 128] = map_eval(P.b_j, col);
    }
    // k offsets: the integer divisions of the maps cost a few hundred cycles, so they are done
    // for KT tiles at once by 128 threads in parallel (every KT-th iteration), not by 16 threads
    // in every tile, which would make one wave the straggler of every barrier
    auto fill_koff = [&](int first_tile) {  // tiles first_tile .. first_tile + KT - 1 (relative)
        if (t < KT * BK) {
            const int buf = (first_tile / KT) & 1, kk = t % BK;
            const int k = (kt_begin + first_tile + t / BK) * BK + kk;
            int kc_a = k < K ? k : ((AV && A_KFAST) ? K - 4 + (kk & 3) : K - 1);
            int kc_b = k < K ? k : ((BV && B_KFAST) ? K - 4 + (kk & 3) : K - 1);
            if (kc_a < 0) kc_a = 0;
            if (kc_b < 0) kc_b = 0;
            koff[buf][0][t] = map_eval(ka1, kc_a);
            if (HAS2) koff[buf][1][t] = map_eval(ka2, kc_a);
            koff[buf][2][t] = map_eval(kb, kc_b);
        }
    };
    fill_koff(0);
    __syncthreads();

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two tiles of operands are in flight in registers: while tile i is multiplied, tile i + 1 (requested one
    // iteration ago) is moved to LDS and tile i + 2 is requested - a load has two iterations to arrive instead of
    // one (the MFMAs of one 16-deep tile last ~0.2 us, a load from L2 / HBM several times that).  The loop is
    // unrolled by two so that the two register sets keep fixed names.
    Operand<BM, A_KFAST, HAS2, AV, REP, BK> A0, A1;
    Operand<BN, B_KFAST, false, BV, 1, BK> B0, B1;
    const long rs1 = P.a1_rs, rs2 = P.a2_rs;
    auto request = [&](auto& A, auto& B, int tile) {  // tile: relative index
        const int kb_ = (tile / KT) & 1, ko_ = (tile % KT) * BK;
        A.load(pa1, pa2, roffA, roffA2, koff[kb_][0] + ko_, koff[kb_][1] + ko_, t, rs1, rs2);
        B.load(pb, nullptr, roffB, nullptr, koff[kb_][2] + ko_, nullptr, t, 0, 0);
    };
    auto stage = [&](const auto& A, const auto& B, int tile) {
        const int k_left = K - (kt_begin + tile) * BK;
        A.store(tiles + (tile & 1) * TILE, t, k_left);
        B.store(tiles + (tile & 1) * TILE + BK * SA, t, k_left);
    };
    if (n_tiles > 0) {
        request(A0, B0, 0);
        if (n_tiles > 1) request(A1, B1, 1);
        stage(A0, B0, 0);
    }
    __syncthreads();

    const int half = lane >> 5, l31 = lane & 31;
    auto mfma_tile = [&](int cur) {
        const float* As = tiles + cur * TILE;
        const float* Bs = As + BK * SA;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float af[TM], bf[TN];
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) af[ii] = As[(2 * s + half) * SA + wm * 32 * TM + ii * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * s + half) * SB + wn * 32 * TN + j * 32 + l31];
#pragma unroll
            for (int ii = 0; ii < TM; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ii], bf[j], acc[ii][j], 0, 0, 0);
        }
    };
    // iteration i: request tile i + 2 into the set that held tile i (already in LDS), multiply tile i, move tile
    // i + 1 to LDS.  The k offsets of tiles i + KT .. replace those of tiles i - KT .. i - 1 when i is a multiple of
    // KT; the request of tile i + 2 reads the table filled at least KT - 2 iterations (and one barrier) earlier.
    auto iteration = [&](auto& Aold, auto& Bold, const auto& Anext, const auto& Bnext, int i) {
        if (i + 2 < n_tiles) request(Aold, Bold, i + 2);
        mfma_tile(i & 1);
        if (i % KT == 0 && i + KT < n_tiles) fill_koff(i + KT);
        stage(Anext, Bnext, i + 1);
        __syncthreads();
    };
    {
        int i = 0;
        for (; i + 2 < n_tiles; i += 2) {
            iteration(A0, B0, A1, B1, i);
            iteration(A1, B1, A0, B0, i + 1);
        }
        if (i + 1 < n_tiles) iteration(A0, B0, A1, B1, i);
    }
    if (n_tiles > 0) mfma_tile((n_tiles - 1) & 1);

    // epilogue: acc[i][j][reg] is C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + wn * 32 * TN + j * 32 + l31;
        if (col >= P.n) continue;
        const long co = map_eval(P.c_j, col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = wm * 32 * TM + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int row = r0 + lr;
                if (row >= P.m) continue;
                if (P.splits > 1) {
                    P.slab[((long)z * P.m + row) * P.n + col] = acc[i][j][reg];
                } else {
                    const long o = roffC[lr] + co;
                    P.c[o] = epilogue_apply(P, acc[i][j][reg], o);
                }
            }
        }
    }
}

// second stage of split-K: ordered sum of the partial slabs, then the usual epilogue
// (Done INSIDE the product's launch instead - a counter per output tile, the workgroup that finishes a tile's last K
// range adds the partial tiles after an agent-scope release / acquire pair - it was built and measured in round 3:
// correct on every test, and the step doubles, config 3 2.73 -> 5.62 ms, o3.Linear 0.146 -> 0.037 of the matrix peak:
// on gfx950 that fence is a write-back and an invalidation of the XCD's L2, once per workgroup.  A launch of its own
// costs 5 - 9 us.)
// A workgroup owns 64 consecutive elements; its four waves each sum a quarter of the slabs (a
// contiguous z range, ascending), and wave 0 adds the four quarter sums in order: a fixed
// association, so the result does not depend on the launch, and the serial chain of a long split
// (up to 64 slabs) is a quarter as long and spread over four times as many CUs for small outputs.
__global__ __launch_bounds__(256) void tfc_gemm_reduce_kernel(const Group g) {
    const Problem& P = g.p[blockIdx.y];
    if (P.splits <= 1) return;
    __shared__ float part[4][64];
    const long total = (long)P.m * P.n;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int per = (P.splits + 3) >> 2;
    const int z0 = q * per, z1 = min(P.splits, z0 + per);
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long i = base + lane;
        float s = 0.f;
        if (i < total) {
            const float* src = P.slab + i;
            int z = z0;
            for (; z + 4 <= z1; z += 4) {
                const float v0 = src[(long)z * total], v1 = src[(long)(z + 1) * total];
                const float v2 = src[(long)(z + 2) * total], v3 = src[(long)(z + 3) * total];
                s = ((s + v0) + v1) + v2 + v3;
            }
            for (; z < z1; ++z) s += src[(long)z * total];
        }
        part[q][lane] = s;
        __syncthreads();
        if (q == 0 && i < total) {
            s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
            const int row = (int)(i / P.n), col = (int)(i - (long)row * P.n);
            const long o = map_eval(P.c_r, row) + map_eval(P.c_j, col);
            P.c[o] = epilogue_apply(P, s, o);
        }
        __syncthreads();
    }
}

int convert_map(const tfc_imap& in, Map* out, const char* name) {
    TFC_REQUIRE(in.d1 >= 1 && in.d2 >= 1, TFC_EINVAL, "tfc_gemm: map %s has d1=%d d2=%d", name, in.d1, in.d2);
    out->d1 = in.d1;
    out->d2 = in.d2;
    out->s0 = in.s0;
    out->s1 = in.s1;
    out->s2 = in.s2;
    return TFC_OK;
}

long host_map_eval(const tfc_imap& m, long t) {
    const long q2 = t / m.d2, r2 = t - q2 * m.d2;
    const long q1 = q2 / m.d1, r1 = q2 - q1 * m.d1;
    return q1 * m.s0 + r1 * m.s1 + r2 * m.s2;
}

// 128-byte lines one wave-wide load instruction of the staging code touches: 16 k x 4 rows when
// the threads walk k, 64 rows at one k otherwise (sampled at the origin of the operand)
int lines_touched(const tfc_imap& r, const tfc_imap& k, long n_rows, long n_k, bool kfast) {
    long seen[64];
    int n = 0;
    auto add = [&](long off) {
        const long line = off >= 0 ? off >> 5 : -((-off) >> 5) - 1;
        for (int i = 0; i < n; ++i)
            if (seen[i] == line) return;
        seen[n++] = line;
    };
    if (kfast) {
        for (long x = 0; x < std::min<long>(4, n_rows); ++x)
            for (long kk = 0; kk < std::min<long>(16, n_k); ++kk) add(host_map_eval(r, x) + host_map_eval(k, kk));
    } else {
        for (long x = 0; x < std::min<long>(64, n_rows); ++x) add(host_map_eval(r, x));
    }
    return n;
}

// Stride of the map inside every aligned group of 4 consecutive indices, if it is 0 or 1 there
// (-1 otherwise).
int group_stride(const tfc_imap& m) {
    long s;
    if (m.d2 > 1) {
        if (m.d2 % 4) return -1;
        s = m.s2;
    } else if (m.d1 > 1) {
        if (m.d1 % 4) return -1;
        s = m.s1;
    } else {
        s = m.s0;
    }
    return s == 0 ? 0 : (s == 1 ? 1 : -1);
}

// every value of the map at a multiple of `step` (1: any index) is a multiple of 4
bool offsets_mult4(const tfc_imap& m, long extent, bool fast_unit) {
    // fast_unit: the innermost active stride is 1 and only group starts (multiples of 4) matter
    bool inner_done = !fast_unit;
    if (m.d2 > 1) {
        if (inner_done && m.s2 % 4) return false;
        inner_done = true;
    }
    if (m.d1 > 1) {
        if (inner_done && m.s1 % 4) return false;
        inner_done = true;
    }
    if (extent > (long)m.d1 * m.d2 || (m.d1 == 1 && m.d2 == 1)) {
        if (inner_done && m.s0 % 4) return false;
    }
    return true;
}

// Can one factor (base p, map `fast` along the walked index, `slow` along the other) be staged
// in 16-byte groups?  *bc = the factor is constant over a group.
bool factor_vec_ok(const float* p, const tfc_imap& fast, long n_fast, const tfc_imap& slow, long n_slow, int* bc) {
    const int gs = group_stride(fast);
    if (gs < 0) return false;
    *bc = gs == 0;
    if (gs == 0) return true;
    return ((uintptr_t)p % 16 == 0) && offsets_mult4(fast, n_fast, true) && offsets_mult4(slow, n_slow, false);
}

struct Flags {
    bool a_kfast, b_kfast, has2, av, bv;
    int rep;
    bool operator==(const Flags& o) const {
        return a_kfast == o.a_kfast && b_kfast == o.b_kfast && has2 == o.has2 && av == o.av && bv == o.bv &&
               rep == o.rep;
    }
};

// Walk each operand in the order that costs the fewest vector-cache cycles: the staging loads are
// gathers, so the 128-byte lines a wave's load instruction touches (one per cycle), not the
// matrix pipe, set the speed when the 64 addresses are spread out; 16-byte loads need a quarter
// of the instructions.
// *swap12: the 16-byte path wants the broadcast factor first; set when a1 and a2 must trade places
Flags flags_of(const tfc_gemm_desc& d, bool* swap12) {
    Flags f;
    f.has2 = d.a2 != nullptr;
    f.rep = d.a_rep > 1 ? d.a_rep : 1;
    int cost[2];
    bool vec[2], swp[2] = {false, false};
    for (int kf = 0; kf < 2; ++kf) {
        const tfc_imap& fast1 = kf ? d.a1_k : d.a1_r;
        const tfc_imap& slow1 = kf ? d.a1_r : d.a1_k;
        const long nf = kf ? d.k : d.m, ns = kf ? d.m : d.k;
        int lines = lines_touched(d.a1_r, d.a1_k, d.m, d.k, kf);
        int bc1 = 0, bc2 = 0;
        bool ok = nf >= 4 && nf % 4 == 0 && factor_vec_ok(d.a1, fast1, nf, slow1, ns, &bc1);
        if (f.has2) {
            lines += lines_touched(d.a2_r, d.a2_k, d.m, d.k, kf);
            ok = ok && factor_vec_ok(d.a2, kf ? d.a2_k : d.a2_r, nf, kf ? d.a2_r : d.a2_k, ns, &bc2);
            ok = ok && (bc1 != bc2);  // one broadcast, one contiguous
            swp[kf] = ok && bc2;
        } else {
            ok = ok && !bc1;
        }
        vec[kf] = ok;
        cost[kf] = lines * (ok ? 1 : 3);
    }
    f.a_kfast = cost[1] <= cost[0];
    f.av = vec[f.a_kfast];
    *swap12 = swp[f.a_kfast];
    for (int kf = 0; kf < 2; ++kf) {
        const long nf = kf ? d.k : d.n, ns = kf ? d.n : d.k;
        int bc = 0;
        const bool ok = nf >= 4 && nf % 4 == 0 &&
                        factor_vec_ok(d.b, kf ? d.b_k : d.b_j, nf, kf ? d.b_j : d.b_k, ns, &bc) && !bc;
        vec[kf] = ok;
        cost[kf] = lines_touched(d.b_j, d.b_k, d.n, d.k, kf) * (ok ? 1 : 3);
    }
    f.b_kfast = cost[1] < cost[0];
    f.bv = vec[f.b_kfast];
    return f;
}

int validate(const tfc_gemm_desc* d) {
    TFC_REQUIRE(d->m >= 0 && d->n >= 0 && d->k >= 0, TFC_EINVAL, "tfc_gemm: negative size");
    TFC_REQUIRE(d->m < (1LL << 31) && d->n < (1LL << 31) && d->k < (1LL << 31), TFC_EUNSUPPORTED,
                "tfc_gemm: dimension >= 2^31");
    if (d->m == 0 || d->n == 0) return TFC_OK;
    TFC_REQUIRE(d->c, TFC_EINVAL, "tfc_gemm: null C");
    TFC_REQUIRE(d->k == 0 || (d->a1 && d->b), TFC_EINVAL, "tfc_gemm: null operand");
    TFC_REQUIRE(d->epilogue >= TFC_EPI_NONE && d->epilogue <= TFC_EPI_SSP_GRAD, TFC_EINVAL,
                "tfc_gemm: unknown epilogue %d", d->epilogue);
    TFC_REQUIRE(d->epilogue != TFC_EPI_SSP_GRAD || d->aux, TFC_EINVAL, "tfc_gemm: SSP_GRAD needs aux");
    return TFC_OK;
}

int convert(const tfc_gemm_desc* d, Problem* p) {
    p->a1 = d->a1;
    p->a2 = d->a2;
    p->b = d->b;
    p->c = d->c;
    p->aux = d->aux;
    p->slab = nullptr;
    p->m = (int)d->m;
    p->n = (int)d->n;
    p->k = (int)d->k;
    p->splits = 1;
    p->block_begin = 0;
    p->alpha = d->alpha;
    p->act_cst = d->act_cst;
    p->accumulate = d->accumulate;
    p->epilogue = d->epilogue;
    p->a_rep = d->a_rep > 1 ? d->a_rep : 1;
    p->a1_rs = d->a1_rep_stride;
    p->a2_rs = d->a2_rep_stride;
    int rc;
    if ((rc = convert_map(d->a1_r, &p->a1_r, "a1_r"))) return rc;
    if ((rc = convert_map(d->a1_k, &p->a1_k, "a1_k"))) return rc;
    if (d->a2) {
        if ((rc = convert_map(d->a2_r, &p->a2_r, "a2_r"))) return rc;
        if ((rc = convert_map(d->a2_k, &p->a2_k, "a2_k"))) return rc;
    } else {
        p->a2_r = p->a1_r;
        p->a2_k = p->a1_k;
    }
    if ((rc = convert_map(d->b_k, &p->b_k, "b_k"))) return rc;
    if ((rc = convert_map(d->b_j, &p->b_j, "b_j"))) return rc;
    if ((rc = convert_map(d->c_r, &p->c_r, "c_r"))) return rc;
    if ((rc = convert_map(d->c_j, &p->c_j, "c_j"))) return rc;
    return TFC_OK;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

template <int TM, int TN, bool... B>
void dispatch(const Group& g, dim3 grid, hipStream_t stream) {
    tfc_gemm_kernel<TM, TN, 1, B...><<<grid, dim3(256), 0, stream>>>(g);
}
// repeated two-factor operand (a_rep = 3, 5, 7): 64 x 64 tile, A walked along k in 16-byte groups
template <int REP>
void dispatch_rep(const Group& g, dim3 grid, hipStream_t stream, bool b_kfast, bool bv) {
    if (b_kfast) {
        if (bv) tfc_gemm_kernel<1, 1, REP, true, true, true, true, true><<<grid, dim3(256), 0, stream>>>(g);
        else tfc_gemm_kernel<1, 1, REP, true, true, true, true, false><<<grid, dim3(256), 0, stream>>>(g);
    } else {
        if (bv) tfc_gemm_kernel<1, 1, REP, true, false, true, true, true><<<grid, dim3(256), 0, stream>>>(g);
        else tfc_gemm_kernel<1, 1, REP, true, false, true, true, false><<<grid, dim3(256), 0, stream>>>(g);
    }
}
template <int TM, int TN, bool... B, typename... Rest>
void dispatch(const Group& g, dim3 grid, hipStream_t stream, bool first, Rest... rest) {
    if (first)
        dispatch<TM, TN, B..., true>(g, grid, stream, rest...);
    else
        dispatch<TM, TN, B..., false>(g, grid, stream, rest...);
}

template <int TM, int TN>
int launch_tile(Group& g, const Flags& f, const char* tag, hipStream_t stream) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    long nb = 0;
    bool any_split = false;
    for (int i = 0; i < g.n; ++i) {
        g.p[i].block_begin = (int)nb;
        nb += cdiv(g.p[i].m, BM) * cdiv(g.p[i].n, BN) * g.p[i].splits;
        any_split |= g.p[i].splits > 1;
        TFC_REQUIRE(nb < (1LL << 31), TFC_EUNSUPPORTED, "tfc_gemm: grid too large");
    }
    const dim3 grid((unsigned)nb), block(256);
    TfcProfScope prof(tag, stream);
    if (f.rep == 1) dispatch<TM, TN>(g, grid, stream, f.a_kfast, f.b_kfast, f.has2, f.av, f.bv);
    else if (f.rep == 3) dispatch_rep<3>(g, grid, stream, f.b_kfast, f.bv);
    else if (f.rep == 5) dispatch_rep<5>(g, grid, stream, f.b_kfast, f.bv);
    else dispatch_rep<7>(g, grid, stream, f.b_kfast, f.bv);
    TFC_LAUNCH_CHECK("tfc_gemm_kernel");
    if (any_split) {
        long big = 1;
        for (int i = 0; i < g.n; ++i)
            if (g.p[i].splits > 1) big = std::max(big, (long)g.p[i].m * g.p[i].n);
        const dim3 rgrid((unsigned)std::min<long>(cdiv(big, 64), 8192), (unsigned)g.n);
        tfc_gemm_reduce_kernel<<<rgrid, block, 0, stream>>>(g);
        TFC_LAUNCH_CHECK("tfc_gemm_reduce_kernel");
    }
    return TFC_OK;
}

// Upper bound on the K ranges a problem may be cut into: ranges of at least 8 K tiles, at most
// 64 of them, and at most 2^24 floats (64 MiB) of partial tiles.
long max_splits(long m, long n, long k) {
    const long nkt = cdiv(k, BK);
    if (nkt < 16) return 1;
    const long mn = std::max<long>(m * n, 1);
    return std::max<long>(1, std::min<long>(std::min<long>(nkt / 8, 64), (1L << 24) / mn));
}

// one launch: problems with equal operand walking order, at most MAXP of them
int launch_group(Group& g, const Flags& f, float* workspace, long workspace_floats, const char* tag,
                 hipStream_t stream) {
    // tile: the largest one that still fills the chip several times over (counting the K ranges a
    // problem can be cut into) and does not pad the problems much (a 128-wide tile on a 64-column
    // problem would spend half its MFMAs on padding)
    const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    long blocks[3], padded[3];
    for (int c = 0; c < 3; ++c) {
        blocks[c] = padded[c] = 0;
        for (int i = 0; i < g.n; ++i) {
            const long tiles = cdiv(g.p[i].m, cand[c][0]) * cdiv(g.p[i].n, cand[c][1]);
            blocks[c] += tiles * (workspace ? max_splits(g.p[i].m, g.p[i].n, g.p[i].k) : 1);
            padded[c] += tiles * cand[c][0] * cand[c][1] * cdiv(g.p[i].k, BK);
        }
    }
    int pick = 2;
    // a larger tile must still leave ~8 workgroups per CU: below that the launch is latency-bound and
    // twice as many 64 x 64 workgroups finish sooner (measured: config 3 step 4.19 -> 3.97 ms, config 5
    // 5.62 -> 5.42, config 4 unchanged; no threshold at all loses 1 % at config 4)
    static const long min_blocks = getenv("TFC_GEMM_MIN_BLOCKS") ? atol(getenv("TFC_GEMM_MIN_BLOCKS")) : 2048;
    for (int c = 1; c >= 0 && f.rep == 1; --c)  // the repeated-factor staging exists for the 64 x 64 tile only
        if (blocks[c] >= min_blocks && padded[c] * 100 <= padded[2] * 112) pick = c;
    const int bm = cand[pick][0], bn = cand[pick][1];
    // K ranges.  All blocks of a launch should carry a similar number of K tiles (the problems of
    // a group differ in K by up to 2 l + 1 and in the number of tiles by more), and a launch with
    // few tiles and a long K must still fill the chip: cut every problem into ranges of at most
    // `len` K tiles, where `len` spreads the launch's work over ~4 blocks per CU.  Deterministic:
    // partial tiles go to the caller's workspace and are summed in a fixed order.
    if (workspace) {
        long work = 0;
        for (int i = 0; i < g.n; ++i)
            work += cdiv(g.p[i].m, bm) * cdiv(g.p[i].n, bn) * cdiv(g.p[i].k, BK);
        // (round 3 sweep, ms per step at configs 3 / 5 / 4 - node-side products at 256: 2.76 / 4.21 / -, 384: 2.73 / - / 38.5,
        // 512: 2.69 / 3.82 / 38.4, 768: 2.73 / - / 38.8, 1024: 2.75 / 3.90 / 38.5; the radial network's, whose outputs are
        // 10 - 100 x larger, lose at 512 (config 5: 3.94) and gain nothing at 2048: fewer, longer K ranges mean fewer partial
        // tiles to write and to sum again)
        static const long target_node = getenv("TFC_GEMM_TARGET_BLOCKS") ? atol(getenv("TFC_GEMM_TARGET_BLOCKS")) : 512;
        static const long target_mlp = getenv("TFC_GEMM_TARGET_BLOCKS_MLP") ? atol(getenv("TFC_GEMM_TARGET_BLOCKS_MLP")) : 1024;
        const long target = (tag && !strncmp(tag, "gemm_mlp", 8)) ? target_mlp : target_node;
        const long len = std::max<long>(8, cdiv(work, target));
        long used = 0;
        for (int i = 0; i < g.n; ++i) {
            Problem& p = g.p[i];
            const long nkt = cdiv(p.k, BK);
            const long mn = std::max<long>((long)p.m * p.n, 1);
            // a range a little longer than `len` is cheaper than a second slab and a reduction pass
            long splits = nkt * 2 > len * 3 ? std::min<long>(cdiv(nkt, len), max_splits(p.m, p.n, p.k)) : 1;
            splits = std::min<long>(splits, (workspace_floats - used) / mn);
            if (splits > 1) splits = cdiv(nkt, cdiv(nkt, splits));  // no empty range
            if (splits > 1) {
                p.splits = (int)splits;
                p.slab = workspace + used;
                used += splits * mn;
            }
        }
    }
    if (bm == 128 && bn == 128) return launch_tile<2, 2>(g, f, tag, stream);
    if (bm == 128) return launch_tile<2, 1>(g, f, tag, stream);
    return launch_tile<1, 1>(g, f, tag, stream);
}

}  // namespace

extern "C" int tfc_gemm_group(const tfc_gemm_desc* descs, int32_t n_descs, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_descs >= 0, TFC_EINVAL, "tfc_gemm_group: negative count");
    if (n_descs == 0) return TFC_OK;
    TFC_REQUIRE(descs, TFC_EINVAL, "tfc_gemm_group: null descriptors");
    for (int i = 0; i < n_descs; ++i) {
        const int rc = validate(&descs[i]);
        if (rc) return rc;
    }
    // launches: runs of problems (in the given order) that walk their operands the same way
    std::vector<char> done(n_descs, 0);
    for (int first = 0; first < n_descs; ++first) {
        if (done[first]) continue;
        if (descs[first].m == 0 || descs[first].n == 0) {
            done[first] = 1;
            continue;
        }
        bool swap12 = false;
        const Flags f = flags_of(descs[first], &swap12);
        if (f.rep > 1) {
            const tfc_gemm_desc& d0 = descs[first];
            TFC_REQUIRE((f.rep == 3 || f.rep == 5 || f.rep == 7) && f.has2 && f.av && f.a_kfast, TFC_EUNSUPPORTED,
                        "tfc_gemm: a_rep=%d needs two factors that can be staged in 16-byte groups along k "
                        "(one constant, one contiguous over each group) and a_rep in {3,5,7}", f.rep);
            TFC_REQUIRE((swap12 ? d0.a1_rep_stride : d0.a2_rep_stride) % 4 == 0, TFC_EUNSUPPORTED,
                        "tfc_gemm: repetition stride of the contiguous factor must be a multiple of 4");
        }
        Group g;
        g.n = 0;
        float* ws = nullptr;
        long ws_floats = 0;
        const char* tag = descs[first].tag ? descs[first].tag : "tfc_gemm_kernel";
        for (int i = first; i < n_descs && g.n < MAXP; ++i) {
            if (done[i] || descs[i].m == 0 || descs[i].n == 0 || !(flags_of(descs[i], &swap12) == f)) continue;
            const int rc = convert(&descs[i], &g.p[g.n]);
            if (rc) return rc;
            if (swap12) {
                std::swap(g.p[g.n].a1, g.p[g.n].a2);
                std::swap(g.p[g.n].a1_r, g.p[g.n].a2_r);
                std::swap(g.p[g.n].a1_k, g.p[g.n].a2_k);
                std::swap(g.p[g.n].a1_rs, g.p[g.n].a2_rs);
            }
            if (descs[i].workspace && descs[i].workspace_floats > ws_floats) {
                ws = descs[i].workspace;
                ws_floats = descs[i].workspace_floats;
            }
            done[i] = 1;
            ++g.n;
        }
        const int rc = launch_group(g, f, ws, ws_floats, tag, stream);
        if (rc) return rc;
    }
    return TFC_OK;
}

extern "C" int tfc_gemm(const tfc_gemm_desc* d, tfc_stream_t stream) {
    TFC_REQUIRE(d, TFC_EINVAL, "tfc_gemm: null descriptor");
    return tfc_gemm_group(d, 1, stream);
}

extern "C" int64_t tfc_gemm_workspace_floats(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const long splits = max_splits(m, n, k);
    return splits > 1 ? splits * m * n : 0;
}
