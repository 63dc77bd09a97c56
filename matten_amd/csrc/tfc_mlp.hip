// Fused forward of the radial FullyConnectedNet (eigenn/nn/utils.py:247-252, called at :262):
//
//     h_0 = edge_embedding [E, B];  h_{i+1} = cst * ssp(h_i W_i / sqrt(dim h_i));  w = h_L W_L / sqrt(dim h_L)
//
// One workgroup owns 128 edges for the whole network.  The hidden activations never leave the
// CU (they live k-major in LDS, exactly the layout the next layer's MFMA A operand wants; a copy
// goes to HBM once because backward needs it), and the last layer - K = hidden size only, so a
// stand-alone GEMM would spend most of its time in prologue and epilogue - streams W_L through
// LDS in 64-column tiles while the 128 x hidden activation tile stays put.  fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fmaf chains).
//
// Limits of this kernel (the caller falls back to tfc_gemm otherwise): at least one hidden
// layer, every input / hidden width <= 64, output width a multiple of 4.
#include "tfc_common.h"

#include <cstdlib>

// the [E, W] weights are written once and read by another kernel later: streaming stores keep them from evicting the
// weight tiles and activations this kernel re-reads (config 3: 2.83 -> 2.80 ms)
#define TFC_MLP_OUT_STORE(dst, v) __builtin_nontemporal_store((v), &(dst))

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int MAXL = 6;   // layers
// NA = accumulator tiles (32 rows each) per wave: the workgroup owns RB = 64 * NA edges.  NA = 2
// halves the number of times W_L is streamed from L2; NA = 1 doubles the number of workgroups and
// lets three of them share a CU, which is what fills the chip when E / 128 is only a few hundred.
template <int NA> struct Tile {
    static constexpr int RB = 64 * NA;   // edges per workgroup
    static constexpr int SA = RB + 4;    // LDS row stride of the activation tile [k][row]
};
constexpr int SB = 64 + 4;  // LDS row stride of a weight tile [k][col]

struct MlpArgs {
    const float* x;
    const float* w[MAXL];
    float* hidden[MAXL];  // hidden[i] = output of layer i (i < n_layers - 1), [E, dims[i+1]]
    float* out;
    long n_rows;
    int n_layers;
    int dims[MAXL + 1];
    float act_cst;
};

// cst * (softplus(v) - ln 2), softplus(v) = max(v, 0) + log(1 + exp(-|v|)) on the hardware exp2 / log2
// units: absolute error ~1e-7 (the argument of the log lies in (1, 2]), against ~100 instructions
// per element for the libm log1pf(expf(v)) this replaces
__device__ __forceinline__ float ssp_act(float v, float cst) {
    const float t = __expf(-fabsf(v));
    const float sp = fmaxf(v, 0.f) + __logf(1.f + t);
    return cst * (sp - 0.69314718055994530942f);
}

// stage a weight tile W[k0.., c0 .. c0+63] (row-major [K, ldw]) into LDS as [k][col], zero padded
__device__ __forceinline__ void stage_weights(const float* __restrict__ W, int K, int ldw, int c0, int ncols,
                                              float* __restrict__ Bs, int t) {
    // 64 k x 16 column quads = 1024 float4: 4 per thread
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c4 = t & 15, k = (t >> 4) + 16 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = c0 + 4 * c4;
        if (k < K) {
            if (c + 3 < ncols && (ldw & 3) == 0) {
                v = *reinterpret_cast<const float4*>(W + (long)k * ldw + c);
            } else {
                if (c + 0 < ncols) v.x = W[(long)k * ldw + c + 0];
                if (c + 1 < ncols) v.y = W[(long)k * ldw + c + 1];
                if (c + 2 < ncols) v.z = W[(long)k * ldw + c + 2];
                if (c + 3 < ncols) v.w = W[(long)k * ldw + c + 3];
            }
        }
        *reinterpret_cast<float4*>(&Bs[k * SB + 4 * c4]) = v;
    }
}

template <int NA>
__global__ __launch_bounds__(256, (NA == 1 ? 3 : 2)) void tfc_radial_mlp_fwd_kernel(const MlpArgs a) {
    constexpr int RB = Tile<NA>::RB, SA = Tile<NA>::SA;
    __shared__ __attribute__((aligned(16))) float act[64 * SA];     // current activations, [k][row]
    __shared__ __attribute__((aligned(16))) float wt[2][64 * SB];   // weight tiles, [k][col]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const long r0 = (long)blockIdx.x * RB;

    // input tile, k-major (rows past the end re-read the last row; nothing of them is stored)
    {
        const int d0 = a.dims[0];
        for (int i = t; i < RB * 64; i += 256) {
            const int row = i / 64, k = i - row * 64;
            long r = r0 + row;
            r = r < a.n_rows ? r : a.n_rows - 1;
            act[k * SA + row] = k < d0 ? a.x[r * d0 + k] : 0.f;
        }
    }
    const bool full_rows = r0 + RB <= a.n_rows;  // no row of this workgroup lies past the end
    // hidden layers
    for (int l = 0; l + 1 < a.n_layers; ++l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        stage_weights(a.w[l], din, dout, 0, dout, wt[0], t);
        __syncthreads();
        floatx16 acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int kp = (din + 1) >> 1;  // k pairs (an odd width reads one zero row of padding)
        for (int s = 0; s < kp; ++s) {
            const float b = wt[0][(2 * s + half) * SB + wn * 32 + l31];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const float av = act[(2 * s + half) * SA + wm * 32 * NA + 32 * i + l31];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave has read the old activations
        const float alpha = 1.0f / sqrtf((float)din);
        const int c = wn * 32 + l31;
        float* hid = a.hidden[l];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = wm * 32 * NA + i * 32 + 8 * g + 4 * half;
                float4 v;
                v.x = c < dout ? ssp_act(alpha * acc[i][4 * g + 0], a.act_cst) : 0.f;
                v.y = c < dout ? ssp_act(alpha * acc[i][4 * g + 1], a.act_cst) : 0.f;
                v.z = c < dout ? ssp_act(alpha * acc[i][4 * g + 2], a.act_cst) : 0.f;
                v.w = c < dout ? ssp_act(alpha * acc[i][4 * g + 3], a.act_cst) : 0.f;
                *reinterpret_cast<float4*>(&act[c * SA + row]) = v;
                if (c < dout) {
                    float* __restrict__ hp = hid + (r0 + row) * dout + c;
                    if (full_rows) {
                        hp[0] = v.x;
                        hp[dout] = v.y;
                        hp[2 * dout] = v.z;
                        hp[3 * dout] = v.w;
                    } else {
                        if (r0 + row + 0 < a.n_rows) hp[0] = v.x;
                        if (r0 + row + 1 < a.n_rows) hp[dout] = v.y;
                        if (r0 + row + 2 < a.n_rows) hp[2 * dout] = v.z;
                        if (r0 + row + 3 < a.n_rows) hp[3 * dout] = v.w;
                    }
                }
            }
        }
        __syncthreads();
    }
    // last layer: stream W_L in 64-column tiles, the activation tile stays in LDS
    {
        const int L = a.n_layers - 1;
        const int din = a.dims[L], W = a.dims[L + 1];
        const float* __restrict__ wl = a.w[L];
        const float alpha = 1.0f / sqrtf((float)din);
        const int kp = (din + 1) >> 1;
        const int nsteps = (W + 63) / 64;
        stage_weights(wl, din, W, 0, W, wt[0], t);
        const bool full = full_rows;
        unsigned ooff[NA][16];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                ooff[i][reg] = (unsigned)(wm * 32 * NA + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * (unsigned)W + wn * 32 + l31;
        __syncthreads();
        for (int n = 0; n < nsteps; ++n) {
            const int cur = n & 1;
            // next tile: global -> registers now, registers -> LDS after the MFMAs
            float4 pre[4];
            const bool more = n + 1 < nsteps;
            if (more) {
                const int c0 = (n + 1) * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = t & 15, k = (t >> 4) + 16 * q;
                    const int c = c0 + 4 * c4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < din && c + 3 < W) v = *reinterpret_cast<const float4*>(wl + (long)k * W + c);
                    pre[q] = v;
                }
            }
            floatx16 acc[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const float* Bs = wt[cur];
            // K = 64 (the usual hidden width): all operands of the tile are fetched from LDS into
            // registers first and the 64 matrix instructions then run back to back, waiting only for
            // the operand they are about to use (a rolled loop waits out the LDS latency before every
            // pair of them: the matrix core then idles more than half of the time)
            if (kp == 32) {
                float bv[32], av[NA][32];
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    bv[s] = Bs[(2 * s + half) * SB + wn * 32 + l31];
#pragma unroll
                    for (int i = 0; i < NA; ++i) av[i][s] = act[(2 * s + half) * SA + wm * 32 * NA + 32 * i + l31];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from sinking the reads back to their uses
#pragma unroll
                for (int s = 0; s < 32; ++s) {
#pragma unroll
                    for (int i = 0; i < NA; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[s], acc[i], 0, 0, 0);
                }
            } else {
                for (int s = 0; s < kp; ++s) {
                    const float b = Bs[(2 * s + half) * SB + wn * 32 + l31];
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        const float av = act[(2 * s + half) * SA + wm * 32 * NA + 32 * i + l31];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[i], 0, 0, 0);
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c4 = t & 15, k = (t >> 4) + 16 * q;
                    *reinterpret_cast<float4*>(&wt[cur ^ 1][k * SB + 4 * c4]) = pre[q];
                }
            }
            const int c = n * 64 + wn * 32 + l31;
            if (c < W) {
                // uniform base of the tile + per-lane 32-bit offsets that do not change from tile to tile
                float* __restrict__ ob = a.out + r0 * W + n * 64;
                if (full) {
#pragma unroll
                    for (int i = 0; i < NA; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) TFC_MLP_OUT_STORE(ob[ooff[i][reg]], alpha * acc[i][reg]);
                } else {
#pragma unroll
                    for (int i = 0; i < NA; ++i)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const long row = r0 + wm * 32 * NA + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                            if (row < a.n_rows) TFC_MLP_OUT_STORE(ob[ooff[i][reg]], alpha * acc[i][reg]);
                        }
                }
            }
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fused backward of the hidden chain (what autograd derives from eigenn/nn/utils.py:247-252,262):
//
//     g_L = gout [E, W];   g_i = (g_{i+1} W_i^T / sqrt(d_i)) * act'(h_i)   (i = L-1 .. 1);   dx = g_1 W_0^T / sqrt(d_0)
//
// One workgroup owns RB = 64 NA edges for the whole chain.  The one long product - gout [RB, W] times
// W_{L-1}^T [W, d_{L-1}], K = W up to 8192 - streams both operands in 32-column chunks.  Both are
// contiguous along K in memory (rows of gout, rows of W_{L-1}), so the chunks go to LDS as they lie,
// row-major [row][32 + 4 pad] with 16-byte stores, and a lane fetches its MFMA operands with one
// ds_read_b128 per four matrix instructions: lane (half h, row l) of step group q reads k = 8q + 4h .. + 3
// of its row, so that instruction e of the group contracts k = 8q + e (h = 0) and k = 8q + 4 + e (h = 1) -
// the same assignment on both operands, hence the same sum.  Two chunks are in flight in registers while
// a third is multiplied (as in tfc_gemm.hip).  The short products that follow (K, N <= 64) run on the
// tile kept in LDS: the gradients of the hidden layers go to HBM once (the weight-gradient products
// read them) and never come back.  Replaces three tfc_gemm launches per network.
struct MlpBwdArgs {
    const float* gout;
    const float* w[MAXL];
    const float* hidden[MAXL];  // hidden[i] = output of layer i (i < n_layers - 1), [E, dims[i+1]]
    float* dhidden[MAXL];       // dhidden[i] = gradient of the pre-activation of layer i, [E, dims[i+1]]
    float* dx;                  // nullable
    long n_rows;
    int n_layers;
    int dims[MAXL + 1];
    float act_cst;
};

constexpr int BKC = 32;       // K chunk of the long product
constexpr int SK = BKC + 4;   // LDS row stride of a chunk [row][k]: 9 (odd) 16-byte units => conflict-free b128 reads
constexpr int SX = 64 + 4;    // LDS row stride of the [row][<= 64] tiles of the short products: 17 units

// d act / d pre from the activation OUTPUT h = cst (softplus(pre) - ln 2):  cst (1 - exp(-softplus(pre)))
__device__ __forceinline__ float ssp_grad_from_out(float h, float cst) {
    return cst * (1.f - 0.5f * __expf(-h / cst));
}

template <int NA>
__global__ __launch_bounds__(256, 2) void tfc_radial_mlp_bwd_kernel(const MlpBwdArgs a) {
    constexpr int RB = 64 * NA;
    constexpr int A_TILE = RB * SK, B_TILE = 64 * SK;
    static_assert(RB * SX <= 2 * A_TILE && 64 * SX <= 2 * B_TILE, "the short products reuse the chunk buffers");
    __shared__ __attribute__((aligned(16))) float As[2 * A_TILE];
    __shared__ __attribute__((aligned(16))) float Bs[2 * B_TILE];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const long r0 = (long)blockIdx.x * RB;
    const int L = a.n_layers - 1;
    const int W = a.dims[L + 1], H = a.dims[L];
    const float* __restrict__ gout = a.gout;
    const float* __restrict__ wl = a.w[L];

    floatx16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- the long product ---------------------------------------------------------------------
    const int c4 = t & 7, rr = t >> 3;  // a thread moves float4 number c4 of rows rr, rr + 32, ...
    const float* ga[NA * 2];
#pragma unroll
    for (int q = 0; q < NA * 2; ++q) {
        long r = r0 + rr + 32 * q;
        r = r < a.n_rows ? r : a.n_rows - 1;  // rows past the end re-read the last one; nothing of them is stored
        ga[q] = gout + r * W + 4 * c4;
    }
    const float* gb[2];
    bool okb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = rr + 32 * q;
        okb[q] = j < H;
        gb[q] = wl + (long)(okb[q] ? j : 0) * W + 4 * c4;
    }
    struct Regs {
        floatx4 va[NA * 2], vb[2];
    };
    Regs R0, R1;
    const int nchunk = (W + BKC - 1) / BKC;
    auto request = [&](Regs& R, int ch) {
        const int c0 = ch * BKC;
        const bool okc = c0 + 4 * c4 < W;  // (W is a multiple of 4: a float4 is inside or outside as a whole)
#pragma unroll
        for (int q = 0; q < NA * 2; ++q)
            R.va[q] = okc ? __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(ga[q] + c0)) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
            R.vb[q] = (okc && okb[q]) ? *reinterpret_cast<const floatx4*>(gb[q] + c0) : floatx4{0.f, 0.f, 0.f, 0.f};
    };
    auto stage = [&](const Regs& R, int ch) {
        float* Ad = As + (ch & 1) * A_TILE;
        float* Bd = Bs + (ch & 1) * B_TILE;
#pragma unroll
        for (int q = 0; q < NA * 2; ++q) *reinterpret_cast<floatx4*>(&Ad[(rr + 32 * q) * SK + 4 * c4]) = R.va[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<floatx4*>(&Bd[(rr + 32 * q) * SK + 4 * c4]) = R.vb[q];
    };
    auto multiply = [&](int ch) {
        const float* Ad = As + (ch & 1) * A_TILE + (wm * 32 * NA + l31) * SK + 4 * half;
        const float* Bd = Bs + (ch & 1) * B_TILE + (wn * 32 + l31) * SK + 4 * half;
#pragma unroll
        for (int q = 0; q < BKC / 8; ++q) {
            float4 av[NA];
            const float4 bv = *reinterpret_cast<const float4*>(Bd + 8 * q);
#pragma unroll
            for (int i = 0; i < NA; ++i) av[i] = *reinterpret_cast<const float4*>(Ad + 32 * i * SK + 8 * q);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv.w, acc[i], 0, 0, 0);
            }
        }
    };
    request(R0, 0);
    if (nchunk > 1) request(R1, 1);
    stage(R0, 0);
    __syncthreads();
    {
        auto iteration = [&](Regs& Rold, const Regs& Rnext, int i) {
            if (i + 2 < nchunk) request(Rold, i + 2);
            multiply(i);
            stage(Rnext, i + 1);
            __syncthreads();
        };
        int i = 0;
        for (; i + 2 < nchunk; i += 2) {
            iteration(R0, R1, i);
            iteration(R1, R0, i + 1);
        }
        if (i + 1 < nchunk) iteration(R0, R1, i);
        multiply(nchunk - 1);
    }
    __syncthreads();  // the chunk buffers are reused below

    // ---- the short products -------------------------------------------------------------------
    float* Xs = As;  // [RB][SX]: gradient tile, A operand of the next product
    float* Ws = Bs;  // [64][SX]: W_{li-1} as it lies ([d_{li-1}][d_li]: contiguous along the contraction index)
    for (int li = L; li >= 0; --li) {
        const int d = a.dims[li];  // columns of acc that mean something
        const float alpha = 1.0f / sqrtf((float)d);
        const int col = wn * 32 + l31;
        if (li == 0) {
            if (a.dx && col < d) {
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const long row = r0 + wm * 32 * NA + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                        if (row < a.n_rows) a.dx[row * d + col] = alpha * acc[i][reg];
                    }
            }
            break;
        }
        const float* __restrict__ h = a.hidden[li - 1];
        float* __restrict__ dh = a.dhidden[li - 1];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = wm * 32 * NA + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const long row = r0 + lr;
                float v = 0.f;
                if (col < d && row < a.n_rows) {
                    v = alpha * acc[i][reg] * ssp_grad_from_out(h[row * d + col], a.act_cst);
                    dh[row * d + col] = v;
                }
                Xs[lr * SX + col] = v;
            }
        if (li == 1 && !a.dx) break;
        // W_{li-1} [dn][d] -> Ws[j'][j], zero padded to 64 x 64
        const int dn = a.dims[li - 1];
        const float* __restrict__ wp = a.w[li - 1];
        for (int idx = t; idx < 64 * 64; idx += 256) {
            const int jn = idx >> 6, j = idx & 63;
            Ws[jn * SX + j] = (jn < dn && j < d) ? wp[jn * d + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        {
            const float* Ad = Xs + (wm * 32 * NA + l31) * SX + 4 * half;
            const float* Bd = Ws + (wn * 32 + l31) * SX + 4 * half;
            const int ng = (d + 7) >> 3;
            for (int q = 0; q < ng; ++q) {
                float4 av[NA];
                const float4 bv = *reinterpret_cast<const float4*>(Bd + 8 * q);
#pragma unroll
                for (int i = 0; i < NA; ++i) av[i] = *reinterpret_cast<const float4*>(Ad + 32 * i * SX + 8 * q);
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv.x, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv.y, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv.z, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv.w, acc[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave has read Xs / Ws before the next layer overwrites them
    }
}

}  // namespace

extern "C" int tfc_radial_mlp_fwd(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                                  const float* const* weights, float act_cst, float* const* hidden, float* out,
                                  tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0, TFC_EINVAL, "tfc_radial_mlp_fwd: negative row count");
    TFC_REQUIRE(dims && weights && hidden, TFC_EINVAL, "tfc_radial_mlp_fwd: null table");
    TFC_REQUIRE(n_layers >= 2 && n_layers <= MAXL, TFC_EUNSUPPORTED,
                "tfc_radial_mlp_fwd: %d layers (this kernel takes 2..%d)", n_layers, MAXL);
    for (int i = 0; i <= n_layers; ++i) TFC_REQUIRE(dims[i] > 0, TFC_EINVAL, "tfc_radial_mlp_fwd: width %d", dims[i]);
    for (int i = 0; i < n_layers; ++i)
        TFC_REQUIRE(dims[i] <= 64, TFC_EUNSUPPORTED, "tfc_radial_mlp_fwd: input/hidden width %d > 64", dims[i]);
    TFC_REQUIRE(dims[n_layers] % 4 == 0, TFC_EUNSUPPORTED, "tfc_radial_mlp_fwd: output width %d not a multiple of 4",
                dims[n_layers]);
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && out, TFC_EINVAL, "tfc_radial_mlp_fwd: null tensor");
    TFC_REQUIRE(((uintptr_t)weights[n_layers - 1] & 15) == 0, TFC_EUNSUPPORTED,
                "tfc_radial_mlp_fwd: last weight matrix not 16-byte aligned");
    MlpArgs a;
    a.x = x;
    a.out = out;
    a.n_rows = n_rows;
    a.n_layers = n_layers;
    a.act_cst = act_cst;
    for (int i = 0; i < MAXL; ++i) {
        a.w[i] = nullptr;
        a.hidden[i] = nullptr;
    }
    for (int i = 0; i <= n_layers; ++i) a.dims[i] = dims[i];
    for (int i = 0; i < n_layers; ++i) {
        TFC_REQUIRE(weights[i], TFC_EINVAL, "tfc_radial_mlp_fwd: null weight %d", i);
        a.w[i] = weights[i];
        if (i + 1 < n_layers) {
            TFC_REQUIRE(hidden[i], TFC_EINVAL, "tfc_radial_mlp_fwd: null hidden buffer %d", i);
            a.hidden[i] = hidden[i];
        }
    }
    // 64- or 128-edge workgroups: whichever needs fewer rounds x rows on 256 CUs (3 resp. 2 workgroups
    // resident per CU); a tie goes to 128, which streams W_L half as often
    static const int forced = getenv("TFC_MLP_RB") ? atoi(getenv("TFC_MLP_RB")) : 0;
    const long nb128 = (n_rows + 127) / 128, nb64 = (n_rows + 63) / 64;
    const long cost128 = (nb128 + 511) / 512 * 128, cost64 = (nb64 + 767) / 768 * 64;
    const bool small = forced ? forced == 64 : cost64 < cost128;
    const int rb = small ? 64 : 128;
    const long nb = (n_rows + rb - 1) / rb;
    TFC_REQUIRE(nb < (1LL << 31), TFC_EUNSUPPORTED, "tfc_radial_mlp_fwd: grid too large");
    TfcProfScope prof("mlp_fused_fwd", stream);
    if (small)
        tfc_radial_mlp_fwd_kernel<1><<<(unsigned)nb, 256, 0, stream>>>(a);
    else
        tfc_radial_mlp_fwd_kernel<2><<<(unsigned)nb, 256, 0, stream>>>(a);
    TFC_LAUNCH_CHECK("tfc_radial_mlp_fwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_radial_mlp_bwd_chain_supported(int32_t n_layers, const int32_t* dims) {
    if (!dims || n_layers < 1 || n_layers > MAXL) return 0;
    for (int i = 0; i < n_layers; ++i)
        if (dims[i] <= 0 || dims[i] > 64) return 0;
    return dims[n_layers] > 0 && dims[n_layers] % 4 == 0;
}

// The hidden-gradient chain of tfc_radial_mlp_bwd in one launch (called from tfc_dense.cpp when
// tfc_radial_mlp_bwd_chain_supported and the pointers are 16-byte aligned).
int tfc_radial_mlp_bwd_chain(int64_t n_rows, int32_t n_layers, const int32_t* dims, const float* const* weights,
                             float act_cst, const float* const* hidden, const float* gout, float* const* dhidden,
                             float* dx, hipStream_t stream) {
    MlpBwdArgs a;
    a.gout = gout;
    a.dx = dx;
    a.n_rows = n_rows;
    a.n_layers = n_layers;
    a.act_cst = act_cst;
    for (int i = 0; i < MAXL; ++i) {
        a.w[i] = nullptr;
        a.hidden[i] = nullptr;
        a.dhidden[i] = nullptr;
    }
    for (int i = 0; i <= n_layers; ++i) a.dims[i] = dims[i];
    for (int i = 0; i < n_layers; ++i) {
        a.w[i] = weights[i];
        if (i + 1 < n_layers) {
            a.hidden[i] = hidden[i];
            a.dhidden[i] = dhidden[i];
        }
    }
    static const int forced = getenv("TFC_MLP_BWD_RB") ? atoi(getenv("TFC_MLP_BWD_RB")) : 0;
    // 128- or 64-edge workgroups: the MFMA time of a CU is set by the edges it owns, so take the size whose
    // busiest CU owns fewer (two 128-edge or four 64-edge workgroups resident per CU); a tie goes to 128,
    // which streams W_L half as often
    const long nb128 = (n_rows + 127) / 128, nb64 = (n_rows + 63) / 64;
    const long cost128 = (nb128 + 255) / 256 * 128, cost64 = (nb64 + 255) / 256 * 64;
    const bool small = forced ? forced == 64 : cost64 < cost128;
    const long nb = small ? nb64 : nb128;
    TFC_REQUIRE(nb < (1LL << 31), TFC_EUNSUPPORTED, "tfc_radial_mlp_bwd: grid too large");
    TfcProfScope prof("mlp_fused_bwd", stream);
    if (small)
        tfc_radial_mlp_bwd_kernel<1><<<(unsigned)nb, 256, 0, stream>>>(a);
    else
        tfc_radial_mlp_bwd_kernel<2><<<(unsigned)nb, 256, 0, stream>>>(a);
    TFC_LAUNCH_CHECK("tfc_radial_mlp_bwd_kernel");
    return TFC_OK;
}
