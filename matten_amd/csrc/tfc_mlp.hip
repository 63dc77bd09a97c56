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
#ifdef TFC_MLP_PLAIN_STORE
#define TFC_MLP_OUT_STORE(dst, v) ((dst) = (v))
#else
#define TFC_MLP_OUT_STORE(dst, v) __builtin_nontemporal_store((v), &(dst))
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int MAXL = 6;   // layers
// NA = accumulator tiles (32 rows each) per wave: the workgroup owns RB = 64 * NA edges.  NA = 2
// halves the number of times W_L is streamed from L2; NA = 1 doubles the number of workgroups and
// lets three of them share a CU, which is what fills the chip when E / 128 is only a few hundred.
// WM = rows of waves (two waves side by side per row): 2 in the fp32 kernels (256 threads), 4 in the bf16-piece
// kernel (512 threads, RB = 256).
template <int NA, int WM = 2> struct Tile {
    static constexpr int RB = 32 * NA * WM;  // edges per workgroup
    static constexpr int SA = RB + 4;        // LDS row stride of the activation tile [k][row]
    static constexpr int NT = 128 * WM;      // threads per workgroup
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
template <int NT = 256>
__device__ __forceinline__ void stage_weights(const float* __restrict__ W, int K, int ldw, int c0, int ncols,
                                              float* __restrict__ Bs, int t) {
    // 64 k x 16 column quads = 1024 float4: 1024 / NT per thread
#pragma unroll
    for (int q = 0; q < 1024 / NT; ++q) {
        const int c4 = t & 15, k = (t >> 4) + (NT / 16) * q;
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

// Input tile and hidden layers of the forward: on return act holds the last hidden activations of the
// workgroup's RB edges, [k][row] with k < 64 (zero beyond the layer's width), and every wave has passed the
// barrier that follows their last write.
template <int NA, int WM = 2>
__device__ __forceinline__ void mlp_hidden_forward(const MlpArgs& a, float* __restrict__ act, float* __restrict__ wt0,
                                                   const int t, const long r0, const bool full_rows) {
    constexpr int RB = Tile<NA, WM>::RB, SA = Tile<NA, WM>::SA, NT = Tile<NA, WM>::NT;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    // input tile, k-major (rows past the end re-read the last row; nothing of them is stored)
    {
        const int d0 = a.dims[0];
        for (int i = t; i < RB * 64; i += NT) {
            const int row = i / 64, k = i - row * 64;
            long r = r0 + row;
            r = r < a.n_rows ? r : a.n_rows - 1;
            act[k * SA + row] = k < d0 ? a.x[r * d0 + k] : 0.f;
        }
    }
    for (int l = 0; l + 1 < a.n_layers; ++l) {
        const int din = a.dims[l], dout = a.dims[l + 1];
        stage_weights<NT>(a.w[l], din, dout, 0, dout, wt0, t);
        __syncthreads();
        floatx16 acc[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int kp = (din + 1) >> 1;  // k pairs (an odd width reads one zero row of padding)
        for (int s = 0; s < kp; ++s) {
            const float b = wt0[(2 * s + half) * SB + wn * 32 + l31];
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
    const bool full_rows = r0 + RB <= a.n_rows;  // no row of this workgroup lies past the end
    mlp_hidden_forward<NA>(a, act, wt[0], t, r0, full_rows);
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
// fp32 products on the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16, 16 x the rate of the fp32 form).
//
// A float is cut into three bf16 pieces, x = h + m + l: h = x truncated to its upper 16 bits (never overflows),
// m = bf16(x - h), l = bf16(x - h - m), round to nearest.  x - h and x - h - m are exact in fp32 and the last
// remainder has at most 8 significant bits, so the three pieces add up to x EXACTLY.  A product a b is then the sum
// of nine piece products, each exact in the fp32 accumulator of the instruction; the three smallest (m l, l m, l l:
// below 2^-24 |a b|) are dropped, the other six are accumulated - the five small ones into an accumulator of their
// own, added to the h h sum at the end.  Measured against fp64 (tools/selftest/bf16x3_probe.hip, K = 16 .. 65536,
// normal operands): max error / sum |a b| = 0.6 - 0.9e-7, against 1.6 - 2.2e-7 for the fmaf chain of
// v_mfma_f32_32x32x2_f32 - no precision is given up.  Six instructions of 32 cycles contract K = 16 where the fp32
// form needs eight of 64 cycles: 2.67 x the fp32 matrix rate, which moves these products from the matrix cores'
// roofline to HBM's (their [E, W] operand).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

struct Pieces8 {  // eight consecutive k of one row / column, as MFMA operand fragments
    uintx4 h, m, l;
};

__device__ __forceinline__ Pieces8 split8(const float* x) {
    Pieces8 r;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned u0 = __float_as_uint(x[2 * p]), u1 = __float_as_uint(x[2 * p + 1]);
        r.h[p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // (u1 & 0xffff0000) | (u0 >> 16)
        const float a0 = x[2 * p] - __uint_as_float(u0 & 0xffff0000u);
        const float a1 = x[2 * p + 1] - __uint_as_float(u1 & 0xffff0000u);
        const bf16x2 mm = {(__bf16)a0, (__bf16)a1};
        const unsigned mu = __builtin_bit_cast(unsigned, mm);
        r.m[p] = mu;
        const float b0 = a0 - __uint_as_float(mu << 16);
        const float b1 = a1 - __uint_as_float(mu & 0xffff0000u);
        const bf16x2 ll = {(__bf16)b0, (__bf16)b1};
        r.l[p] = __builtin_bit_cast(unsigned, ll);
    }
    return r;
}

// acc += a_h b_h ; lo += the five next terms (smallest first)
__device__ __forceinline__ void mfma_x3(const Pieces8& a, const Pieces8& b, floatx16& acc, floatx16& lo) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.h), am = __builtin_bit_cast(bf16x8, a.m), al = __builtin_bit_cast(bf16x8, a.l);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.h), bm = __builtin_bit_cast(bf16x8, b.m), bl = __builtin_bit_cast(bf16x8, b.l);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, lo, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// LDS image of an operand tile as pieces: three planes [h | m | l], each [row or column][64 k] bf16 with k contiguous
// and a row stride of 144 bytes (9 sixteen-byte units: ds_read_b128 / ds_write_b128 of consecutive rows are free of
// bank conflicts).
constexpr int X3_ROW = 72;                 // bf16 per row (64 + 8 pad)
constexpr int X3_PLANE64 = 64 * X3_ROW;    // bf16 per plane of a 64-row tile

__device__ __forceinline__ void x3_store(__bf16* tile, int plane_elems, int row, int k0, const Pieces8& p) {
    *reinterpret_cast<uintx4*>(tile + row * X3_ROW + k0) = p.h;
    *reinterpret_cast<uintx4*>(tile + plane_elems + row * X3_ROW + k0) = p.m;
    *reinterpret_cast<uintx4*>(tile + 2 * plane_elems + row * X3_ROW + k0) = p.l;
}
__device__ __forceinline__ Pieces8 x3_load(const __bf16* tile, int plane_elems, int row, int k0) {
    Pieces8 p;
    p.h = *reinterpret_cast<const uintx4*>(tile + row * X3_ROW + k0);
    p.m = *reinterpret_cast<const uintx4*>(tile + plane_elems + row * X3_ROW + k0);
    p.l = *reinterpret_cast<const uintx4*>(tile + 2 * plane_elems + row * X3_ROW + k0);
    return p;
}

constexpr int X3_TILE = 3 * X3_PLANE64;         // bf16 per 64-row tile image (27 648 bytes)
constexpr int X3_TILE_VEC = X3_TILE * 2 / 16;    // 16-byte units per image

// Pre-pass: a weight matrix cut into pieces and laid out as the LDS images the product kernels copy in, one
// contiguous image per 64-wide tile:  img[tile][plane][r][k] = piece(src[tile * 64 * (on_r ? rs : ks) + r rs + k ks]).
// The kernels that stream the matrix then read consecutive memory (the rows of the matrix itself lie W floats apart,
// and at W = 8192 every row of a tile sits on the same L2 channel: 1564 workgroups fetching their copies of a tile
// took three times as long as the products on it), and nobody repeats the cutting.
__global__ __launch_bounds__(256) void tfc_x3_pack_kernel(const float* __restrict__ src, __bf16* __restrict__ img, int ntiles,
                                                           long rs, long ks, int r_lim, int k_lim, int on_r) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int tile = (int)(idx >> 9);
    if (tile >= ntiles) return;
    const int in = (int)(idx & 511);
    const int r = on_r ? (in & 63) : (in >> 3), o = on_r ? (in >> 6) : (in & 7);
    const int rr = (on_r ? tile * 64 : 0) + r, k0 = (on_r ? 0 : tile * 64) + 8 * o;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (rr < r_lim && k0 + j < k_lim) ? src[rr * rs + (long)(k0 + j) * ks] : 0.f;
    // forward images (on_r): column c of the tile goes to image row (c >> 1) + 32 (c & 1), so that lane l of the kernel
    // ends up with the adjacent columns 2 l and 2 l + 1 in its two accumulators and stores them as one 8-byte value
    const int ir = on_r ? (r >> 1) + 32 * (r & 1) : r;
    x3_store(img + (long)tile * X3_TILE, X3_PLANE64, ir, 8 * o, split8(x));
}

// Forward with the last layer on the bf16 instruction.  512 threads own 256 edges; the hidden layers as in
// tfc_radial_mlp_fwd_kernel.  Then wave i keeps the pieces of its 32 rows of the last hidden activations (A operand,
// K = 64) in registers for the rest of the kernel, and W_L goes by in 64-column tiles whose images the pre-pass made:
// straight 16-byte copies into LDS (double buffered); a wave multiplies its rows with both 32-column halves of the
// tile (24 matrix instructions each) and streams the results out.
__global__ __launch_bounds__(512, 1) void tfc_radial_mlp_fwd_x3_kernel(const MlpArgs a, const uintx4* __restrict__ wimg) {
    constexpr int NA = 2, WM = 4, RB = Tile<NA, WM>::RB, SA = Tile<NA, WM>::SA, NT = Tile<NA, WM>::NT;
    constexpr int SMEM_FLOATS = 64 * SA + 64 * SB;
    static_assert(2 * X3_TILE * 2 <= SMEM_FLOATS * 4, "the weight tiles reuse the hidden layers' LDS");
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* act = smem;
    float* wt0 = smem + 64 * SA;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const long r0 = (long)blockIdx.x * RB;
    const bool full_rows = r0 + RB <= a.n_rows;
    mlp_hidden_forward<NA, WM>(a, act, wt0, t, r0, full_rows);

    const int L = a.n_layers - 1;
    const int din = a.dims[L], W = a.dims[L + 1];
    const float alpha = 1.0f / sqrtf((float)din);
    const int ntiles = (W + 63) / 64;

    // this wave's rows of the activations as operand fragments: k step s, lane (row l31, half) holds k = 16 s + 8 half + j
    Pieces8 A[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = act[(16 * s + 8 * half + j) * SA + 32 * wave + l31];
        A[s] = split8(x);
    }
    __syncthreads();  // the activations are in registers: their LDS becomes the weight tiles
    uintx4* tiles = reinterpret_cast<uintx4*>(smem);

    constexpr int NQ = (X3_TILE_VEC + NT - 1) / NT;  // 16-byte units per thread and tile (the last one partly)
    uintx4 pre[NQ];
    auto request = [&](int n) {
        const uintx4* __restrict__ src = wimg + (long)n * X3_TILE_VEC;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = t + NT * q;
            pre[q] = src[i < X3_TILE_VEC ? i : X3_TILE_VEC - 1];
        }
    };
    auto stage = [&](int n) {
        uintx4* dst = tiles + (n & 1) * X3_TILE_VEC;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = t + NT * q;
            if (q + 1 < NQ || i < X3_TILE_VEC) dst[i] = pre[q];
        }
    };
    // Workgroups walk the tiles from different starting points.  They run in step, and rows of the output lie W floats
    // apart: if all of them wrote column block n at the same time, the whole chip's stores (and its reads of the image)
    // would land on the few memory channels that block maps to (W = 8192: a 32 KB stride).
#ifdef TFC_X3_NOSTAGGER
    const int first = 0;
#else
    const int first = (int)(blockIdx.x % (unsigned)ntiles);
#endif
    auto tile_of = [&](int i) { const int n = first + i; return n < ntiles ? n : n - ntiles; };
    request(tile_of(0));
    stage(0);
    __syncthreads();
    const long row_base = r0 + 32 * wave + 4 * half;  // + (reg & 3) + 8 (reg >> 2)
    floatx2 outv[16];
    float* __restrict__ prev_ob = a.out;
    bool fast_prev = false;  // per lane: outv holds a tile that still has to be stored
    for (int it = 0; it < ntiles; ++it) {
        const bool more = it + 1 < ntiles;
        const int n = tile_of(it);
        if (more) request(tile_of(it + 1));
        const __bf16* tile = reinterpret_cast<const __bf16*>(tiles + (it & 1) * X3_TILE_VEC);
        // The results of a tile wait in registers and leave while the NEXT tile is multiplied, two stores behind every
        // group of six matrix instructions: issued in one burst after the products they fill the store queue, the
        // waves stall on it and the matrix cores idle until it drains (measured: products and stores added up).
        floatx16 acc[2], lo[2];
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n2][r] = 0.f, lo[n2][r] = 0.f;
            Pieces8 B[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) B[s] = x3_load(tile, X3_PLANE64, 32 * n2 + l31, 16 * s + 8 * half);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#ifndef TFC_X3_NOMFMA
                mfma_x3(A[s], B[s], acc[n2], lo[n2]);
#else
                acc[n2][s] += __uint_as_float(B[s].h[0] ^ A[s].h[1]), lo[n2][s] += __uint_as_float(B[s].m[1] ^ B[s].l[2]);
#endif
                if (fast_prev) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int reg = 2 * (4 * n2 + s) + q;
                        const int dr = (reg & 3) + 8 * (reg >> 2);
#ifdef TFC_X3_NOSTORE
                        if (outv[reg].x == 12345.678f)
#endif
                        TFC_MLP_OUT_STORE(*reinterpret_cast<floatx2*>(prev_ob + (long)dr * W), outv[reg]);
                    }
                }
            }
        }
        // lane l31 holds columns 2 l31 (accumulator 0) and 2 l31 + 1 (accumulator 1) of the tile: 256 contiguous
        // bytes per row and half wave
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) outv[reg] = floatx2{alpha * (acc[0][reg] + lo[0][reg]), alpha * (acc[1][reg] + lo[1][reg])};
        const int c = n * 64 + 2 * l31;
        prev_ob = a.out + row_base * W + (c < W ? c : 0);
        fast_prev = full_rows && c < W;
        if (!full_rows && c < W) {  // the workgroup with rows past the end: checked stores, right away
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int dr = (reg & 3) + 8 * (reg >> 2);
                if (row_base + dr < a.n_rows) TFC_MLP_OUT_STORE(*reinterpret_cast<floatx2*>(prev_ob + (long)dr * W), outv[reg]);
            }
        }
        if (more) stage(it + 1);
        __syncthreads();
    }
    if (fast_prev) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int dr = (reg & 3) + 8 * (reg >> 2);
            TFC_MLP_OUT_STORE(*reinterpret_cast<floatx2*>(prev_ob + (long)dr * W), outv[reg]);
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

extern "C" int64_t tfc_radial_mlp_fwd_workspace_floats(int32_t n_layers, const int32_t* dims) {
    if (!dims || n_layers < 1 || dims[n_layers] <= 0) return -1;
    return (int64_t)((dims[n_layers] + 63) / 64) * (X3_TILE * 2 / 4);
}

extern "C" int tfc_radial_mlp_fwd(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                                  const float* const* weights, float act_cst, float* const* hidden, float* out,
                                  tfc_stream_t stream) {
    return tfc_radial_mlp_fwd_ws(x, n_rows, n_layers, dims, weights, act_cst, hidden, out, nullptr, 0, stream);
}

extern "C" int tfc_radial_mlp_fwd_ws(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                                     const float* const* weights, float act_cst, float* const* hidden, float* out,
                                     float* workspace, int64_t workspace_floats, tfc_stream_t stream_) {
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
    // the last layer on the bf16 matrix instruction (three-piece operands, see split8) when the caller brought the
    // scratch for the images of W_L; TFC_MLP_X3=0 keeps the fp32 form
    static const bool x3 = !(getenv("TFC_MLP_X3") && atoi(getenv("TFC_MLP_X3")) == 0);
    const int64_t need = tfc_radial_mlp_fwd_workspace_floats(n_layers, dims);
    if (x3 && workspace && workspace_floats >= need && ((uintptr_t)workspace & 15) == 0) {
        const int L = n_layers - 1, W = dims[n_layers], ntiles = (W + 63) / 64;
        tfc_x3_pack_kernel<<<(unsigned)((ntiles * 512 + 255) / 256), 256, 0, stream>>>(
            weights[L], reinterpret_cast<__bf16*>(workspace), ntiles, 1, W, W, dims[L], 1);
        TFC_LAUNCH_CHECK("tfc_x3_pack_kernel");
        const long nbx = (n_rows + 255) / 256;
        tfc_radial_mlp_fwd_x3_kernel<<<(unsigned)nbx, 512, 0, stream>>>(a, reinterpret_cast<const uintx4*>(workspace));
        TFC_LAUNCH_CHECK("tfc_radial_mlp_fwd_x3_kernel");
        return TFC_OK;
    }
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
    // busiest CU owns fewer (two 128-edge or four 64-edge workgroups resident per CU); a tie goes to 64: twice
    // as many workgroups hide the latency of the gout stream better than the halved W_L traffic (L2-resident) is
    // worth (each kernel alone: config 3 126 -> 114 us, config 5 205 -> 179, config 4 1259 -> 1161)
    const long nb128 = (n_rows + 127) / 128, nb64 = (n_rows + 63) / 64;
    const long cost128 = (nb128 + 255) / 256 * 128, cost64 = (nb64 + 255) / 256 * 64;
    const bool small = forced ? forced == 64 : cost64 <= cost128;
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
