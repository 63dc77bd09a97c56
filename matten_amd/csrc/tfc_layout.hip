// Channel-contiguous copy of e3nn feature rows.
//
// e3nn stores a block mul x (l, p) of a feature row as [mul][2l+1] (m fastest), so the channel
// index u - the contraction index of o3.Linear and of the self-connection
// FullyConnectedTensorProduct (eigenn/nn/point_conv.py:56-61,130-135,159-164) - has stride 2l+1.
// The dense kernels stage their operands in 16-byte groups along the contraction index, which
// needs stride 1.  This pass writes block b as [2l+1][n_rows][mul] (u fastest):
//
//     out[off_b * n_rows + (m * n_rows + n) * mul_b + u] = x[n * d_row + off_b + u * dim_b + m]
//
// It moves a node-feature table once (a few MB, L2-resident) and turns every later gather of it
// into a coalesced stream.
#include "tfc_common.h"

#include <algorithm>

namespace {

constexpr int MAXB = 32;
struct Blocks {
    int n;
    tfc_ir_block b[MAXB];
};

// One thread per (row n, channel u) of a block: it reads the 2l+1 consecutive values of that copy
// (consecutive lanes read consecutive copies: one contiguous span per wave) and writes one value to
// each of the 2l+1 planes (consecutive lanes write consecutive u: 256 contiguous bytes per wave and
// plane).  blockIdx.y = block of the row.
template <int DIM>
__device__ __forceinline__ void ir_transpose_block(const tfc_ir_block b, const float* __restrict__ x, long n_rows,
                                                   int d_row, float* __restrict__ out) {
    const long total = n_rows * (long)b.mul;
    const int dim = DIM > 0 ? DIM : b.dim;
    float* ob = out + (long)b.off * n_rows;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        long n;
        int u;
        if (total < (1L << 31)) {
            const unsigned tt = (unsigned)t, q = tt / (unsigned)b.mul;
            n = q;
            u = (int)(tt - q * (unsigned)b.mul);
        } else {
            n = t / b.mul;
            u = (int)(t - n * b.mul);
        }
        const float* src = x + n * (long)d_row + b.off + (long)u * dim;
        if (DIM > 0) {
            float v[DIM > 0 ? DIM : 1];
#pragma unroll
            for (int m = 0; m < DIM; ++m) v[m] = src[m];
#pragma unroll
            for (int m = 0; m < DIM; ++m) ob[((long)m * n_rows + n) * b.mul + u] = v[m];
        } else {
            for (int m = 0; m < dim; ++m) ob[((long)m * n_rows + n) * b.mul + u] = src[m];
        }
    }
}

__global__ __launch_bounds__(256) void tfc_ir_transpose_kernel(const Blocks bl, const float* __restrict__ x,
                                                               long n_rows, int d_row, float* __restrict__ out) {
    const tfc_ir_block b = bl.b[blockIdx.y];
    switch (b.dim) {
        case 1: ir_transpose_block<1>(b, x, n_rows, d_row, out); break;
        case 3: ir_transpose_block<3>(b, x, n_rows, d_row, out); break;
        case 5: ir_transpose_block<5>(b, x, n_rows, d_row, out); break;
        case 7: ir_transpose_block<7>(b, x, n_rows, d_row, out); break;
        default: ir_transpose_block<0>(b, x, n_rows, d_row, out); break;
    }
}

// out[i] = sum_p part[p * numel + i], p ascending (fixed order)
__global__ __launch_bounds__(256) void tfc_sum_parts_kernel(const float* __restrict__ part, int n_parts, long numel,
                                                            float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long)gridDim.x * blockDim.x) {
        float s = part[i];
        for (int p = 1; p < n_parts; ++p) s += part[(long)p * numel + i];
        out[i] = s;
    }
}

// da[n, v] = inv_count[cls[n]] * sum_i part[i, cls[n], v], i ascending
__global__ __launch_bounds__(256) void tfc_class_rows_kernel(const float* __restrict__ part, int n_parts, int r, int d_a,
                                                             const float* __restrict__ inv_count,
                                                             const int64_t* __restrict__ cls, long n_rows,
                                                             float* __restrict__ da) {
    const long total = n_rows * (long)d_a;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / d_a;
        const int v = (int)(i - n * d_a);
        const long c = cls[n];
        float s = 0.f;
        if (c >= 0 && c < r) {
            const float* p = part + c * d_a + v;
            s = p[0];
            for (int k = 1; k < n_parts; ++k) s += p[(long)k * r * d_a];
            s *= inv_count[c];
        }
        da[i] = s;
    }
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- class table of a labelling with a known number of classes (tfc_class_table) --------------------------------
// work[0..r) = first row of each class (INT_MAX: none yet), work[r..2r) = rows in the class; integer atomics, so the
// result does not depend on the order in which the rows arrive
__global__ __launch_bounds__(64) void tfc_class_init_kernel(int* __restrict__ work, int r) {
    const int s = threadIdx.x + blockIdx.x * blockDim.x;
    if (s < r) {
        work[s] = 0x7fffffff;
        work[r + s] = 0;
    }
}

// (per workgroup the first rows and the counts are gathered in LDS - up to CLASS_LDS classes - and only then merged
// into `work`: a few thousand rows on a handful of classes would otherwise queue on as many global addresses)
constexpr int CLASS_LDS = 256;

__global__ __launch_bounds__(256) void tfc_class_scan_kernel(const int64_t* __restrict__ cls, long n_rows, int n_classes, int r,
                                                             int* __restrict__ work, float* __restrict__ onehot) {
    __shared__ int s_first[CLASS_LDS];
    __shared__ int s_count[CLASS_LDS];
    const bool local = n_classes <= CLASS_LDS;
    if (local) {
        for (int s = threadIdx.x; s < n_classes; s += blockDim.x) {
            s_first[s] = 0x7fffffff;
            s_count[s] = 0;
        }
        __syncthreads();
    }
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < n_rows; n += (long)gridDim.x * blockDim.x) {
        const long c = cls[n];
        const bool ok = c >= 0 && c < n_classes;
        float* row = onehot + n * r;
        for (int s = 0; s < r; s += 4) {
            floatx4 v;
            v[0] = (ok && c == s) ? 1.f : 0.f;
            v[1] = (ok && c == s + 1) ? 1.f : 0.f;
            v[2] = (ok && c == s + 2) ? 1.f : 0.f;
            v[3] = (ok && c == s + 3) ? 1.f : 0.f;
            *reinterpret_cast<floatx4*>(row + s) = v;
        }
        if (ok) {
            if (local) {
                atomicMin(s_first + c, (int)n);
                atomicAdd(s_count + c, 1);
            } else {
                atomicMin(work + c, (int)n);
                atomicAdd(work + r + c, 1);
            }
        }
    }
    if (local) {
        __syncthreads();
        for (int s = threadIdx.x; s < n_classes; s += blockDim.x) {
            if (s_count[s] > 0) {
                atomicMin(work + s, s_first[s]);
                atomicAdd(work + r + s, s_count[s]);
            }
        }
    }
}

__global__ __launch_bounds__(64) void tfc_class_finish_kernel(const int* __restrict__ work, int r, int64_t* __restrict__ rep_idx,
                                                              float* __restrict__ rep_mask, float* __restrict__ inv_count) {
    const int s = threadIdx.x + blockIdx.x * blockDim.x;
    if (s < r) {
        const int first = work[s], cnt = work[r + s];
        rep_idx[s] = cnt > 0 ? (int64_t)first : 0;
        rep_mask[s] = cnt > 0 ? 1.f : 0.f;
        inv_count[s] = 1.f / (float)(cnt > 0 ? cnt : 1);
    }
}

// ---- weight re-associations of the self connection's low-rank route (tfc_fctp_rows_*) ------------------------------
// With attributes that are rows of a small table R [r, d_a] (a[n] = R[cls[n]]) the self connection contracts over
// W'[u, s, w] = sum_v R[s, v] W[u, v, w].  W' and dW = R^T dW' are products with r = 4 .. rows / terms: as GEMMs they cost a
// launch of 13 us each at config 3 (94 % of a 64-row tile is padding), six per step; here they are one short kernel
// each (config 3: 13 -> 8 us).  Every sum runs in a fixed order.  (dR = dW' . W, r x mul_v outputs of mul_u mul_w terms,
// was written the same way and measured no faster than its split-K GEMM - 26 against 25 us, 133 against 92 at config
// 4: every workgroup re-reads all of dW' - and stays a GEMM.)
constexpr int ROWS_MAX_INS = 32;
struct RowsIns {
    int mu, mv, mw, a_off, w_off, p_off;  // p_off: offset of the instruction's block in W' / dW'
};
struct RowsArgs {
    int n, r, d_a;
    RowsIns q[ROWS_MAX_INS];
};

// W'[p_off + (u r + s) mw + w] = sum_v R[s, a_off + v] W[w_off + (u mv + v) mw + w];  grid (tiles of 64 (u, w), ins, r / 4)
// The four waves of a workgroup take a quarter of v each for the same 64 (u, w) pairs (16 - 32 loads per thread, all in
// flight, and four times as many workgroups as one thread per pair would give); wave 0 adds the quarters in order.
__global__ __launch_bounds__(256) void tfc_fctp_rows_w_kernel(const RowsArgs A, const float* __restrict__ rows,
                                                              const float* __restrict__ weight, float* __restrict__ wp) {
    const RowsIns q = A.q[blockIdx.y];
    const int s0 = 4 * blockIdx.z;
    __shared__ float Rl[4][256];
    __shared__ float part[4][4][64];
    for (int i = threadIdx.x; i < 4 * q.mv; i += blockDim.x) {
        const int k = i / q.mv, v = i - k * q.mv;
        Rl[k][v] = rows[(long)(s0 + k) * A.d_a + q.a_off + v];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_p = q.mu * q.mw;
    const int p = blockIdx.x * 64 + lane;
    const int pc = p < n_p ? p : n_p - 1;  // (lanes past the end shadow the last pair: valid addresses, nothing stored)
    const int u = pc / q.mw, w = pc - u * q.mw;
    const float* wr = weight + q.w_off + (long)u * q.mv * q.mw + w;
    const int per = (q.mv + 3) >> 2;
    const int v0 = wave * per, v1 = min(q.mv, v0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int v = v0;
    for (; v + 8 <= v1; v += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wr[(long)(v + j) * q.mw];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = fmaf(Rl[0][v + j], x[j], a0);
            a1 = fmaf(Rl[1][v + j], x[j], a1);
            a2 = fmaf(Rl[2][v + j], x[j], a2);
            a3 = fmaf(Rl[3][v + j], x[j], a3);
        }
    }
    for (; v < v1; ++v) {
        const float x = wr[(long)v * q.mw];
        a0 = fmaf(Rl[0][v], x, a0);
        a1 = fmaf(Rl[1][v], x, a1);
        a2 = fmaf(Rl[2][v], x, a2);
        a3 = fmaf(Rl[3][v], x, a3);
    }
    part[wave][0][lane] = a0;
    part[wave][1][lane] = a1;
    part[wave][2][lane] = a2;
    part[wave][3][lane] = a3;
    __syncthreads();
    if (wave == 0 && p < n_p) {
        float* o = wp + q.p_off + ((long)u * A.r + s0) * q.mw + w;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[(long)k * q.mw] = ((part[0][k][lane] + part[1][k][lane]) + part[2][k][lane]) + part[3][k][lane];
    }
}

// dW[w_off + (u mv + v) mw + w] = sum_s R[s, a_off + v] dW'[p_off + (u r + s) mw + w];  grid (tiles of (u, v, w), ins)
__global__ __launch_bounds__(256) void tfc_fctp_rows_dw_kernel(const RowsArgs A, const float* __restrict__ rows,
                                                               const float* __restrict__ dwp, float* __restrict__ dw) {
    const RowsIns q = A.q[blockIdx.y];
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;  // (32-bit: a weight block has fewer than 2^31 entries)
    if (p >= (unsigned)(q.mu * q.mv * q.mw)) return;
    const unsigned t = p / (unsigned)q.mw;
    const int w = (int)(p - t * (unsigned)q.mw);
    const int u = (int)(t / (unsigned)q.mv), v = (int)(t - (unsigned)u * (unsigned)q.mv);
    const float* g = dwp + q.p_off + (long)u * A.r * q.mw + w;
    const float* rv = rows + q.a_off + v;
    float acc = 0.f;
    int s = 0;
    for (; s + 4 <= A.r; s += 4) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = rv[(long)(s + j) * A.d_a];
            b[j] = g[(long)(s + j) * q.mw];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(a[j], b[j], acc);
    }
    for (; s < A.r; ++s) acc = fmaf(rv[(long)s * A.d_a], g[(long)s * q.mw], acc);
    dw[q.w_off + p] = acc;
}

}  // namespace

extern "C" int tfc_fctp_class_rows(const float* part, int32_t n_parts, int32_t r, int32_t d_a, const float* inv_count,
                                   const int64_t* cls, int64_t n_rows, float* da, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_parts > 0 && r > 0 && d_a > 0 && n_rows >= 0, TFC_EINVAL, "tfc_fctp_class_rows: bad size");
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(part && inv_count && cls && da, TFC_EINVAL, "tfc_fctp_class_rows: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d_a + 255) / 256, 4096);
    TfcProfScope prof("tfc_class_rows_kernel", stream);
    tfc_class_rows_kernel<<<grid, 256, 0, stream>>>(part, n_parts, r, d_a, inv_count, cls, (long)n_rows, da);
    TFC_LAUNCH_CHECK("tfc_class_rows_kernel");
    return TFC_OK;
}

namespace {
int rows_args(const tfc_fctp_ins* ins, int n_ins, int r, int d_a, long w_numel, RowsArgs* A, long* wp_numel, int* mv_max,
              long* uw_max, long* uvw_max, const char* who) {
    TFC_REQUIRE(ins && n_ins > 0 && n_ins <= ROWS_MAX_INS, TFC_EUNSUPPORTED, "%s: 1 .. %d instructions, got %d", who, ROWS_MAX_INS,
                n_ins);
    TFC_REQUIRE(r >= 4 && r % 4 == 0 && d_a > 0, TFC_EINVAL, "%s: r must be a positive multiple of 4", who);
    A->n = n_ins, A->r = r, A->d_a = d_a;
    long off = 0;
    *mv_max = 0, *uw_max = 0, *uvw_max = 0;
    for (int i = 0; i < n_ins; ++i) {
        const tfc_fctp_ins& q = ins[i];
        TFC_REQUIRE(q.mul_u > 0 && q.mul_v > 0 && q.mul_v <= 256 && q.mul_w > 0, TFC_EUNSUPPORTED,
                    "%s: instruction %d: multiplicities must be positive, second input at most 256", who, i);
        TFC_REQUIRE(q.a_off >= 0 && q.a_off + q.mul_v <= d_a, TFC_EINVAL, "%s: instruction %d reads outside the attribute row", who, i);
        TFC_REQUIRE(q.w_off >= 0 && (long)q.w_off + (long)q.mul_u * q.mul_v * q.mul_w <= w_numel, TFC_EINVAL,
                    "%s: instruction %d reads outside the weights", who, i);
        TFC_REQUIRE(off + (long)q.mul_u * r * q.mul_w < (1L << 31), TFC_EUNSUPPORTED, "%s: table too large", who);
        A->q[i] = RowsIns{q.mul_u, q.mul_v, q.mul_w, q.a_off, q.w_off, (int)off};
        off += (long)q.mul_u * r * q.mul_w;
        *mv_max = std::max(*mv_max, q.mul_v);
        *uw_max = std::max(*uw_max, (long)q.mul_u * q.mul_w);
        *uvw_max = std::max(*uvw_max, (long)q.mul_u * q.mul_v * q.mul_w);
    }
    *wp_numel = off;
    return TFC_OK;
}
}  // namespace

extern "C" int64_t tfc_fctp_rows_numel(const tfc_fctp_ins* ins, int32_t n_ins, int32_t r) {
    if (!ins || n_ins <= 0 || r <= 0) return -1;
    int64_t off = 0;
    for (int i = 0; i < n_ins; ++i) off += (int64_t)ins[i].mul_u * r * ins[i].mul_w;
    return off;
}

extern "C" int tfc_fctp_rows_w(const tfc_fctp_ins* ins, int32_t n_ins, const float* rows, int32_t r, int32_t d_a,
                               const float* weight, int64_t weight_numel, float* wp, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RowsArgs A;
    long n_wp, uw, uvw;
    int mv;
    const int rc = rows_args(ins, n_ins, r, d_a, weight_numel, &A, &n_wp, &mv, &uw, &uvw, "tfc_fctp_rows_w");
    if (rc) return rc;
    TFC_REQUIRE(rows && weight && wp, TFC_EINVAL, "tfc_fctp_rows_w: null tensor");
    TfcProfScope prof("gemm_fctp_lowrank_w", stream);
    tfc_fctp_rows_w_kernel<<<dim3((unsigned)((uw + 63) / 64), n_ins, r / 4), 256, 0, stream>>>(A, rows, weight, wp);
    TFC_LAUNCH_CHECK("tfc_fctp_rows_w_kernel");
    return TFC_OK;
}

extern "C" int tfc_fctp_rows_dw(const tfc_fctp_ins* ins, int32_t n_ins, const float* rows, int32_t r, int32_t d_a,
                                const float* dwp, int64_t weight_numel, float* dw, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RowsArgs A;
    long n_wp, uw, uvw;
    int mv;
    const int rc = rows_args(ins, n_ins, r, d_a, weight_numel, &A, &n_wp, &mv, &uw, &uvw, "tfc_fctp_rows_dw");
    if (rc) return rc;
    TFC_REQUIRE(rows && dwp && dw, TFC_EINVAL, "tfc_fctp_rows_dw: null tensor");
    TfcProfScope prof("gemm_fctp_lowrank_w", stream);
    tfc_fctp_rows_dw_kernel<<<dim3((unsigned)((uvw + 255) / 256), n_ins), 256, 0, stream>>>(A, rows, dwp, dw);
    TFC_LAUNCH_CHECK("tfc_fctp_rows_dw_kernel");
    return TFC_OK;
}

extern "C" int tfc_class_table(const int64_t* cls, int64_t n_rows, int32_t n_classes, int32_t r, int32_t* work, float* onehot,
                               int64_t* rep_idx, float* rep_mask, float* inv_count, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && n_rows < 0x7fffffffL && n_classes > 0 && r >= n_classes && r % 4 == 0, TFC_EINVAL,
                "tfc_class_table: need 0 < n_classes <= r, r a multiple of 4, fewer than 2^31 rows");
    TFC_REQUIRE(work && rep_idx && rep_mask && inv_count && (n_rows == 0 || (cls && onehot)), TFC_EINVAL,
                "tfc_class_table: null tensor");
    TfcProfScope prof("tfc_class_table", stream);
    tfc_class_init_kernel<<<(r + 63) / 64, 64, 0, stream>>>(work, r);
    TFC_LAUNCH_CHECK("tfc_class_init_kernel");
    if (n_rows > 0) {
        const int grid = (int)std::min<long>((n_rows + 255) / 256, 2048);
        tfc_class_scan_kernel<<<grid, 256, 0, stream>>>(cls, (long)n_rows, n_classes, r, work, onehot);
        TFC_LAUNCH_CHECK("tfc_class_scan_kernel");
    }
    tfc_class_finish_kernel<<<(r + 63) / 64, 64, 0, stream>>>(work, r, rep_idx, rep_mask, inv_count);
    TFC_LAUNCH_CHECK("tfc_class_finish_kernel");
    return TFC_OK;
}

// internal: zero-fill by a kernel (see tfc_common.h for why not a memset)
namespace {
__global__ __launch_bounds__(256) void tfc_zero_kernel(uint32_t* __restrict__ p, long n_words) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

int tfc_zero_async(void* p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return TFC_OK;
    TFC_REQUIRE(p && bytes % 4 == 0 && ((uintptr_t)p & 3) == 0, TFC_EINVAL, "tfc_zero_async: %zu bytes at %p", bytes, p);
    const long n_words = (long)(bytes / 4);
    const int grid = (int)std::min<long>((n_words + 255) / 256, 4096);
    tfc_zero_kernel<<<grid, 256, 0, stream>>>((uint32_t*)p, n_words);
    TFC_LAUNCH_CHECK("tfc_zero_kernel");
    return TFC_OK;
}

// internal (tfc_dense.cpp): ordered sum of n_parts slabs of numel floats
int tfc_sum_parts(const float* part, int n_parts, long numel, float* out, hipStream_t stream) {
    if (numel <= 0 || n_parts <= 0) return TFC_OK;
    const int grid = (int)std::min<long>((numel + 255) / 256, 4096);
    tfc_sum_parts_kernel<<<grid, 256, 0, stream>>>(part, n_parts, numel, out);
    TFC_LAUNCH_CHECK("tfc_sum_parts_kernel");
    return TFC_OK;
}

extern "C" int tfc_ir_transpose(const float* x, int64_t n_rows, int32_t d_row, const tfc_ir_block* blocks,
                                int32_t n_blocks, float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_row > 0, TFC_EINVAL, "tfc_ir_transpose: bad size");
    TFC_REQUIRE(n_blocks >= 0 && n_blocks <= MAXB && (n_blocks == 0 || blocks), TFC_EINVAL,
                "tfc_ir_transpose: %d blocks (at most %d)", n_blocks, MAXB);
    Blocks bl;
    bl.n = n_blocks;
    for (int i = 0; i < n_blocks; ++i) {
        const tfc_ir_block& b = blocks[i];
        TFC_REQUIRE(b.off >= 0 && b.mul > 0 && b.dim > 0 && (long)b.off + (long)b.mul * b.dim <= d_row, TFC_EINVAL,
                    "tfc_ir_transpose: block %d outside the row", i);
        bl.b[i] = b;
    }
    if (n_rows == 0 || n_blocks == 0) return TFC_OK;
    TFC_REQUIRE(x && out, TFC_EINVAL, "tfc_ir_transpose: null tensor");
    int max_mul = 1;
    for (int i = 0; i < n_blocks; ++i) max_mul = std::max(max_mul, bl.b[i].mul);
    const dim3 grid((unsigned)std::min<long>((n_rows * (long)max_mul + 255) / 256, 4096), (unsigned)n_blocks);
    TfcProfScope prof("tfc_ir_transpose_kernel", stream);
    tfc_ir_transpose_kernel<<<grid, 256, 0, stream>>>(bl, x, (long)n_rows, d_row, out);
    TFC_LAUNCH_CHECK("tfc_ir_transpose_kernel");
    return TFC_OK;
}
