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

__global__ __launch_bounds__(256) void tfc_ir_transpose_kernel(const Blocks bl, const float* __restrict__ x,
                                                               long n_rows, int d_row, float* __restrict__ out) {
    const long total = n_rows * (long)d_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long n = i / d_row;
        const int c = (int)(i - n * d_row);
        for (int s = 0; s < bl.n; ++s) {
            const tfc_ir_block b = bl.b[s];
            const int idx = c - b.off;
            if (idx >= 0 && idx < b.mul * b.dim) {
                const int u = idx / b.dim, m = idx - u * b.dim;
                out[(long)b.off * n_rows + ((long)m * n_rows + n) * b.mul + u] = x[i];
                break;
            }
        }
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

}  // namespace

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
    const int grid = (int)std::min<long>((n_rows * (long)d_row + 255) / 256, 8192);
    TfcProfScope prof("tfc_ir_transpose_kernel", stream);
    tfc_ir_transpose_kernel<<<grid, 256, 0, stream>>>(bl, x, (long)n_rows, d_row, out);
    TFC_LAUNCH_CHECK("tfc_ir_transpose_kernel");
    return TFC_OK;
}
