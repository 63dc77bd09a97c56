// Edge ordering: group the edges by one row of edge_index so that the convolution reduces a
// node's edges in registers (replaces the atomics of scatter_add_, eigenn/nn/utils.py:351-357).
// Inside a node the edges keep ascending original id, so every run sums in the same order.
#include "tfc_common.h"

#include <algorithm>
#include <mutex>

// Scratch of tfc_csr_build_noalloc (cursors + the unsorted edge queue), one copy per GPU the library
// runs on, held until the process ends.  In host code this name is only a stand-in for the symbol:
// the device address is asked for once per device (csr_static_scratch) and used from then on.
static __device__ __align__(256) int32_t tfc_csr_scratch[TFC_CSR_NOALLOC_MAX_INTS];

__global__ __launch_bounds__(256) void tfc_csr_count_kernel(const int64_t* __restrict__ key,
                                                            const int64_t* __restrict__ other, long n_edges,
                                                            long n_nodes, int* __restrict__ row_ptr,
                                                            int* __restrict__ status) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (long)gridDim.x * blockDim.x) {
        const int64_t k = key[e], o = other[e];
        if (k < 0 || k >= n_nodes || o < 0 || o >= n_nodes) {
            *status = 1;
            continue;
        }
        atomicAdd(&row_ptr[k + 1], 1);
    }
}

// single-block inclusive scan of row_ptr[1..n]; also seeds the fill cursors
__global__ __launch_bounds__(1024) void tfc_csr_scan_kernel(int* __restrict__ row_ptr, int* __restrict__ cursor,
                                                            long n_nodes) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long base = 0; base < n_nodes; base += 1024) {
        const long i = base + threadIdx.x;
        const int v = (i < n_nodes) ? row_ptr[i + 1] : 0;
        int s = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        if (lane == 63) wave_tot[wave] = s;
        __syncthreads();
        int off = carry_s;
        for (int w2 = 0; w2 < wave; ++w2) off += wave_tot[w2];
        if (i < n_nodes) {
            row_ptr[i + 1] = off + s;
            cursor[i] = off + s - v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + s;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void tfc_csr_fill_kernel(const int64_t* __restrict__ key,
                                                           const int64_t* __restrict__ other, long n_edges,
                                                           long n_nodes, int* __restrict__ cursor,
                                                           int* __restrict__ tmp) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (long)gridDim.x * blockDim.x) {
        const int64_t k = key[e], o = other[e];
        if (k < 0 || k >= n_nodes || o < 0 || o >= n_nodes) continue;
        const int pos = atomicAdd(&cursor[k], 1);
        tmp[pos] = (int)e;
    }
}

// one wave per node: rank-sort the node's edge ids (ascending) from tmp into eid, fetch nbr
__global__ __launch_bounds__(256) void tfc_csr_sort_kernel(const int* __restrict__ row_ptr,
                                                           const int* __restrict__ tmp,
                                                           const int64_t* __restrict__ other, long n_nodes,
                                                           int* __restrict__ eid, int* __restrict__ nbr) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long n = wave0; n < n_nodes; n += (long)gridDim.x * 4) {
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        for (int i = beg + lane; i < end; i += 64) {
            const int v = tmp[i];
            int rank = 0;
            for (int j = beg; j < end; ++j) rank += (tmp[j] < v) ? 1 : 0;
            eid[beg + rank] = v;
            nbr[beg + rank] = (int)other[v];
        }
    }
}

// out[n][c] = sum over the rows of node n (fixed order) of src[eid[k]][c]
__global__ __launch_bounds__(256) void tfc_segment_sum_kernel(const float* __restrict__ src,
                                                              const int* __restrict__ row_ptr,
                                                              const int* __restrict__ eid, long n_nodes, int d,
                                                              float* __restrict__ out) {
    const int chunks = (d + 255) / 256;
    for (long b = blockIdx.x; b < n_nodes * chunks; b += gridDim.x) {
        const long n = b / chunks;
        const int c = (int)(b - n * chunks) * 256 + threadIdx.x;
        if (c >= d) continue;
        float s = 0.f;
        for (int k = row_ptr[n]; k < row_ptr[n + 1]; ++k) s += src[(long)eid[k] * d + c];
        out[n * d + c] = s;
    }
}

__global__ __launch_bounds__(256) void tfc_gather_rows_kernel(const float* __restrict__ src,
                                                              const int64_t* __restrict__ index, long n_rows,
                                                              long n_src, int d, float* __restrict__ out) {
    const long total = n_rows * (long)d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / d;
        const int c = (int)(i - r * d);
        const int64_t s = index[r];
        out[i] = (s >= 0 && s < n_src) ? src[s * d + c] : 0.f;
    }
}

extern "C" int tfc_segment_sum(const float* src, const int32_t* row_ptr, const int32_t* eid, int64_t n_nodes,
                               int32_t d, float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_nodes >= 0 && d > 0, TFC_EINVAL, "tfc_segment_sum: bad size");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(row_ptr && out, TFC_EINVAL, "tfc_segment_sum: null tensor");
    const long blocks = n_nodes * ((d + 255) / 256);
    tfc_segment_sum_kernel<<<(unsigned)std::min<long>(blocks, 1 << 20), 256, 0, stream>>>(src, row_ptr, eid,
                                                                                         (long)n_nodes, d, out);
    TFC_LAUNCH_CHECK("tfc_segment_sum_kernel");
    return TFC_OK;
}

extern "C" int tfc_gather_rows(const float* src, const int64_t* index, int64_t n_rows, int64_t n_src, int32_t d,
                               float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && n_src >= 0 && d > 0, TFC_EINVAL, "tfc_gather_rows: bad size");
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(src && index && out, TFC_EINVAL, "tfc_gather_rows: null tensor");
    const long total = n_rows * (long)d;
    tfc_gather_rows_kernel<<<(unsigned)std::min<long>((total + 255) / 256, 8192), 256, 0, stream>>>(
        src, index, (long)n_rows, (long)n_src, d, out);
    TFC_LAUNCH_CHECK("tfc_gather_rows_kernel");
    return TFC_OK;
}

extern "C" int64_t tfc_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    if (n_nodes < 0 || n_edges < 0) return -1;
    return (n_nodes + n_edges + 2) * (int64_t)sizeof(int32_t);
}

static int csr_build_impl(const int64_t* key, const int64_t* other, int64_t n_edges, int64_t n_nodes,
                          int32_t* row_ptr, int32_t* eid, int32_t* nbr, int32_t* status, int* workspace,
                          hipStream_t stream) {
    int* cursor = workspace;
    int* tmp = cursor + n_nodes + 1;
    { const int zrc = tfc_zero_async(row_ptr, (size_t)(n_nodes + 1) * sizeof(int32_t), stream); if (zrc != TFC_OK) return zrc; }
    { const int zrc = tfc_zero_async(status, sizeof(int32_t), stream); if (zrc != TFC_OK) return zrc; }
    if (n_edges == 0 || n_nodes == 0) return TFC_OK;
    const int eg = (int)std::min<long>((n_edges + 255) / 256, 4096);
    tfc_csr_count_kernel<<<eg, 256, 0, stream>>>(key, other, (long)n_edges, (long)n_nodes, row_ptr, status);
    TFC_LAUNCH_CHECK("tfc_csr_count_kernel");
    tfc_csr_scan_kernel<<<1, 1024, 0, stream>>>(row_ptr, cursor, (long)n_nodes);
    TFC_LAUNCH_CHECK("tfc_csr_scan_kernel");
    tfc_csr_fill_kernel<<<eg, 256, 0, stream>>>(key, other, (long)n_edges, (long)n_nodes, cursor, tmp);
    TFC_LAUNCH_CHECK("tfc_csr_fill_kernel");
    const int ng = (int)std::min<long>((n_nodes + 3) / 4, 8192);
    tfc_csr_sort_kernel<<<ng, 256, 0, stream>>>(row_ptr, tmp, other, (long)n_nodes, eid, nbr);
    TFC_LAUNCH_CHECK("tfc_csr_sort_kernel");
    return TFC_OK;
}

extern "C" int tfc_csr_build(const int64_t* key, const int64_t* other, int64_t n_edges, int64_t n_nodes,
                             int32_t* row_ptr, int32_t* eid, int32_t* nbr, int32_t* status, void* workspace,
                             tfc_stream_t stream_) {
    TFC_REQUIRE(n_edges >= 0 && n_nodes >= 0, TFC_EINVAL, "tfc_csr_build: negative size");
    TFC_REQUIRE(n_edges < (1LL << 31) - 1 && n_nodes < (1LL << 31) - 1, TFC_EUNSUPPORTED,
                "tfc_csr_build: more than 2^31 nodes or edges");
    TFC_REQUIRE(row_ptr && status && workspace, TFC_EINVAL, "tfc_csr_build: null output");
    TFC_REQUIRE(n_edges == 0 || (key && other && eid && nbr), TFC_EINVAL, "tfc_csr_build: null tensor");
    return csr_build_impl(key, other, n_edges, n_nodes, row_ptr, eid, nbr, status, (int*)workspace,
                          (hipStream_t)stream_);
}

// Device address of this GPU's copy of tfc_csr_scratch: looked up on the first call on a device (with that
// device current), kept per device afterwards.
static int csr_static_scratch(int32_t** out) {
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static int32_t* ptrs[kMaxDevices] = {};
    int dev = -1;
    TFC_HIP(hipGetDevice(&dev));
    TFC_REQUIRE(dev >= 0 && dev < kMaxDevices, TFC_EUNSUPPORTED, "tfc_csr_build_noalloc: device index %d", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void* p = nullptr;
        TFC_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(tfc_csr_scratch)));
        TFC_REQUIRE(p, TFC_ELAUNCH, "tfc_csr_build_noalloc: no device address for the scratch array");
        ptrs[dev] = (int32_t*)p;
    }
    *out = ptrs[dev];
    return TFC_OK;
}

extern "C" int tfc_csr_build_noalloc(const int64_t* key, const int64_t* other, int64_t n_edges, int64_t n_nodes,
                                     int32_t* row_ptr, int32_t* eid, int32_t* nbr, int32_t* status,
                                     tfc_stream_t stream_) {
    TFC_REQUIRE(n_edges >= 0 && n_nodes >= 0, TFC_EINVAL, "tfc_csr_build_noalloc: negative size");
    TFC_REQUIRE(n_nodes + n_edges + 2 <= (int64_t)TFC_CSR_NOALLOC_MAX_INTS, TFC_EUNSUPPORTED,
                "tfc_csr_build_noalloc: %lld nodes + %lld edges + 2 exceed the static scratch of %d entries "
                "(use tfc_csr_build with a caller workspace)",
                (long long)n_nodes, (long long)n_edges, (int)TFC_CSR_NOALLOC_MAX_INTS);
    TFC_REQUIRE(row_ptr && status, TFC_EINVAL, "tfc_csr_build_noalloc: null output");
    TFC_REQUIRE(n_edges == 0 || (key && other && eid && nbr), TFC_EINVAL, "tfc_csr_build_noalloc: null tensor");
    int32_t* scratch = nullptr;
    const int rc = csr_static_scratch(&scratch);
    if (rc) return rc;
    // nothing in the scratch has to start at zero: the scan seeds every cursor, the fill writes every queue
    // slot the sort reads
    return csr_build_impl(key, other, n_edges, n_nodes, row_ptr, eid, nbr, status, scratch, (hipStream_t)stream_);
}
