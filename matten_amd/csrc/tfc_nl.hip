// Periodic neighbour list of a batch of structures on the GPU, with the semantics of the
// reference's neighbor_list_and_relative_vec (eigenn/data/data.py:400-528): every (i, j, S) with
// |pos[j] - pos[i] + S @ cell| < r_max (strict, fp64), self images kept, (i == j, S == 0) dropped.
// The edges come out in the canonical order (i, j, Sx, Sy, Sz) of matten_amd/data/neighbor.py, so
// the list is reproducible (ASE's own order inside equal i is not) and needs no sort: a node's
// candidates (j, S) are walked in exactly that order by one wave, first to count, then to write at
// ballot-prefix positions.  All distance arithmetic is fp64 without contraction, in the operation
// order of the host implementation, so the cutoff decisions are the same.
//
// tfc_nl_ordering then derives both edge groupings the convolution reduces over (by sender and by
// receiver) from the symmetry of the list, (i, j, S) <-> (j, i, -S), without the sort of
// tfc_csr_build.
#include "tfc_common.h"

#include <algorithm>

namespace {

constexpr int NL_MAX_IMAGES = 2048;  // per-wave LDS list of surviving images (4 waves: 32 KB); longer tables are not pruned

struct NlArgs {
    const double* pos;         // [N,3]
    const int* node_graph;     // [N]
    const int* graph_ptr;      // [G+1] first node of each graph
    const int* shift_ptr;      // [G+1] first row of each graph's image table
    const int* shifts;         // [S_total,3] integer images, ascending lexicographic per graph
    const double* shift_vec;   // [S_total,3] = shifts @ cell, fp64 (host-computed)
    const double* graph_box;   // [G,6] min xyz, max xyz of each graph's positions, or null (no pruning)
    long n_nodes;
    double r_max, r2_hi;
};

__device__ __forceinline__ bool nl_keep(const NlArgs& a, double xi, double yi, double zi, int j, int srow, bool same,
                                       bool valid) {
#pragma clang fp contract(off)
    if (!valid) return false;
    const double* pj = a.pos + 3L * j;
    const double* sv = a.shift_vec + 3L * srow;
    // d = (pos[j] + shift_vec) - pos[i], |d|^2 = (dx^2 + dy^2) + dz^2 : no fused multiply-add
    const double dx = __dsub_rn(__dadd_rn(pj[0], sv[0]), xi);
    const double dy = __dsub_rn(__dadd_rn(pj[1], sv[1]), yi);
    const double dz = __dsub_rn(__dadd_rn(pj[2], sv[2]), zi);
    const double d2 = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    if (!(d2 < a.r2_hi)) return false;
    if (!(__dsqrt_rn(d2) < a.r_max)) return false;  // the reference compares distances, not squares
    if (same) {
        const int* s = a.shifts + 3L * srow;
        if (s[0] == 0 && s[1] == 0 && s[2] == 0) return false;
    }
    return true;
}

// FILL = false: counts[i] = number of edges with sender i.
// FILL = true : writes the edges of sender i at row_ptr[i] ... in candidate order.
template <bool FILL>
__global__ __launch_bounds__(256) void tfc_nl_kernel(const NlArgs a, int* __restrict__ counts,
                                                     const int* __restrict__ row_ptr, long n_edges,
                                                     int64_t* __restrict__ edge_index, float* __restrict__ edge_shift) {
    __shared__ int img[4][NL_MAX_IMAGES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wave0 = (long)blockIdx.x * 4 + wv;
    for (long i = wave0; i < a.n_nodes; i += (long)gridDim.x * 4) {
        const int g = a.node_graph[i];
        const int n0 = a.graph_ptr[g], ng = a.graph_ptr[g + 1] - n0;
        const int s0 = a.shift_ptr[g], S = a.shift_ptr[g + 1] - s0;
        const double xi = a.pos[3 * i], yi = a.pos[3 * i + 1], zi = a.pos[3 * i + 2];
        if (S <= 0 || ng <= 0) continue;  // rejected on the host; counts stay 0
        // Images that cannot hold a neighbour of i are dropped first: the box of the structure's
        // positions, moved by the image vector, must come within r_max of pos[i] (with a relative margin
        // far above rounding, so the test only ever removes candidates the exact test would reject).
        // The survivors keep their ascending order in this wave's LDS list.
        const bool pruned = a.graph_box != nullptr && S <= NL_MAX_IMAGES;
        int Sp = S;
        if (pruned) {
            const double* bx = a.graph_box + 6L * g;
            const double r2_box = a.r2_hi * (1.0 + 1e-9);
            int cnt = 0;
            for (int t0 = 0; t0 < S; t0 += 64) {
                const int t = t0 + lane;
                bool ok = false;
                if (t < S) {
                    const double* sv = a.shift_vec + 3L * (s0 + t);
                    const double ex = fmax(fmax(bx[0] + sv[0] - xi, xi - (bx[3] + sv[0])), 0.0);
                    const double ey = fmax(fmax(bx[1] + sv[1] - yi, yi - (bx[4] + sv[1])), 0.0);
                    const double ez = fmax(fmax(bx[2] + sv[2] - zi, zi - (bx[5] + sv[2])), 0.0);
                    ok = ex * ex + ey * ey + ez * ez <= r2_box;
                }
                const unsigned long long m = __ballot(ok);
                if (ok) img[wv][cnt + __popcll(m & ((1ULL << lane) - 1ULL))] = t;
                cnt += __popcll(m);
            }
            Sp = cnt;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (Sp == 0) {
            if (!FILL && lane == 0) counts[i + 1] = 0;
            continue;
        }
        const long total = (long)ng * Sp;
        // candidate c = j_local * Sp + t (t-th surviving image); (j, t) advance by 64 candidates per iteration
        int j = lane / Sp, s = lane - j * Sp;
        const int dj = 64 / Sp, ds = 64 - dj * Sp;
        int running = 0;
        const int base = FILL ? row_ptr[i] : 0;
        for (long c0 = 0; c0 < total; c0 += 64) {
            const bool valid = c0 + lane < total;
            const int srow = s0 + (valid ? (pruned ? img[wv][s] : s) : 0);
            const bool keep = nl_keep(a, xi, yi, zi, n0 + (valid ? j : 0), srow, n0 + j == i, valid);
            const unsigned long long mask = __ballot(keep);
            if (FILL) {
                if (keep) {
                    const long p = (long)base + running + __popcll(mask & ((1ULL << lane) - 1ULL));
                    if (p < n_edges) {
                        edge_index[p] = i;
                        edge_index[n_edges + p] = n0 + j;
                        const int* sh = a.shifts + 3L * srow;
                        edge_shift[3 * p] = (float)sh[0];
                        edge_shift[3 * p + 1] = (float)sh[1];
                        edge_shift[3 * p + 2] = (float)sh[2];
                    }
                }
            }
            running += __popcll(mask);
            j += dj;
            s += ds;
            if (s >= Sp) {
                s -= Sp;
                ++j;
            }
        }
        if (!FILL && lane == 0) counts[i + 1] = running;
    }
}

// single-block inclusive scan of row_ptr[1..n] (row_ptr[0] = 0 is written here too)
__global__ __launch_bounds__(1024) void tfc_nl_scan_kernel(int* __restrict__ row_ptr, long n_nodes) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry_s = 0;
        row_ptr[0] = 0;
    }
    __syncthreads();
    for (long base = 0; base < n_nodes; base += 1024) {
        const long i = base + threadIdx.x;
        const int v = (i < n_nodes) ? row_ptr[i + 1] : 0;
        int s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        if (lane == 63) wave_tot[wave] = s;
        __syncthreads();
        int off = carry_s;
        for (int w2 = 0; w2 < wave; ++w2) off += wave_tot[w2];
        if (i < n_nodes) row_ptr[i + 1] = off + s;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + s;
        __syncthreads();
    }
}

// One thread per edge k = (r -> i, S), the t-th edge of the group of r's edges that go to i.  Its
// reverse (i -> r, -S) is the (size - 1 - t)-th edge of the group (i -> r) in row i (negation
// reverses the lexicographic order of the images), and the edges INTO r, in ascending edge id, are
// the reverses of r's outgoing edges taken group by group (ascending i), each group backwards.
__global__ __launch_bounds__(256) void tfc_nl_ordering_kernel(const int64_t* __restrict__ edge_index,
                                                              const int* __restrict__ row_ptr, long n_edges,
                                                              long n_nodes, int* __restrict__ src_eid,
                                                              int* __restrict__ src_nbr, int* __restrict__ dst_eid,
                                                              int* __restrict__ dst_nbr, int* __restrict__ status) {
    const int64_t* snd = edge_index;
    const int64_t* rcv = edge_index + n_edges;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n_edges; k += (long)gridDim.x * blockDim.x) {
        const int64_t r64 = snd[k], i64 = rcv[k];
        if (r64 < 0 || r64 >= n_nodes || i64 < 0 || i64 >= n_nodes) {
            *status = 2;
            continue;
        }
        const int r = (int)r64, i = (int)i64;
        src_eid[k] = (int)k;
        src_nbr[k] = i;
        // group of k inside row r
        const int rb = row_ptr[r], re = row_ptr[r + 1];
        int g0 = (int)k, g1 = (int)k + 1;
        while (g0 > rb && rcv[g0 - 1] == i) --g0;
        while (g1 < re && rcv[g1] == i) ++g1;
        // group (i -> r) inside row i: lower bound of r among the receivers of row i
        int lo = row_ptr[i], hi = row_ptr[i + 1];
        const int ie = hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (rcv[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        int h1 = lo;
        while (h1 < ie && rcv[h1] == r) ++h1;
        if (h1 - lo != g1 - g0) {  // the list is not symmetric (a pair within rounding of the cutoff)
            *status = 1;
            continue;
        }
        const int partner = lo + (g1 - 1 - (int)k);
        const int slot = g0 + (g1 - 1 - (int)k);  // position among the edges into r
        dst_eid[slot] = partner;
        dst_nbr[slot] = i;
    }
}

int nl_check(const double* pos, const int32_t* node_graph, const int32_t* graph_ptr, const int32_t* shift_ptr,
             const int32_t* shifts, const double* shift_vec, int64_t n_nodes, int64_t n_graphs, double r_max,
             const char* who) {
    TFC_REQUIRE(n_nodes >= 0 && n_graphs >= 0, TFC_EINVAL, "%s: negative size", who);
    TFC_REQUIRE(n_nodes < (1LL << 31) - 1, TFC_EUNSUPPORTED, "%s: more than 2^31 nodes", who);
    TFC_REQUIRE(r_max > 0.0, TFC_EINVAL, "%s: r_max must be positive", who);
    TFC_REQUIRE(n_nodes == 0 || (pos && node_graph && graph_ptr && shift_ptr && shifts && shift_vec), TFC_EINVAL,
                "%s: null tensor", who);
    return TFC_OK;
}

}  // namespace

extern "C" int tfc_nl_count(const double* pos, const int32_t* node_graph, const int32_t* graph_ptr,
                            const int32_t* shift_ptr, const int32_t* shifts, const double* shift_vec,
                            const double* graph_box, int64_t n_nodes, int64_t n_graphs, double r_max,
                            int32_t* row_ptr, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int rc = nl_check(pos, node_graph, graph_ptr, shift_ptr, shifts, shift_vec, n_nodes, n_graphs, r_max,
                            "tfc_nl_count");
    if (rc) return rc;
    TFC_REQUIRE(row_ptr, TFC_EINVAL, "tfc_nl_count: null output");
    { const int zrc = tfc_zero_async(row_ptr, (size_t)(n_nodes + 1) * sizeof(int32_t), stream); if (zrc != TFC_OK) return zrc; }
    if (n_nodes == 0) return TFC_OK;
    const NlArgs a{pos, node_graph, graph_ptr, shift_ptr, shifts, shift_vec, graph_box, (long)n_nodes, r_max,
                   r_max * r_max * (1.0 + 1e-12)};
    const int grid = (int)std::min<long>((n_nodes + 3) / 4, 1 << 16);
    {
        TfcProfScope prof("tfc_nl_count_kernel", stream);
        tfc_nl_kernel<false><<<grid, 256, 0, stream>>>(a, row_ptr, nullptr, 0, nullptr, nullptr);
        TFC_LAUNCH_CHECK("tfc_nl_kernel<count>");
    }
    tfc_nl_scan_kernel<<<1, 1024, 0, stream>>>(row_ptr, (long)n_nodes);
    TFC_LAUNCH_CHECK("tfc_nl_scan_kernel");
    return TFC_OK;
}

extern "C" int tfc_nl_fill(const double* pos, const int32_t* node_graph, const int32_t* graph_ptr,
                           const int32_t* shift_ptr, const int32_t* shifts, const double* shift_vec,
                           const double* graph_box, int64_t n_nodes, int64_t n_graphs, double r_max,
                           const int32_t* row_ptr, int64_t n_edges,
                           int64_t* edge_index, float* edge_shift, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int rc = nl_check(pos, node_graph, graph_ptr, shift_ptr, shifts, shift_vec, n_nodes, n_graphs, r_max,
                            "tfc_nl_fill");
    if (rc) return rc;
    TFC_REQUIRE(n_edges >= 0 && n_edges < (1LL << 31) - 1, TFC_EUNSUPPORTED, "tfc_nl_fill: bad edge count");
    if (n_nodes == 0 || n_edges == 0) return TFC_OK;
    TFC_REQUIRE(row_ptr && edge_index && edge_shift, TFC_EINVAL, "tfc_nl_fill: null tensor");
    const NlArgs a{pos, node_graph, graph_ptr, shift_ptr, shifts, shift_vec, graph_box, (long)n_nodes, r_max,
                   r_max * r_max * (1.0 + 1e-12)};
    const int grid = (int)std::min<long>((n_nodes + 3) / 4, 1 << 16);
    TfcProfScope prof("tfc_nl_fill_kernel", stream);
    tfc_nl_kernel<true><<<grid, 256, 0, stream>>>(a, nullptr, row_ptr, (long)n_edges, edge_index, edge_shift);
    TFC_LAUNCH_CHECK("tfc_nl_kernel<fill>");
    return TFC_OK;
}

extern "C" int tfc_nl_ordering(const int64_t* edge_index, const int32_t* row_ptr, int64_t n_edges, int64_t n_nodes,
                               int32_t* src_eid, int32_t* src_nbr, int32_t* dst_eid, int32_t* dst_nbr, int32_t* status,
                               tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_edges >= 0 && n_nodes >= 0, TFC_EINVAL, "tfc_nl_ordering: negative size");
    TFC_REQUIRE(n_edges < (1LL << 31) - 1 && n_nodes < (1LL << 31) - 1, TFC_EUNSUPPORTED,
                "tfc_nl_ordering: more than 2^31 nodes or edges");
    TFC_REQUIRE(status, TFC_EINVAL, "tfc_nl_ordering: null status");
    { const int zrc = tfc_zero_async(status, sizeof(int32_t), stream); if (zrc != TFC_OK) return zrc; }
    if (n_edges == 0) return TFC_OK;
    TFC_REQUIRE(edge_index && row_ptr && src_eid && src_nbr && dst_eid && dst_nbr, TFC_EINVAL,
                "tfc_nl_ordering: null tensor");
    const int grid = (int)std::min<long>((n_edges + 255) / 256, 8192);
    TfcProfScope prof("tfc_nl_ordering_kernel", stream);
    tfc_nl_ordering_kernel<<<grid, 256, 0, stream>>>(edge_index, row_ptr, (long)n_edges, (long)n_nodes, src_eid,
                                                     src_nbr, dst_eid, dst_nbr, status);
    TFC_LAUNCH_CHECK("tfc_nl_ordering_kernel");
    return TFC_OK;
}
