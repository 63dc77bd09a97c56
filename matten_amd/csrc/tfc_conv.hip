// Fused gather + 'uvu' tensor product + per-node sum, forward and backward, for gfx950.
//
// Replaces the chain  node_feats[edge_src] -> o3.TensorProduct('uvu', per-edge weights) ->
// scatter_add -> /sqrt(avg_num_neighbors)  of the reference (eigenn/nn/point_conv.py:82,
// eigenn/nn/utils.py:265, eigenn/nn/point_conv.py:173-178) and what autograd derives from it.
//
// Data flow:
//   * edges are grouped by their centre node (receiver for forward, sender for backward) by
//     tfc_csr_build; one 64-lane wave owns one (centre node, work item) and keeps the sums of
//     that item in registers over all edges of the node: no atomics, fixed summation order.
//   * lane = channel u.  Everything that does not depend on u is wave-uniform and is kept out
//     of the vector registers: the edge ids come in through scalar loads, and the contraction
//     of the Clebsch-Gordan tensor with the edge's spherical harmonics,
//         M_e[m1][m3] = sum_m2 C[m1][m2][m3] * y[e][m2],
//     is computed once per edge by a small pre-pass and then read through the scalar cache,
//     so the inner loop is (2l1+1)(2l3+1) v_fma with an SGPR operand per edge and path.
//   * HBM traffic is the per-edge weight row (read once, fully coalesced: 256 B per wave and
//     path); neighbour rows are gathered from L2 / Infinity Cache.
#include "tfc_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <tuple>
#include <vector>

namespace {

struct PathDev {
    int l1, l2, l3, mul;
    int x_off, y_off, out_off, w_off;
    int m_off;   // offset of the (l1,l2,l3,y_off) block in an M row
    int cg_off;  // offset of the dense C[m1][m2][m3] table of that triple
    int py_off;  // first partial-dy slot of this path (chunk 0)
    float coeff;
};

struct FwdItem {
    int path, chunk;
};

struct BwdItem {
    int grp_start, grp_count;  // range in grp_paths: the paths reading this in1 block
    int chunk, l1;
    int x_off, mul;
};

struct MEnt {
    int cg_row;  // offset in cgm of the n2 coefficients of this (m1, m3)
    int y_off;
    int n2;  // 0 for padding entries
};

}  // namespace

struct tfc_conv_plan {
    int n_paths, d_in, d_sh, d_out, w_numel;
    int msz;  // floats per edge of the M buffer (multiple of 4)
    int py;   // floats per edge of the partial-dy buffer
    int n_fwd_items, n_bwd_items;
    int dx_covered;  // every column of an input row belongs to an in1 block some path reads (dx needs no zero fill)
    int lmax;  // largest l1, l2 or l3 of any path
    long m_sig;  // hash of the layout of an M row (the list of (l1, l2, l3, y_off) blocks)
    int device;
    PathDev* paths;
    FwdItem* fwd_items;
    BwdItem* bwd_items;
    int* grp_paths;
    MEnt* ments;
    float* cgm;  // per (m1,m3): C[m1][0..n2)[m3]
    float* cgd;  // dense C[m1][m2][m3]
    int* ylist_ptr;  // [d_sh + 1]
    int* ylist;      // partial-dy columns contributing to each edge-attribute column
    void* blob;
    int* tout;  // [n_paths][3] (block off, block mul, first channel) of each path's output in the
                // channel-contiguous copy (tfc_conv_plan_set_out_blocks), or null
};

// ------------------------------------------------------------------------------------------
// M pre-pass: M[e][j] for edge e (the caller's edge order) and M-row entry j.  One M serves the forward (edges
// grouped by receiver) and the backward (grouped by sender): both kernels look a row up by edge id.
// ------------------------------------------------------------------------------------------
// One thread per M-row entry j: its descriptor and Clebsch-Gordan coefficients are read once and stay in
// registers; the workgroup then walks the edges in groups of EPB, whose harmonics it stages in LDS (one contiguous
// read), and every thread writes its entry of each of those edges - consecutive lanes, consecutive addresses.
// (The first version took one thread per (edge, j): a 64-bit division and 3 + 2 n2 loads for every element.)
constexpr int M_EPB = 16;
__global__ __launch_bounds__(1024) void tfc_m_prepass_kernel(const float* __restrict__ y,
                                                             const MEnt* __restrict__ ments,
                                                             const float* __restrict__ cgm, long n_edges,
                                                             int msz, int d_sh, float* __restrict__ M) {
    constexpr int MAXN2 = 2 * TFC_LMAX + 1;
    extern __shared__ float ys[];  // [M_EPB][d_sh]
    for (long e0 = (long)blockIdx.x * M_EPB; e0 < n_edges; e0 += (long)gridDim.x * M_EPB) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < M_EPB * d_sh; idx += blockDim.x) {
            const int ee = idx / d_sh, c = idx - ee * d_sh;
            ys[idx] = e0 + ee < n_edges ? y[(e0 + ee) * d_sh + c] : 0.f;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < msz; j += blockDim.x) {
            const MEnt me = ments[j];
            float c[MAXN2];
#pragma unroll
            for (int m2 = 0; m2 < MAXN2; ++m2) c[m2] = m2 < me.n2 ? cgm[me.cg_row + m2] : 0.f;
            float* Mj = M + e0 * msz + j;
#pragma unroll 4
            for (int ee = 0; ee < M_EPB; ++ee) {
                float v = 0.f;
#pragma unroll
                for (int m2 = 0; m2 < MAXN2; ++m2)
                    if (m2 < me.n2) v = fmaf(c[m2], ys[ee * d_sh + me.y_off + m2], v);
                if (e0 + ee < n_edges) Mj[(long)ee * msz] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------
// UNR edges of one path: all gathers first, then the FMAs.  The ids of up to 64 edges of the node
// sit in one VGPR each (lane j = edge c0 + j, loaded with one vector load) and are read out with
// v_readlane, so that no step waits on a scalar load of its own.
template <int L1, int L3, int UNR>
__device__ __forceinline__ void fwd_edges(const float* __restrict__ xb, const float* __restrict__ wb,
                                          const float* __restrict__ Mb, int ev, int sv, int j0, int k,
                                          int d_in, int w_numel, int msz, float (&acc)[2 * L3 + 1]) {
    constexpr int D1 = 2 * L1 + 1, D3 = 2 * L3 + 1;
    float xv[UNR][D1], wv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int e = __builtin_amdgcn_readlane(ev, j0 + j);
        const int s = __builtin_amdgcn_readlane(sv, j0 + j);
        const float* xr = xb + (size_t)s * d_in;
#pragma unroll
        for (int m1 = 0; m1 < D1; ++m1) xv[j][m1] = xr[m1];
        wv[j] = __builtin_nontemporal_load(&wb[(size_t)e * w_numel]);  // every weight is read once: streaming load
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const float* Mk = Mb + (size_t)__builtin_amdgcn_readlane(ev, j0 + j) * msz;  // M rows are in edge order
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) {
            float z = 0.f;
#pragma unroll
            for (int m1 = 0; m1 < D1; ++m1) z = fmaf(Mk[m1 * D3 + m3], xv[j][m1], z);
            acc[m3] = fmaf(wv[j], z, acc[m3]);
        }
    }
}

template <int L1, int L3>
__device__ __forceinline__ void fwd_item(const float* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ M, const int* __restrict__ eid,
                                         const int* __restrict__ nbr, int beg, int end, const PathDev p,
                                         int u, bool active, int node, int d_in, int d_out, int w_numel,
                                         int msz, float scale, float* __restrict__ out,
                                         const int* __restrict__ tout, float* __restrict__ out_t, int n_nodes) {
    constexpr int D1 = 2 * L1 + 1, D3 = 2 * L3 + 1;
    // edges in flight per step: 4 / 2 (was 8 / 4: the M blocks of that many edges need more scalar registers than
    // there are, and the compiler spills them through v_writelane; config-3 last layer 174 -> 161 us)
#ifndef TFC_FWD_UNR_A
#define TFC_FWD_UNR_A 4
#define TFC_FWD_UNR_B 2
#endif
    constexpr int UNR = D1 <= 3 ? TFC_FWD_UNR_A : TFC_FWD_UNR_B;
    const int lane = threadIdx.x & 63;
    const int uu = active ? u : 0;  // idle lanes shadow channel 0: valid addresses, nothing stored
    const float* xb = x + p.x_off + uu * D1;
    const float* wb = w + p.w_off + uu;
    const float* Mb = M + p.m_off;
    float acc[D3];
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3) acc[m3] = 0.f;
    for (int c0 = beg; c0 < end; c0 += 64) {
        const int cn = end - c0 < 64 ? end - c0 : 64;
        const int kl = c0 + (lane < cn ? lane : cn - 1);
        const int ev = eid[kl], sv = nbr[kl];
        int j = 0;
        for (; j + UNR <= cn; j += UNR) fwd_edges<L1, L3, UNR>(xb, wb, Mb, ev, sv, j, c0 + j, d_in, w_numel, msz, acc);
        for (; j < cn; ++j) fwd_edges<L1, L3, 1>(xb, wb, Mb, ev, sv, j, c0 + j, d_in, w_numel, msz, acc);
    }
    if (active) {
        const float cs = p.coeff * scale;
        float* o = out + (size_t)node * d_out + p.out_off + u * D3;
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) o[m3] = cs * acc[m3];
        if (out_t) {  // second copy, channel-contiguous: block [2l3+1][n_nodes][block mul] (256 B per wave and plane)
            const int t_off = tout[0], t_mul = tout[1], t_ch = tout[2];
            float* ot = out_t + (size_t)t_off * n_nodes + (size_t)node * t_mul + t_ch + u;
            const size_t plane = (size_t)n_nodes * t_mul;
#pragma unroll
            for (int m3 = 0; m3 < D3; ++m3) ot[m3 * plane] = cs * acc[m3];
        }
    }
}

// Task -> (node, item).  Nodes are taken in groups of `group`; inside a group the tasks run item
// by item over the group's nodes, so that the waves in flight on an XCD work on the SAME columns
// of the gathered rows for a set of neighbouring nodes (crystal neighbours are each other's
// neighbours): a row segment is then re-read by its ~deg consumers while it is still in that
// XCD's L2.  group = 1 is the plain node-major order.
__device__ __forceinline__ void tfc_task_split(long task, int n_items, int n_nodes, int group, int& node, int& item) {
    const long per = (long)group * n_items;
    const long g = task / per;
    const int local = (int)(task - g * per);
    const int g0 = (int)(g * group);
    const int gsz = min(group, n_nodes - g0);
    item = local / gsz;
    node = g0 + local - item * gsz;
}

__global__ __launch_bounds__(256) void tfc_conv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ M,
    const int* __restrict__ row_ptr, const int* __restrict__ eid, const int* __restrict__ nbr,
    const PathDev* __restrict__ paths, const FwdItem* __restrict__ items, int n_items, long n_tasks,
    int d_in, int d_out, int w_numel, int msz, float scale, float* __restrict__ out, int n_nodes, int group,
    const int* __restrict__ tout, float* __restrict__ out_t) {
    const unsigned vb = tfc_xcd_remap(blockIdx.x, gridDim.x);
    const int wave = tfc_uniform((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const long task = (long)vb * 4 + wave;
    if (task >= n_tasks) return;
    int node, item;
    tfc_task_split(task, n_items, n_nodes, group, node, item);
    node = tfc_uniform(node);
    item = tfc_uniform(item);
    const FwdItem it = items[item];
    const PathDev p = paths[it.path];
    const int u = it.chunk * 64 + lane;
    const bool active = u < p.mul;
    const int beg = tfc_uniform(row_ptr[node]);
    const int end = tfc_uniform(row_ptr[node + 1]);
    const int* tp = tout + 3 * it.path;  // read after the edge loop only

#define TFC_FWD_CASE(A, B)                                                                          \
    case (A)*4 + (B):                                                                               \
        fwd_item<A, B>(x, w, M, eid, nbr, beg, end, p, u, active, node, d_in, d_out, w_numel, msz, \
                       scale, out, tp, out_t, n_nodes);                                             \
        break;
    switch (p.l1 * 4 + p.l3) {
        TFC_FWD_CASE(0, 0) TFC_FWD_CASE(0, 1) TFC_FWD_CASE(0, 2) TFC_FWD_CASE(0, 3)
        TFC_FWD_CASE(1, 0) TFC_FWD_CASE(1, 1) TFC_FWD_CASE(1, 2) TFC_FWD_CASE(1, 3)
        TFC_FWD_CASE(2, 0) TFC_FWD_CASE(2, 1) TFC_FWD_CASE(2, 2) TFC_FWD_CASE(2, 3)
        TFC_FWD_CASE(3, 0) TFC_FWD_CASE(3, 1) TFC_FWD_CASE(3, 2) TFC_FWD_CASE(3, 3)
        default: break;
    }
#undef TFC_FWD_CASE
}

// ------------------------------------------------------------------------------------------
// Forward with the radial network's LAST layer folded in ("measure both"): the per-edge
// weights w [E, W] are never read; each wave forms the 64 weights of its (path, chunk) for an edge from the last
// hidden activations h [E, H] (wave-uniform: scalar loads) and its own H x 64 slice of W_L, one column per lane
// held in registers:  w[e, w_off + u] = alpha * sum_k h[e, k] * W_L[k, w_off + u].
// Algorithmic bytes per edge drop by 4 W (the w row) and grow by 4 H (the h row); the price is H extra FMAs and
// H scalar operands per (edge, path), and H more live registers.  H = 64 only.
template <int L1, int L3, int UNR, int H>
__device__ __forceinline__ void fwdw_edges(const float* __restrict__ xb, const float* __restrict__ h,
                                           const float* __restrict__ Mb, int ev, int sv, int j0, int k, int d_in,
                                           int msz, const float (&wc)[H], float (&acc)[2 * L3 + 1]) {
    constexpr int D1 = 2 * L1 + 1, D3 = 2 * L3 + 1;
    float xv[UNR][D1], wv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int s = __builtin_amdgcn_readlane(sv, j0 + j);
        const float* xr = xb + (size_t)s * d_in;
#pragma unroll
        for (int m1 = 0; m1 < D1; ++m1) xv[j][m1] = xr[m1];
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int e = __builtin_amdgcn_readlane(ev, j0 + j);
        const float* hr = h + (size_t)e * H;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;  // four chains instead of one of length H
#pragma unroll
        for (int q = 0; q < H; q += 4) {
            p0 = fmaf(hr[q], wc[q], p0);
            p1 = fmaf(hr[q + 1], wc[q + 1], p1);
            p2 = fmaf(hr[q + 2], wc[q + 2], p2);
            p3 = fmaf(hr[q + 3], wc[q + 3], p3);
        }
        wv[j] = (p0 + p1) + (p2 + p3);
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const float* Mk = Mb + (size_t)__builtin_amdgcn_readlane(ev, j0 + j) * msz;
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) {
            float z = 0.f;
#pragma unroll
            for (int m1 = 0; m1 < D1; ++m1) z = fmaf(Mk[m1 * D3 + m3], xv[j][m1], z);
            acc[m3] = fmaf(wv[j], z, acc[m3]);
        }
    }
}

template <int L1, int L3, int H>
__device__ __forceinline__ void fwdw_item(const float* __restrict__ x, const float* __restrict__ h,
                                          const float* __restrict__ wl, const float* __restrict__ M,
                                          const int* __restrict__ eid, const int* __restrict__ nbr, int beg, int end,
                                          const PathDev p, int u, bool active, int node, int d_in, int d_out,
                                          int w_numel, int msz, float scale, float alpha, float* __restrict__ out) {
    constexpr int D1 = 2 * L1 + 1, D3 = 2 * L3 + 1;
    constexpr int UNR = D1 <= 3 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int uu = active ? u : 0;
    const float* xb = x + p.x_off + uu * D1;
    const float* Mb = M + p.m_off;
    float wc[H];
#pragma unroll
    for (int q = 0; q < H; ++q) wc[q] = alpha * wl[(size_t)q * w_numel + p.w_off + uu];  // coalesced over lanes
    float acc[D3];
#pragma unroll
    for (int m3 = 0; m3 < D3; ++m3) acc[m3] = 0.f;
    for (int c0 = beg; c0 < end; c0 += 64) {
        const int cn = end - c0 < 64 ? end - c0 : 64;
        const int kl = c0 + (lane < cn ? lane : cn - 1);
        const int ev = eid[kl], sv = nbr[kl];
        int j = 0;
        for (; j + UNR <= cn; j += UNR) fwdw_edges<L1, L3, UNR, H>(xb, h, Mb, ev, sv, j, c0 + j, d_in, msz, wc, acc);
        for (; j < cn; ++j) fwdw_edges<L1, L3, 1, H>(xb, h, Mb, ev, sv, j, c0 + j, d_in, msz, wc, acc);
    }
    if (active) {
        const float cs = p.coeff * scale;
        float* o = out + (size_t)node * d_out + p.out_off + u * D3;
#pragma unroll
        for (int m3 = 0; m3 < D3; ++m3) o[m3] = cs * acc[m3];
    }
}

__global__ __launch_bounds__(256) void tfc_conv_fwd_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ wl,
    const float* __restrict__ M, const int* __restrict__ row_ptr, const int* __restrict__ eid,
    const int* __restrict__ nbr, const PathDev* __restrict__ paths, const FwdItem* __restrict__ items, int n_items,
    long n_tasks, int d_in, int d_out, int w_numel, int msz, float scale, float alpha, float* __restrict__ out,
    int n_nodes, int group) {
    const unsigned vb = tfc_xcd_remap(blockIdx.x, gridDim.x);
    const int wave = tfc_uniform((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const long task = (long)vb * 4 + wave;
    if (task >= n_tasks) return;
    int node, item;
    tfc_task_split(task, n_items, n_nodes, group, node, item);
    node = tfc_uniform(node);
    item = tfc_uniform(item);
    const FwdItem it = items[item];
    const PathDev p = paths[it.path];
    const int u = it.chunk * 64 + lane;
    const bool active = u < p.mul;
    const int beg = tfc_uniform(row_ptr[node]);
    const int end = tfc_uniform(row_ptr[node + 1]);
#define TFC_FWDW_CASE(A, B)                                                                               \
    case (A)*4 + (B):                                                                                     \
        fwdw_item<A, B, 64>(x, h, wl, M, eid, nbr, beg, end, p, u, active, node, d_in, d_out, w_numel, msz, \
                            scale, alpha, out);                                                           \
        break;
    switch (p.l1 * 4 + p.l3) {
        TFC_FWDW_CASE(0, 0) TFC_FWDW_CASE(0, 1) TFC_FWDW_CASE(0, 2) TFC_FWDW_CASE(0, 3)
        TFC_FWDW_CASE(1, 0) TFC_FWDW_CASE(1, 1) TFC_FWDW_CASE(1, 2) TFC_FWDW_CASE(1, 3)
        TFC_FWDW_CASE(2, 0) TFC_FWDW_CASE(2, 1) TFC_FWDW_CASE(2, 2) TFC_FWDW_CASE(2, 3)
        TFC_FWDW_CASE(3, 0) TFC_FWDW_CASE(3, 1) TFC_FWDW_CASE(3, 2) TFC_FWDW_CASE(3, 3)
        default: break;
    }
#undef TFC_FWDW_CASE
}

// ------------------------------------------------------------------------------------------
// Backward.  Centre node = sender: its own row x[n] is fixed for the whole item, the upstream
// gradient rows gout[receiver] are gathered.
// ------------------------------------------------------------------------------------------
// U edges of one path per step: all their gathers are issued before the first use, and the
// 64-lane sums the edge-attribute gradient needs (one per edge and m2) are taken U * (2 l2 + 1)
// at a time by the reduce-scatter butterfly of tfc_common.h instead of one at a time.
template <int D2>
struct BwdUnroll {
#ifndef TFC_BWD_U1
#define TFC_BWD_U1 4
#define TFC_BWD_U3 5
#define TFC_BWD_U5 3
#define TFC_BWD_U7 2
#endif
    static constexpr int U = D2 == 1 ? TFC_BWD_U1 : (D2 == 3 ? TFC_BWD_U3 : (D2 == 5 ? TFC_BWD_U5 : TFC_BWD_U7));  // U * D2 <= 16
    static constexpr int V = D2 == 1 ? 4 : 16;
};

template <int L1, int L2, int L3, bool NEED_DY>
__device__ __forceinline__ void bwd_path(const float* __restrict__ w, const float* __restrict__ gout,
                                         const float* __restrict__ M, const float* __restrict__ cgd,
                                         const int* __restrict__ eid, const int* __restrict__ nbr,
                                         int beg, int end, const PathDev p, int u, int uu, bool active,
                                         int d_out, int w_numel, int msz, int py, float scale,
                                         const float (&xv)[2 * L1 + 1], float (&gx)[2 * L1 + 1],
                                         float* __restrict__ dw, float* __restrict__ dyp, int chunk) {
    constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
    constexpr int U = NEED_DY ? BwdUnroll<D2>::U : 4;
    constexpr int V = BwdUnroll<D2>::V;
    const int lane = threadIdx.x & 63;
    const float cs = p.coeff * scale;
    const float* gb = gout + p.out_off + uu * D3;
    const float* wb = w + p.w_off + uu;
    float* dwb = dw + p.w_off + uu;
    const float* Mb = M + p.m_off;

    // P[m2][m3] = sum_m1 C[m1][m2][m3] * x[n][u][m1]: fixed for the item, used for d/dy
    float P[NEED_DY ? D2 : 1][NEED_DY ? D3 : 1];
    if (NEED_DY) {
        const float* C = cgd + p.cg_off;
#pragma unroll
        for (int m2 = 0; m2 < D2; ++m2)
#pragma unroll
            for (int m3 = 0; m3 < D3; ++m3) {
                float s = 0.f;
#pragma unroll
                for (int m1 = 0; m1 < D1; ++m1) s = fmaf(C[(m1 * D2 + m2) * D3 + m3], xv[m1], s);
                P[m2][m3] = active ? s : 0.f;
            }
    }
    // which reduced value this lane ends up holding, and the edge / component it belongs to
    const int ridx = V == 16 ? tfc_reduce16_index(lane) : tfc_reduce4_index(lane);
    const int rj = ridx / D2, rm2 = ridx - rj * D2;
    const bool rlane = V == 16 ? (lane & 3) == 0 : (lane & 15) == 0;

    for (int k = beg; k < end; k += U) {
        int e[U];
        float g[U][D3], wv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int kk = k + j < end ? k + j : end - 1;  // past the end: re-read the last edge, contribute 0
            e[j] = eid[kk];
            const int d = nbr[kk];
            const float* gr = gb + (size_t)d * d_out;
#pragma unroll
            for (int m3 = 0; m3 < D3; ++m3) g[j][m3] = gr[m3];
            wv[j] = __builtin_nontemporal_load(&wb[(size_t)e[j] * w_numel]);  // every weight is read once
        }
        float q[NEED_DY ? V : 1];
        if (NEED_DY) {
#pragma unroll
            for (int i = 0; i < V; ++i) q[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool ok = k + j < end;
            const float* Mk = Mb + (size_t)e[j] * msz;  // M rows are in edge order
            const float wj = ok ? wv[j] : 0.f;
#pragma unroll
            for (int m3 = 0; m3 < D3; ++m3) g[j][m3] *= cs;
            float gw = 0.f;
#pragma unroll
            for (int m1 = 0; m1 < D1; ++m1) {
                float t = 0.f;
#pragma unroll
                for (int m3 = 0; m3 < D3; ++m3) t = fmaf(Mk[m1 * D3 + m3], g[j][m3], t);
                gw = fmaf(xv[m1], t, gw);
                gx[m1] = fmaf(wj, t, gx[m1]);
            }
            // (written once, read much later by another kernel: streaming store, so that it does not push the
            // gathered rows and the M rows out of L2)
            if (ok && active) __builtin_nontemporal_store(gw, &dwb[(size_t)e[j] * w_numel]);
            if (NEED_DY) {
#pragma unroll
                for (int m2 = 0; m2 < D2; ++m2) {
                    float s = 0.f;
#pragma unroll
                    for (int m3 = 0; m3 < D3; ++m3) s = fmaf(P[m2][m3], g[j][m3], s);
                    q[j * D2 + m2] = wj * s;  // idle lanes hold P = 0
                }
            }
        }
        if (NEED_DY) {
            float r;
            if (V == 16) {
                float (&q16)[16] = reinterpret_cast<float (&)[16]>(q);
                r = tfc_wave_reduce16(q16);
            } else {
                float (&q4)[4] = reinterpret_cast<float (&)[4]>(q);
                r = tfc_wave_reduce4(q4);
            }
            int es = e[0];
#pragma unroll
            for (int j = 1; j < U; ++j) es = rj == j ? e[j] : es;
            if (rlane && rj < U && k + rj < end) dyp[(size_t)es * py + p.py_off + chunk * D2 + rm2] = r;
        }
    }
}

template <int L1, bool NEED_DY, int LM>
__device__ __forceinline__ void bwd_item(const float* __restrict__ x, const float* __restrict__ w,
                                         const float* __restrict__ gout, const float* __restrict__ M,
                                         const float* __restrict__ cgd, const int* __restrict__ eid,
                                         const int* __restrict__ nbr, const PathDev* __restrict__ paths,
                                         const int* __restrict__ grp_paths, const BwdItem it, int beg,
                                         int end, int node, int lane, int d_in, int d_out, int w_numel,
                                         int msz, int py, float scale, float* __restrict__ dx,
                                         float* __restrict__ dw, float* __restrict__ dyp,
                                         float* __restrict__ dx_t, int n_nodes) {
    constexpr int D1 = 2 * L1 + 1;
    const int u = it.chunk * 64 + lane;
    const bool active = u < it.mul;
    const int uu = active ? u : 0;
    float xv[D1], gx[D1];
    const float* xr = x + (size_t)node * d_in + it.x_off + uu * D1;
#pragma unroll
    for (int m1 = 0; m1 < D1; ++m1) {
        xv[m1] = xr[m1];
        gx[m1] = 0.f;
    }
#ifndef TFC_BWD_NO_WARM
    // Warm-up: the M blocks this wave reads through the scalar cache are touched exactly once per backward pass,
    // so every group of edges would wait for a scalar miss to HBM with 3-4 waves per SIMD to hide it.  Lane j
    // first touches every 128-byte line of the M row of edge j (all edges of the node at once through the vector
    // path); the scalar loads below then find the rows in L2.
    {
        float t = 0.f;
        for (int k0 = beg; k0 < end; k0 += 64) {
            const int kk = k0 + lane < end ? k0 + lane : end - 1;
            const float* mr = M + (size_t)eid[kk] * msz;
            for (int o = 0; o < msz; o += 32) t += mr[o];
        }
        asm volatile("" ::"v"(t));
    }
#endif
    for (int gi = 0; gi < it.grp_count; ++gi) {
        const PathDev p = paths[grp_paths[it.grp_start + gi]];
#define TFC_BWD_CASE(B, C)                                                                            \
    case (B)*4 + (C):                                                                                 \
        if constexpr ((B) <= LM && (C) <= LM)                                                         \
            bwd_path<L1, B, C, NEED_DY>(w, gout, M, cgd, eid, nbr, beg, end, p, u, uu, active, d_out, \
                                        w_numel, msz, py, scale, xv, gx, dw, dyp, it.chunk);          \
        break;
        switch (p.l2 * 4 + p.l3) {
            TFC_BWD_CASE(0, 0) TFC_BWD_CASE(0, 1) TFC_BWD_CASE(0, 2) TFC_BWD_CASE(0, 3)
            TFC_BWD_CASE(1, 0) TFC_BWD_CASE(1, 1) TFC_BWD_CASE(1, 2) TFC_BWD_CASE(1, 3)
            TFC_BWD_CASE(2, 0) TFC_BWD_CASE(2, 1) TFC_BWD_CASE(2, 2) TFC_BWD_CASE(2, 3)
            TFC_BWD_CASE(3, 0) TFC_BWD_CASE(3, 1) TFC_BWD_CASE(3, 2) TFC_BWD_CASE(3, 3)
            default: break;
        }
#undef TFC_BWD_CASE
    }
    if (active) {
        float* o = dx + (size_t)node * d_in + it.x_off + u * D1;
#pragma unroll
        for (int m1 = 0; m1 < D1; ++m1) o[m1] = gx[m1];
        if (dx_t) {  // second copy, channel-contiguous: in1 block as [2l1+1][n_nodes][mul]
            float* ot = dx_t + (size_t)it.x_off * n_nodes + (size_t)node * it.mul + u;
            const size_t plane = (size_t)n_nodes * it.mul;
#pragma unroll
            for (int m1 = 0; m1 < D1; ++m1) ot[m1 * plane] = gx[m1];
        }
    }
}

// LM: largest l any path of the plan uses.  The register allocation of the kernel is that of its
// widest (l1, l2, l3) case, so plans without l = 3 get their own, leaner instantiation.
template <bool NEED_DY, int LM>
__global__ __launch_bounds__(256) void tfc_conv_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ gout,
    const float* __restrict__ M, const float* __restrict__ cgd, const int* __restrict__ row_ptr,
    const int* __restrict__ eid, const int* __restrict__ nbr, const PathDev* __restrict__ paths,
    const BwdItem* __restrict__ items, const int* __restrict__ grp_paths, int n_items, long n_tasks,
    int d_in, int d_out, int w_numel, int msz, int py, float scale, float* __restrict__ dx,
    float* __restrict__ dw, float* __restrict__ dyp, int n_nodes, int group, float* __restrict__ dx_t) {
    const unsigned vb = tfc_xcd_remap(blockIdx.x, gridDim.x);
    const int wave = tfc_uniform((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const long task = (long)vb * 4 + wave;
    if (task >= n_tasks) return;
    int node, item;
    tfc_task_split(task, n_items, n_nodes, group, node, item);
    node = tfc_uniform(node);
    item = tfc_uniform(item);
    const BwdItem it = items[item];
    const int beg = tfc_uniform(row_ptr[node]);
    const int end = tfc_uniform(row_ptr[node + 1]);
    switch (it.l1) {
        case 0:
            if constexpr (0 <= LM)
                bwd_item<0, NEED_DY, LM>(x, w, gout, M, cgd, eid, nbr, paths, grp_paths, it, beg, end, node, lane,
                                 d_in, d_out, w_numel, msz, py, scale, dx, dw, dyp, dx_t, n_nodes);
            break;
        case 1:
            if constexpr (1 <= LM)
                bwd_item<1, NEED_DY, LM>(x, w, gout, M, cgd, eid, nbr, paths, grp_paths, it, beg, end, node, lane,
                                 d_in, d_out, w_numel, msz, py, scale, dx, dw, dyp, dx_t, n_nodes);
            break;
        case 2:
            if constexpr (2 <= LM)
                bwd_item<2, NEED_DY, LM>(x, w, gout, M, cgd, eid, nbr, paths, grp_paths, it, beg, end, node, lane,
                                 d_in, d_out, w_numel, msz, py, scale, dx, dw, dyp, dx_t, n_nodes);
            break;
        case 3:
            if constexpr (3 <= LM)
                bwd_item<3, NEED_DY, LM>(x, w, gout, M, cgd, eid, nbr, paths, grp_paths, it, beg, end, node, lane,
                                 d_in, d_out, w_numel, msz, py, scale, dx, dw, dyp, dx_t, n_nodes);
            break;
        default: break;
    }
}

// dy[e][j] = sum of the partial-dy columns that belong to edge-attribute column j
__global__ __launch_bounds__(256) void tfc_dy_reduce_kernel(const float* __restrict__ dyp,
                                                            const int* __restrict__ ylist_ptr,
                                                            const int* __restrict__ ylist, long n_edges,
                                                            int d_sh, int py, float* __restrict__ dy) {
    const long total = n_edges * (long)d_sh;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long e = i / d_sh;
        const int j = (int)(i - e * d_sh);
        const float* r = dyp + e * (long)py;
        float s = 0.f;
        for (int t = ylist_ptr[j]; t < ylist_ptr[j + 1]; ++t) s += r[ylist[t]];
        dy[i] = s;
    }
}

// zero-fill of rows no item writes (input blocks read by no path)
__global__ __launch_bounds__(256) void tfc_fill_kernel(float* __restrict__ p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        p[i] = v;
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" int tfc_conv_plan_create(const tfc_path* paths, int32_t n_paths, int32_t d_in, int32_t d_sh,
                                    int32_t d_out, int32_t w_numel, tfc_conv_plan** plan_out) {
    TFC_REQUIRE(paths && plan_out, TFC_EINVAL, "tfc_conv_plan_create: null argument");
    TFC_REQUIRE(n_paths > 0 && n_paths <= 4096, TFC_EINVAL, "tfc_conv_plan_create: n_paths=%d", n_paths);
    TFC_REQUIRE(d_in > 0 && d_sh > 0 && d_out > 0 && w_numel > 0, TFC_EINVAL,
                "tfc_conv_plan_create: non-positive row size");
    *plan_out = nullptr;

    std::vector<PathDev> pd(n_paths);
    std::map<std::tuple<int, int, int, int>, int> triple_index;  // (l1,l2,l3,y_off) -> index
    std::vector<std::tuple<int, int, int, int>> triples;
    std::vector<char> out_used(d_out, 0);
    for (int i = 0; i < n_paths; ++i) {
        const tfc_path& p = paths[i];
        TFC_REQUIRE(p.l1 >= 0 && p.l2 >= 0 && p.l3 >= 0 && p.l1 <= TFC_LMAX && p.l2 <= TFC_LMAX &&
                        p.l3 <= TFC_LMAX,
                    TFC_EUNSUPPORTED, "path %d: l=(%d,%d,%d) outside 0..%d", i, p.l1, p.l2, p.l3, TFC_LMAX);
        TFC_REQUIRE(std::abs(p.l1 - p.l2) <= p.l3 && p.l3 <= p.l1 + p.l2, TFC_EINVAL,
                    "path %d: (%d,%d,%d) violates the triangle rule", i, p.l1, p.l2, p.l3);
        TFC_REQUIRE(p.mul > 0, TFC_EINVAL, "path %d: mul=%d", i, p.mul);
        TFC_REQUIRE(p.x_off >= 0 && p.x_off + p.mul * (2 * p.l1 + 1) <= d_in, TFC_EINVAL,
                    "path %d: in1 block outside the node-feature row", i);
        TFC_REQUIRE(p.y_off >= 0 && p.y_off + (2 * p.l2 + 1) <= d_sh, TFC_EINVAL,
                    "path %d: in2 block outside the edge-attribute row", i);
        TFC_REQUIRE(p.out_off >= 0 && p.out_off + p.mul * (2 * p.l3 + 1) <= d_out, TFC_EINVAL,
                    "path %d: output block outside the output row", i);
        TFC_REQUIRE(p.w_off >= 0 && p.w_off + p.mul <= w_numel, TFC_EINVAL,
                    "path %d: weights outside the weight row", i);
        for (int c = p.out_off; c < p.out_off + p.mul * (2 * p.l3 + 1); ++c) {
            TFC_REQUIRE(!out_used[c], TFC_EUNSUPPORTED, "path %d shares its output block with another path",
                        i);
            out_used[c] = 1;
        }
        auto key = std::make_tuple(p.l1, p.l2, p.l3, p.y_off);
        if (!triple_index.count(key)) {
            triple_index[key] = (int)triples.size();
            triples.push_back(key);
        }
        pd[i] = PathDev{p.l1, p.l2, p.l3, p.mul, p.x_off, p.y_off, p.out_off, p.w_off, 0, 0, 0, p.coeff};
    }

    // M-row layout + CG tables per triple
    std::vector<int> t_moff(triples.size()), t_cgoff(triples.size());
    std::vector<MEnt> ments;
    std::vector<float> cgm, cgd;
    std::vector<double> cg;
    for (size_t t = 0; t < triples.size(); ++t) {
        int l1, l2, l3, yo;
        std::tie(l1, l2, l3, yo) = triples[t];
        const int D1 = 2 * l1 + 1, D2 = 2 * l2 + 1, D3 = 2 * l3 + 1;
        cg.assign((size_t)D1 * D2 * D3, 0.0);
        int rc = tfc_real_cg(l1, l2, l3, cg.data());
        if (rc != TFC_OK) return rc;
        t_moff[t] = (int)ments.size();
        t_cgoff[t] = (int)cgd.size();
        for (double v : cg) cgd.push_back((float)v);
        for (int m1 = 0; m1 < D1; ++m1)
            for (int m3 = 0; m3 < D3; ++m3) {
                MEnt me{(int)cgm.size(), yo, D2};
                for (int m2 = 0; m2 < D2; ++m2) cgm.push_back((float)cg[((size_t)m1 * D2 + m2) * D3 + m3]);
                ments.push_back(me);
            }
        while (ments.size() % 4) ments.push_back(MEnt{0, 0, 0});
    }
    const int msz = (int)ments.size();
    if (cgm.empty()) cgm.push_back(0.f);

    // forward items: (path, 64-channel chunk); partial-dy slots in the same order
    std::vector<FwdItem> fwd_items;
    int py = 0;
    std::vector<std::vector<int>> ycols(d_sh);
    for (int i = 0; i < n_paths; ++i) {
        auto key = std::make_tuple(pd[i].l1, pd[i].l2, pd[i].l3, pd[i].y_off);
        const int t = triple_index[key];
        pd[i].m_off = t_moff[t];
        pd[i].cg_off = t_cgoff[t];
        pd[i].py_off = py;
        const int chunks = (pd[i].mul + 63) / 64;
        for (int c = 0; c < chunks; ++c) {
            fwd_items.push_back(FwdItem{i, c});
            for (int m2 = 0; m2 < 2 * pd[i].l2 + 1; ++m2) ycols[pd[i].y_off + m2].push_back(py + m2);
            py += 2 * pd[i].l2 + 1;
        }
    }
    // order forward items by in1 block so that waves sharing gathered lines are neighbours
    std::stable_sort(fwd_items.begin(), fwd_items.end(), [&](const FwdItem& a, const FwdItem& b) {
        return std::make_pair(pd[a.path].x_off, a.chunk) < std::make_pair(pd[b.path].x_off, b.chunk);
    });

    // backward items: (in1 block, chunk) with the list of paths reading the block
    std::map<std::tuple<int, int, int>, std::vector<int>> groups;  // (x_off, l1, mul) -> paths
    for (int i = 0; i < n_paths; ++i) groups[std::make_tuple(pd[i].x_off, pd[i].l1, pd[i].mul)].push_back(i);
    std::vector<BwdItem> bwd_items;
    std::vector<int> grp_paths;
    for (auto& kv : groups) {
        int xo, l1, mul;
        std::tie(xo, l1, mul) = kv.first;
        const int start = (int)grp_paths.size();
        for (int pi : kv.second) grp_paths.push_back(pi);
        for (int c = 0; c < (mul + 63) / 64; ++c)
            bwd_items.push_back(BwdItem{start, (int)kv.second.size(), c, l1, xo, mul});
    }
    // in1 blocks must not partially overlap (same offset => same block)
    {
        std::vector<char> used(d_in, 0);
        for (auto& kv : groups) {
            int xo, l1, mul;
            std::tie(xo, l1, mul) = kv.first;
            for (int c = xo; c < xo + mul * (2 * l1 + 1); ++c) {
                TFC_REQUIRE(!used[c], TFC_EUNSUPPORTED, "overlapping in1 blocks at column %d", c);
                used[c] = 1;
            }
        }
    }

    std::vector<int> ylist_ptr(d_sh + 1, 0), ylist;
    for (int j = 0; j < d_sh; ++j) {
        ylist_ptr[j] = (int)ylist.size();
        for (int c : ycols[j]) ylist.push_back(c);
    }
    ylist_ptr[d_sh] = (int)ylist.size();
    if (ylist.empty()) ylist.push_back(0);

    // one device blob for all tables
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t o_paths = 0;
    size_t o_fwd = al(o_paths + pd.size() * sizeof(PathDev));
    size_t o_bwd = al(o_fwd + fwd_items.size() * sizeof(FwdItem));
    size_t o_grp = al(o_bwd + bwd_items.size() * sizeof(BwdItem));
    size_t o_ment = al(o_grp + grp_paths.size() * sizeof(int));
    size_t o_cgm = al(o_ment + ments.size() * sizeof(MEnt));
    size_t o_cgd = al(o_cgm + cgm.size() * sizeof(float));
    size_t o_ylp = al(o_cgd + cgd.size() * sizeof(float));
    size_t o_yl = al(o_ylp + ylist_ptr.size() * sizeof(int));
    size_t total = al(o_yl + ylist.size() * sizeof(int));
    std::vector<char> host(total, 0);
    std::memcpy(host.data() + o_paths, pd.data(), pd.size() * sizeof(PathDev));
    std::memcpy(host.data() + o_fwd, fwd_items.data(), fwd_items.size() * sizeof(FwdItem));
    std::memcpy(host.data() + o_bwd, bwd_items.data(), bwd_items.size() * sizeof(BwdItem));
    std::memcpy(host.data() + o_grp, grp_paths.data(), grp_paths.size() * sizeof(int));
    std::memcpy(host.data() + o_ment, ments.data(), ments.size() * sizeof(MEnt));
    std::memcpy(host.data() + o_cgm, cgm.data(), cgm.size() * sizeof(float));
    std::memcpy(host.data() + o_cgd, cgd.data(), cgd.size() * sizeof(float));
    std::memcpy(host.data() + o_ylp, ylist_ptr.data(), ylist_ptr.size() * sizeof(int));
    std::memcpy(host.data() + o_yl, ylist.data(), ylist.size() * sizeof(int));

    tfc_conv_plan* pl = new (std::nothrow) tfc_conv_plan();
    TFC_REQUIRE(pl, TFC_ENOMEM, "tfc_conv_plan_create: out of host memory");
    hipError_t e = hipGetDevice(&pl->device);
    if (e == hipSuccess) e = hipMalloc(&pl->blob, total);
    if (e == hipSuccess) e = hipMemcpy(pl->blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (pl->blob) (void)hipFree(pl->blob);
        delete pl;
        return tfc_set_error(e == hipErrorOutOfMemory ? TFC_ENOMEM : TFC_ELAUNCH,
                             "tfc_conv_plan_create: device table upload failed: %s", hipGetErrorString(e));
    }
    char* b = (char*)pl->blob;
    pl->n_paths = n_paths;
    pl->d_in = d_in;
    pl->d_sh = d_sh;
    pl->d_out = d_out;
    pl->w_numel = w_numel;
    pl->msz = msz;
    pl->py = py;
    pl->n_fwd_items = (int)fwd_items.size();
    pl->n_bwd_items = (int)bwd_items.size();
    {
        std::vector<char> used(d_in, 0);
        for (int i = 0; i < n_paths; ++i)
            for (int c = pd[i].x_off; c < pd[i].x_off + pd[i].mul * (2 * pd[i].l1 + 1); ++c) used[c] = 1;
        pl->dx_covered = 1;
        for (int c = 0; c < d_in; ++c) pl->dx_covered &= used[c];
    }
    pl->m_sig = 1469598103934665603L;
    for (auto& t : triples) {
        int v[4];
        std::tie(v[0], v[1], v[2], v[3]) = t;
        for (int q = 0; q < 4; ++q) pl->m_sig = (pl->m_sig ^ (long)(v[q] + 1)) * 1099511628211L;
    }
    pl->m_sig &= 0x7fffffffffffffffL;
    pl->lmax = 0;
    for (int i = 0; i < n_paths; ++i) pl->lmax = std::max(pl->lmax, std::max(pd[i].l1, std::max(pd[i].l2, pd[i].l3)));
    pl->paths = (PathDev*)(b + o_paths);
    pl->fwd_items = (FwdItem*)(b + o_fwd);
    pl->bwd_items = (BwdItem*)(b + o_bwd);
    pl->grp_paths = (int*)(b + o_grp);
    pl->ments = (MEnt*)(b + o_ment);
    pl->cgm = (float*)(b + o_cgm);
    pl->cgd = (float*)(b + o_cgd);
    pl->ylist_ptr = (int*)(b + o_ylp);
    pl->ylist = (int*)(b + o_yl);
    *plan_out = pl;
    return TFC_OK;
}

extern "C" int tfc_conv_plan_destroy(tfc_conv_plan* plan) {
    if (!plan) return TFC_OK;
    if (plan->blob) (void)hipFree(plan->blob);
    if (plan->tout) (void)hipFree(plan->tout);
    delete plan;
    return TFC_OK;
}

extern "C" int tfc_conv_plan_set_out_blocks(tfc_conv_plan* pl, const tfc_ir_block* blocks, int32_t n_blocks) {
    TFC_REQUIRE(pl && blocks && n_blocks > 0, TFC_EINVAL, "tfc_conv_plan_set_out_blocks: null argument");
    std::vector<PathDev> pd(pl->n_paths);
    TFC_HIP(hipMemcpy(pd.data(), pl->paths, sizeof(PathDev) * pd.size(), hipMemcpyDeviceToHost));
    std::vector<int> t(3 * (size_t)pl->n_paths);
    for (int i = 0; i < pl->n_paths; ++i) {
        const PathDev& p = pd[i];
        const int d3 = 2 * p.l3 + 1;
        int found = -1;
        for (int b = 0; b < n_blocks; ++b) {
            const tfc_ir_block& k = blocks[b];
            TFC_REQUIRE(k.off >= 0 && k.mul > 0 && k.dim > 0 && (long)k.off + (long)k.mul * k.dim <= pl->d_out, TFC_EINVAL,
                        "tfc_conv_plan_set_out_blocks: block %d outside the output row", b);
            if (p.out_off >= k.off && p.out_off + p.mul * d3 <= k.off + k.mul * k.dim) found = b;
        }
        TFC_REQUIRE(found >= 0, TFC_EINVAL, "tfc_conv_plan_set_out_blocks: the output of path %d lies in no block", i);
        const tfc_ir_block& k = blocks[found];
        TFC_REQUIRE(k.dim == d3 && (p.out_off - k.off) % d3 == 0, TFC_EINVAL,
                    "tfc_conv_plan_set_out_blocks: path %d (l3=%d) does not fit block %d (dim %d)", i, p.l3, found, k.dim);
        t[3 * i] = k.off;
        t[3 * i + 1] = k.mul;
        t[3 * i + 2] = (p.out_off - k.off) / d3;
    }
    int dev = 0;
    TFC_HIP(hipGetDevice(&dev));
    TFC_REQUIRE(dev == pl->device, TFC_EINVAL, "tfc_conv_plan_set_out_blocks: plan lives on device %d, current is %d",
                pl->device, dev);
    if (!pl->tout) TFC_HIP(hipMalloc(&pl->tout, sizeof(int) * t.size()));
    TFC_HIP(hipMemcpy(pl->tout, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice));
    return TFC_OK;
}

extern "C" int64_t tfc_conv_workspace_floats(const tfc_conv_plan* plan, int64_t n_edges, int need_dy) {
    if (!plan || n_edges < 0) return -1;
    int64_t f = n_edges * (int64_t)plan->msz;
    if (need_dy) f += n_edges * (int64_t)plan->py;
    return f > 0 ? f : 1;
}

static inline int grid_for(long n, int per_block, int cap = 8192) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" int64_t tfc_conv_m_signature(const tfc_conv_plan* plan) { return plan ? plan->m_sig : -1; }

extern "C" int64_t tfc_conv_m_floats(const tfc_conv_plan* plan, int64_t n_edges) {
    if (!plan || n_edges < 0) return -1;
    const int64_t f = n_edges * (int64_t)plan->msz;
    return f > 0 ? f : 1;
}

extern "C" int tfc_conv_m_build(const tfc_conv_plan* pl, const float* y, int64_t n_edges, float* M,
                                tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_m_build: null plan");
    TFC_REQUIRE(n_edges >= 0 && n_edges < (1LL << 31), TFC_EINVAL, "tfc_conv_m_build: bad edge count");
    if (n_edges == 0) return TFC_OK;
    TFC_REQUIRE(y && M, TFC_EINVAL, "tfc_conv_m_build: null tensor");
    TfcProfScope prof("tfc_m_prepass_kernel", stream);
    const int threads = std::min(1024, round_up(pl->msz, 64));
    TFC_REQUIRE(pl->d_sh <= 1024, TFC_EUNSUPPORTED, "tfc_conv_m_build: %d edge-attribute columns (at most 1024)", pl->d_sh);
    tfc_m_prepass_kernel<<<grid_for(n_edges, M_EPB, 4096), threads, sizeof(float) * M_EPB * pl->d_sh, stream>>>(
        y, pl->ments, pl->cgm, (long)n_edges, pl->msz, pl->d_sh, M);
    TFC_LAUNCH_CHECK("tfc_m_prepass_kernel");
    return TFC_OK;
}

// Nodes per task group (tfc_task_split).  Forward: 16 (mild gain).  Backward: as many nodes as keep
// the column range one item gathers from the group's upstream-gradient rows near 1 MB (a quarter
// of an XCD's L2), between 16 and 256, then adjusted to divide the slab of nodes one XCD walks
// (n_nodes / 8) evenly, so that every XCD still sees every item type equally often (items differ
// in cost).  Measured: config 3 layer 2 481 -> 432 us (group 256; 512 and more lose to that
// imbalance), config 4 5.65 -> 4.95 ms (groups 27..108 within 2 %).
// TFC_CONV_GROUP="fwd,bwd" overrides for experiments.
static int conv_group(int which, long n_nodes, int n_items = 1, int d_out = 1) {
    static int g[2] = {-1, -1};
    if (g[0] < 0) {
        int a = 0, b = 0;
        if (const char* e = getenv("TFC_CONV_GROUP")) {
            if (sscanf(e, "%d,%d", &a, &b) < 2) b = a;
        }
        g[0] = a < 0 ? 0 : a;
        g[1] = b < 0 ? 0 : b;
    }
    if (g[which] > 0) return g[which];
    if (which == 0) return 16;
    const long want = std::min<long>(256, std::max<long>(16, ((long)n_items << 18) / std::max(d_out, 1)));
    const long slab = (n_nodes + 7) / 8;
    const long parts = std::max<long>(1, (slab + want / 2) / want);
    return (int)std::max<long>(1, (slab + parts - 1) / parts);
}

extern "C" int tfc_conv_fwd_m(const tfc_conv_plan* pl, const float* x, const float* M, const float* w,
                              const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                              int64_t n_edges, float scale, float* out, tfc_stream_t stream) {
    return tfc_conv_fwd_m_t(pl, x, M, w, row_ptr, eid, nbr, n_nodes, n_edges, scale, out, nullptr, stream);
}

extern "C" int tfc_conv_fwd_m_t(const tfc_conv_plan* pl, const float* x, const float* M, const float* w,
                                const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                                int64_t n_edges, float scale, float* out, float* out_t, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_fwd: null plan");
    TFC_REQUIRE(!out_t || pl->tout, TFC_EINVAL, "tfc_conv_fwd: out_t needs tfc_conv_plan_set_out_blocks first");
    TFC_REQUIRE(n_nodes >= 0 && n_edges >= 0, TFC_EINVAL, "tfc_conv_fwd: negative size");
    TFC_REQUIRE(n_nodes < (1LL << 31) && n_edges < (1LL << 31), TFC_EUNSUPPORTED,
                "tfc_conv_fwd: more than 2^31 nodes or edges");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(x && out && row_ptr, TFC_EINVAL, "tfc_conv_fwd: null tensor");
    TFC_REQUIRE(n_edges == 0 || (M && w && eid && nbr), TFC_EINVAL, "tfc_conv_fwd: null tensor");
    const long n_tasks = (long)n_nodes * pl->n_fwd_items;
    const long n_blocks = (n_tasks + 3) / 4;
    TFC_REQUIRE(n_blocks < (1LL << 31), TFC_EUNSUPPORTED, "tfc_conv_fwd: grid too large");
    {
        TfcProfScope prof("tfc_conv_fwd_kernel", stream);
        tfc_conv_fwd_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(
            x, w, M, row_ptr, eid, nbr, pl->paths, pl->fwd_items, pl->n_fwd_items, n_tasks, pl->d_in, pl->d_out,
            pl->w_numel, pl->msz, scale, out, (int)n_nodes, conv_group(0, n_nodes), pl->tout, out_t);
    }
    TFC_LAUNCH_CHECK("tfc_conv_fwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_conv_fwd_fused_m(const tfc_conv_plan* pl, const float* x, const float* M, const float* h,
                                    const float* w_last, int32_t hidden, float alpha, const int32_t* row_ptr,
                                    const int32_t* eid, const int32_t* nbr, int64_t n_nodes, int64_t n_edges,
                                    float scale, float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_fwd_fused: null plan");
    TFC_REQUIRE(hidden == 64, TFC_EUNSUPPORTED, "tfc_conv_fwd_fused: hidden width %d (built for 64)", hidden);
    TFC_REQUIRE(n_nodes >= 0 && n_edges >= 0, TFC_EINVAL, "tfc_conv_fwd_fused: negative size");
    TFC_REQUIRE(n_nodes < (1LL << 31) && n_edges < (1LL << 31), TFC_EUNSUPPORTED,
                "tfc_conv_fwd_fused: more than 2^31 nodes or edges");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(x && out && row_ptr && w_last, TFC_EINVAL, "tfc_conv_fwd_fused: null tensor");
    TFC_REQUIRE(n_edges == 0 || (M && h && eid && nbr), TFC_EINVAL, "tfc_conv_fwd_fused: null tensor");
    const long n_tasks = (long)n_nodes * pl->n_fwd_items;
    const long n_blocks = (n_tasks + 3) / 4;
    TFC_REQUIRE(n_blocks < (1LL << 31), TFC_EUNSUPPORTED, "tfc_conv_fwd_fused: grid too large");
    {
        TfcProfScope prof("tfc_conv_fwd_fused_kernel", stream);
        tfc_conv_fwd_fused_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(
            x, h, w_last, M, row_ptr, eid, nbr, pl->paths, pl->fwd_items, pl->n_fwd_items, n_tasks, pl->d_in,
            pl->d_out, pl->w_numel, pl->msz, scale, alpha, out, (int)n_nodes, conv_group(0, n_nodes));
    }
    TFC_LAUNCH_CHECK("tfc_conv_fwd_fused_kernel");
    return TFC_OK;
}

extern "C" int tfc_conv_fwd(const tfc_conv_plan* pl, const float* x, const float* y, const float* w,
                            const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                            int64_t n_edges, float scale, float* out, float* workspace, tfc_stream_t stream) {
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_fwd: null plan");
    TFC_REQUIRE(n_edges <= 0 || n_nodes <= 0 || (y && eid && workspace), TFC_EINVAL, "tfc_conv_fwd: null tensor");
    if (n_nodes > 0 && n_edges > 0) {
        const int rc = tfc_conv_m_build(pl, y, n_edges, workspace, stream);
        if (rc) return rc;
    }
    return tfc_conv_fwd_m(pl, x, workspace, w, row_ptr, eid, nbr, n_nodes, n_edges, scale, out, stream);
}

extern "C" int64_t tfc_conv_dy_floats(const tfc_conv_plan* plan, int64_t n_edges) {
    if (!plan || n_edges < 0) return -1;
    const int64_t f = n_edges * (int64_t)plan->py;
    return f > 0 ? f : 1;
}

extern "C" int tfc_conv_bwd_m(const tfc_conv_plan* pl, const float* x, const float* M, const float* w,
                              const float* gout, const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr,
                              int64_t n_nodes, int64_t n_edges, float scale, float* dx, float* dw, float* dy,
                              float* dyp, tfc_stream_t stream) {
    return tfc_conv_bwd_m_t(pl, x, M, w, gout, row_ptr, eid, nbr, n_nodes, n_edges, scale, dx, nullptr, dw, dy, dyp,
                            stream);
}

extern "C" int tfc_conv_bwd_m_t(const tfc_conv_plan* pl, const float* x, const float* M, const float* w,
                                const float* gout, const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr,
                                int64_t n_nodes, int64_t n_edges, float scale, float* dx, float* dx_t, float* dw,
                                float* dy, float* dyp, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_bwd: null plan");
    TFC_REQUIRE(n_nodes >= 0 && n_edges >= 0, TFC_EINVAL, "tfc_conv_bwd: negative size");
    TFC_REQUIRE(n_nodes < (1LL << 31) && n_edges < (1LL << 31), TFC_EUNSUPPORTED,
                "tfc_conv_bwd: more than 2^31 nodes or edges");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(x && gout && dx && row_ptr, TFC_EINVAL, "tfc_conv_bwd: null tensor");
    TFC_REQUIRE(n_edges == 0 || (M && w && dw && eid && nbr && (!dy || dyp)), TFC_EINVAL,
                "tfc_conv_bwd: null tensor");
    // columns of dx that no path reads must still be defined (none in the layers the reference builds: every
    // block of the sorted, simplified input irreps has a path with the scalar spherical harmonic)
    if (!pl->dx_covered) {
        tfc_fill_kernel<<<grid_for((long)n_nodes * pl->d_in, 256, 2048), 256, 0, stream>>>(
            dx, (long)n_nodes * pl->d_in, 0.f);
        TFC_LAUNCH_CHECK("tfc_fill_kernel");
    }
    const long n_tasks = (long)n_nodes * pl->n_bwd_items;
    const long n_blocks = (n_tasks + 3) / 4;
    TFC_REQUIRE(n_blocks < (1LL << 31), TFC_EUNSUPPORTED, "tfc_conv_bwd: grid too large");
#define TFC_BWD_LAUNCH(DY, LM)                                                                          \
    tfc_conv_bwd_kernel<DY, LM><<<(unsigned)n_blocks, 256, 0, stream>>>(                                \
        x, w, gout, M, pl->cgd, row_ptr, eid, nbr, pl->paths, pl->bwd_items, pl->grp_paths, pl->n_bwd_items, \
        n_tasks, pl->d_in, pl->d_out, pl->w_numel, pl->msz, pl->py, scale, dx, dw, dy ? dyp : nullptr,  \
        (int)n_nodes, conv_group(1, n_nodes, pl->n_bwd_items, pl->d_out), dx_t)
    {
        TfcProfScope prof("tfc_conv_bwd_kernel", stream);
        if (dy) {
            if (pl->lmax <= 2) TFC_BWD_LAUNCH(true, 2); else TFC_BWD_LAUNCH(true, 3);
        } else {
            if (pl->lmax <= 2) TFC_BWD_LAUNCH(false, 2); else TFC_BWD_LAUNCH(false, 3);
        }
    }
#undef TFC_BWD_LAUNCH
    TFC_LAUNCH_CHECK("tfc_conv_bwd_kernel");
    if (dy && n_edges > 0) {
        TfcProfScope prof("tfc_dy_reduce_kernel", stream);
        tfc_dy_reduce_kernel<<<grid_for(n_edges * (long)pl->d_sh, 256, 2048), 256, 0, stream>>>(
            dyp, pl->ylist_ptr, pl->ylist, (long)n_edges, pl->d_sh, pl->py, dy);
        TFC_LAUNCH_CHECK("tfc_dy_reduce_kernel");
    }
    return TFC_OK;
}

extern "C" int tfc_conv_bwd(const tfc_conv_plan* pl, const float* x, const float* y, const float* w,
                            const float* gout, const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr,
                            int64_t n_nodes, int64_t n_edges, float scale, float* dx, float* dw, float* dy,
                            float* workspace, tfc_stream_t stream) {
    TFC_REQUIRE(pl, TFC_EINVAL, "tfc_conv_bwd: null plan");
    TFC_REQUIRE(n_edges <= 0 || n_nodes <= 0 || (y && eid && workspace), TFC_EINVAL, "tfc_conv_bwd: null tensor");
    if (n_nodes > 0 && n_edges > 0) {
        const int rc = tfc_conv_m_build(pl, y, n_edges, workspace, stream);
        if (rc) return rc;
    }
    float* dyp = workspace ? workspace + (size_t)(n_edges > 0 ? n_edges : 0) * pl->msz : nullptr;
    return tfc_conv_bwd_m(pl, x, workspace, w, gout, row_ptr, eid, nbr, n_nodes, n_edges, scale, dx, dw, dy, dyp,
                          stream);
}
