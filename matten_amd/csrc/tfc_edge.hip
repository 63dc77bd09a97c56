// Per-edge encoders in one pass: edge vectors (+ periodic shift), real spherical harmonics
// (component normalisation, l <= 3) and the Bessel x polynomial-cutoff radial embedding.
// Replaces eigenn/nn/_nequip.py:214-266 (with_edge_vectors), :167-175 (o3.SphericalHarmonics
// with normalize=True, normalization="component") and :117-128, :63-78, :206-208 (BesselBasis,
// PolynomialCutoff, their product), plus the backward autograd derives from them.
// HBM-bound elementwise work: one thread per edge, coalesced stores; executed once per model,
// not per layer.
#include "tfc_common.h"

#include <algorithm>

namespace {

constexpr float SQ3 = 1.7320508075688772f;
constexpr float SQ5 = 2.23606797749979f;
constexpr float SQ7 = 2.6457513110645907f;
constexpr float SQ15 = 3.872983346207417f;
constexpr float A3 = 2.6457513110645907f * 0.9128709291752769f;   // sqrt(7)*sqrt(5/6)
constexpr float B3 = 5.916079783099616f;                          // sqrt(35)
constexpr float C3 = 2.6457513110645907f * 0.6123724356957945f;   // sqrt(7)*sqrt(3/8)

__device__ __forceinline__ void sh_eval(int lmax, float x, float y, float z, float* __restrict__ o) {
    o[0] = 1.f;
    if (lmax >= 1) {
        o[1] = SQ3 * x;
        o[2] = SQ3 * y;
        o[3] = SQ3 * z;
    }
    if (lmax >= 2) {
        const float y2 = y * y, x2z2 = x * x + z * z;
        const float s20 = SQ3 * x * z, s24 = 0.5f * SQ3 * (z * z - x * x);
        o[4] = SQ5 * s20;
        o[5] = SQ15 * x * y;
        o[6] = SQ5 * (y2 - 0.5f * x2z2);
        o[7] = SQ15 * y * z;
        o[8] = SQ5 * s24;
        if (lmax >= 3) {
            o[9] = A3 * (s20 * z + s24 * x);
            o[10] = B3 * s20 * y;
            o[11] = C3 * (4.f * y2 - x2z2) * x;
            o[12] = SQ7 * 0.5f * y * (2.f * y2 - 3.f * x2z2);
            o[13] = C3 * z * (4.f * y2 - x2z2);
            o[14] = B3 * s24 * y;
            o[15] = A3 * (s24 * z - s20 * x);
        }
    }
}

// gradient of sum_lm d[lm] * Y_lm(n) with respect to the (unconstrained) polynomial argument n
__device__ __forceinline__ void sh_grad(int lmax, float x, float y, float z, const float* __restrict__ d,
                                        float& gx, float& gy, float& gz) {
    gx = gy = gz = 0.f;
    if (lmax >= 1) {
        gx += SQ3 * d[1];
        gy += SQ3 * d[2];
        gz += SQ3 * d[3];
    }
    if (lmax >= 2) {
        gx += SQ15 * z * d[4] + SQ15 * y * d[5] - SQ5 * x * d[6] - SQ15 * x * d[8];
        gy += SQ15 * x * d[5] + 2.f * SQ5 * y * d[6] + SQ15 * z * d[7];
        gz += SQ15 * x * d[4] - SQ5 * z * d[6] + SQ15 * y * d[7] + SQ15 * z * d[8];
    }
    if (lmax >= 3) {
        const float a = A3 * SQ3, b = B3 * SQ3;
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        // Y30 = a (1.5 x z^2 - 0.5 x^3)
        gx += d[9] * a * 1.5f * (z2 - x2);
        gz += d[9] * a * 3.f * x * z;
        // Y31 = b x y z
        gx += d[10] * b * y * z;
        gy += d[10] * b * x * z;
        gz += d[10] * b * x * y;
        // Y32 = C3 (4 y^2 - x^2 - z^2) x
        gx += d[11] * C3 * (4.f * y2 - 3.f * x2 - z2);
        gy += d[11] * C3 * 8.f * x * y;
        gz -= d[11] * C3 * 2.f * x * z;
        // Y33 = sqrt7/2 y (2 y^2 - 3 x^2 - 3 z^2)
        gx -= d[12] * 3.f * SQ7 * x * y;
        gy += d[12] * 0.5f * SQ7 * (6.f * y2 - 3.f * (x2 + z2));
        gz -= d[12] * 3.f * SQ7 * y * z;
        // Y34 = C3 z (4 y^2 - x^2 - z^2)
        gx -= d[13] * C3 * 2.f * x * z;
        gy += d[13] * C3 * 8.f * y * z;
        gz += d[13] * C3 * (4.f * y2 - x2 - 3.f * z2);
        // Y35 = b/2 (z^2 - x^2) y
        gx -= d[14] * b * x * y;
        gy += d[14] * 0.5f * b * (z2 - x2);
        gz += d[14] * b * y * z;
        // Y36 = a (0.5 z^3 - 1.5 x^2 z)
        gx -= d[15] * a * 3.f * x * z;
        gz += d[15] * a * 1.5f * (z2 - x2);
    }
}

__global__ __launch_bounds__(256) void tfc_edge_fwd_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ ei, const float* __restrict__ shift,
    const float* __restrict__ cell, const int64_t* __restrict__ batch, long n_edges, long n_nodes, long n_graphs,
    int lmax, int sh_norm, const float* __restrict__ bw, int n_basis, float r_max, float p, float* vec,
    float* len, const float* len_in, float* __restrict__ sh, float* __restrict__ emb) {
    const int S = (lmax + 1) * (lmax + 1);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (long)gridDim.x * blockDim.x) {
        float vx, vy, vz, r;
        if (pos) {
            long i = ei[e], j = ei[n_edges + e];
            i = i < 0 ? 0 : (i >= n_nodes ? n_nodes - 1 : i);  // invalid ids are reported by tfc_csr_build
            j = j < 0 ? 0 : (j >= n_nodes ? n_nodes - 1 : j);
            vx = pos[3 * j] - pos[3 * i];
            vy = pos[3 * j + 1] - pos[3 * i + 1];
            vz = pos[3 * j + 2] - pos[3 * i + 2];
            if (shift) {
                long g = (batch && n_graphs > 1) ? batch[i] : 0;
                g = g < 0 ? 0 : (g >= n_graphs ? n_graphs - 1 : g);
                const float* c = cell + 9 * g;
                const float s0 = shift[3 * e], s1 = shift[3 * e + 1], s2 = shift[3 * e + 2];
                vx += s0 * c[0] + s1 * c[3] + s2 * c[6];
                vy += s0 * c[1] + s1 * c[4] + s2 * c[7];
                vz += s0 * c[2] + s1 * c[5] + s2 * c[8];
            }
            r = sqrtf(vx * vx + vy * vy + vz * vz);
            vec[3 * e] = vx;
            vec[3 * e + 1] = vy;
            vec[3 * e + 2] = vz;
            len[e] = r;
        } else {  // vectors (and lengths) supplied by the caller
            vx = vec[3 * e];
            vy = vec[3 * e + 1];
            vz = vec[3 * e + 2];
            r = len_in ? len_in[e] : sqrtf(vx * vx + vy * vy + vz * vz);
            if (!len_in && len) len[e] = r;
        }
        if (sh) {
            // normalize=False (edge_sh_normalize): the same polynomials at the vector itself, Y_l(v) = |v|^l Y_l(v/|v|)
            const float inv = sh_norm ? 1.f / fmaxf(r, 1e-12f) : 1.f;
            float o[16];
            sh_eval(lmax, vx * inv, vy * inv, vz * inv, o);
            float* so = sh + e * S;
            for (int q = 0; q < S; ++q) so[q] = o[q];
        }
        if (emb) {
            const float u = r / r_max;
            const float up = powf(u, p);
            float env = 1.f - 0.5f * (p + 1.f) * (p + 2.f) * up + p * (p + 2.f) * up * u -
                        0.5f * p * (p + 1.f) * up * u * u;
            env = (r < r_max) ? env : 0.f;
            const float pref = 2.f / r_max;
            float* eo = emb + e * n_basis;
            for (int n = 0; n < n_basis; ++n) eo[n] = pref * (sinf(bw[n] * r / r_max) / r) * env;
        }
    }
}

constexpr int EDGE_BWD_BLOCK = 256;
constexpr int MAX_BASIS = 64;

__global__ __launch_bounds__(EDGE_BWD_BLOCK) void tfc_edge_bwd_kernel(
    const float* __restrict__ vec, const float* __restrict__ len, long n_edges, int lmax, int sh_norm,
    const float* __restrict__ bw, int n_basis, float r_max, float p, const float* __restrict__ dsh,
    const float* __restrict__ demb, const float* __restrict__ dvec_add, const float* __restrict__ dlen,
    float* __restrict__ dvec, float* __restrict__ dbw_partial) {
    __shared__ float red[4][MAX_BASIS];
    const int S = (lmax + 1) * (lmax + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dbw_acc[MAX_BASIS > 8 ? 8 : MAX_BASIS];  // register path for n_basis <= 8
#pragma unroll
    for (int n = 0; n < 8; ++n) dbw_acc[n] = 0.f;
    // every thread runs the same trip count so that the wave reductions below see all lanes
    const long stride = (long)gridDim.x * blockDim.x;
    const long trips = (n_edges + stride - 1) / stride;
    for (long it = 0; it < trips; ++it) {
        const long e = it * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = e < n_edges;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (live) {
            const float vx = vec[3 * e], vy = vec[3 * e + 1], vz = vec[3 * e + 2];
            const float r = len ? len[e] : sqrtf(vx * vx + vy * vy + vz * vz);
            const float inv = 1.f / fmaxf(r, 1e-12f);
            const float x = vx * inv, y = vy * inv, z = vz * inv;
            if (dsh) {
                float d[16];
                const float* dr = dsh + e * S;
                for (int q = 0; q < S; ++q) d[q] = dr[q];
                float px, py, pz;
                if (sh_norm) {
                    sh_grad(lmax, x, y, z, d, px, py, pz);
                    const float dot = px * x + py * y + pz * z;
                    // d n / d v = (I - n n^T) / |v|  (F.normalize; zero vectors stay zero)
                    const float k = r > 1e-12f ? inv : 0.f;
                    gx += (px - x * dot) * k;
                    gy += (py - y * dot) * k;
                    gz += (pz - z * dot) * k;
                } else {  // polynomials of the vector itself
                    sh_grad(lmax, vx, vy, vz, d, px, py, pz);
                    gx += px;
                    gy += py;
                    gz += pz;
                }
            }
            if (demb) {
                const float u = r / r_max;
                const bool inside = r < r_max;
                const float up1 = powf(u, p - 1.f), up = up1 * u;
                const float c1 = 0.5f * (p + 1.f) * (p + 2.f), c2 = p * (p + 2.f), c3 = 0.5f * p * (p + 1.f);
                const float env = inside ? 1.f - c1 * up + c2 * up * u - c3 * up * u * u : 0.f;
                const float denv =
                    inside ? (-c1 * p * up1 + c2 * (p + 1.f) * up - c3 * (p + 2.f) * up * u) / r_max : 0.f;
                const float pref = 2.f / r_max;
                const float* de = demb + e * n_basis;
                float dr = 0.f;
                for (int n = 0; n < n_basis; ++n) {
                    const float arg = bw[n] * r / r_max;
                    float sn, cs;
                    sincosf(arg, &sn, &cs);
                    const float g = de[n];
                    const float sr = sn / r;
                    // d/dr [ pref * sin(b r / rc) / r * env(r) ]
                    dr += g * pref * (((bw[n] / r_max) * cs / r - sr / r) * env + sr * denv);
                    if (n < 8) dbw_acc[n] += g * pref * cs / r_max * env;
                }
                gx += dr * x;
                gy += dr * y;
                gz += dr * z;
            }
            if (dlen) {  // d|v| / dv = n
                const float g = dlen[e];
                gx += g * x;
                gy += g * y;
                gz += g * z;
            }
            if (dvec_add) {
                gx += dvec_add[3 * e];
                gy += dvec_add[3 * e + 1];
                gz += dvec_add[3 * e + 2];
            }
            dvec[3 * e] = gx;
            dvec[3 * e + 1] = gy;
            dvec[3 * e + 2] = gz;
        }
    }
    if (dbw_partial) {
        // fixed-shape two-stage sum: lanes -> wave (DPP), waves -> block (LDS); deterministic
        for (int n = 0; n < n_basis && n < 8; ++n) {
            const float s = tfc_wave_sum(dbw_acc[n]);
            if (lane == 0) red[wave][n] = s;
        }
        __syncthreads();
        if (threadIdx.x < n_basis && threadIdx.x < 8)
            dbw_partial[(long)blockIdx.x * n_basis + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// dpos[n] = sum_{e: ei[1,e]=n} dvec[e] - sum_{e: ei[0,e]=n} dvec[e]
__global__ __launch_bounds__(256) void tfc_edge_vec_bwd_kernel(const float* __restrict__ dvec,
                                                               const int* __restrict__ rp_src,
                                                               const int* __restrict__ eid_src,
                                                               const int* __restrict__ rp_dst,
                                                               const int* __restrict__ eid_dst, long n_nodes,
                                                               float* __restrict__ dpos) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = t / 3;
    const int c = (int)(t - 3 * n);
    if (n >= n_nodes) return;
    float s = 0.f;
    for (int k = rp_dst[n]; k < rp_dst[n + 1]; ++k) s += dvec[3 * (long)eid_dst[k] + c];
    float m = 0.f;
    for (int k = rp_src[n]; k < rp_src[n + 1]; ++k) m += dvec[3 * (long)eid_src[k] + c];
    dpos[3 * n + c] = s - m;
}

}  // namespace

extern "C" int tfc_edge_encode_fwd(const float* pos, const int64_t* edge_index, const float* shift,
                                   const float* cell, const int64_t* batch, int64_t n_edges, int64_t n_nodes,
                                   int64_t n_graphs, int32_t lmax, int32_t sh_normalize, const float* bw,
                                   int32_t n_basis, float r_max, float p, float* vec, float* len, float* sh,
                                   float* emb, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_edges >= 0 && n_nodes >= 0, TFC_EINVAL, "tfc_edge_encode_fwd: negative size");
    if (n_edges == 0) return TFC_OK;
    TFC_REQUIRE(!pos || n_nodes > 0, TFC_EINVAL, "tfc_edge_encode_fwd: edges without nodes");
    TFC_REQUIRE(vec, TFC_EINVAL, "tfc_edge_encode_fwd: null vec");
    TFC_REQUIRE(!pos || (edge_index && len), TFC_EINVAL, "tfc_edge_encode_fwd: null tensor");
    TFC_REQUIRE(!sh || (lmax >= 0 && lmax <= TFC_LMAX), TFC_EUNSUPPORTED,
                "tfc_edge_encode_fwd: lmax=%d outside 0..%d", lmax, TFC_LMAX);
    TFC_REQUIRE(!emb || (bw && n_basis > 0 && r_max > 0.f), TFC_EINVAL, "tfc_edge_encode_fwd: bad radial basis");
    TFC_REQUIRE(!shift || (cell && n_graphs >= 1), TFC_EINVAL, "tfc_edge_encode_fwd: shift without cell");
    TFC_REQUIRE(!shift || n_graphs == 1 || batch, TFC_EINVAL, "tfc_edge_encode_fwd: batched cell without batch");
    const int grid = (int)std::min<long>((n_edges + 255) / 256, 8192);
    TfcProfScope prof("tfc_edge_fwd_kernel", stream);
    tfc_edge_fwd_kernel<<<grid, 256, 0, stream>>>(pos, edge_index, shift, cell, batch, (long)n_edges,
                                                  (long)n_nodes, (long)n_graphs, sh ? lmax : 0, sh_normalize, bw, n_basis,
                                                  r_max, p, vec, pos ? len : nullptr, pos ? nullptr : len, sh,
                                                  emb);
    TFC_LAUNCH_CHECK("tfc_edge_fwd_kernel");
    return TFC_OK;
}

extern "C" int64_t tfc_edge_bwd_blocks(int64_t n_edges) {
    if (n_edges <= 0) return 1;
    return std::min<int64_t>((n_edges + EDGE_BWD_BLOCK - 1) / EDGE_BWD_BLOCK, 1024);
}

extern "C" int tfc_edge_encode_bwd(const float* vec, const float* len, int64_t n_edges, int32_t lmax,
                                   int32_t sh_normalize, const float* bw, int32_t n_basis, float r_max, float p, const float* dsh,
                                   const float* demb, const float* dvec_add, const float* dlen, float* dvec,
                                   float* dbw_partial, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_edges >= 0, TFC_EINVAL, "tfc_edge_encode_bwd: negative size");
    TFC_REQUIRE(!dsh || (lmax >= 0 && lmax <= TFC_LMAX), TFC_EUNSUPPORTED,
                "tfc_edge_encode_bwd: lmax=%d outside 0..%d", lmax, TFC_LMAX);
    TFC_REQUIRE(!demb || (bw && n_basis > 0 && n_basis <= 8 && r_max > 0.f), TFC_EUNSUPPORTED,
                "tfc_edge_encode_bwd: n_basis=%d outside 1..8", n_basis);
    TFC_REQUIRE(!dbw_partial || demb, TFC_EINVAL, "tfc_edge_encode_bwd: dbw without demb");
    const int grid = (int)tfc_edge_bwd_blocks(n_edges);
    if (n_edges == 0) {
        if (dbw_partial) TFC_HIP(hipMemsetAsync(dbw_partial, 0, sizeof(float) * n_basis, stream));
        return TFC_OK;
    }
    TFC_REQUIRE(vec && dvec, TFC_EINVAL, "tfc_edge_encode_bwd: null tensor");
    TfcProfScope prof("tfc_edge_bwd_kernel", stream);
    tfc_edge_bwd_kernel<<<grid, EDGE_BWD_BLOCK, 0, stream>>>(vec, len, (long)n_edges, dsh ? lmax : 0, sh_normalize, bw, n_basis,
                                                             r_max, p, dsh, demb, dvec_add, dlen, dvec,
                                                             dbw_partial);
    TFC_LAUNCH_CHECK("tfc_edge_bwd_kernel");
    return TFC_OK;
}

// part[n][3 a + b] = sum_{e: ei[0,e] = n} shift[e][a] * dvec[e][b]: the sender's share of d/dcell of
// vec[e] = ... + shift[e] @ cell[batch[ei[0,e]]]; fixed order (ascending edge id), no atomics
__global__ __launch_bounds__(256) void tfc_edge_cell_bwd_kernel(const float* __restrict__ dvec,
                                                                const float* __restrict__ shift,
                                                                const int* __restrict__ rp_src,
                                                                const int* __restrict__ eid_src, long n_nodes,
                                                                float* __restrict__ part) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = t / 9;
    const int c = (int)(t - 9 * n), a = c / 3, b = c - 3 * a;
    if (n >= n_nodes) return;
    float s = 0.f;
    for (int k = rp_src[n]; k < rp_src[n + 1]; ++k) {
        const long e = eid_src[k];
        s = fmaf(shift[3 * e + a], dvec[3 * e + b], s);
    }
    part[9 * n + c] = s;
}

extern "C" int tfc_edge_cell_bwd(const float* dvec, const float* shift, const int32_t* rp_src, const int32_t* eid_src,
                                 int64_t n_nodes, float* part, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_nodes >= 0, TFC_EINVAL, "tfc_edge_cell_bwd: negative size");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(dvec && shift && rp_src && eid_src && part, TFC_EINVAL, "tfc_edge_cell_bwd: null tensor");
    const long t = 9 * (long)n_nodes;
    tfc_edge_cell_bwd_kernel<<<(unsigned)((t + 255) / 256), 256, 0, stream>>>(dvec, shift, rp_src, eid_src,
                                                                              (long)n_nodes, part);
    TFC_LAUNCH_CHECK("tfc_edge_cell_bwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_edge_vec_bwd(const float* dvec, const int32_t* rp_src, const int32_t* eid_src,
                                const int32_t* rp_dst, const int32_t* eid_dst, int64_t n_nodes, float* dpos,
                                tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_nodes >= 0, TFC_EINVAL, "tfc_edge_vec_bwd: negative size");
    if (n_nodes == 0) return TFC_OK;
    TFC_REQUIRE(dvec && rp_src && eid_src && rp_dst && eid_dst && dpos, TFC_EINVAL, "tfc_edge_vec_bwd: null tensor");
    const long t = 3 * (long)n_nodes;
    tfc_edge_vec_bwd_kernel<<<(unsigned)((t + 255) / 256), 256, 0, stream>>>(dvec, rp_src, eid_src, rp_dst, eid_dst,
                                                                             (long)n_nodes, dpos);
    TFC_LAUNCH_CHECK("tfc_edge_vec_bwd_kernel");
    return TFC_OK;
}
