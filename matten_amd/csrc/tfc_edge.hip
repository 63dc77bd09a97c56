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
// Sixteen lanes per node: each walks every sixteenth edge of the node's two segments (the loads of a segment are
// then in flight together instead of one dependent pair after the other: 24 -> 6 us at config 3), and the sixteen
// partial sums are added in a fixed butterfly, so the result does not depend on the launch.
__global__ __launch_bounds__(256) void tfc_edge_vec_bwd_kernel(const float* __restrict__ dvec,
                                                               const int* __restrict__ rp_src,
                                                               const int* __restrict__ eid_src,
                                                               const int* __restrict__ rp_dst,
                                                               const int* __restrict__ eid_dst, long n_nodes,
                                                               float* __restrict__ dpos) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = t >> 4;
    const int sub = (int)(t & 15);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (n < n_nodes) {
        for (int k = rp_dst[n] + sub, e = rp_dst[n + 1]; k < e; k += 16) {
            const float* v = dvec + 3 * (long)eid_dst[k];
            s0 += v[0];
            s1 += v[1];
            s2 += v[2];
        }
        for (int k = rp_src[n] + sub, e = rp_src[n + 1]; k < e; k += 16) {
            const float* v = dvec + 3 * (long)eid_src[k];
            s0 -= v[0];
            s1 -= v[1];
            s2 -= v[2];
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {  // (all 64 lanes take part: no early return above)
        s0 += __shfl_xor(s0, o, 16);
        s1 += __shfl_xor(s1, o, 16);
        s2 += __shfl_xor(s2, o, 16);
    }
    if (n < n_nodes && sub == 0) {
        dpos[3 * n + 0] = s0;
        dpos[3 * n + 1] = s1;
        dpos[3 * n + 2] = s2;
    }
}


// ------------------------------------------------------------------------------------------
// Second order (double backward: training on forces F = -dE/dpos).  The backward above is the gradient of
//     phi(v, b) = sum_q dsh[q] Y_q(v) + sum_n demb[n] emb_n(|v|, b_n) + dlen |v|
// with respect to the edge vector v and the Bessel frequencies b.  What autograd needs from its backward is
// the product of phi's Hessian with the incoming cotangents (gv for dvec, gb for dbw) and the Jacobians of
// Y, emb and |v| applied to them.  Both are directional derivatives along (gv, gb), so the kernel evaluates
// the forward and the backward formulas once more in dual-number arithmetic, a + eps b with eps^2 = 0: the
// eps parts are exactly those products.
struct Dual {
    float a, b;
};
__device__ __forceinline__ Dual mk(float a, float b = 0.f) { return Dual{a, b}; }
__device__ __forceinline__ Dual operator+(Dual x, Dual y) { return Dual{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ Dual operator-(Dual x, Dual y) { return Dual{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ Dual operator-(Dual x) { return Dual{-x.a, -x.b}; }
__device__ __forceinline__ Dual operator*(Dual x, Dual y) { return Dual{x.a * y.a, fmaf(x.a, y.b, x.b * y.a)}; }
__device__ __forceinline__ Dual operator*(float c, Dual y) { return Dual{c * y.a, c * y.b}; }
__device__ __forceinline__ Dual operator*(Dual y, float c) { return Dual{c * y.a, c * y.b}; }
__device__ __forceinline__ Dual operator+(Dual x, float c) { return Dual{x.a + c, x.b}; }
__device__ __forceinline__ Dual operator+(float c, Dual x) { return Dual{x.a + c, x.b}; }
__device__ __forceinline__ Dual operator-(float c, Dual x) { return Dual{c - x.a, -x.b}; }
__device__ __forceinline__ Dual operator-(Dual x, float c) { return Dual{x.a - c, x.b}; }
__device__ __forceinline__ Dual operator/(Dual x, Dual y) {
    const float q = x.a / y.a;
    return Dual{q, (x.b - q * y.b) / y.a};
}
__device__ __forceinline__ Dual operator/(Dual x, float c) { return Dual{x.a / c, x.b / c}; }
__device__ __forceinline__ Dual& operator+=(Dual& x, Dual y) {
    x.a += y.a;
    x.b += y.b;
    return x;
}
__device__ __forceinline__ Dual& operator-=(Dual& x, Dual y) {
    x.a -= y.a;
    x.b -= y.b;
    return x;
}
__device__ __forceinline__ Dual dsqrt(Dual x) {
    const float s = sqrtf(x.a);
    return Dual{s, 0.5f * x.b / s};
}
__device__ __forceinline__ Dual dpow(Dual x, float p) {  // x^p, p constant
    const float q = powf(x.a, p - 1.f);
    return Dual{q * x.a, p * q * x.b};
}
__device__ __forceinline__ void dsincos(Dual x, Dual& sn, Dual& cs) {
    float s, c;
    sincosf(x.a, &s, &c);
    sn = Dual{s, c * x.b};
    cs = Dual{c, -s * x.b};
}

__device__ __forceinline__ void sh_eval_dual(int lmax, Dual x, Dual y, Dual z, Dual* __restrict__ o) {
    o[0] = mk(1.f);
    if (lmax >= 1) {
        o[1] = SQ3 * x;
        o[2] = SQ3 * y;
        o[3] = SQ3 * z;
    }
    if (lmax >= 2) {
        const Dual y2 = y * y, x2z2 = x * x + z * z;
        const Dual s20 = SQ3 * (x * z), s24 = (0.5f * SQ3) * (z * z - x * x);
        o[4] = SQ5 * s20;
        o[5] = SQ15 * (x * y);
        o[6] = SQ5 * (y2 - 0.5f * x2z2);
        o[7] = SQ15 * (y * z);
        o[8] = SQ5 * s24;
        if (lmax >= 3) {
            o[9] = A3 * (s20 * z + s24 * x);
            o[10] = B3 * (s20 * y);
            o[11] = C3 * ((4.f * y2 - x2z2) * x);
            o[12] = (SQ7 * 0.5f) * (y * (2.f * y2 - 3.f * x2z2));
            o[13] = C3 * (z * (4.f * y2 - x2z2));
            o[14] = B3 * (s24 * y);
            o[15] = A3 * (s24 * z - s20 * x);
        }
    }
}

// sh_grad in dual arithmetic (same polynomials, same coefficients d)
__device__ __forceinline__ void sh_grad_dual(int lmax, Dual x, Dual y, Dual z, const float* __restrict__ d, Dual& gx,
                                             Dual& gy, Dual& gz) {
    gx = gy = gz = mk(0.f);
    if (lmax >= 1) {
        gx = gx + SQ3 * d[1];
        gy = gy + SQ3 * d[2];
        gz = gz + SQ3 * d[3];
    }
    if (lmax >= 2) {
        gx += (SQ15 * d[4]) * z + (SQ15 * d[5]) * y - (SQ5 * d[6]) * x - (SQ15 * d[8]) * x;
        gy += (SQ15 * d[5]) * x + (2.f * SQ5 * d[6]) * y + (SQ15 * d[7]) * z;
        gz += (SQ15 * d[4]) * x - (SQ5 * d[6]) * z + (SQ15 * d[7]) * y + (SQ15 * d[8]) * z;
    }
    if (lmax >= 3) {
        const float a = A3 * SQ3, b = B3 * SQ3;
        const Dual x2 = x * x, y2 = y * y, z2 = z * z;
        gx += (d[9] * a * 1.5f) * (z2 - x2);
        gz += (d[9] * a * 3.f) * (x * z);
        gx += (d[10] * b) * (y * z);
        gy += (d[10] * b) * (x * z);
        gz += (d[10] * b) * (x * y);
        gx += (d[11] * C3) * (4.f * y2 - 3.f * x2 - z2);
        gy += (d[11] * C3 * 8.f) * (x * y);
        gz -= (d[11] * C3 * 2.f) * (x * z);
        gx -= (d[12] * 3.f * SQ7) * (x * y);
        gy += (d[12] * 0.5f * SQ7) * (6.f * y2 - 3.f * (x2 + z2));
        gz -= (d[12] * 3.f * SQ7) * (y * z);
        gx -= (d[13] * C3 * 2.f) * (x * z);
        gy += (d[13] * C3 * 8.f) * (y * z);
        gz += (d[13] * C3) * (4.f * y2 - x2 - 3.f * z2);
        gx -= (d[14] * b) * (x * y);
        gy += (d[14] * 0.5f * b) * (z2 - x2);
        gz += (d[14] * b) * (y * z);
        gx -= (d[15] * a * 3.f) * (x * z);
        gz += (d[15] * a * 1.5f) * (z2 - x2);
    }
}

__global__ __launch_bounds__(EDGE_BWD_BLOCK) void tfc_edge_bwd2_kernel(
    const float* __restrict__ vec, long n_edges, int lmax, int sh_norm, const float* __restrict__ bw, int n_basis,
    float r_max, float p, const float* __restrict__ dsh, const float* __restrict__ demb,
    const float* __restrict__ dlen, const float* __restrict__ gv, const float* __restrict__ gb,
    float* __restrict__ g_vec, float* __restrict__ g_dsh, float* __restrict__ g_demb, float* __restrict__ g_dlen,
    float* __restrict__ g_bw_partial) {
    __shared__ float red[4][8];
    const int S = (lmax + 1) * (lmax + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gbw_acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) gbw_acc[n] = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    const long trips = (n_edges + stride - 1) / stride;
    for (long it = 0; it < trips; ++it) {
        const long e = it * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (e >= n_edges) continue;
        const Dual vx = mk(vec[3 * e], gv ? gv[3 * e] : 0.f), vy = mk(vec[3 * e + 1], gv ? gv[3 * e + 1] : 0.f),
                   vz = mk(vec[3 * e + 2], gv ? gv[3 * e + 2] : 0.f);
        const Dual r2 = vx * vx + vy * vy + vz * vz;
        const bool tiny = r2.a <= 1e-24f;  // F.normalize leaves a zero vector at zero: no dependence on it
        const Dual r = tiny ? mk(1e-12f) : dsqrt(r2);
        const Dual inv = mk(1.f) / r;
        const Dual x = vx * inv, y = vy * inv, z = vz * inv;
        Dual Gx = mk(0.f), Gy = mk(0.f), Gz = mk(0.f);
        if (dsh) {
            float d[16];
            const float* dr = dsh + e * S;
            for (int q = 0; q < S; ++q) d[q] = dr[q];
            Dual o[16], px, py, pz;
            if (sh_norm) {
                sh_eval_dual(lmax, x, y, z, o);
                sh_grad_dual(lmax, x, y, z, d, px, py, pz);
                const Dual dot = px * x + py * y + pz * z;
                if (!tiny) {
                    Gx += (px - x * dot) * inv;
                    Gy += (py - y * dot) * inv;
                    Gz += (pz - z * dot) * inv;
                }
            } else {
                sh_eval_dual(lmax, vx, vy, vz, o);
                sh_grad_dual(lmax, vx, vy, vz, d, px, py, pz);
                Gx += px;
                Gy += py;
                Gz += pz;
            }
            if (g_dsh)
                for (int q = 0; q < S; ++q) g_dsh[e * S + q] = o[q].b;
        }
        if (demb) {
            const Dual u = r / r_max;
            const bool inside = r.a < r_max;
            const Dual up1 = dpow(u, p - 1.f), up = up1 * u;
            const float c1 = 0.5f * (p + 1.f) * (p + 2.f), c2 = p * (p + 2.f), c3 = 0.5f * p * (p + 1.f);
            const Dual env = inside ? 1.f - c1 * up + c2 * (up * u) - c3 * (up * u * u) : mk(0.f);
            const Dual denv = inside ? ((-c1 * p) * up1 + (c2 * (p + 1.f)) * up - (c3 * (p + 2.f)) * (up * u)) / r_max : mk(0.f);
            const float pref = 2.f / r_max;
            const float* de = demb + e * n_basis;
            Dual dr = mk(0.f);
            for (int n = 0; n < n_basis; ++n) {
                const Dual b = mk(bw[n], gb ? gb[n] : 0.f);
                Dual sn, cs;
                dsincos(b * r / r_max, sn, cs);
                const Dual sr = sn / r;
                const float g = de[n];
                if (g_demb) g_demb[e * n_basis + n] = (pref * (sr * env)).b;
                dr += (g * pref) * (((b / r_max) * cs / r - sr / r) * env + sr * denv);
                if (n < 8) gbw_acc[n] += ((g * pref / r_max) * (cs * env)).b;
            }
            Gx += dr * x;
            Gy += dr * y;
            Gz += dr * z;
        }
        if (dlen) {
            const float g = dlen[e];
            Gx += g * x;
            Gy += g * y;
            Gz += g * z;
        }
        if (g_dlen) g_dlen[e] = tiny ? 0.f : r.b;
        if (g_vec) {
            g_vec[3 * e] = Gx.b;
            g_vec[3 * e + 1] = Gy.b;
            g_vec[3 * e + 2] = Gz.b;
        }
    }
    if (g_bw_partial) {
        for (int n = 0; n < n_basis && n < 8; ++n) {
            const float s = tfc_wave_sum(gbw_acc[n]);
            if (lane == 0) red[wave][n] = s;
        }
        __syncthreads();
        if (threadIdx.x < n_basis && threadIdx.x < 8)
            g_bw_partial[(long)blockIdx.x * n_basis + threadIdx.x] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
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
        if (dbw_partial) { const int zrc = tfc_zero_async(dbw_partial, sizeof(float) * n_basis, stream); if (zrc != TFC_OK) return zrc; }
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
    const long t = 16 * (long)n_nodes;  // sixteen lanes per node
    tfc_edge_vec_bwd_kernel<<<(unsigned)((t + 255) / 256), 256, 0, stream>>>(dvec, rp_src, eid_src, rp_dst, eid_dst,
                                                                             (long)n_nodes, dpos);
    TFC_LAUNCH_CHECK("tfc_edge_vec_bwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_edge_encode_bwd2(const float* vec, int64_t n_edges, int32_t lmax, int32_t sh_normalize,
                                    const float* bw, int32_t n_basis, float r_max, float p, const float* dsh,
                                    const float* demb, const float* dlen, const float* gv, const float* gb,
                                    float* g_vec, float* g_dsh, float* g_demb, float* g_dlen, float* g_bw_partial,
                                    tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_edges >= 0, TFC_EINVAL, "tfc_edge_encode_bwd2: negative size");
    TFC_REQUIRE(!dsh || (lmax >= 0 && lmax <= TFC_LMAX), TFC_EUNSUPPORTED, "tfc_edge_encode_bwd2: lmax=%d outside 0..%d",
                lmax, TFC_LMAX);
    TFC_REQUIRE(!demb || (bw && n_basis > 0 && n_basis <= 8 && r_max > 0.f), TFC_EUNSUPPORTED,
                "tfc_edge_encode_bwd2: n_basis=%d outside 1..8", n_basis);
    TFC_REQUIRE((!g_dsh || dsh) && (!g_demb || demb) && (!g_bw_partial || demb), TFC_EINVAL,
                "tfc_edge_encode_bwd2: a cotangent is asked for an input that was not given");
    const int grid = (int)tfc_edge_bwd_blocks(n_edges);
    if (n_edges == 0) {
        if (g_bw_partial) { const int zrc = tfc_zero_async(g_bw_partial, sizeof(float) * n_basis, stream); if (zrc != TFC_OK) return zrc; }
        return TFC_OK;
    }
    TFC_REQUIRE(vec, TFC_EINVAL, "tfc_edge_encode_bwd2: null tensor");
    TfcProfScope prof("tfc_edge_bwd2_kernel", stream);
    tfc_edge_bwd2_kernel<<<grid, EDGE_BWD_BLOCK, 0, stream>>>(vec, (long)n_edges, dsh ? lmax : 0, sh_normalize, bw, n_basis,
                                                              r_max, p, dsh, demb, dlen, gv, gb, g_vec, g_dsh, g_demb,
                                                              g_dlen, g_bw_partial);
    TFC_LAUNCH_CHECK("tfc_edge_bwd2_kernel");
    return TFC_OK;
}
