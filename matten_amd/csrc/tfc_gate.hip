// Gate non-linearity: out = [ act(scalars) | act(gates) * gated ].
// Replaces e3nn.nn.Gate as built at eigenn/nn/utils.py:134-140 and applied at
// eigenn/nn/point_conv.py:339 (Activation with normalize2mom constants, then an
// ElementwiseTensorProduct that is a plain multiply), and its autograd backward.
// Elementwise and HBM-bound: one thread per output (forward) / input (backward) column.
#include "tfc_common.h"

#include <algorithm>

namespace {

constexpr int MAX_SEGS = 32;

struct GateSegs {
    tfc_gate_seg s[MAX_SEGS];
    int n;
};

__device__ __forceinline__ float act_eval(int act, float x) {
    switch (act) {
        case TFC_ACT_SSP: return (x > 20.f ? x : log1pf(expf(x))) - 0.69314718055994530942f;
        case TFC_ACT_SILU: return x / (1.f + expf(-x));
        case TFC_ACT_TANH: return tanhf(x);
        case TFC_ACT_ABS: return fabsf(x);
        case TFC_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}

__device__ __forceinline__ float act_grad(int act, float x) {
    switch (act) {
        case TFC_ACT_SSP: return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
        case TFC_ACT_SILU: {
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        case TFC_ACT_TANH: {
            const float t = tanhf(x);
            return 1.f - t * t;
        }
        case TFC_ACT_ABS: return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        case TFC_ACT_SIGMOID: {
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f - s);
        }
        default: return 1.f;
    }
}

__device__ __forceinline__ float act_grad2(int act, float x) {  // second derivative
    switch (act) {
        case TFC_ACT_SSP: {
            if (x > 20.f) return 0.f;
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f - s);
        }
        case TFC_ACT_SILU: {
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f - s) * (2.f + x * (1.f - 2.f * s));
        }
        case TFC_ACT_TANH: {
            const float t = tanhf(x);
            return -2.f * t * (1.f - t * t);
        }
        case TFC_ACT_SIGMOID: {
            const float s = 1.f / (1.f + expf(-x));
            return s * (1.f - s) * (1.f - 2.f * s);
        }
        default: return 0.f;  // identity, abs
    }
}

// Second, channel-contiguous copy of a result row (the layout of tfc_ir_transpose): element c of row r, inside
// block (off, mul, dim), goes to t[off * n_rows + (m * n_rows + r) * mul + u], c = off + u * dim + m.
struct GateBlocks {
    tfc_ir_block b[MAX_SEGS];
    int n;
};

__device__ __forceinline__ void store_transposed(const GateBlocks& bl, long r, int c, long n_rows, float v,
                                                 float* __restrict__ t) {
    for (int k = 0; k < bl.n; ++k) {
        const tfc_ir_block b = bl.b[k];
        const int idx = c - b.off;
        if (idx < 0 || idx >= b.mul * b.dim) continue;
        const int u = idx / b.dim, m = idx - u * b.dim;
        t[(long)b.off * n_rows + ((long)m * n_rows + r) * b.mul + u] = v;
        return;
    }
}

template <bool T>
__global__ __launch_bounds__(256) void tfc_gate_fwd_kernel(const GateSegs g, const float* __restrict__ x,
                                                           long n_rows, int d_in, int d_out,
                                                           float* __restrict__ out, const GateBlocks bl,
                                                           float* __restrict__ out_t) {
    const long total = n_rows * (long)d_out;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / d_out;
        const int c = (int)(i - r * d_out);
        const float* xr = x + r * (long)d_in;
        float v = 0.f;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg sg = g.s[s];
            const int idx = c - sg.out_off;
            if (idx < 0 || idx >= sg.count * sg.dim) continue;
            if (sg.gate_off < 0) {
                v = sg.cst * act_eval(sg.act, xr[sg.in_off + idx]);
            } else {
                const int u = idx / sg.dim;
                v = sg.cst * act_eval(sg.act, xr[sg.gate_off + u]) * xr[sg.in_off + idx];
            }
            break;
        }
        out[i] = v;
        if (T) store_transposed(bl, r, c, n_rows, v, out_t);
    }
}

template <bool T>
__global__ __launch_bounds__(256) void tfc_gate_bwd_kernel(const GateSegs g, const float* __restrict__ x,
                                                           const float* __restrict__ gout, long n_rows, int d_in,
                                                           int d_out, float* __restrict__ dx, const GateBlocks bl,
                                                           float* __restrict__ dx_t) {
    const long total = n_rows * (long)d_in;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / d_in;
        const int c = (int)(i - r * d_in);
        const float* xr = x + r * (long)d_in;
        const float* gr = gout + r * (long)d_out;
        float v = 0.f;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg sg = g.s[s];
            const int idx = c - sg.in_off;
            if (idx >= 0 && idx < sg.count * sg.dim) {
                if (sg.gate_off < 0) {
                    v = gr[sg.out_off + idx] * sg.cst * act_grad(sg.act, xr[c]);
                } else {
                    const int u = idx / sg.dim;
                    v = gr[sg.out_off + idx] * sg.cst * act_eval(sg.act, xr[sg.gate_off + u]);
                }
                break;
            }
            const int gu = c - sg.gate_off;
            if (sg.gate_off >= 0 && gu >= 0 && gu < sg.count) {
                float dot = 0.f;
                for (int m = 0; m < sg.dim; ++m)
                    dot = fmaf(gr[sg.out_off + gu * sg.dim + m], xr[sg.in_off + gu * sg.dim + m], dot);
                v = dot * sg.cst * act_grad(sg.act, xr[c]);
                break;
            }
        }
        dx[i] = v;
        if (T) store_transposed(bl, r, c, n_rows, v, dx_t);
    }
}

// The same two passes walking the rows copy by copy (one thread per (row, u) of a block, like tfc_ir_transpose):
// the 2l+1 values of a copy are read and written as one contiguous run, the gate's activation is evaluated once
// per copy, and the channel-contiguous copy gets one coalesced store per plane.  Needs blocks that tile the row.
__device__ __forceinline__ bool find_copy(const GateBlocks& bl, int j, tfc_ir_block& b, int& u) {
    for (int k = 0; k < bl.n; ++k) {
        if (j < bl.b[k].mul) {
            b = bl.b[k];
            u = j;
            return true;
        }
        j -= bl.b[k].mul;
    }
    return false;
}

constexpr int GATE_MAXD = 2 * TFC_LMAX + 1;

__global__ __launch_bounds__(256) void tfc_gate_fwd_copies_kernel(const GateSegs g, const float* __restrict__ x,
                                                                  long n_rows, int d_in, int d_out, int copies,
                                                                  float* __restrict__ out, const GateBlocks bl,
                                                                  float* __restrict__ out_t) {
    const long total = n_rows * (long)copies;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / copies;
        tfc_ir_block b;
        int u;
        if (!find_copy(bl, (int)(i - r * copies), b, u)) continue;
        const int c0 = b.off + u * b.dim;
        const float* xr = x + r * (long)d_in;
        float v[GATE_MAXD];
#pragma unroll
        for (int m = 0; m < GATE_MAXD; ++m) v[m] = 0.f;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg sg = g.s[s];
            const int idx = c0 - sg.out_off;
            if (idx < 0 || idx >= sg.count * sg.dim) continue;
            if (sg.gate_off < 0) {
                v[0] = sg.cst * act_eval(sg.act, xr[sg.in_off + idx]);
            } else {
                const float a = sg.cst * act_eval(sg.act, xr[sg.gate_off + idx / sg.dim]);
#pragma unroll
                for (int m = 0; m < GATE_MAXD; ++m)
                    if (m < b.dim) v[m] = a * xr[sg.in_off + idx + m];
            }
            break;
        }
        float* o = out + r * (long)d_out + c0;
        float* t = out_t + (long)b.off * n_rows + r * b.mul + u;
        const long plane = n_rows * (long)b.mul;
#pragma unroll
        for (int m = 0; m < GATE_MAXD; ++m)
            if (m < b.dim) {
                o[m] = v[m];
                t[m * plane] = v[m];
            }
    }
}

__global__ __launch_bounds__(256) void tfc_gate_bwd_copies_kernel(const GateSegs g, const float* __restrict__ x,
                                                                  const float* __restrict__ gout, long n_rows,
                                                                  int d_in, int d_out, int copies,
                                                                  float* __restrict__ dx, const GateBlocks bl,
                                                                  float* __restrict__ dx_t) {
    const long total = n_rows * (long)copies;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / copies;
        tfc_ir_block b;
        int u;
        if (!find_copy(bl, (int)(i - r * copies), b, u)) continue;
        const int c0 = b.off + u * b.dim;
        const float* xr = x + r * (long)d_in;
        const float* gr = gout + r * (long)d_out;
        float v[GATE_MAXD];
#pragma unroll
        for (int m = 0; m < GATE_MAXD; ++m) v[m] = 0.f;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg sg = g.s[s];
            const int idx = c0 - sg.in_off;
            if (idx >= 0 && idx < sg.count * sg.dim) {
                if (sg.gate_off < 0) {
                    v[0] = gr[sg.out_off + idx] * sg.cst * act_grad(sg.act, xr[c0]);
                } else {
                    const float a = act_eval(sg.act, xr[sg.gate_off + idx / sg.dim]);
#pragma unroll
                    for (int m = 0; m < GATE_MAXD; ++m)
                        if (m < b.dim) v[m] = gr[sg.out_off + idx + m] * sg.cst * a;  // (same rounding as per column)
                }
                break;
            }
            const int gu = c0 - sg.gate_off;
            if (sg.gate_off >= 0 && gu >= 0 && gu < sg.count) {  // a gate scalar (its block has dim 1)
                float dot = 0.f;
                for (int m = 0; m < sg.dim; ++m)
                    dot = fmaf(gr[sg.out_off + gu * sg.dim + m], xr[sg.in_off + gu * sg.dim + m], dot);
                v[0] = dot * sg.cst * act_grad(sg.act, xr[c0]);
                break;
            }
        }
        float* o = dx + r * (long)d_in + c0;
        float* t = dx_t + (long)b.off * n_rows + r * b.mul + u;
        const long plane = n_rows * (long)b.mul;
#pragma unroll
        for (int m = 0; m < GATE_MAXD; ++m)
            if (m < b.dim) {
                o[m] = v[m];
                t[m * plane] = v[m];
            }
    }
}

// Second order: the backward above is dx = J(x)^T gout.  Given a cotangent v of dx, autograd needs
//   ggout = J(x) v                     (one thread per output column), and
//   gx    = d/dx ( v^T J(x)^T gout )   (one thread per input column).
template <bool OUT_SIDE>
__global__ __launch_bounds__(256) void tfc_gate_bwd2_kernel(const GateSegs g, const float* __restrict__ x,
                                                            const float* __restrict__ gout,
                                                            const float* __restrict__ v, long n_rows, int d_in,
                                                            int d_out, float* __restrict__ res) {
    const int d = OUT_SIDE ? d_out : d_in;
    const long total = n_rows * (long)d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / d;
        const int c = (int)(i - r * d);
        const float* xr = x + r * (long)d_in;
        const float* vr = v + r * (long)d_in;
        const float* gr = gout + r * (long)d_out;
        float o = 0.f;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg sg = g.s[s];
            if (OUT_SIDE) {
                const int idx = c - sg.out_off;
                if (idx < 0 || idx >= sg.count * sg.dim) continue;
                if (sg.gate_off < 0) {
                    o = vr[sg.in_off + idx] * sg.cst * act_grad(sg.act, xr[sg.in_off + idx]);
                } else {
                    const int u = idx / sg.dim;
                    const float t = xr[sg.gate_off + u];
                    o = sg.cst * (vr[sg.in_off + idx] * act_eval(sg.act, t) +
                                  vr[sg.gate_off + u] * act_grad(sg.act, t) * xr[sg.in_off + idx]);
                }
                break;
            } else {
                const int idx = c - sg.in_off;
                if (idx >= 0 && idx < sg.count * sg.dim) {
                    if (sg.gate_off < 0) {
                        o = vr[c] * gr[sg.out_off + idx] * sg.cst * act_grad2(sg.act, xr[c]);
                    } else {
                        const int u = idx / sg.dim;
                        o = vr[sg.gate_off + u] * sg.cst * act_grad(sg.act, xr[sg.gate_off + u]) * gr[sg.out_off + idx];
                    }
                    break;
                }
                const int gu = c - sg.gate_off;
                if (sg.gate_off >= 0 && gu >= 0 && gu < sg.count) {
                    float vg = 0.f, gx = 0.f;
                    for (int m = 0; m < sg.dim; ++m) {
                        const float gg = gr[sg.out_off + gu * sg.dim + m];
                        vg = fmaf(vr[sg.in_off + gu * sg.dim + m], gg, vg);
                        gx = fmaf(gg, xr[sg.in_off + gu * sg.dim + m], gx);
                    }
                    o = sg.cst * (act_grad(sg.act, xr[c]) * vg + vr[c] * act_grad2(sg.act, xr[c]) * gx);
                    break;
                }
            }
        }
        res[i] = o;
    }
}

// Norm activation (e3nn.nn.NormActivation, eigenn/nn/utils.py:142-150): per irrep copy (u of a
// block), n = sqrt(max(sum_m x^2, eps^2)), out = act(n) [/ n] * x.  One thread per column; the
// 2l+1 values of its copy are re-read from cache.
__device__ __forceinline__ bool norm_find(const GateSegs& g, int c, tfc_gate_seg& sg, int& base) {
    for (int s = 0; s < g.n; ++s) {
        const int idx = c - g.s[s].in_off;
        if (idx >= 0 && idx < g.s[s].count * g.s[s].dim) {
            sg = g.s[s];
            base = sg.in_off + idx / sg.dim * sg.dim;
            return true;
        }
    }
    return false;
}

template <bool BWD>
__global__ __launch_bounds__(256) void tfc_norm_act_kernel(const GateSegs g, const float* __restrict__ x,
                                                           const float* __restrict__ gout, long n_rows, int d,
                                                           float eps2, int normalize, float* __restrict__ out) {
    const long total = n_rows * (long)d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long r = i / d;
        const int c = (int)(i - r * d);
        tfc_gate_seg sg;
        int base;
        float v = 0.f;
        if (norm_find(g, c, sg, base)) {
            const float* xr = x + r * (long)d + base;
            float n2 = 0.f, dot = 0.f;
            for (int m = 0; m < sg.dim; ++m) {
                n2 = fmaf(xr[m], xr[m], n2);
                if (BWD) dot = fmaf(gout[r * (long)d + base + m], xr[m], dot);
            }
            const bool clamped = eps2 > 0.f && n2 < eps2;
            // without epsilon e3nn takes the norm itself (its gradient at 0 is undefined there too)
            const float n = sqrtf(clamped ? eps2 : n2);
            const float a = act_eval(sg.act, n);
            const float scale = normalize ? a / n : a;
            if (!BWD) {
                v = scale * x[i];
            } else {
                v = scale * gout[i];
                if (!clamped) {
                    const float da = act_grad(sg.act, n);
                    // d scale / d n, times d n / d x[c] = x[c] / n
                    const float ds = normalize ? (da * n - a) / (n * n) : da;
                    v = fmaf(dot * ds / n, x[i], v);
                }
            }
        }
        out[i] = v;
    }
}

// Second order of the norm activation.  Per copy out = s(n) x with n = |x| and s = act(n) [/ n], so the Jacobian
// J = s I + q x x^T (q = s'(n) / n) is symmetric: the cotangent of gout under a cotangent v of dx is J v, i.e. the
// first-order backward applied to v (no kernel of its own), and the cotangent of x is
//   d/dx ( v^T J g ) = ( q (v.g) + q'(n)/n (x.v)(x.g) ) x + q ( (x.g) v + (x.v) g ),
// zero for a clamped copy (its scale is the constant act(eps) [/ eps]).
__global__ __launch_bounds__(256) void tfc_norm_act_bwd2_kernel(const GateSegs g, const float* __restrict__ x,
                                                                const float* __restrict__ gout, const float* __restrict__ v,
                                                                long n_rows, int d, float eps2, int normalize,
                                                                float* __restrict__ gx) {
    const long total = n_rows * (long)d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / d;
        const int c = (int)(i - r * d);
        tfc_gate_seg sg;
        int base;
        float o = 0.f;
        if (norm_find(g, c, sg, base)) {
            const long b = r * (long)d + base;
            float n2 = 0.f, xg = 0.f, xv = 0.f, vg = 0.f;
            for (int m = 0; m < sg.dim; ++m) {
                const float xm = x[b + m], gm = gout[b + m], vm = v[b + m];
                n2 = fmaf(xm, xm, n2);
                xg = fmaf(xm, gm, xg);
                xv = fmaf(xm, vm, xv);
                vg = fmaf(vm, gm, vg);
            }
            if (!(eps2 > 0.f && n2 < eps2)) {
                const float n = sqrtf(n2);
                const float a = act_eval(sg.act, n), a1 = act_grad(sg.act, n), a2 = act_grad2(sg.act, n);
                const float s1 = normalize ? (a1 * n - a) / (n * n) : a1;
                const float s2 = normalize ? (a2 * n * n - 2.f * a1 * n + 2.f * a) / (n * n * n) : a2;
                const float q = s1 / n, qp = (s2 * n - s1) / (n * n);
                o = (q * vg + qp / n * xv * xg) * x[i] + q * (xg * v[i] + xv * gout[i]);
            }
        }
        gx[i] = o;
    }
}

int pack_norm_segs(const tfc_gate_seg* segs, int32_t n_segs, int32_t d, GateSegs* out) {
    TFC_REQUIRE(segs && n_segs > 0 && n_segs <= MAX_SEGS, TFC_EINVAL, "tfc_norm_act: n_segs=%d outside 1..%d", n_segs,
                MAX_SEGS);
    for (int i = 0; i < n_segs; ++i) {
        const tfc_gate_seg& s = segs[i];
        TFC_REQUIRE(s.count >= 0 && s.dim >= 1 && s.dim <= 2 * TFC_LMAX + 1 && s.in_off >= 0 && s.in_off == s.out_off &&
                        s.in_off + s.count * s.dim <= d && s.act >= TFC_ACT_IDENTITY && s.act <= TFC_ACT_SIGMOID,
                    TFC_EINVAL, "tfc_norm_act: segment %d is inconsistent", i);
        out->s[i] = s;
    }
    out->n = n_segs;
    return TFC_OK;
}

int pack_segs(const tfc_gate_seg* segs, int32_t n_segs, int32_t d_in, int32_t d_out, GateSegs* out) {
    TFC_REQUIRE(segs && n_segs > 0 && n_segs <= MAX_SEGS, TFC_EINVAL, "tfc_gate: n_segs=%d outside 1..%d", n_segs,
                MAX_SEGS);
    for (int i = 0; i < n_segs; ++i) {
        const tfc_gate_seg& s = segs[i];
        TFC_REQUIRE(s.count >= 0 && s.dim >= 1 && s.in_off >= 0 && s.out_off >= 0 &&
                        s.in_off + s.count * s.dim <= d_in && s.out_off + s.count * s.dim <= d_out &&
                        (s.gate_off < 0 || s.gate_off + s.count <= d_in) && (s.gate_off >= 0 || s.dim == 1) &&
                        s.act >= TFC_ACT_IDENTITY && s.act <= TFC_ACT_SIGMOID,
                    TFC_EINVAL, "tfc_gate: segment %d is inconsistent", i);
        out->s[i] = s;
    }
    out->n = n_segs;
    return TFC_OK;
}

int pack_blocks(const tfc_ir_block* blocks, int32_t n_blocks, int32_t d_row, const float* t, GateBlocks* out) {
    out->n = 0;
    if (!t) return TFC_OK;
    TFC_REQUIRE(blocks && n_blocks > 0 && n_blocks <= MAX_SEGS, TFC_EINVAL, "tfc_gate: n_blocks=%d outside 1..%d",
                n_blocks, MAX_SEGS);
    for (int i = 0; i < n_blocks; ++i) {
        const tfc_ir_block& b = blocks[i];
        TFC_REQUIRE(b.off >= 0 && b.mul > 0 && b.dim > 0 && (long)b.off + (long)b.mul * b.dim <= d_row, TFC_EINVAL,
                    "tfc_gate: block %d outside the row", i);
        out->b[i] = b;
    }
    out->n = n_blocks;
    return TFC_OK;
}

// copies per row when the blocks, in the order given, tile [0, d_row) with copies of at most GATE_MAXD values and no
// segment boundary falls inside a copy; 0 otherwise (the per-column kernels then do the job)
int tiling_copies(const GateBlocks& bl, int d_row, const GateSegs& g, bool in_side) {
    int at = 0, copies = 0;
    for (int k = 0; k < bl.n; ++k) {
        const tfc_ir_block& b = bl.b[k];
        if (b.off != at || b.dim > GATE_MAXD) return 0;
        at += b.mul * b.dim;
        copies += b.mul;
        for (int s = 0; s < g.n; ++s) {
            const tfc_gate_seg& sg = g.s[s];
            const int so = in_side ? sg.in_off : sg.out_off;
            const bool overlap = so < b.off + b.mul * b.dim && so + sg.count * sg.dim > b.off;
            if (overlap && sg.count > 0 && (sg.dim != b.dim || (so - b.off) % b.dim != 0)) return 0;
            if (in_side && sg.gate_off >= 0 && sg.count > 0 && sg.gate_off < b.off + b.mul * b.dim &&
                sg.gate_off + sg.count > b.off && b.dim != 1)
                return 0;
        }
    }
    return at == d_row ? copies : 0;
}

}  // namespace

extern "C" int tfc_gate_fwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows, int32_t d_in,
                            int32_t d_out, float* out, tfc_stream_t stream) {
    return tfc_gate_fwd_t(segs, n_segs, x, n_rows, d_in, d_out, out, nullptr, 0, nullptr, stream);
}

extern "C" int tfc_gate_fwd_t(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows, int32_t d_in,
                              int32_t d_out, float* out, const tfc_ir_block* blocks, int32_t n_blocks, float* out_t,
                              tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_in > 0 && d_out > 0, TFC_EINVAL, "tfc_gate_fwd: bad size");
    GateSegs g;
    int rc = pack_segs(segs, n_segs, d_in, d_out, &g);
    if (rc) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && out, TFC_EINVAL, "tfc_gate_fwd: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d_out + 255) / 256, 8192);
    GateBlocks bl;
    rc = pack_blocks(blocks, n_blocks, d_out, out_t, &bl);
    if (rc) return rc;
    TfcProfScope prof("tfc_gate_fwd_kernel", stream);
    const int copies = out_t ? tiling_copies(bl, d_out, g, false) : 0;
    if (copies > 0) {
        const int cgrid = (int)std::min<long>((n_rows * (long)copies + 255) / 256, 8192);
        tfc_gate_fwd_copies_kernel<<<cgrid, 256, 0, stream>>>(g, x, (long)n_rows, d_in, d_out, copies, out, bl, out_t);
    } else if (out_t)
        tfc_gate_fwd_kernel<true><<<grid, 256, 0, stream>>>(g, x, (long)n_rows, d_in, d_out, out, bl, out_t);
    else
        tfc_gate_fwd_kernel<false><<<grid, 256, 0, stream>>>(g, x, (long)n_rows, d_in, d_out, out, bl, nullptr);
    TFC_LAUNCH_CHECK("tfc_gate_fwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_gate_bwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                            int64_t n_rows, int32_t d_in, int32_t d_out, float* dx, tfc_stream_t stream) {
    return tfc_gate_bwd_t(segs, n_segs, x, gout, n_rows, d_in, d_out, dx, nullptr, 0, nullptr, stream);
}

extern "C" int tfc_gate_bwd_t(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                              int64_t n_rows, int32_t d_in, int32_t d_out, float* dx, const tfc_ir_block* blocks,
                              int32_t n_blocks, float* dx_t, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_in > 0 && d_out > 0, TFC_EINVAL, "tfc_gate_bwd: bad size");
    GateSegs g;
    int rc = pack_segs(segs, n_segs, d_in, d_out, &g);
    if (rc) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && gout && dx, TFC_EINVAL, "tfc_gate_bwd: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d_in + 255) / 256, 8192);
    GateBlocks bl;
    rc = pack_blocks(blocks, n_blocks, d_in, dx_t, &bl);
    if (rc) return rc;
    TfcProfScope prof("tfc_gate_bwd_kernel", stream);
    const int copies = dx_t ? tiling_copies(bl, d_in, g, true) : 0;
    if (copies > 0) {
        const int cgrid = (int)std::min<long>((n_rows * (long)copies + 255) / 256, 8192);
        tfc_gate_bwd_copies_kernel<<<cgrid, 256, 0, stream>>>(g, x, gout, (long)n_rows, d_in, d_out, copies, dx, bl, dx_t);
    } else if (dx_t)
        tfc_gate_bwd_kernel<true><<<grid, 256, 0, stream>>>(g, x, gout, (long)n_rows, d_in, d_out, dx, bl, dx_t);
    else
        tfc_gate_bwd_kernel<false><<<grid, 256, 0, stream>>>(g, x, gout, (long)n_rows, d_in, d_out, dx, bl, nullptr);
    TFC_LAUNCH_CHECK("tfc_gate_bwd_kernel");
    return TFC_OK;
}

extern "C" int tfc_gate_bwd2(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                             const float* v, int64_t n_rows, int32_t d_in, int32_t d_out, float* ggout, float* gx,
                             tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_in > 0 && d_out > 0, TFC_EINVAL, "tfc_gate_bwd2: bad size");
    GateSegs g;
    int rc = pack_segs(segs, n_segs, d_in, d_out, &g);
    if (rc) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && gout && v && (ggout || gx), TFC_EINVAL, "tfc_gate_bwd2: null tensor");
    TfcProfScope prof("tfc_gate_bwd2_kernel", stream);
    if (ggout) {
        const int grid = (int)std::min<long>((n_rows * (long)d_out + 255) / 256, 8192);
        tfc_gate_bwd2_kernel<true><<<grid, 256, 0, stream>>>(g, x, gout, v, (long)n_rows, d_in, d_out, ggout);
        TFC_LAUNCH_CHECK("tfc_gate_bwd2_kernel<out>");
    }
    if (gx) {
        const int grid = (int)std::min<long>((n_rows * (long)d_in + 255) / 256, 8192);
        tfc_gate_bwd2_kernel<false><<<grid, 256, 0, stream>>>(g, x, gout, v, (long)n_rows, d_in, d_out, gx);
        TFC_LAUNCH_CHECK("tfc_gate_bwd2_kernel<in>");
    }
    return TFC_OK;
}

extern "C" int tfc_norm_act_fwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows, int32_t d,
                                float eps_squared, int32_t normalize, float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d > 0 && eps_squared >= 0.f, TFC_EINVAL, "tfc_norm_act_fwd: bad size");
    GateSegs g;
    int rc = pack_norm_segs(segs, n_segs, d, &g);
    if (rc) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && out, TFC_EINVAL, "tfc_norm_act_fwd: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d + 255) / 256, 8192);
    TfcProfScope prof("tfc_norm_act_fwd_kernel", stream);
    tfc_norm_act_kernel<false><<<grid, 256, 0, stream>>>(g, x, nullptr, (long)n_rows, d, eps_squared, normalize, out);
    TFC_LAUNCH_CHECK("tfc_norm_act_kernel<fwd>");
    return TFC_OK;
}

extern "C" int tfc_norm_act_bwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                                int64_t n_rows, int32_t d, float eps_squared, int32_t normalize, float* dx,
                                tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d > 0 && eps_squared >= 0.f, TFC_EINVAL, "tfc_norm_act_bwd: bad size");
    GateSegs g;
    int rc = pack_norm_segs(segs, n_segs, d, &g);
    if (rc) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && gout && dx, TFC_EINVAL, "tfc_norm_act_bwd: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d + 255) / 256, 8192);
    TfcProfScope prof("tfc_norm_act_bwd_kernel", stream);
    tfc_norm_act_kernel<true><<<grid, 256, 0, stream>>>(g, x, gout, (long)n_rows, d, eps_squared, normalize, dx);
    TFC_LAUNCH_CHECK("tfc_norm_act_kernel<bwd>");
    return TFC_OK;
}

extern "C" int tfc_norm_act_bwd2(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout, const float* v,
                                 int64_t n_rows, int32_t d, float eps_squared, int32_t normalize, float* gx,
                                 tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d > 0 && eps_squared >= 0.f, TFC_EINVAL, "tfc_norm_act_bwd2: bad size");
    GateSegs g;
    int rc = pack_norm_segs(segs, n_segs, d, &g);
    if (rc != TFC_OK) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && gout && v && gx, TFC_EINVAL, "tfc_norm_act_bwd2: null tensor");
    const int grid = (int)std::min<long>((n_rows * (long)d + 255) / 256, 8192);
    TfcProfScope prof("tfc_norm_act_bwd2_kernel", stream);
    tfc_norm_act_bwd2_kernel<<<grid, 256, 0, stream>>>(g, x, gout, v, (long)n_rows, d, eps_squared, normalize, gx);
    TFC_LAUNCH_CHECK("tfc_norm_act_bwd2_kernel");
    return TFC_OK;
}
