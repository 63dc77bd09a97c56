// o3.Linear and the backward of the radial FullyConnectedNet as single C entry points: the
// descriptor lists for tfc_gemm_group are built here, so that a caller binds one function per
// reference call site (eigenn/nn/point_conv.py:56-61,79,130-135,180 for the linears,
// eigenn/nn/utils.py:247-252,262 for the MLP) instead of re-deriving the index maps.
// Host code only; the kernels are those of tfc_gemm.hip / tfc_layout.hip.
#include "tfc_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <vector>

int tfc_sum_parts(const float* part, int n_parts, long numel, float* out, hipStream_t stream);  // tfc_layout.hip
extern "C" int tfc_radial_mlp_bwd_chain_supported(int32_t n_layers, const int32_t* dims);              // tfc_mlp.hip
int tfc_radial_mlp_bwd_chain(int64_t n_rows, int32_t n_layers, const int32_t* dims, const float* const* weights,
                             float act_cst, const float* const* hidden, const float* gout, float* const* dhidden,
                             float* dx, hipStream_t stream);

namespace {

tfc_imap imap(int d1, int d2, long s0, long s1, long s2) { return tfc_imap{d1, d2, s0, s1, s2}; }
tfc_imap lin(long stride) { return tfc_imap{1, 1, stride, 0, 0}; }

tfc_gemm_desc desc(long m, long n, long k, const float* a1, const float* b, float* c, tfc_imap a1_r, tfc_imap a1_k,
                   tfc_imap b_k, tfc_imap b_j, tfc_imap c_r, tfc_imap c_j, float alpha, bool accumulate,
                   const char* tag) {
    tfc_gemm_desc d{};
    d.m = m, d.n = n, d.k = k;
    d.a1 = a1, d.a2 = nullptr, d.b = b, d.c = c;
    d.a1_r = a1_r, d.a1_k = a1_k, d.a2_r = lin(0), d.a2_k = lin(0);
    d.b_k = b_k, d.b_j = b_j, d.c_r = c_r, d.c_j = c_j;
    d.alpha = alpha;
    d.accumulate = accumulate ? 1 : 0;
    d.epilogue = TFC_EPI_NONE;
    d.act_cst = 1.f;
    d.tag = tag;
    d.a_rep = 1;
    return d;
}

int check_ins(const tfc_linear_ins* ins, int n_ins, int d_in, int d_out, long w_numel, const char* who) {
    TFC_REQUIRE(ins && n_ins > 0 && n_ins <= 4096, TFC_EINVAL, "%s: %d instructions", who, n_ins);
    for (int i = 0; i < n_ins; ++i) {
        const tfc_linear_ins& q = ins[i];
        TFC_REQUIRE(q.mul_in > 0 && q.mul_out > 0 && q.dim > 0, TFC_EINVAL, "%s: instruction %d is empty", who, i);
        TFC_REQUIRE(q.in_off >= 0 && (long)q.in_off + (long)q.mul_in * q.dim <= d_in, TFC_EINVAL,
                    "%s: instruction %d reads outside the input row", who, i);
        TFC_REQUIRE(q.out_off >= 0 && (long)q.out_off + (long)q.mul_out * q.dim <= d_out, TFC_EINVAL,
                    "%s: instruction %d writes outside the output row", who, i);
        TFC_REQUIRE(q.w_off >= 0 && (long)q.w_off + (long)q.mul_in * q.mul_out <= w_numel, TFC_EINVAL,
                    "%s: instruction %d reads outside the weights", who, i);
    }
    return TFC_OK;
}

// Problems of one launch must write disjoint outputs: the n-th writer of a block goes to the n-th
// wave and accumulates.
template <class KeyOf, class Make>
int run_waves(const tfc_linear_ins* ins, int n_ins, KeyOf key_of, Make make, float* ws, long ws_floats,
              hipStream_t stream) {
    std::map<int, int> seen;
    std::vector<std::vector<tfc_gemm_desc>> waves;
    for (int i = 0; i < n_ins; ++i) {
        const int w = seen[key_of(ins[i])]++;
        if ((int)waves.size() <= w) waves.emplace_back();
        tfc_gemm_desc d = make(ins[i], w > 0);
        d.workspace = ws;
        d.workspace_floats = ws_floats;
        waves[w].push_back(d);
    }
    for (auto& g : waves) {
        int rc = tfc_gemm_group(g.data(), (int)g.size(), (tfc_stream_t)stream);
        if (rc != TFC_OK) return rc;
    }
    return TFC_OK;
}

int transpose_blocks(const float* x, long n_rows, int d_row, std::vector<tfc_ir_block> blocks, float* out,
                     hipStream_t stream) {
    // distinct blocks only, at most 32 per call
    std::map<int, tfc_ir_block> uniq;
    for (auto& b : blocks) uniq[b.off] = b;
    std::vector<tfc_ir_block> v;
    for (auto& kv : uniq) v.push_back(kv.second);
    for (size_t i = 0; i < v.size(); i += 32) {
        int rc = tfc_ir_transpose(x, n_rows, d_row, v.data() + i, (int)std::min<size_t>(32, v.size() - i), out,
                                  (tfc_stream_t)stream);
        if (rc != TFC_OK) return rc;
    }
    return TFC_OK;
}

}  // namespace

extern "C" int64_t tfc_linear_workspace_floats(const tfc_linear_ins* ins, int32_t n_ins, int64_t n_rows) {
    if (!ins || n_ins <= 0 || n_rows < 0) return -1;
    int64_t need = 0;
    for (int i = 0; i < n_ins; ++i)  // only the weight gradient has a long K
        need += tfc_gemm_workspace_floats(ins[i].mul_in, ins[i].mul_out, n_rows * ins[i].dim);
    return need;
}

extern "C" int tfc_linear_fwd(const tfc_linear_ins* ins, int32_t n_ins, const float* x, const float* weight,
                              int64_t n_rows, int32_t d_in, int32_t d_out, int64_t weight_numel, float* xt,
                              float* out, tfc_stream_t stream) {
    return tfc_linear_fwd_t(ins, n_ins, x, weight, n_rows, d_in, d_out, weight_numel, xt, 0, out, stream);
}

extern "C" int tfc_linear_fwd_t(const tfc_linear_ins* ins, int32_t n_ins, const float* x, const float* weight,
                                int64_t n_rows, int32_t d_in, int32_t d_out, int64_t weight_numel, float* xt,
                                int32_t xt_ready, float* out, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_in > 0 && d_out > 0, TFC_EINVAL, "tfc_linear_fwd: bad size");
    int rc = check_ins(ins, n_ins, d_in, d_out, weight_numel, "tfc_linear_fwd");
    if (rc != TFC_OK) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE((x || xt_ready) && weight && xt && out, TFC_EINVAL, "tfc_linear_fwd: null tensor");
    const long n = n_rows;
    if (!xt_ready) {
        std::vector<tfc_ir_block> blocks;
        for (int i = 0; i < n_ins; ++i) blocks.push_back(tfc_ir_block{ins[i].in_off, ins[i].mul_in, ins[i].dim});
        rc = transpose_blocks(x, n, d_in, blocks, xt, stream);
        if (rc != TFC_OK) return rc;
    }
    // out[(k, n), w] = pw * sum_u xt[k][n][u] W[u][w]     rows r = n * dim + k
    return run_waves(
        ins, n_ins, [](const tfc_linear_ins& q) { return q.out_off; },
        [&](const tfc_linear_ins& q, bool acc) {
            return desc(n * q.dim, q.mul_out, q.mul_in, xt + (long)q.in_off * n, weight + q.w_off, out + q.out_off,
                        imap(1, q.dim, q.mul_in, 0, n * q.mul_in), lin(1), lin(q.mul_out), lin(1),
                        imap(1, q.dim, d_out, 0, 1), lin(q.dim), q.path_weight, acc, "gemm_linear_fwd");
        },
        nullptr, 0, stream);
}

extern "C" int tfc_linear_bwd(const tfc_linear_ins* ins, int32_t n_ins, const float* xt, const float* gout,
                              const float* weight, int64_t n_rows, int32_t d_in, int32_t d_out,
                              int64_t weight_numel, float* gt, int32_t gt_ready, float* dx, float* dw,
                              float* workspace, int64_t workspace_floats, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_in > 0 && d_out > 0, TFC_EINVAL, "tfc_linear_bwd: bad size");
    int rc = check_ins(ins, n_ins, d_in, d_out, weight_numel, "tfc_linear_bwd");
    if (rc != TFC_OK) return rc;
    TFC_REQUIRE(gt && (gt_ready || gout), TFC_EINVAL, "tfc_linear_bwd: null gradient");
    TFC_REQUIRE(!dw || xt, TFC_EINVAL, "tfc_linear_bwd: the weight gradient needs the forward's xt");
    TFC_REQUIRE(!dx || weight, TFC_EINVAL, "tfc_linear_bwd: null weights");
    const long n = n_rows;
    if (n == 0) {
        if (dw) { const int zrc = tfc_zero_async(dw, sizeof(float) * (size_t)weight_numel, stream); if (zrc != TFC_OK) return zrc; }
        return TFC_OK;
    }
    if (!gt_ready) {
        std::vector<tfc_ir_block> blocks;
        for (int i = 0; i < n_ins; ++i) blocks.push_back(tfc_ir_block{ins[i].out_off, ins[i].mul_out, ins[i].dim});
        rc = transpose_blocks(gout, n, d_out, blocks, gt, stream);
        if (rc != TFC_OK) return rc;
    }
    if (dx) {
        // dx[(k, n), u] = pw * sum_w gt[k][n][w] W[u][w]
        rc = run_waves(
            ins, n_ins, [](const tfc_linear_ins& q) { return q.in_off; },
            [&](const tfc_linear_ins& q, bool acc) {
                return desc(n * q.dim, q.mul_in, q.mul_out, gt + (long)q.out_off * n, weight + q.w_off,
                            dx + q.in_off, imap(1, q.dim, q.mul_out, 0, n * q.mul_out), lin(1), lin(1),
                            lin(q.mul_out), imap(1, q.dim, d_in, 0, 1), lin(q.dim), q.path_weight, acc,
                            "gemm_linear_dx");
            },
            nullptr, 0, stream);
        if (rc != TFC_OK) return rc;
    }
    if (dw) {
        // dW[u][w] = pw * sum_{(k, n)} xt[k][n][u] gt[k][n][w]; every instruction owns its weight block
        std::vector<tfc_gemm_desc> g;
        for (int i = 0; i < n_ins; ++i) {
            const tfc_linear_ins& q = ins[i];
            tfc_gemm_desc d = desc(q.mul_in, q.mul_out, n * q.dim, xt + (long)q.in_off * n, gt + (long)q.out_off * n,
                                   dw + q.w_off, lin(1), imap(1, q.dim, q.mul_in, 0, n * q.mul_in),
                                   imap(1, q.dim, q.mul_out, 0, n * q.mul_out), lin(1), lin(q.mul_out), lin(1),
                                   q.path_weight, false, "gemm_linear_dw");
            d.workspace = workspace;
            d.workspace_floats = workspace ? workspace_floats : 0;
            g.push_back(d);
        }
        rc = tfc_gemm_group(g.data(), (int)g.size(), stream_);
        if (rc != TFC_OK) return rc;
    }
    return TFC_OK;
}

extern "C" int64_t tfc_radial_mlp_bwd_workspace_floats(int64_t n_rows, int32_t n_layers, const int32_t* dims) {
    if (!dims || n_layers <= 0 || n_rows < 0) return -1;
    // the weight gradients share one launch (their needs add up); the hidden gradients run one after
    // the other on the same stream and reuse the same scratch
    int64_t dw = 0, dh = 0;
    for (int i = 0; i < n_layers; ++i) {
        dw += tfc_gemm_workspace_floats(dims[i], dims[i + 1], n_rows);
        dh = std::max<int64_t>(dh, tfc_gemm_workspace_floats(n_rows, dims[i], dims[i + 1]));
    }
    return std::max(dw, dh);
}

extern "C" int tfc_radial_mlp_bwd(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                                  const float* const* weights, float act_cst, const float* const* hidden,
                                  const float* gout, float* const* dhidden, float* dx, float* const* dweights,
                                  float* workspace, int64_t workspace_floats, tfc_stream_t stream) {
    TFC_REQUIRE(dims && weights && hidden && dhidden && dweights, TFC_EINVAL, "tfc_radial_mlp_bwd: null argument");
    TFC_REQUIRE(n_layers >= 1 && n_layers <= 16 && n_rows >= 0, TFC_EINVAL, "tfc_radial_mlp_bwd: bad size");
    for (int i = 0; i <= n_layers; ++i)
        TFC_REQUIRE(dims[i] > 0, TFC_EINVAL, "tfc_radial_mlp_bwd: layer width %d", dims[i]);
    const long E = n_rows;
    if (E == 0) {
        for (int i = 0; i < n_layers; ++i)
            if (dweights[i])
                { const int zrc = tfc_zero_async(dweights[i], sizeof(float) * (size_t)dims[i] * dims[i + 1], (hipStream_t)stream); if (zrc != TFC_OK) return zrc; }
        return TFC_OK;
    }
    TFC_REQUIRE(x && gout, TFC_EINVAL, "tfc_radial_mlp_bwd: null tensor");
    // h_0 = x, h_i = hidden[i-1]; g_L = gout, g_i = dhidden[i-1] = (g_{i+1} W_i^T / sqrt(d_i)) * act'(h_i)
    std::vector<tfc_gemm_desc> dws;
    const float* g = gout;
    // the whole chain of hidden gradients in one launch when the network fits the fused kernel (tfc_mlp.hip)
    static const bool no_chain = getenv("TFC_MLP_BWD_FUSED") && atoi(getenv("TFC_MLP_BWD_FUSED")) == 0;
    bool chain = !no_chain && tfc_radial_mlp_bwd_chain_supported(n_layers, dims) && ((uintptr_t)gout & 15) == 0 &&
                 weights[n_layers - 1] && ((uintptr_t)weights[n_layers - 1] & 15) == 0;
    for (int i = 0; chain && i < n_layers; ++i)
        chain = weights[i] && (i + 1 == n_layers || (hidden[i] && dhidden[i]));
    if (chain && (n_layers > 1 || dx)) {
        int rc = tfc_radial_mlp_bwd_chain(E, n_layers, dims, weights, act_cst, hidden, gout, dhidden, dx, (hipStream_t)stream);
        if (rc != TFC_OK) return rc;
    }
    for (int i = n_layers - 1; i >= 0; --i) {
        const int hin = dims[i], hout = dims[i + 1];
        const float alpha = 1.f / std::sqrt((float)hin);
        const float* h = i == 0 ? x : hidden[i - 1];
        TFC_REQUIRE(weights[i] && h, TFC_EINVAL, "tfc_radial_mlp_bwd: null tensor at layer %d", i);
        if (dweights[i]) {
            tfc_gemm_desc d = desc(hin, hout, E, h, g, dweights[i], lin(1), lin(hin), lin(hout), lin(1), lin(hout),
                                   lin(1), alpha, false, "gemm_mlp_dw");
            d.workspace = workspace;
            d.workspace_floats = workspace ? workspace_floats : 0;
            dws.push_back(d);
        }
        float* dh = i == 0 ? dx : dhidden[i - 1];
        if (!dh) {
            TFC_REQUIRE(i == 0, TFC_EINVAL, "tfc_radial_mlp_bwd: dhidden[%d] is null", i - 1);
            break;
        }
        if (chain) {  // already written by the fused kernel
            g = dh;
            continue;
        }
        tfc_gemm_desc d = desc(E, hin, hout, g, weights[i], dh, lin(hout), lin(1), lin(1), lin(hout), lin(hin), lin(1),
                               alpha, false, "gemm_mlp_dh");
        d.workspace = workspace;
        d.workspace_floats = workspace ? workspace_floats : 0;
        if (i > 0) {
            d.epilogue = TFC_EPI_SSP_GRAD;
            d.aux = h;
            d.act_cst = act_cst;
        }
        int rc = tfc_gemm(&d, stream);
        if (rc != TFC_OK) return rc;
        g = dh;
    }
    // all weight gradients (h_i^T g_{i+1}, K = rows) in one grouped launch
    if (!dws.empty()) return tfc_gemm_group(dws.data(), (int)dws.size(), stream);
    return TFC_OK;
}

// ---------------------------------------------------------------------------------------------
// FullyConnectedTensorProduct with a scalar second input (the self connection,
// eigenn/nn/point_conv.py:159-164,185-188).
// ---------------------------------------------------------------------------------------------
namespace {

int check_fctp(const tfc_fctp_ins* ins, int n_ins, int d_x, int d_a, int d_out, long w_numel, const char* who) {
    TFC_REQUIRE(ins && n_ins > 0 && n_ins <= 4096, TFC_EINVAL, "%s: %d instructions", who, n_ins);
    for (int i = 0; i < n_ins; ++i) {
        const tfc_fctp_ins& q = ins[i];
        TFC_REQUIRE(q.mul_u > 0 && q.mul_v > 0 && q.mul_w > 0 && q.dim > 0, TFC_EINVAL, "%s: instruction %d is empty", who, i);
        TFC_REQUIRE(q.x_off >= 0 && (long)q.x_off + (long)q.mul_u * q.dim <= d_x, TFC_EINVAL,
                    "%s: instruction %d reads outside the first input row", who, i);
        TFC_REQUIRE(q.a_off >= 0 && q.a_off + q.mul_v <= d_a, TFC_EINVAL,
                    "%s: instruction %d reads outside the second input row", who, i);
        TFC_REQUIRE(q.out_off >= 0 && (long)q.out_off + (long)q.mul_w * q.dim <= d_out, TFC_EINVAL,
                    "%s: instruction %d writes outside the output row", who, i);
        TFC_REQUIRE(q.w_off >= 0 && (long)q.w_off + (long)q.mul_u * q.mul_v * q.mul_w <= w_numel, TFC_EINVAL,
                    "%s: instruction %d reads outside the weights", who, i);
    }
    return TFC_OK;
}

template <class KeyOf, class Make>
int run_fctp_waves(const tfc_fctp_ins* ins, int n_ins, KeyOf key_of, Make make, float* ws, long ws_floats,
                   hipStream_t stream) {
    std::map<int, int> seen;
    std::vector<std::vector<tfc_gemm_desc>> waves;
    for (int i = 0; i < n_ins; ++i) {
        const int w = seen[key_of(ins[i])]++;
        if ((int)waves.size() <= w) waves.emplace_back();
        tfc_gemm_desc d = make(ins[i], w > 0);
        d.workspace = ws;
        d.workspace_floats = ws ? ws_floats : 0;
        waves[w].push_back(d);
    }
    for (auto& g : waves) {
        int rc = tfc_gemm_group(g.data(), (int)g.size(), (tfc_stream_t)stream);
        if (rc != TFC_OK) return rc;
    }
    return TFC_OK;
}

}  // namespace

extern "C" int tfc_fctp_fwd(const tfc_fctp_ins* ins, int32_t n_ins, const float* x, const float* a,
                            const float* weight, int64_t n_rows, int32_t d_x, int32_t d_a, int32_t d_out,
                            int64_t weight_numel, int32_t accumulate, float* out, float* workspace,
                            int64_t workspace_floats, tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_x > 0 && d_a > 0 && d_out > 0, TFC_EINVAL, "tfc_fctp_fwd: bad size");
    int rc = check_fctp(ins, n_ins, d_x, d_a, d_out, weight_numel, "tfc_fctp_fwd");
    if (rc != TFC_OK) return rc;
    if (n_rows == 0) return TFC_OK;
    TFC_REQUIRE(x && a && weight && out, TFC_EINVAL, "tfc_fctp_fwd: null tensor");
    const long n = n_rows;
    // out[(n,k), w] (+)= alpha sum_{(u,v)} x[n,u,k] a[n,v] W[(u,v), w]; the operand x (x) a is formed while staging
    return run_fctp_waves(
        ins, n_ins, [](const tfc_fctp_ins& q) { return q.out_off; },
        [&](const tfc_fctp_ins& q, bool acc) {
            tfc_gemm_desc d = desc(n * q.dim, q.mul_w, (long)q.mul_u * q.mul_v, x + q.x_off, weight + q.w_off,
                                   out + q.out_off, imap(1, q.dim, d_x, 0, 1), imap(1, q.mul_v, q.dim, 0, 0),
                                   lin(q.mul_w), lin(1), imap(1, q.dim, d_out, 0, 1), lin(q.dim), q.alpha,
                                   acc || accumulate, "gemm_fctp_fwd");
            d.a2 = a + q.a_off;
            d.a2_r = imap(1, q.dim, d_a, 0, 0);
            d.a2_k = imap(1, q.mul_v, 0, 0, 1);
            return d;
        },
        workspace, workspace_floats, stream);
}

extern "C" int64_t tfc_fctp_workspace_floats(const tfc_fctp_ins* ins, int32_t n_ins, int64_t n_rows) {
    if (!ins || n_ins <= 0 || n_rows < 0) return -1;
    // each of the four products is one grouped launch (needs add up); they follow one another
    int64_t fw = 0, dx = 0, dw = 0, da = 0;
    for (int i = 0; i < n_ins; ++i) {
        const tfc_fctp_ins& q = ins[i];
        fw += tfc_gemm_workspace_floats(n_rows * q.dim, q.mul_w, (long)q.mul_u * q.mul_v);
        dx += tfc_gemm_workspace_floats(n_rows * q.dim, q.mul_u, (long)q.mul_v * q.mul_w);
        dw += tfc_gemm_workspace_floats((long)q.mul_u * q.mul_v, q.mul_w, n_rows * q.dim);
        da += tfc_gemm_workspace_floats(n_rows, q.mul_v, (long)q.dim * q.mul_u * q.mul_w);
    }
    return std::max(std::max(fw, dx), std::max(dw, da));
}

extern "C" int tfc_fctp_bwd(const tfc_fctp_ins* ins, int32_t n_ins, const float* x, const float* a,
                            const float* weight, const float* gout, int64_t n_rows, int32_t d_x, int32_t d_a,
                            int32_t d_out, int64_t weight_numel, float* gt, int32_t gt_ready, float* dx, float* dw,
                            float* da, float* xt, float* da_part, float* workspace, int64_t workspace_floats,
                            tfc_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TFC_REQUIRE(n_rows >= 0 && d_x > 0 && d_a > 0 && d_out > 0, TFC_EINVAL, "tfc_fctp_bwd: bad size");
    int rc = check_fctp(ins, n_ins, d_x, d_a, d_out, weight_numel, "tfc_fctp_bwd");
    if (rc != TFC_OK) return rc;
    TFC_REQUIRE(gt && (gt_ready || gout), TFC_EINVAL, "tfc_fctp_bwd: null gradient");
    TFC_REQUIRE(!da || (xt && da_part), TFC_EINVAL, "tfc_fctp_bwd: the attribute gradient needs the xt and da_part scratch");
    const long n = n_rows;
    if (n == 0) {
        if (dw) { const int zrc = tfc_zero_async(dw, sizeof(float) * (size_t)weight_numel, stream); if (zrc != TFC_OK) return zrc; }
        return TFC_OK;
    }
    TFC_REQUIRE(x && a && weight, TFC_EINVAL, "tfc_fctp_bwd: null tensor");
    if (!gt_ready) {
        std::vector<tfc_ir_block> blocks;
        for (int i = 0; i < n_ins; ++i) blocks.push_back(tfc_ir_block{ins[i].out_off, ins[i].mul_w, ins[i].dim});
        rc = transpose_blocks(gout, n, d_out, blocks, gt, stream);
        if (rc != TFC_OK) return rc;
    }
    // channel-contiguous copy of x ([dim][N][mul_u] per block, block of instruction q at xt + q.x_off * n), made once
    bool xt_done = false;
    auto ensure_xt = [&]() -> int {
        if (xt_done) return TFC_OK;
        std::vector<tfc_ir_block> blocks;
        for (int i = 0; i < n_ins; ++i) blocks.push_back(tfc_ir_block{ins[i].x_off, ins[i].mul_u, ins[i].dim});
        xt_done = true;
        return transpose_blocks(x, n, d_x, blocks, xt, stream);
    };
    if (dx) {
        // dx[(n,k), u] = alpha sum_{(v,w)} a[n,v] g[n,w,k] W[u,v,w]
        rc = run_fctp_waves(
            ins, n_ins, [](const tfc_fctp_ins& q) { return q.x_off; },
            [&](const tfc_fctp_ins& q, bool acc) {
                tfc_gemm_desc d = desc(n * q.dim, q.mul_u, (long)q.mul_v * q.mul_w, a + q.a_off, weight + q.w_off,
                                       dx + q.x_off, imap(1, q.dim, d_a, 0, 0), imap(1, q.mul_w, 1, 0, 0), lin(1),
                                       lin((long)q.mul_v * q.mul_w), imap(1, q.dim, d_x, 0, 1), lin(q.dim), q.alpha,
                                       acc, "gemm_fctp_dx");
                d.a2 = gt + (long)q.out_off * n;
                d.a2_r = imap(1, q.dim, q.mul_w, 0, n * q.mul_w);
                d.a2_k = imap(1, q.mul_w, 0, 0, 1);
                return d;
            },
            workspace, workspace_floats, stream);
        if (rc != TFC_OK) return rc;
    }
    if (da) {
        // da[n, v] = alpha sum_{(k,u,w)} x[n,u,k] g[n,w,k] W[u,v,w]: every instruction writes its own partial
        // [n_rows, d_a] in one grouped launch, the partials are then added in instruction order
        rc = ensure_xt();
        if (rc != TFC_OK) return rc;
        { const int zrc = tfc_zero_async(da_part, sizeof(float) * (size_t)n_ins * n * d_a, stream); if (zrc != TFC_OK) return zrc; }
        std::vector<tfc_gemm_desc> g;
        for (int i = 0; i < n_ins; ++i) {
            const tfc_fctp_ins& q = ins[i];
            const long mu = q.mul_u, mv = q.mul_v, mw = q.mul_w;
            float* c = da_part + (long)i * n * d_a + q.a_off;
            tfc_gemm_desc d;
            // the sum over the 2l+1 components folded into the operand where that was measured to pay
            const bool fold = (q.dim == 3 || q.dim == 5) && mv <= 64 && mw % 4 == 0 && ((long)q.out_off * n) % 4 == 0 &&
                              (n * mw) % 4 == 0 && ((uintptr_t)gt & 15) == 0;
            if (fold) {
                d = desc(n, mv, mu * mw, xt + (long)q.x_off * n, weight + q.w_off, c, lin(mu), imap(1, (int)mw, 1, 0, 0),
                         imap(1, (int)mw, mv * mw, 0, 1), lin(mw), lin(d_a), lin(1), q.alpha, false, "gemm_fctp_da");
                d.a2 = gt + (long)q.out_off * n;
                d.a2_r = lin(mw);
                d.a2_k = imap(1, (int)mw, 0, 0, 1);
                d.a_rep = q.dim;
                d.a1_rep_stride = n * mu;
                d.a2_rep_stride = n * mw;
            } else {
                d = desc(n, mv, q.dim * mu * mw, xt + (long)q.x_off * n, weight + q.w_off, c, lin(mu),
                         imap((int)mu, (int)mw, n * mu, 1, 0), imap((int)mu, (int)mw, 0, mv * mw, 1), lin(mw), lin(d_a),
                         lin(1), q.alpha, false, "gemm_fctp_da");
                d.a2 = gt + (long)q.out_off * n;
                d.a2_r = lin(mw);
                d.a2_k = imap((int)mu, (int)mw, n * mw, 0, 1);
            }
            d.workspace = workspace;
            d.workspace_floats = workspace ? workspace_floats : 0;
            g.push_back(d);
        }
        rc = tfc_gemm_group(g.data(), (int)g.size(), stream_);
        if (rc != TFC_OK) return rc;
        rc = tfc_sum_parts(da_part, n_ins, n * d_a, da, stream);
        if (rc != TFC_OK) return rc;
    }
    if (dw) {
        // dW[(u,v), w] = alpha sum_{(n,k)} x[n,u,k] a[n,v] g[n,w,k]; every instruction owns its weight block
        std::vector<tfc_gemm_desc> g;
        for (int i = 0; i < n_ins; ++i) {
            const tfc_fctp_ins& q = ins[i];
            // (the same product from the channel-contiguous copy of x, rows ordered (v, u) so that four rows share
            // a[n, v] and read 16 bytes of x, was measured in round 3: 271 against 273 us per launch at config 4, and
            // three more transposing passes per step - not kept)
            tfc_gemm_desc d;
            d = desc((long)q.mul_u * q.mul_v, q.mul_w, n * q.dim, x + q.x_off, gt + (long)q.out_off * n,
                     dw + q.w_off, imap(1, q.mul_v, q.dim, 0, 0), imap(1, q.dim, d_x, 0, 1),
                     imap(1, q.dim, q.mul_w, 0, n * q.mul_w), lin(1), lin(q.mul_w), lin(1), q.alpha, false,
                     "gemm_fctp_dw");
            d.a2 = a + q.a_off;
            d.a2_r = imap(1, q.mul_v, 0, 0, 1);
            d.a2_k = imap(1, q.dim, d_a, 0, 0);
            d.workspace = workspace;
            d.workspace_floats = workspace ? workspace_floats : 0;
            g.push_back(d);
        }
        rc = tfc_gemm_group(g.data(), (int)g.size(), stream_);
        if (rc != TFC_OK) return rc;
    }
    return TFC_OK;
}
