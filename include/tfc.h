/*
 * tfc.h - C ABI of libtfc: MI355X (gfx950) kernels for eigenn's tensor-field convolution
 * message-passing layer.
 *
 * The reference (mjwen/matten, package `eigenn`) is pure Python and has no FFI of its own:
 * every number on this path is produced by e3nn / torch / torch_scatter calls.  Each entry
 * point below therefore cites the reference CALL SITE it replaces (paths relative to the
 * reference root).  The reference-side binding a maintainer would add is a ctypes stub; see
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative TFC_E* code otherwise; the message of the
 *     last failure on the calling thread is available from tfc_last_error().
 *   - all tensor arguments are DEVICE pointers to contiguous row-major buffers owned by the
 *     caller (PyTorch).  fp32 data, int64 indices as delivered by the reference's data
 *     pipeline (eigenn/data/_dtype.py:3-4), int32 for the orderings this library builds.
 *   - the library never allocates or frees device memory inside a compute call: scratch comes
 *     in through `workspace` arguments whose size the *_workspace_* functions report.  Only
 *     tfc_conv_plan_create() and tfc_conv_plan_set_out_blocks() allocate (small constant tables
 *     of the plan) and tfc_conv_plan_destroy() frees them.
 *   - all launches are asynchronous on the caller's `stream` (a hipStream_t passed as void*,
 *     0 = the null stream).  No call synchronises the host.  Plans are immutable after
 *     creation, so calls are re-entrant across host threads and streams; the only global
 *     mutable state is the thread-local error string.
 *   - a feature row with irreps sum_b mul_b x (l_b, p_b) is the concatenation of the blocks b,
 *     each stored row-major as [mul_b][2 l_b + 1] (e3nn layout).
 */
#ifndef TFC_H
#define TFC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFC_OK 0
#define TFC_EINVAL (-1)       /* null pointer, negative size, inconsistent table ... */
#define TFC_EUNSUPPORTED (-2) /* irreps outside what the kernels are built for (l > 3, ...) */
#define TFC_ELAUNCH (-3)      /* a HIP call or kernel launch failed */
#define TFC_ENOMEM (-4)

#define TFC_LMAX 3

typedef void* tfc_stream_t;
typedef struct tfc_conv_plan tfc_conv_plan;

int tfc_version(void);
const char* tfc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Host-only helper: real Clebsch-Gordan tensor C[m1][m2][m3] (fp64, unit Frobenius norm) in
 * the basis of the spherical harmonics below.  Replaces e3nn.o3.wigner_3j as used inside
 * o3.TensorProduct (eigenn/nn/utils.py:231-238).  `out` is a HOST buffer of
 * (2l1+1)(2l2+1)(2l3+1) doubles.
 */
int tfc_real_cg(int l1, int l2, int l3, double* out);

/* ------------------------------------------------------------------------------------------
 * Edge ordering.  Groups the edges by `key` (a row of edge_index) so that the convolution can
 * reduce over each node's edges in registers instead of scatter_add_'s global atomics
 * (eigenn/nn/utils.py:351-357, called at eigenn/nn/point_conv.py:173-175).
 *
 *   key, other : [E] int64   the grouping row of edge_index and the opposite row
 *   row_ptr    : [N+1] int32 out: edges of node n are sorted positions row_ptr[n]..row_ptr[n+1]
 *   eid        : [E] int32   out: original edge id at each sorted position (ascending inside a
 *                            node => the summation order, hence the result, is deterministic)
 *   nbr        : [E] int32   out: other[eid[k]]
 *   status     : [1] int32   out: 0, or 1 if any index was outside [0, N) (such edges are dropped
 *                            from the ordering; the caller decides when to read the flag)
 *   workspace  : tfc_csr_workspace_bytes(N, E) bytes
 */
int64_t tfc_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int tfc_csr_build(const int64_t* key, const int64_t* other, int64_t n_edges, int64_t n_nodes,
                  int32_t* row_ptr, int32_t* eid, int32_t* nbr, int32_t* status, void* workspace,
                  tfc_stream_t stream);

/* tfc_csr_build for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and
 * results as tfc_csr_build minus `workspace`: the cursors and the unsorted edge queue live in a static
 * device array of TFC_CSR_NOALLOC_MAX_INTS int32 (16 MiB per GPU the library runs on, held until the
 * process ends), so n_nodes + n_edges + 2 must not exceed that (TFC_EUNSUPPORTED otherwise).  The call
 * allocates and frees nothing; on its first use on a device it looks up that device's address of the
 * array.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race - order them.  Nothing in the array has to be cleared between calls. */
#define TFC_CSR_NOALLOC_MAX_INTS (1 << 22)
int tfc_csr_build_noalloc(const int64_t* key, const int64_t* other, int64_t n_edges, int64_t n_nodes,
                          int32_t* row_ptr, int32_t* eid, int32_t* nbr, int32_t* status, tfc_stream_t stream);

/* The unfused route, for callers that materialise per-edge messages and sum them afterwards
 * (scatter_add, eigenn/nn/utils.py:351-357, and its backward, a row gather):
 *   tfc_segment_sum: out[n] = sum_{k in rows(n)} src[eid[k]]   src [E,d] -> out [N,d], no atomics
 *   tfc_gather_rows: out[r] = src[index[r]]                    src [n_src,d], index [n_rows] int64
 */
int tfc_segment_sum(const float* src, const int32_t* row_ptr, const int32_t* eid, int64_t n_nodes,
                    int32_t d, float* out, tfc_stream_t stream);
int tfc_gather_rows(const float* src, const int64_t* index, int64_t n_rows, int64_t n_src, int32_t d,
                    float* out, tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Periodic neighbour list of a batch of structures.  Replaces neighbor_list_and_relative_vec
 * (eigenn/data/data.py:400-528): ase.neighborlist.primitive_neighbor_list("ijS", pbc, cell, pos,
 * cutoff=r_max, self_interaction=True) (:480-489) followed by the removal of the (i == j, S == 0)
 * edges (:491-502).  The result is the set of all (i, j, S) with
 * |pos[j] - pos[i] + S @ cell| < r_max (strict; fp64, no fused multiply-add), in the canonical order
 * (i, j, Sx, Sy, Sz) - ASE's own order inside equal i is not reproducible, so parity is equality of
 * sets; edge_index[0] is non-decreasing as ASE delivers it.
 *
 *   pos        : [N,3] double   positions of all structures, concatenated
 *   node_graph : [N] int32      structure of each node
 *   graph_ptr  : [G+1] int32    nodes of structure g are graph_ptr[g] .. graph_ptr[g+1]
 *   shift_ptr  : [G+1] int32    rows of `shifts` / `shift_vec` that belong to structure g
 *   shifts     : [S,3] int32    periodic images to test, ascending lexicographic inside a structure
 *                               (the caller enumerates them: ceil(r_max / face distance) + spread per
 *                               periodic axis, matten_amd/data/neighbor.py), (0,0,0) included
 *   shift_vec  : [S,3] double   shifts @ cell of that structure
 *   graph_box  : [G,6] double   (min x,y,z, max x,y,z) of each structure's positions, or null.  With it,
 *                               every node first drops the images whose moved box stays farther than
 *                               r_max away (a conservative test: the result is the same set), which
 *                               leaves 1-8 of the 27-125 images of the bench cells.
 *
 * tfc_nl_count -> row_ptr [N+1] int32: edges of sender i will occupy row_ptr[i] .. row_ptr[i+1];
 * the caller reads E = row_ptr[N], allocates, and calls tfc_nl_fill with the same arguments ->
 * edge_index [2,E] int64 (row 0 sender i, row 1 receiver j, global node ids) and edge_shift [E,3]
 * float32 (the dtype the reference stores, data.py:509-513).  No sort and no atomics: one wave walks
 * a node's candidates in canonical order and writes at ballot-prefix positions.
 *
 * tfc_nl_ordering: both groupings the convolution needs (tfc_csr_build's outputs for key = row 0 and
 * key = row 1), derived from the symmetry (i, j, S) <-> (j, i, -S) of such a list instead of a sort;
 * both share `row_ptr`.  status [1] int32: 0, 1 if the list is not symmetric (a pair within rounding
 * of the cutoff: use tfc_csr_build), 2 if an index is outside [0, N); on non-zero status the dst
 * arrays are incomplete and must not be used.
 */
int tfc_nl_count(const double* pos, const int32_t* node_graph, const int32_t* graph_ptr,
                 const int32_t* shift_ptr, const int32_t* shifts, const double* shift_vec,
                 const double* graph_box, int64_t n_nodes, int64_t n_graphs, double r_max, int32_t* row_ptr,
                 tfc_stream_t stream);
int tfc_nl_fill(const double* pos, const int32_t* node_graph, const int32_t* graph_ptr,
                const int32_t* shift_ptr, const int32_t* shifts, const double* shift_vec,
                const double* graph_box, int64_t n_nodes, int64_t n_graphs, double r_max, const int32_t* row_ptr,
                int64_t n_edges, int64_t* edge_index, float* edge_shift, tfc_stream_t stream);
int tfc_nl_ordering(const int64_t* edge_index, const int32_t* row_ptr, int64_t n_edges, int64_t n_nodes,
                    int32_t* src_eid, int32_t* src_nbr, int32_t* dst_eid, int32_t* dst_nbr, int32_t* status,
                    tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 'uvu' tensor product with per-edge weights, fused with the gather of sender features and
 * the sum over each receiver's edges.
 *
 * Replaces, in one kernel: node_feats[edge_src] (eigenn/nn/point_conv.py:82), the e3nn
 * o3.TensorProduct(..., 'uvu', shared_weights=False) call (eigenn/nn/utils.py:265), scatter_add
 * (eigenn/nn/point_conv.py:173-175) and the 1/sqrt(avg_num_neighbors) scale (:177-178).
 *
 *   out[n, out_off_p + u*(2l3+1) + m3] = scale * coeff_p * sum_{e : centre(e) = n} w[e, w_off_p + u]
 *        * sum_{m1,m2} C^{l1 l2 l3}[m1,m2,m3] * x[nbr(e), x_off_p + u*(2l1+1) + m1] * y[e, y_off_p + m2]
 *
 * One tfc_path per e3nn instruction, in the instruction (= weight slot) order built at
 * eigenn/nn/utils.py:205-229.  Restrictions (TFC_EUNSUPPORTED otherwise): l1,l2,l3 <= TFC_LMAX,
 * the edge-attribute block of every path has multiplicity 1, and every path owns its output
 * block (both hold for every UVUTensorProduct the reference builds from spherical-harmonic
 * edge attributes).
 */
typedef struct tfc_path {
    int32_t l1, l2, l3;
    int32_t mul;     /* channels u of the in1 block (= multiplicity of the output block) */
    int32_t x_off;   /* offset of the in1 block inside a node-feature row */
    int32_t y_off;   /* offset of the in2 block inside an edge-attribute row */
    int32_t out_off; /* offset of the output block inside an output row */
    int32_t w_off;   /* offset of this path's `mul` weights inside a per-edge weight row */
    float coeff;     /* sqrt(alpha): e3nn's path normalisation (= sqrt(2 l3 + 1) here) */
} tfc_path;

int tfc_conv_plan_create(const tfc_path* paths, int32_t n_paths, int32_t d_in, int32_t d_sh,
                         int32_t d_out, int32_t w_numel, tfc_conv_plan** plan);
int tfc_conv_plan_destroy(tfc_conv_plan* plan);

/* floats of scratch a forward / backward call needs for E edges */
int64_t tfc_conv_workspace_floats(const tfc_conv_plan* plan, int64_t n_edges, int need_dy);

/* Forward.  (row_ptr, eid, nbr) = tfc_csr_build(key = edge_index[1] (receiver), other =
 * edge_index[0] (sender)).  x [N, d_in], y [E, d_sh], w [E, w_numel] in the caller's edge order;
 * out [N, d_out] is fully overwritten. */
int tfc_conv_fwd(const tfc_conv_plan* plan, const float* x, const float* y, const float* w,
                 const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                 int64_t n_edges, float scale, float* out, float* workspace, tfc_stream_t stream);

/* Backward of the above (what autograd derives in the reference).  Here the edges are grouped
 * by SENDER: (row_ptr, eid, nbr) = tfc_csr_build(key = edge_index[0], other = edge_index[1]).
 * gout [N, d_out]; dx [N, d_in] and dw [E, w_numel] are fully overwritten; dy [E, d_sh] is
 * overwritten when non-null (pass null when the edge attributes need no gradient). */
int tfc_conv_bwd(const tfc_conv_plan* plan, const float* x, const float* y, const float* w,
                 const float* gout, const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr,
                 int64_t n_nodes, int64_t n_edges, float scale, float* dx, float* dw, float* dy,
                 float* workspace, tfc_stream_t stream);

/* The same two calls with the per-edge matrices M_e[m1][m3] = sum_m2 C[m1][m2][m3] y[e][m2] (what
 * the kernels actually read; rows in the caller's EDGE order, looked up by edge id) built by the caller
 * beforehand.  M depends on the edge attributes and the set of (l1, l2, l3) blocks of the plan only, so
 * the layers of a model that share both (every PointConvMessagePassing after the first, whose
 * e3nn TensorProducts use the same Clebsch-Gordan blocks, eigenn/nn/utils.py:205-229) share ONE M,
 * forward and backward, instead of rebuilding it in every call.
 *   tfc_conv_m_signature : equal for two plans <=> their M rows have the same layout
 *   tfc_conv_m_floats    : size of M for E edges;  tfc_conv_dy_floats: size of `dyp` (backward scratch
 *                          for the partial edge-attribute gradients, needed when dy is non-null)
 *   tfc_conv_m_build     : M from y [E, d_sh] */
int64_t tfc_conv_m_signature(const tfc_conv_plan* plan);
int64_t tfc_conv_m_floats(const tfc_conv_plan* plan, int64_t n_edges);
int64_t tfc_conv_dy_floats(const tfc_conv_plan* plan, int64_t n_edges);
int tfc_conv_m_build(const tfc_conv_plan* plan, const float* y, int64_t n_edges, float* M, tfc_stream_t stream);
int tfc_conv_fwd_m(const tfc_conv_plan* plan, const float* x, const float* M, const float* w,
                   const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                   int64_t n_edges, float scale, float* out, tfc_stream_t stream);
int tfc_conv_bwd_m(const tfc_conv_plan* plan, const float* x, const float* M, const float* w, const float* gout,
                   const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                   int64_t n_edges, float scale, float* dx, float* dw, float* dy, float* dyp,
                   tfc_stream_t stream);

/* The same two calls writing a SECOND, channel-contiguous copy of their node-row result (the layout
 * tfc_ir_transpose produces, which tfc_linear_fwd / tfc_linear_bwd consume), so that the o3.Linear that follows
 * the convolution (linear_2, eigenn/nn/point_conv.py:130-135) or precedes it (linear_1, :56-61, in the backward)
 * needs no transposing pass of its own.  No counterpart in the reference (layout plumbing between two kernels).
 *   out_t [d_out * N]: block b = (off, mul, dim) of the output row, as given to tfc_conv_plan_set_out_blocks
 *                      (every path's output must lie inside one block of its own 2l3+1), is written at
 *                      out_t[off * N + (m * N + n) * mul + channel]; may be null
 *   dx_t  [d_in * N] : the in1 blocks (x_off, mul, 2l1+1) of the paths in the same layout; columns no path
 *                      reads are left untouched; may be null */
struct tfc_ir_block; /* (off, mul, dim), defined with tfc_ir_transpose below */
int tfc_conv_plan_set_out_blocks(tfc_conv_plan* plan, const struct tfc_ir_block* blocks, int32_t n_blocks);
int tfc_conv_fwd_m_t(const tfc_conv_plan* plan, const float* x, const float* M, const float* w,
                     const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                     int64_t n_edges, float scale, float* out, float* out_t, tfc_stream_t stream);
int tfc_conv_bwd_m_t(const tfc_conv_plan* plan, const float* x, const float* M, const float* w, const float* gout,
                     const int32_t* row_ptr, const int32_t* eid, const int32_t* nbr, int64_t n_nodes,
                     int64_t n_edges, float scale, float* dx, float* dx_t, float* dw, float* dy, float* dyp,
                     tfc_stream_t stream);

/* tfc_conv_fwd_m with the LAST layer of the radial network folded in (eigenn/nn/utils.py:262 is where the
 * weights are born, :265 where they are consumed): the per-edge weights are never materialised,
 *     w[e, j] = alpha * sum_k h[e, k] * w_last[k, j],   h [E, hidden] the last hidden activations,
 *     w_last [hidden, w_numel] row-major, alpha = 1 / sqrt(hidden).
 * Forward only (inference; training needs w for the backward, see DESIGN.md §4 for the measurement that keeps
 * the unfused call as the default).  hidden = 64 only (TFC_EUNSUPPORTED otherwise). */
int tfc_conv_fwd_fused_m(const tfc_conv_plan* plan, const float* x, const float* M, const float* h,
                         const float* w_last, int32_t hidden, float alpha, const int32_t* row_ptr,
                         const int32_t* eid, const int32_t* nbr, int64_t n_nodes, int64_t n_edges, float scale,
                         float* out, tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-edge encoders.
 *
 * Replaces with_edge_vectors (eigenn/nn/_nequip.py:214-266), o3.SphericalHarmonics(irreps,
 * normalize=True, normalization="component") (:167-175) and BesselBasis x PolynomialCutoff
 * (:117-128, :63-78, :206-208) in one pass over the edges:
 *
 *   vec[e] = pos[ei[1,e]] - pos[ei[0,e]] + shift[e] @ cell[batch[ei[0,e]]]
 *   sh [e, l*l + m]  = Y_l^m(vec/|vec|), l = 0..lmax, component normalisation; with sh_normalize = 0
 *                      (edge_sh_normalize=False, _nequip.py:150,169) the same polynomials at vec itself
 *   emb[e, n]        = (2/r_max) sin(bw[n] |vec| / r_max) / |vec| * env_p(|vec| / r_max) * (|vec| < r_max)
 *
 * edge_index [2,E] int64; shift [E,3] (may be null: no periodic images); cell [G,3,3] (null iff
 * shift is null); batch [N] int64 (may be null when G == 1); bw [n_basis] the (trainable) Bessel
 * frequencies.  Outputs vec [E,3], len [E], sh [E,(lmax+1)^2], emb [E,n_basis]; sh or emb may be
 * null to skip that encoder.  When `pos` is null the edge vectors are taken from `vec` as given
 * (the reference's cached EDGE_VECTORS branch, _nequip.py:224-229) and `len`, when non-null, is
 * read as their lengths.
 */
int tfc_edge_encode_fwd(const float* pos, const int64_t* edge_index, const float* shift,
                        const float* cell, const int64_t* batch, int64_t n_edges, int64_t n_nodes,
                        int64_t n_graphs, int32_t lmax, int32_t sh_normalize, const float* bw,
                        int32_t n_basis, float r_max, float p, float* vec, float* len, float* sh, float* emb,
                        tfc_stream_t stream);

/* Backward: per-edge gradient of the edge vector from the gradients of sh and emb (either may
 * be null), and the gradient of the Bessel frequencies.
 *   dvec [E,3] out;  dbw_partial [n_blocks, n_basis] out (null to skip), where
 *   n_blocks = tfc_edge_bwd_blocks(E): the caller sums the rows (deterministic two-stage sum). */
int64_t tfc_edge_bwd_blocks(int64_t n_edges);
int tfc_edge_encode_bwd(const float* vec, const float* len, int64_t n_edges, int32_t lmax,
                        int32_t sh_normalize, const float* bw, int32_t n_basis, float r_max, float p, const float* dsh,
                        const float* demb, const float* dvec_add, const float* dlen, float* dvec,
                        float* dbw_partial, tfc_stream_t stream);

/* Second order of the encoders, for training on forces (a loss on F = -dE/dpos needs the derivative of the
 * backward above; the reference gets it from autograd's double backward through e3nn / torch,
 * eigenn/nn/_nequip.py:231-235).  With phi = sum dsh Y(vec) + sum demb emb(|vec|, bw) + dlen |vec|, the call
 * above returns dvec = d phi / d vec and dbw = d phi / d bw.  Given cotangents gv [E,3] of dvec and gb [n_basis]
 * of dbw (either may be null = zero) this returns the products autograd asks for:
 *   g_vec [E,3], g_bw_partial [tfc_edge_bwd_blocks(E), n_basis] : Hessian of phi times (gv, gb)
 *   g_dsh [E,(lmax+1)^2], g_demb [E,n_basis], g_dlen [E]        : Jacobians of Y, emb, |vec| times (gv, gb)
 * (every output nullable).  Lengths are recomputed from vec.  One pass in dual-number arithmetic. */
int tfc_edge_encode_bwd2(const float* vec, int64_t n_edges, int32_t lmax, int32_t sh_normalize, const float* bw,
                         int32_t n_basis, float r_max, float p, const float* dsh, const float* demb,
                         const float* dlen, const float* gv, const float* gb, float* g_vec, float* g_dsh,
                         float* g_demb, float* g_dlen, float* g_bw_partial, tfc_stream_t stream);

/* Scatter of the per-edge vector gradient to positions (and cells):
 *   dpos[n] = sum_{e: ei[1,e]=n} dvec[e] - sum_{e: ei[0,e]=n} dvec[e]
 * using the two orderings of tfc_csr_build (no atomics).  (rp_dst, eid_dst) group by ei[1],
 * (rp_src, eid_src) by ei[0].  dpos [N,3] is overwritten. */
int tfc_edge_vec_bwd(const float* dvec, const int32_t* rp_src, const int32_t* eid_src,
                     const int32_t* rp_dst, const int32_t* eid_dst, int64_t n_nodes, float* dpos,
                     tfc_stream_t stream);

/* Gradient of the periodic cell (with_edge_vectors is differentiable in `cell` too, _nequip.py:236-262):
 *   part[n][3a+b] = sum_{e: ei[0,e]=n} shift[e][a] * dvec[e][b]      part [N,9]
 * per sender node, in ascending edge id; the caller sums the rows of each structure's nodes
 * (tfc_csr_build over `batch` + tfc_segment_sum) to get dcell [G,3,3].  No atomics. */
int tfc_edge_cell_bwd(const float* dvec, const float* shift, const int32_t* rp_src, const int32_t* eid_src,
                      int64_t n_nodes, float* part, tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense contractions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains).
 *
 * One strided GEMM serves the radial FullyConnectedNet (eigenn/nn/utils.py:247-252,262), the
 * o3.Linear channel mixers (eigenn/nn/point_conv.py:56-61,79 and :130-135,180) and the
 * FullyConnectedTensorProduct self connection (:159-164,185-188), forward and backward:
 *
 *   C[addrC(r,j)] (+)= epilogue( alpha * sum_k A1[addrA1(r,k)] * (A2 ? A2[addrA2(r,k)] : 1) * B[addrB(k,j)] )
 *   (see a_rep below for an A element that is itself a short sum of such products)
 *
 * Every address is a sum of per-index maps  f(t) = (t / (d1*d2))*s0 + ((t / d2) % d1)*s1 + (t % d2)*s2,
 * which is what lets one kernel walk e3nn's [node][mul][2l+1] blocks without materialising
 * transposes or outer products.
 */
typedef struct tfc_imap {
    int32_t d1, d2;      /* >= 1 */
    int64_t s0, s1, s2;  /* element strides */
} tfc_imap;

typedef struct tfc_gemm_desc {
    int64_t m, n, k;
    const float* a1;
    const float* a2; /* optional second factor of the A operand (null: none) */
    const float* b;
    float* c;
    tfc_imap a1_r, a1_k, a2_r, a2_k, b_k, b_j, c_r, c_j;
    float alpha;
    int32_t accumulate; /* 1: C += ..., 0: C = ... */
    int32_t epilogue;   /* TFC_EPI_* */
    const float* aux;   /* TFC_EPI_SSP_GRAD: activation output h, addressed like C */
    float act_cst;      /* normalize2mom constant of the activation */
    float* workspace;   /* optional scratch for split-K (small m x n, long k); null: never split */
    int64_t workspace_floats;
    const char* tag;    /* optional name under which tfc_profile_report() books this call */
    /* a_rep > 1 (two factors only): the A element is
     *     sum_{m < a_rep} A1[a1R(r)+a1K(k)+m*a1_rep_stride] * A2[a2R(r)+a2K(k)+m*a2_rep_stride],
     * i.e. a short contraction (the 2l+1 components of an irrep) folded into the operand instead of
     * lengthening K.  Supported for a_rep in {3,5,7} when both factors can be staged in 16-byte
     * groups along k (TFC_EUNSUPPORTED otherwise; 0 or 1: no repetition). */
    int32_t a_rep;
    int64_t a1_rep_stride, a2_rep_stride;
} tfc_gemm_desc;

#define TFC_EPI_NONE 0
#define TFC_EPI_SSP 1      /* v = act_cst * (softplus(v) - ln 2)                              */
#define TFC_EPI_SSP_GRAD 2 /* v = v * d act/d pre, recovered from aux = act output            */

int tfc_gemm(const tfc_gemm_desc* desc, tfc_stream_t stream);
/* Several independent problems in as few launches as possible (problems that walk their operands
 * the same way share a launch, so the many small per-irrep products of one o3.Linear or one
 * FullyConnectedTensorProduct fill the chip together).  Equivalent to calling tfc_gemm on each
 * descriptor, PROVIDED no two problems of one call write the same element of C (not checked):
 * the order in which the problems of a call run is unspecified.  The workspace of the group is
 * the largest one found in the descriptors. */
int tfc_gemm_group(const tfc_gemm_desc* descs, int32_t n_descs, tfc_stream_t stream);
/* scratch floats with which tfc_gemm can cut K to fill the chip (0: the problem is not split).
 * Split sums are deterministic: partial tiles are stored and added in a fixed order. */
int64_t tfc_gemm_workspace_floats(int64_t m, int64_t n, int64_t k);

/* The radial FullyConnectedNet forward (eigenn/nn/utils.py:247-252,262) in ONE launch:
 *     h_0 = x [n_rows, dims[0]];  h_{i+1} = act_cst * ssp(h_i W_i / sqrt(dims[i]))  (i < n_layers - 1);
 *     out = h_L W_L / sqrt(dims[L]),  L = n_layers - 1,  W_i row-major [dims[i], dims[i+1]].
 * `weights` and `hidden` are HOST arrays of device pointers; hidden[i] [n_rows, dims[i+1]] receives
 * the output of layer i (what backward needs; i < n_layers - 1).  The hidden activations stay in
 * LDS between layers.  TFC_EUNSUPPORTED (use tfc_gemm layer by layer) unless 2 <= n_layers <= 6,
 * dims[0 .. n_layers-1] <= 64 and dims[n_layers] % 4 == 0. */
int tfc_radial_mlp_fwd(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                       const float* const* weights, float act_cst, float* const* hidden, float* out,
                       tfc_stream_t stream);
/* The same call with caller scratch of tfc_radial_mlp_fwd_workspace_floats floats (16-byte aligned): the last layer
 * - all but a few per cent of the network's arithmetic - then runs on v_mfma_f32_32x32x16_bf16 with every fp32 operand
 * cut into three bf16 pieces that add up to it exactly (six piece products per product: fp32 accuracy, see
 * csrc/tfc_mlp.hip), at 2.67 x the rate of the fp32 matrix instruction.  Null or short scratch: as tfc_radial_mlp_fwd. */
int64_t tfc_radial_mlp_fwd_workspace_floats(int32_t n_layers, const int32_t* dims);
int tfc_radial_mlp_fwd_ws(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                          const float* const* weights, float act_cst, float* const* hidden, float* out,
                          float* workspace, int64_t workspace_floats, tfc_stream_t stream);

/* Backward of the radial FullyConnectedNet (what autograd derives from eigenn/nn/utils.py:247-252,262).
 * Arguments as for the forward; `hidden` are the forward's saved activations, `gout` [n_rows,
 * dims[n_layers]] the gradient of the output.  dhidden[i] [n_rows, dims[i+1]] (i < n_layers - 1) is
 * caller scratch that receives the gradient of the pre-activation-scaled hidden layer; dx (nullable)
 * [n_rows, dims[0]] the gradient of the input; dweights[i] (nullable entries) [dims[i], dims[i+1]]
 * the weight gradients, all formed by ONE grouped launch at the end.  `workspace`: optional split-K
 * scratch of tfc_radial_mlp_bwd_workspace_floats floats (null: the long-K products are not split). */
int64_t tfc_radial_mlp_bwd_workspace_floats(int64_t n_rows, int32_t n_layers, const int32_t* dims);
int tfc_radial_mlp_bwd(const float* x, int64_t n_rows, int32_t n_layers, const int32_t* dims,
                       const float* const* weights, float act_cst, const float* const* hidden,
                       const float* gout, float* const* dhidden, float* dx, float* const* dweights,
                       float* workspace, int64_t workspace_floats, tfc_stream_t stream);

/* o3.Linear (eigenn/nn/point_conv.py:56-61,79 linear_1; :130-135,180 linear_2; eigenn/nn/nodewise.py:112-117):
 * one tfc_linear_ins per e3nn instruction (a pair of equal irreps, in e3nn's order):
 *     out[n, out_off + w*dim + k] (+)= path_weight * sum_u x[n, in_off + u*dim + k] * W[w_off + u*mul_out + w]
 * forward: x [n_rows, d_in] -> out [n_rows, d_out]; columns of out that no instruction writes are
 * left untouched (zero them beforehand).  `xt` [n_rows * d_in] receives the channel-contiguous copy
 * of x (tfc_ir_transpose) the products read; keep it for the backward.
 * backward: gout [n_rows, d_out]; `gt` [n_rows * d_out] is scratch for its channel-contiguous copy
 * (gt_ready != 0: gt already holds it, gout is not read - lets two consumers of one gradient share
 * the copy); dx (nullable) [n_rows, d_in], columns no instruction reaches left untouched; dw (nullable)
 * [weight_numel] fully written for the blocks of the instructions.  `workspace`: optional split-K
 * scratch of tfc_linear_workspace_floats floats for the weight gradient. */
typedef struct tfc_linear_ins {
    int32_t in_off, mul_in, out_off, mul_out, dim, w_off;
    float path_weight;
} tfc_linear_ins;
int64_t tfc_linear_workspace_floats(const tfc_linear_ins* ins, int32_t n_ins, int64_t n_rows);
int tfc_linear_fwd(const tfc_linear_ins* ins, int32_t n_ins, const float* x, const float* weight, int64_t n_rows,
                   int32_t d_in, int32_t d_out, int64_t weight_numel, float* xt, float* out, tfc_stream_t stream);
/* the same with xt_ready != 0: `xt` already holds the channel-contiguous copy (e.g. written by
 * tfc_conv_fwd_m_t) and x is not read */
int tfc_linear_fwd_t(const tfc_linear_ins* ins, int32_t n_ins, const float* x, const float* weight, int64_t n_rows,
                     int32_t d_in, int32_t d_out, int64_t weight_numel, float* xt, int32_t xt_ready, float* out,
                     tfc_stream_t stream);
int tfc_linear_bwd(const tfc_linear_ins* ins, int32_t n_ins, const float* xt, const float* gout,
                   const float* weight, int64_t n_rows, int32_t d_in, int32_t d_out, int64_t weight_numel,
                   float* gt, int32_t gt_ready, float* dx, float* dw, float* workspace, int64_t workspace_floats,
                   tfc_stream_t stream);

/* o3.FullyConnectedTensorProduct(in1, in2, out) with a SCALAR second input and shared internal
 * weights: the self connection (eigenn/nn/point_conv.py:159-164, called at :185-188).  One
 * tfc_fctp_ins per e3nn instruction (i1, i2, i_out) with ir_out = ir1 (l2 = 0), in e3nn's order:
 *     out[n, out_off + w*dim + k] (+)= alpha * sum_{u,v} W[w_off + (u*mul_v + v)*mul_w + w]
 *                                                * x[n, x_off + u*dim + k] * a[n, a_off + v]
 * forward: accumulate != 0 adds to what `out` holds (e.g. linear_2's result, saving the separate
 * add of point_conv.py:188); otherwise columns no instruction writes are left untouched.
 * backward: gout/gt/gt_ready as for tfc_linear_bwd; dx [n_rows, d_x], dw [weight_numel], da
 * [n_rows, d_a] are each optional (null: not computed).  da needs two scratch buffers: xt
 * [n_rows * d_x] (channel-contiguous copy of x, written by the call) and da_part [n_ins * n_rows * d_a]
 * (one partial per instruction, added in instruction order: deterministic).  `workspace`: optional split-K
 * scratch of tfc_fctp_workspace_floats floats.  The low-rank route of DESIGN.md §4 (attributes that
 * are copies of few rows) is these same calls on the re-associated operands. */
typedef struct tfc_fctp_ins {
    int32_t x_off, mul_u, a_off, mul_v, out_off, mul_w, dim, w_off;
    float alpha;
} tfc_fctp_ins;
int64_t tfc_fctp_workspace_floats(const tfc_fctp_ins* ins, int32_t n_ins, int64_t n_rows);
int tfc_fctp_fwd(const tfc_fctp_ins* ins, int32_t n_ins, const float* x, const float* a, const float* weight,
                 int64_t n_rows, int32_t d_x, int32_t d_a, int32_t d_out, int64_t weight_numel, int32_t accumulate,
                 float* out, float* workspace, int64_t workspace_floats, tfc_stream_t stream);
int tfc_fctp_bwd(const tfc_fctp_ins* ins, int32_t n_ins, const float* x, const float* a, const float* weight,
                 const float* gout, int64_t n_rows, int32_t d_x, int32_t d_a, int32_t d_out, int64_t weight_numel,
                 float* gt, int32_t gt_ready, float* dx, float* dw, float* da, float* xt, float* da_part,
                 float* workspace, int64_t workspace_floats, tfc_stream_t stream);

/* Last step of the attribute gradient when the node attributes are rows of a per-class table (a = table[cls],
 * eigenn/nn/embedding.py:59-75): every row receives the mean gradient of its class,
 *     da[n, v] = inv_count[cls[n]] * sum_i part[i, cls[n], v]      (part [n_parts, r, d_a], i ascending)
 * rows with a class outside [0, r) get zeros.  One launch instead of a sum, a scale and a row gather. */
int tfc_fctp_class_rows(const float* part, int32_t n_parts, int32_t r, int32_t d_a, const float* inv_count,
                        const int64_t* cls, int64_t n_rows, float* da, tfc_stream_t stream);

/* Weight re-associations of the self connection when the node attributes are rows of a small table (a[n] = rows[cls[n]],
 * rows [r, d_a], r a multiple of 4; eigenn/nn/embedding.py:59-75): the layer then contracts over
 *     W'[u, s, w] = sum_v rows[s, a_off + v] W[u, v, w]                      (tfc_fctp_rows_w;  W' block of instruction i at
 *                                                                             sum_{j<i} mul_u_j r mul_w_j, [mul_u][r][mul_w])
 * with tfc_fctp_fwd / _bwd on (one-hot classes, W'), and the weight gradient comes back through
 *     dW[u, v, w]  = sum_s rows[s, a_off + v] dW'[u, s, w]                   (tfc_fctp_rows_dw)
 * - products with r rows / r terms, one short kernel each instead of a GEMM launch whose tile is 94 % padding.  At most 32
 * instructions, mul_v <= 256 (otherwise TFC_EUNSUPPORTED: use tfc_gemm_group).  Fixed summation order. */
int64_t tfc_fctp_rows_numel(const tfc_fctp_ins* ins, int32_t n_ins, int32_t r);
int tfc_fctp_rows_w(const tfc_fctp_ins* ins, int32_t n_ins, const float* rows, int32_t r, int32_t d_a, const float* weight,
                    int64_t weight_numel, float* wp, tfc_stream_t stream);
int tfc_fctp_rows_dw(const tfc_fctp_ins* ins, int32_t n_ins, const float* rows, int32_t r, int32_t d_a, const float* dwp,
                     int64_t weight_numel, float* dw, tfc_stream_t stream);

/* The class table of node attributes that are rows of a per-class table (eigenn/nn/embedding.py:59-75: NODE_ATTRS =
 * embedding(species), so the classes are the species indices and their number is known): in one call, with no host
 * synchronisation (it runs inside captured steps),
 *     onehot[n, s]  = 1 if cls[n] == s else 0          [n_rows, r] fp32, r >= n_classes a multiple of 4
 *     rep_idx[s]    = first row of class s (0 if none), rep_mask[s] = 1 if the class has a row else 0
 *     inv_count[s]  = 1 / max(rows in class s, 1)
 * rows with a class outside [0, n_classes) get a zero one-hot row and count for nothing.  `work`: 2 r int32 of device
 * memory (overwritten).  Counting uses integer atomics: the result does not depend on scheduling. */
int tfc_class_table(const int64_t* cls, int64_t n_rows, int32_t n_classes, int32_t r, int32_t* work, float* onehot,
                    int64_t* rep_idx, float* rep_mask, float* inv_count, tfc_stream_t stream);

/* Channel-contiguous copy of feature rows, for the operands of the dense contractions above:
 * e3nn stores a block as [mul][2l+1] (eigenn feature layout, see the conventions at the top), which
 * gives the contraction index u of o3.Linear / FullyConnectedTensorProduct the stride 2l+1; block b
 * of `out` is [dim_b][n_rows][mul_b] with u fastest:
 *     out[off_b * n_rows + (m * n_rows + n) * mul_b + u] = x[n * d_row + off_b + u * dim_b + m]
 * Columns of x outside every block are not copied. */
typedef struct tfc_ir_block {
    int32_t off; /* first column of the block in a row of x */
    int32_t mul; /* channels */
    int32_t dim; /* 2l+1 */
} tfc_ir_block;
int tfc_ir_transpose(const float* x, int64_t n_rows, int32_t d_row, const tfc_ir_block* blocks,
                     int32_t n_blocks, float* out, tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gate non-linearity (e3nn.nn.Gate built at eigenn/nn/utils.py:134-140, applied at
 * eigenn/nn/point_conv.py:339).  The input row is [scalars | gates | gated]; the output row is
 * [act(scalars) | gate * gated].  One tfc_gate_seg per run of columns sharing an activation.
 */
#define TFC_ACT_IDENTITY 0
#define TFC_ACT_SSP 1
#define TFC_ACT_SILU 2
#define TFC_ACT_TANH 3
#define TFC_ACT_ABS 4
#define TFC_ACT_SIGMOID 5

typedef struct tfc_gate_seg {
    int32_t in_off;   /* first input column of the segment */
    int32_t out_off;  /* first output column */
    int32_t count;    /* number of scalar channels (scalars) or gated channels (mul) */
    int32_t dim;      /* 1 for scalars; 2l+1 for a gated block */
    int32_t gate_off; /* gated block: input column of its first gate; scalars: -1 */
    int32_t act;      /* TFC_ACT_* applied to the scalar / to the gate */
    float cst;        /* normalize2mom constant of that activation */
} tfc_gate_seg;

int tfc_gate_fwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows,
                 int32_t d_in, int32_t d_out, float* out, tfc_stream_t stream);
int tfc_gate_bwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                 int64_t n_rows, int32_t d_in, int32_t d_out, float* dx, tfc_stream_t stream);
/* The same two calls writing a second, channel-contiguous copy of their result (see tfc_conv_fwd_m_t): the
 * (off, mul, dim) `blocks` of the out / dx row go to out_t / dx_t in the layout of tfc_ir_transpose, for the
 * o3.Linear / self connection of the next layer (forward) or of this layer's update (backward).  A null
 * out_t / dx_t makes them the plain calls. */
int tfc_gate_fwd_t(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows, int32_t d_in,
                   int32_t d_out, float* out, const struct tfc_ir_block* blocks, int32_t n_blocks, float* out_t,
                   tfc_stream_t stream);
int tfc_gate_bwd_t(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout, int64_t n_rows,
                   int32_t d_in, int32_t d_out, float* dx, const struct tfc_ir_block* blocks, int32_t n_blocks,
                   float* dx_t, tfc_stream_t stream);

/* Second order of the gate (double backward, see tfc_edge_encode_bwd2): tfc_gate_bwd returns dx = J(x)^T gout;
 * for a cotangent v [n_rows, d_in] of dx this returns ggout = J(x) v [n_rows, d_out] and
 * gx = d/dx (v^T J(x)^T gout) [n_rows, d_in] (either nullable). */
int tfc_gate_bwd2(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout, const float* v,
                  int64_t n_rows, int32_t d_in, int32_t d_out, float* ggout, float* gx, tfc_stream_t stream);

/* Norm activation: the `activation_type="norm"` branch of ActivationLayer (eigenn/nn/utils.py:142-150),
 * e3nn.nn.NormActivation(irreps_in, scalar_nonlinearity, normalize=True, epsilon=1e-8, bias=False):
 * for every irrep copy u of a block, n = sqrt(max(sum_m x[u,m]^2, eps_squared)) (a clamped copy passes
 * no gradient through n), out[u,m] = act(n) / n * x[u,m] (without `normalize`: act(n) * x[u,m]).
 * One tfc_gate_seg per block of the row: in_off = out_off, count = mul, dim = 2l+1, act = TFC_ACT_*
 * (applied as is - NormActivation does not rescale it), gate_off and cst unused.  Same irreps in and
 * out: x, gout, out, dx are [n_rows, d].  eps_squared = 0: no clamp. */
int tfc_norm_act_fwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, int64_t n_rows, int32_t d,
                     float eps_squared, int32_t normalize, float* out, tfc_stream_t stream);
int tfc_norm_act_bwd(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout,
                     int64_t n_rows, int32_t d, float eps_squared, int32_t normalize, float* dx,
                     tfc_stream_t stream);
/* Second order (what autograd derives when the reference trains on a gradient, eigenn/nn/utils.py:142-150): `v` is a
 * cotangent of dx; gx receives the cotangent of x, d/dx (v^T J(x) gout).  The Jacobian is symmetric, so the cotangent
 * of gout is tfc_norm_act_bwd with v in the place of gout. */
int tfc_norm_act_bwd2(const tfc_gate_seg* segs, int32_t n_segs, const float* x, const float* gout, const float* v,
                      int64_t n_rows, int32_t d, float eps_squared, int32_t normalize, float* gx, tfc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Opt-in kernel timing (no counterpart in the reference, whose only timer is the epoch wall
 * clock, eigenn/model/utils.py:4-35).  While enabled, every main kernel launch is bracketed by
 * HIP events on its own stream.  tfc_profile_report() synchronises on those events and writes
 * one "kernel_name launches total_ms" line per kernel into `buf`; it returns the number of
 * lines (>= 0) or a negative TFC_E* code.  This is the library's only global mutable state
 * besides the error string (mutex protected); off by default.  tfc_profile_enable(1) times every
 * kernel, (2) only the convolution kernels (tfc_conv_*: two event records per launch are a
 * measurable host cost when hundreds of short kernels run per step), (0) none.
 */
int tfc_profile_enable(int on);
int tfc_profile_reset(void);
int tfc_profile_report(char* buf, int64_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* TFC_H */
