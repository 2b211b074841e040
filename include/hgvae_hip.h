/* libhgvae_hip.so - C ABI of the MI355X (gfx950) graph-VAE message-passing kernels.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (inputs, outputs, workspaces);
 *    the library never allocates, frees or synchronises.  Every call is asynchronous on
 *    the caller's stream.
 *  - Return value: 0 on success, negative on error (HGV_E*); hgv_last_error() returns the
 *    message of the last failing call made by the calling host thread.
 *  - Matrices are row-major fp32 with an explicit leading dimension (elements per row).
 *    16-byte vector paths are used when base pointers are 16-byte aligned and leading
 *    dimensions are multiples of 4 and cover ceil(F/4)*4 columns (pad columns are then read
 *    and written); otherwise scalar paths run.
 *  - Index arrays inside the library are int32; edge_index at the boundary is int64 as in
 *    PyTorch Geometric.
 *
 * Each entry point cites the reference interface (file:line under the reference tree) whose
 * CPU operator sequence it replaces.  The reference delegates these to PyTorch Geometric
 * (SAGEConv, GATConv, SAGPooling, GCNConv, Batch) and torch.nn.
 */
#ifndef HGVAE_HIP_H
#define HGVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hgv_stream_t; /* == hipStream_t */

#define HGV_OK 0
#define HGV_EINVAL (-1)
#define HGV_EUNSUPPORTED (-2)
#define HGV_ELAUNCH (-3)
#define HGV_EWORKSPACE (-4)

int hgv_version(void);
const char* hgv_last_error(void);

/* ---- CSR construction (no reference counterpart: PyG consumes COO edge_index directly at
 *      gvae/gvae_skipcon.py:105-118; built once per batch and shared by every layer) -------
 * key row = edge_index[1] (targets) when by_src == 0, edge_index[0] (sources) otherwise.
 *   perm   = stable argsort of the key row                      [E]
 *   col    = the other row, permuted                            [E]
 *   rowptr = [0, cumsum(bincount(key, N))]                      [N+1]
 * edge_index element (r, e) is read at edge_index[r*ld_row + e*ld_col] (any strides).
 * status (nullable, caller-zeroed int32): set to 1 if an index was outside [0, N) (such
 * indices are clamped so no kernel reads out of bounds). */
size_t hgv_csr_build_ws(int64_t E, int64_t N);
int hgv_csr_build(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t N, int by_src,
                  int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* status, void* ws, size_t ws_bytes,
                  hgv_stream_t stream);

/* ---- mini-batch assembly on the device from a packed shard of graphs (replaces the per-item torch.load and the Python
 *      Batch.from_data_list of gvae/gvae_skipcon.py:236-263) ---------------------------------------------------------
 * The shard is the disjoint union of all graphs of a dataset: node_ptr / edge_ptr [G+1] (first node / edge of every
 * graph), x_all [Nt, ldx], ei_all [2][e_total] with shard-global node numbers, and per direction the CSR of the union
 * (hgv_csr_build of ei_all): rowptr_all [Nt+1], col_all, perm_all [e_total].  A batch selects graphs ids[0..B) (any
 * order, B <= 65535); out_nptr / out_eptr [B+1] are the prefix sums of their node / edge counts (all index arrays on
 * the device), max_nodes / max_edges the largest count of a selected graph (sizes the launch).
 * hgv_collate_nodes writes x_out [n_out, ldo] (columns [F, ldo) zeroed), batch_out [n_out] (graph number per node,
 * nullable) and the batch's CSR row pointers [n_out+1] (nullable); hgv_collate_edges writes ei_out [2][e_out] and the
 * CSR col / perm arrays of both directions (each group nullable) and the edge attributes [e_out, ea_dim].
 * The CSR arrays equal hgv_csr_build of ei_out bit for bit: graphs are disjoint, so no sort is needed per batch. */
int hgv_collate_nodes(const float* x_all, int64_t ldx, int F, const int64_t* node_ptr, const int64_t* edge_ptr,
                      const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all, const int64_t* ids, int B,
                      const int64_t* out_nptr, const int64_t* out_eptr, int64_t n_out, int64_t max_nodes, float* x_out,
                      int64_t ldo, int64_t* batch_out, int32_t* rowptr_dst_out, int32_t* rowptr_src_out, hgv_stream_t stream);
int hgv_collate_edges(const int64_t* ei_all, int64_t e_total, const int32_t* col_dst_all, const int32_t* perm_dst_all,
                      const int32_t* col_src_all, const int32_t* perm_src_all, const float* ea_all, int ea_dim,
                      const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* ids, int B, const int64_t* out_nptr,
                      const int64_t* out_eptr, int64_t e_out, int64_t max_edges, int64_t* ei_out, int32_t* col_dst_out, int32_t* perm_dst_out,
                      int32_t* col_src_out, int32_t* perm_src_out, float* ea_out, hgv_stream_t stream);

/* ---- graphs from parsed PDB ATOM records, on the device, in the packed shard layout (replaces pdb_to_pyg_data,
 *      gvae/create_graphs_for_gvae.py:67-144: one-hot(atom name) | one-hot(residue name) | 3 force columns per node, one
 *      directed edge from every ATOM record to the next of its file, edge attribute = float32 of the float64 distance) ----
 * atom_idx / res_idx [Nt] int32 category numbers, forces [Nt, 3] (nullable: zeros), coords [Nt, 3] float64 (nullable when
 * ea is null), node_ptr / edge_ptr [G+1] int64 device arrays (a graph of n nodes has n - 1 edges), max_edges = the largest
 * edge count of a graph.  x [Nt, ldx]: 16-byte aligned, ldx a multiple of 4 >= n_atom + n_res + 3, pad columns zeroed;
 * ei [2][Et] union-wide node numbers; ea [Et] (nullable).  status (nullable, caller-zeroed) is set to 1 if a category
 * number was outside its range (it is clamped). */
int hgv_pdb_build(const int32_t* atom_idx, const int32_t* res_idx, const float* forces, const double* coords,
                  const int64_t* node_ptr, const int64_t* edge_ptr, int G, int64_t Nt, int64_t Et, int64_t max_edges,
                  int n_atom, int n_res, float* x, int64_t ldx, int64_t* ei, float* ea, int32_t* status, hgv_stream_t stream);

/* Stable LSD radix sort of (uint32 key, int32 value) pairs on key bits [bit_lo, bit_hi).
 * vals_in may be null (values = positions).  Used by the per-graph top-k of SAGPooling
 * (autoencoder/vae_with_pool.py:39). */
size_t hgv_sort_pairs_ws(int64_t n);
int hgv_sort_pairs_u32(const uint32_t* keys_in, const int32_t* vals_in, int64_t n, int bit_lo, int bit_hi,
                       uint32_t* keys_out, int32_t* vals_out, void* ws, size_t ws_bytes, hgv_stream_t stream);

/* ---- neighbour aggregation: SAGEConv mean aggregation gvae/gvae_skipcon.py:105,110,115,118;
 *      GraphConv sum autoencoder/vae_with_pool.py:39; GATConv weighted sum
 *      gvae/gvae_skipcon_attention.py:109 --------------------------------------------------
 *   out[i,:] = rs_i * sum_{k in [rowptr[i], rowptr[i+1])} w_k * x[col[k], :]  (+ add[i,:])
 *   rs_i: row_mode 0 -> 1; 1 -> 1/max(row length, 1) (true division); 2 -> row_scale[i]
 *   w_k = edge_w[k] (nullable) * nbr_w[col[k]] (nullable)
 * Rows are summed in CSR order; the unweighted result is bit-identical to a sequential
 * scatter_add_ in stable edge order.  col may be null for a graph without edges. */
int hgv_agg_fwd(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* x, int64_t ldx, int F,
                int row_mode, const float* row_scale, const float* edge_w, const float* nbr_w, const float* add,
                int64_t ldadd, float* out, int64_t ldo, hgv_stream_t stream);
/* Backward w.r.t. x (autograd of the above, loss.backward() gvae/gvae_skipcon.py:183): the same
 * reduction over the TRANSPOSED CSR, inv_deg = 1/max(in-degree,1) per target (null for sums). */
int hgv_agg_bwd(const int32_t* rowptr_t, const int32_t* col_t, int64_t n_rows, const float* g, int64_t ldg, int F,
                const float* inv_deg, const float* edge_w_t, const float* add, int64_t ldadd, float* gx,
                int64_t ldgx, hgv_stream_t stream);
/* The same aggregation (same call sites as hgv_agg_fwd) for graphs with hub nodes.  hgv_agg_fwd gives every row to one
 * lane group, which walks it alone: a row of thousands of entries (the out-edges of a hub in the transposed CSR of
 * loss.backward(), gvae/gvae_skipcon.py:183) then outlasts the rest of the launch.  hgv_csr_long_rows lists, once per CSR,
 * the rows with more than `threshold` entries (rows[0 .. min(count, cap)), count[0] zero on entry, order unspecified,
 * cap >= E / threshold + 1 suffices); hgv_agg_fwd_long leaves those rows out of the row-per-group kernel and, with
 * run_long_pass != 0, sums each of them with a 1024-thread workgroup (16 in-order sixteenths of the segment, partial sums
 * added in fixed order: deterministic, but blocked rather than sequential summation for these rows).  run_long_pass = 0
 * is for callers that know the list is empty. */
int hgv_csr_long_rows(const int32_t* rowptr, int64_t n_rows, int threshold, int32_t* rows, int cap, int32_t* count,
                      hgv_stream_t stream);
int hgv_agg_fwd_long(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* x, int64_t ldx, int F,
                     int row_mode, const float* row_scale, const float* edge_w, const float* nbr_w, const float* add,
                     int64_t ldadd, float* out, int64_t ldo, int threshold, const int32_t* long_rows,
                     const int32_t* long_count, int long_cap, int run_long_pass, hgv_stream_t stream);

/* ---- fused (ReLU) -> Dropout -> BatchNorm1d over nodes: gvae/gvae_skipcon.py:105-107,
 *      110-112,115-117,124,130,136 ----------------------------------------------------------
 * y is the pre-activation [N, H].  Dropout keep mask: explicit uint8 [N,H] (contiguous), or
 * Philox4x32-7 (16 random bits per element) keyed by (seed, offset) when mask is null and p > 0; offset_dev (nullable) is a device
 * word added to offset, so a captured graph can draw fresh noise on every replay.  Biased batch variance
 * for normalisation, unbiased for the running update (torch.nn.BatchNorm1d semantics).  hgv_bn_act_bwd takes an optional
 * second incoming gradient g2 (nullable; summed with g on the fly) for outputs that fed two consumers - in the reference
 * every encoder level's output goes to the next level and, as a skip tensor, to the decoder (gvae/gvae_skipcon.py:119-133). */
size_t hgv_bn_act_ws(int64_t N, int H);
int hgv_bn_act_fwd(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
                   uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma, const float* beta, float eps,
                   float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                   float* save_mean, float* save_rstd, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                   hgv_stream_t stream);
/* hgv_bn_act_fwd without a workspace argument and without any allocation.  No counterpart in the reference: it
 * exists for callers that cannot allocate device memory around the call (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  The partial column statistics go to a device array inside the library
 * (2.25 MiB per GPU, held until the process ends) instead of ws; limits: H <= 288 (larger H is rejected with
 * HGV_EUNSUPPORTED - use hgv_bn_act_fwd), any N.  Same kernels and results as hgv_bn_act_fwd, bit for bit.  All calls
 * on one GPU share that array: calls that overlap on one GPU (different streams or host threads) race; calls on one
 * stream are ordered and safe.  The first call on each device resolves the array's address (hipGetSymbolAddress); make
 * that first call outside a graph capture. */
int hgv_bn_act_fwd_static(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask,
                          uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma, const float* beta,
                          float eps, float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                          float* save_mean, float* save_rstd, float* out, int64_t ldo, hgv_stream_t stream);
int hgv_bn_act_eval(const float* y, int64_t ldy, int64_t N, int H, int relu, const float* gamma, const float* beta,
                    float eps, const float* running_mean, const float* running_var, float* out, int64_t ldo,
                    hgv_stream_t stream);
int hgv_bn_act_bwd(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* y, int64_t ldy, int64_t N, int H, int relu, float p,
                   const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* gamma, const float* save_mean,
                   const float* save_rstd, float* dy, int64_t lddy, float* dgamma, float* dbeta, int accumulate,
                   void* ws, size_t ws_bytes, hgv_stream_t stream);
/* Two dropout draws of ONE pre-activation over one read of it.  The reference encodes every batch twice per train
 * step (gvae/gvae_skipcon.py:166-171: model(...), then model.encode(...) again); nothing stochastic precedes the first
 * convolution, so both passes apply relu -> Dropout -> BatchNorm1d (:105-107) to the same matrix with different masks.
 * hgv_bn_act_fwd2 = hgv_bn_act_fwd with Philox offset `offset` into out, then with `offset2` into out2 (running
 * statistics updated in that order, num_batches_tracked += 2), save_mean / save_rstd = [2][H].  hgv_bn_act_bwd2: g / g2
 * are the incoming gradients of out, h / h2 those of out2 (g2, h2 nullable); dy is the SUM of the two input gradients,
 * dgamma / dbeta the sums over both draws.  Results equal those of the two separate calls bit for bit.  Philox noise with
 * p > 0 and H <= 512 only; every matrix 16-byte aligned with a leading dimension that is a multiple of 4 (HGV_EUNSUPPORTED
 * otherwise). */
int hgv_bn_act_fwd2(const float* y, int64_t ldy, int64_t N, int H, int relu, float p, uint64_t seed, uint64_t offset,
                    uint64_t offset2, const uint64_t* offset_dev, const float* gamma, const float* beta, float eps,
                    float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                    float* save_mean, float* save_rstd, float* out, float* out2, int64_t ldo, void* ws, size_t ws_bytes,
                    hgv_stream_t stream);
int hgv_bn_act_bwd2(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* h, int64_t ldh,
                    const float* h2, int64_t ldh2, const float* y, int64_t ldy, int64_t N, int H, int relu, float p,
                    uint64_t seed, uint64_t offset, uint64_t offset2, const uint64_t* offset_dev, const float* gamma,
                    const float* save_mean, const float* save_rstd, float* dy, int64_t lddy, float* dgamma, float* dbeta,
                    int accumulate, void* ws, size_t ws_bytes, hgv_stream_t stream);

/* ---- reparameterisation / KL / MSE: gvae/gvae_skipcon.py:138-142, :153-156, :174 ----------
 * ml = [mean | log_std], [N, 2Z] with leading dimension ldm.
 * eps: explicit [n_samples, N, Z] draws or null (Philox + Box-Muller from seed/offset).
 * eps_bar [N, Z] (mean of the draws) is an output kept for the backward pass. */
size_t hgv_loss_ws(void);
int hgv_reparam_fwd(const float* ml, int64_t ldm, int64_t N, int Z, int n_samples, const float* eps, uint64_t seed,
                    uint64_t offset, const uint64_t* offset_dev, float* z, int64_t ldz, float* eps_bar,
                    hgv_stream_t stream);
int hgv_reparam_bwd(const float* gz, int64_t ldgz, const float* ml, int64_t ldm, int64_t N, int Z,
                    const float* eps_bar, float* gml, int64_t ldg, hgv_stream_t stream);
int hgv_kl_fwd(const float* ml, int64_t ldm, int64_t N, int Z, float* out, void* ws, size_t ws_bytes,
               hgv_stream_t stream);
int hgv_kl_bwd(const float* ml, int64_t ldm, int64_t N, int Z, const float* gout, float* gml, int64_t ldg,
               int accumulate, hgv_stream_t stream);
int hgv_mse_fwd(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F, float* out, void* ws,
                size_t ws_bytes, hgv_stream_t stream);
int hgv_mse_bwd(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F, const float* gout,
                float* gr, int64_t ldg, hgv_stream_t stream);
/* F.mse_loss (gvae/gvae_skipcon.py:174) with its gradient written by the forward pass: hgv_mse_fwd_grad is hgv_mse_fwd that
 * also writes gr = 2 / (N F) (r - x), the gradient for an upstream gradient of exactly 1 (the reference adds the loss terms with
 * weight one, :176); hgv_mse_bwd_unless_unit rewrites gr as hgv_mse_bwd does when *gout != 1 and returns at once when it is 1
 * (decided on the device). */
int hgv_mse_fwd_grad(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F, float* out, float* gr,
                     int64_t ldg, void* ws, size_t ws_bytes, hgv_stream_t stream);
int hgv_mse_bwd_unless_unit(const float* r, int64_t ldr, const float* x, int64_t ldx, int64_t N, int F, const float* gout,
                            float* gr, int64_t ldg, hgv_stream_t stream);

/* Exclusive prefix sum of int32 (in and out may alias). */
size_t hgv_exclusive_scan_ws(int64_t n);
int hgv_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hgv_stream_t stream);

/* ---- single-head graph attention: GATConv(h, h, heads=1, dropout=0.49) at
 *      gvae/gvae_skipcon_attention.py:68,109 (negative_slope 0.2, self loops added) ----------
 * CSR = by-target CSR of the loop-augmented edge list (edges minus self loops, then one loop per
 * node); perm maps CSR position -> index in that edge list (the index the dropout mask uses).
 *   alpha_k  = softmax over row i of leaky_relu(a_src[col[k]] + a_dst[i])  (+1e-16 in the sum)
 *   alpha'_k = alpha_k * keep_k / (1-p)     (mask: explicit uint8 [E'] by edge-list index, or Philox when null and p > 0)
 *   out[i,:] = sum_k alpha'_k * xw[col[k],:] (+ bias)
 * One kernel: softmax, dropout and the weighted sum per target row; no per-edge array is written.  row_max / row_den [N]
 * receive the row maximum of the logits and the normaliser (sum + 1e-16) for the backward pass.  a_src / a_dst are read
 * with an element stride (ld_src / ld_dst).  xw / out: 16-byte aligned, leading dimensions multiples of 4 that cover
 * ceil(C/4)*4 columns; pad columns of out are written as zeros. */
int hgv_gat_fwd(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N, const float* xw, int64_t ldx,
                int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst, float slope, float p,
                const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* bias,
                float* out, int64_t ldo, float* row_max, float* row_den, hgv_stream_t stream);
/* Backward (autograd of the above, loss.backward() gvae/gvae_skipcon_attention.py:181) for the output gradient g [N, C]:
 * gxw [N, C] = d xw, da_src [N], da_dst [N].  Takes both CSRs of the same edge list (rowptr_t / col_t / perm_t = by
 * source), the forward's noise arguments and row_max / row_den; S [2N] is scratch.  Two kernels (one per CSR direction),
 * each recomputing the attention weights of its edges; same layout contract as hgv_gat_fwd for g / xw / gxw.  da_src /
 * da_dst are written with element stride ld_da.  att_src / att_dst (nullable, padded to ceil(C/4)*4 floats, 16-byte
 * aligned): if the logits are a_src = <xw, att_src>, a_dst = <xw, att_dst> (GATConv's att parameters), the part of d xw
 * that flows through them, da_src[j] att_src + da_dst[j] att_dst, is added to gxw in the same pass. */
int hgv_gat_bwd(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* rowptr_t, const int32_t* col_t,
                const int32_t* perm_t, int64_t N, const float* g, int64_t ldg, const float* xw, int64_t ldx, int C,
                const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst, float slope, float p,
                const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* row_max,
                const float* row_den, float* S, float* gxw, int64_t ldgx, float* da_src, float* da_dst, int64_t ld_da,
                const float* att_src, const float* att_dst, hgv_stream_t stream);

/* hgv_gat_fwd / hgv_gat_bwd with hub rows handled by a workgroup each (GATConv over graphs whose degrees are very uneven,
 * gvae/gvae_skipcon_attention.py:109): rows with more than long_threshold (> 0) entries - listed by hgv_csr_long_rows with
 * the same threshold in long_rows[min(*long_count, long_cap)] - are left out of the lane-group kernels and summed in
 * eight in-order parts by a 512-thread workgroup (same value to rounding; no atomics).  run_long_pass = 0 skips the pass
 * when the caller knows the list is empty.  The backward pass takes one list per CSR (by target; by source: suffix _t). */
int hgv_gat_fwd_long(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t N, const float* xw, int64_t ldx,
                     int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst, float slope, float p,
                     const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, const float* bias,
                     float* out, int64_t ldo, float* row_max, float* row_den, int long_threshold, const int32_t* long_rows,
                     const int32_t* long_count, int long_cap, int run_long_pass, hgv_stream_t stream);
int hgv_gat_bwd_long(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* rowptr_t,
                     const int32_t* col_t, const int32_t* perm_t, int64_t N, const float* g, int64_t ldg, const float* xw,
                     int64_t ldx, int C, const float* a_src, int64_t ld_src, const float* a_dst, int64_t ld_dst, float slope,
                     float p, const uint8_t* mask, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                     const float* row_max, const float* row_den, float* S, float* gxw, int64_t ldgx, float* da_src,
                     float* da_dst, int64_t ld_da, const float* att_src, const float* att_dst, int long_threshold,
                     const int32_t* long_rows, const int32_t* long_count, int long_cap, int run_long_pass,
                     int long_threshold_t, const int32_t* long_rows_t, const int32_t* long_count_t, int long_cap_t,
                     int run_long_pass_t, hgv_stream_t stream);

/* ---- SAGPooling: autoencoder/vae_with_pool.py:25,36,39 and :42-50;
 *      autoencoder/vae_custom_pool.py:56-58 ------------------------------------------------
 * hgv_topk_perm: per graph (ascending graph id) the k[g] highest-scoring nodes, descending score,
 * ties by ascending node id.  batch [N], ptr [B+1], k [B], kptr [B+1] are int64 device arrays.  max_nodes: node count
 * of the largest graph when the caller knows it (<= 0: unknown); when it is <= 16384 every graph is sorted by one
 * workgroup in LDS (one launch, ws unused), otherwise global radix-sort passes over all N scores run.  A max_nodes
 * smaller than a graph's real node count is a caller error: that graph's entries of perm then hold valid node ids of
 * the graph in no particular order (nothing outside the buffers is touched). */
size_t hgv_topk_ws(int64_t N);
int hgv_topk_perm(const float* score, const int64_t* batch, const int64_t* ptr, const int64_t* k,
                  const int64_t* kptr, int64_t N, int64_t B, int64_t max_nodes, int64_t* perm, void* ws, size_t ws_bytes,
                  hgv_stream_t stream);
/* filter_adj: keep edges with both ends in perm, relabel to positions in perm, original order.
 * out_ei: [2, cap] int64 (row stride ld_out, cap >= E suffices); kept_ids (nullable): original
 * edge id of every kept edge; count[0]: number of kept edges.  fill_tail != 0: slots [count, E) of out_ei / kept_ids are
 * written as zeros, so the whole capacity holds valid indices (for callers that do not read count back). */
size_t hgv_filter_adj_ws(int64_t E, int64_t N);
int hgv_filter_adj(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t N,
                   const int64_t* perm, int64_t K, int64_t* out_ei, int64_t ld_out, int32_t* kept_ids,
                   int32_t* count, int fill_tail, void* ws, size_t ws_bytes, hgv_stream_t stream);
/* out[q,:] = x[idx[q],:] * (scale ? scale[idx[q]] : 1)   (x[perm] * score[perm], x[perm]) */
int hgv_gather_rows(const float* x, int64_t ldx, const int64_t* idx, const float* scale, int64_t K, int F,
                    float* out, int64_t ldo, hgv_stream_t stream);
/* Backward: gx[idx[q],:] = g[q,:] * scale[idx[q]] (other rows untouched: caller zeroes gx),
 * gscale[idx[q]] = <g[q,:], x[idx[q],:]>.  idx must hold distinct rows. */
int hgv_gather_rows_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const int64_t* idx,
                        const float* scale, int64_t K, int F, float* gx, int64_t ldgx, float* gscale,
                        hgv_stream_t stream);
int hgv_gather_i64(const int64_t* src, const int64_t* idx, int64_t n, int64_t* out, hgv_stream_t stream);

/* ---- fp32 MFMA projections: SAGEConv.lin_l / lin_r and the decoder Linear layers
 *      (gvae/gvae_skipcon.py:61-100) and their autograd --------------------------------------
 * hgv_gemm_nt:  C[M,N] = A1[M,K1] W1^T (+ A2[M,K2] W2^T) (+ bias[N]);  weight element (n,k) of source s
 * is read at Ws[n*sn_s + k*sk_s] ((ld,1): row-major [N,K] weight; (1,ld): multiply by a row-major [K,N]
 * matrix, i.e. the grad-input product).  A2 may be null.  A1/A2: 16-byte aligned, lda % 4 == 0,
 * lda >= ceil(K/4)*4, pad columns [K, lda) finite.  Columns [N, ldc) of C are written as zeros.  N <= 288.
 * The weights are first packed into ws (hgv_gemm_nt_ws bytes); callers that reuse a weight can pack once
 * with hgv_pack_weight (hgv_pack_bytes bytes per weight) and call hgv_gemm_nt_packed.
 * hgv_gemm_tn:  GW[N,K] (row-major, leading dimension ldgw >= K, so that it can be a column block of a wider
 * weight gradient) = G[M,N]^T A[M,K];  gbias[N] = column sums of G (nullable); accumulate != 0 adds to GW / gbias
 * instead of overwriting them (accumulation into an existing .grad buffer).  G, A: row-major, any 4-byte
 * aligned pointers, leading dimensions >= their widths.  The reduction over M is split over workgroups and
 * combined in a fixed order (reproducible). */
int hgv_pack_rows(int N);
size_t hgv_pack_bytes(int N, int K);
int hgv_pack_weight(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, hgv_stream_t stream);
/* Up to 4 weights (or orientations of weights) packed by one launch; host arrays of njobs entries. */
int hgv_pack_weights(int njobs, const float* const* W, const int64_t* sn, const int64_t* sk, const int* N, const int* K,
                     float* const* out, hgv_stream_t stream);
/* One weight into rows [row_offset, row_offset+N) of the packed image of an [n_total, K] weight that stacks
 * several weights along N (rows of the image not covered by a weight must be zero). */
int hgv_pack_weight_rows(const float* W, int64_t sn, int64_t sk, int N, int K, float* out, int row_offset,
                         int n_total, hgv_stream_t stream);
int hgv_gemm_nt_packed(const float* A1, int64_t lda1, int K1, const float* Wp1, const float* A2, int64_t lda2, int K2,
                       const float* Wp2, const float* bias, int64_t M, int N, float* C, int64_t ldc,
                       hgv_stream_t stream);
size_t hgv_gemm_nt_ws(int N, int K1, int K2);
int hgv_gemm_nt(const float* A1, int64_t lda1, int K1, const float* W1, int64_t sn1, int64_t sk1, const float* A2,
                int64_t lda2, int K2, const float* W2, int64_t sn2, int64_t sk2, const float* bias, int64_t M, int N,
                float* C, int64_t ldc, void* ws, size_t ws_bytes, hgv_stream_t stream);
size_t hgv_gemm_tn_ws(int64_t M, int N, int K);
int hgv_gemm_tn(const float* G, int64_t ldg, int N, const float* A, int64_t lda, int K, int64_t M, float* GW,
                int64_t ldgw, float* gbias, int accumulate, void* ws, size_t ws_bytes, hgv_stream_t stream);
/* Two weight gradients that share G in one pass over it (the halves of a K-concatenated projection:
 * SAGEConv's lin_l / lin_r, gvae/gvae_skipcon.py:61, and a decoder level fed by [h | skip], :92-100):
 * GW1[N,K1] (+)= G^T A1, GW2[N,K2] (+)= G^T A2; GW1 / GW2 may be column blocks of one matrix. */
size_t hgv_gemm_tn2_ws(int64_t M, int N, int K1, int K2);
int hgv_gemm_tn2(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                 const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias, int accumulate,
                 void* ws, size_t ws_bytes, hgv_stream_t stream);

/* Two weight gradients that share A in one pass: GWa[N,K] (+)= Ga^T A, GWb[N,K] (+)= Gb^T A for two gradient matrices of
 * the same width N (2N <= 288), gbias_a / gbias_b (+)= their column sums (both or neither).  The backward pass of a
 * SAGEConv whose projection runs before the aggregation (the mu / log-std convolutions, gvae/gvae_skipcon.py:113-118):
 * W_l meets the transposed aggregate of the output gradient, W_r the output gradient itself, both against the layer input. */
size_t hgv_gemm_tn_gg_ws(int64_t M, int N, int K);
int hgv_gemm_tn_gg(const float* Ga, int64_t ldga, const float* Gb, int64_t ldgb, int N, const float* A, int64_t lda, int K,
                   int64_t M, float* GWa, int64_t ldgwa, float* GWb, int64_t ldgwb, float* gbias_a, float* gbias_b,
                   int accumulate, void* ws, size_t ws_bytes, hgv_stream_t stream);
/* The same on the bf16 matrix cores, as one product: GW[2N, K] (+)= [Ga | Gb]^T A, gbias[2N] (+)= column sums of Ga, then
 * of Gb; N a multiple of 4, shapes as hgv_gemm_tn16_supported(M, 2N, K, 0) says, workspace hgv_gemm_tn16_ws(M, 2N, K, 0). */
int hgv_gemm_tn16_gg(const float* Ga, int64_t ldga, const float* Gb, int64_t ldgb, int N, const float* A, int64_t lda, int K,
                     int64_t M, float* GW, int64_t ldgw, float* gbias, int accumulate, void* ws, size_t ws_bytes,
                     hgv_stream_t stream);

/* ---- the same projections on the bf16 matrix cores at fp32 accuracy ("bf16x6": every operand split into three
 *      bf16 pieces, six piece products per fp32 product, fp32 accumulation) -------------------------------------
 * hgv_gemm_nt16_supported: 1 when the split image of an N x (K1 + K2) weight, whole or in two parts along N,
 * fits in LDS (then hgv_pack16_bytes > 0).  hgv_pack16_weights builds the image of [W1 | W2] (W2 nullable,
 * element (n,k) of source s at Ws[n*sn_s + k*sk_s]); hgv_pack16_weights_multi builds up to 24 images in one
 * launch (arrays of njobs entries; W2[q] may be null with K2[q] = 0); hgv_gemm_nt16 computes
 * C = A1 W1^T (+ A2 W2^T) (+ bias) under the contract of hgv_gemm_nt_packed (A, C 16-byte aligned, leading
 * dimensions multiples of 4) in one launch, whether the image has one part or two. */
int hgv_gemm_nt16_supported(int N, int K1, int K2);
size_t hgv_pack16_bytes(int N, int K1, int K2);
int hgv_pack16_weights(const float* W1, int64_t sn1, int64_t sk1, int K1, const float* W2, int64_t sn2, int64_t sk2, int K2,
                       int N, void* out, hgv_stream_t stream);
int hgv_pack16_weights_multi(int njobs, const float* const* W1, const int64_t* sn1, const int64_t* sk1, const int* K1,
                             const float* const* W2, const int64_t* sn2, const int64_t* sk2, const int* K2, const int* N,
                             void* const* out, hgv_stream_t stream);
int hgv_gemm_nt16(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                  const float* bias, int64_t M, int N, float* C, int64_t ldc, hgv_stream_t stream);

/* ---- the same projections on the fp16 matrix cores with scaled two-piece operands ("f16x3": every operand times a
 *      power of two that brings its row's largest entry to [2^12, 2^13), split into two fp16 pieces, three piece
 *      products per fp32 product, fp32 accumulation, exact rescaling at the end; dropped terms <= 3 * 2^-22 per
 *      product).  Replaces the same reference calls as hgv_gemm_nt16 (the lin_l / lin_r / Linear products of
 *      gvae/gvae_skipcon.py:61-100) at half its matrix-core work.  Same contracts as the hgv_*16 functions above;
 *      the image holds two pieces per element plus one inverse scale per weight row.  Pad columns [K, lda) of A must
 *      be zero (a large finite pad value would set the row's scale). */
int hgv_gemm_nt16h_supported(int N, int K1, int K2);
size_t hgv_pack16h_bytes(int N, int K1, int K2);
int hgv_pack16h_weights_multi(int njobs, const float* const* W1, const int64_t* sn1, const int64_t* sk1, const int* K1,
                              const float* const* W2, const int64_t* sn2, const int64_t* sk2, const int* K2, const int* N,
                              void* const* out, hgv_stream_t stream);
int hgv_gemm_nt16h(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                   const float* bias, int64_t M, int N, float* C, int64_t ldc, hgv_stream_t stream);

/* ---- levels whose normalisation is deferred to the consumers -----------------------------------------------------
 * A level of the reference is  y = conv-or-Linear(x) -> relu? -> Dropout -> BatchNorm1d  and the result feeds the next
 * SAGEConv / Linear (gvae/gvae_skipcon.py:105-118, :122-136).  hgv_bn_stats_act overwrites y with d = dropout(relu(y))
 * (without a relu a dropped element is -0.0f, a kept zero +0.0f) in the pass that takes the column statistics of d and
 * returns the normalisation as two vectors (out = d * col_s + col_t); the consumers then use d with them:
 * hgv_pack16h_weights_affine folds them into the weight image and bias of a following projection, hgv_agg_fwd_affine
 * applies them to the finished rows of a mean aggregation, hgv_wgrad_finish corrects a weight gradient that was formed
 * with d in place of the normalised operand (G^T x = (G^T d) diag(s) + colsum(G) t^T).  The normalised matrix is never
 * written; hgv_bn_apply writes it for a consumer that needs it after all.  hgv_bn_act_bwd_pre is hgv_bn_act_bwd reading d
 * (relu and dropout are read off the stored values). */
int hgv_bn_stats_act(float* y, int64_t ldy, int64_t N, int H, int relu, float p, const uint8_t* mask, uint64_t seed,
                     uint64_t offset, const uint64_t* offset_dev, const float* gamma, const float* beta, float eps,
                     float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked, float* save_mean,
                     float* save_rstd, float* col_s, float* col_t, void* ws, size_t ws_bytes, hgv_stream_t stream);

/* Keep bits of one dropout call over [N, H] - torch.nn.Dropout of the reference's level tails: the decisions
 * hgv_bn_stats_act / hgv_bn_act_fwd take for the same explicit mask ([N, H] bytes, non-zero = keep) or, with mask ==
 * NULL, the same Philox (p, seed, offset, offset_dev).  bits: [N, ld_bits] 32-bit words, ld_bits >= ceil(4 * ceil(H / 4)
 * / 32); bit c & 31 of word c >> 5 = element (row, c) is kept. */
int hgv_dropout_bits(int64_t N, int H, float p, const uint8_t* mask, uint64_t seed, uint64_t offset,
                     const uint64_t* offset_dev, uint32_t* bits, int64_t ld_bits, hgv_stream_t stream);

/* The statistics half of hgv_bn_stats_act for a matrix that already holds d = dropout(relu(y)) (hgv_gemm_nt16h_act stored
 * it): one read, no write; mean / rstd / col_s / col_t / running statistics equal hgv_bn_stats_act's bit for bit. */
int hgv_bn_stats_pre(const float* d, int64_t ldd, int64_t N, int H, const float* gamma, const float* beta, float eps,
                     float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked, float* save_mean,
                     float* save_rstd, float* col_s, float* col_t, void* ws, size_t ws_bytes, hgv_stream_t stream);

/* The merge step of the level statistics on its own: partial[nparts][2][Hp] = (mean, M2) of part p over the rows of the
 * group_rows-row groups p, p + nparts, ... of [0, N).  Writes what hgv_bn_stats_act writes. */
int hgv_bn_stats_merge(const float* partial, int nparts, int group_rows, int64_t N, int H, const float* gamma, const float* beta,
                       float eps, float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked,
                       float* save_mean, float* save_rstd, float* col_s, float* col_t, hgv_stream_t stream);
int hgv_pack16h_weights_affine(const float* W1, int64_t sn1, int64_t sk1, int K1, const float* cs1, const float* ct1,
                               int indicator1, const float* W2, int64_t sn2, int64_t sk2, int K2, const float* cs2,
                               const float* ct2, int N, const float* bias_in, float* bias_out, void* out, hgv_stream_t stream);
/* mean aggregation that also writes a has-neighbours indicator (1.0f / 0.0f) into the pad column ind_col of every output
 * row: mean_j(d_j s + t) = mean_j(d_j) s + t * indicator, so [mean(d) | indicator] is the operand of a projection whose
 * image holds [W diag(s) | W t] (indicator1 of hgv_pack16h_weights_affine) */
int hgv_agg_fwd_ind(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* d, int64_t ldd, int F, float* out,
                    int64_t ldo, int ind_col, int threshold, const int32_t* long_rows, const int32_t* long_count, int long_cap,
                    int run_long_pass, hgv_stream_t stream);
int hgv_agg_fwd_affine(const int32_t* rowptr, const int32_t* col, int64_t n_rows, const float* a, int64_t lda, int F,
                       const float* col_s, const float* col_t, float* out, int64_t ldo, int threshold,
                       const int32_t* long_rows, const int32_t* long_count, int long_cap, int run_long_pass,
                       hgv_stream_t stream);
int hgv_bn_apply(const float* d, int64_t ldd, int64_t N, int H, const float* gamma, const float* beta, const float* save_mean,
                 const float* save_rstd, float* out, int64_t ldo, hgv_stream_t stream);
int hgv_bn_act_bwd_pre(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* d, int64_t ldd, int64_t N,
                       int H, int relu, float p, const float* gamma, const float* save_mean, const float* save_rstd, float* dy,
                       int64_t lddy, float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes,
                       hgv_stream_t stream);
/* hgv_bn_act_bwd_pre without its statistics pass: when every consumer of the level was a projection that took it as
 * (d, s, t), the incoming gradient is g = sum_p G_p W_p and the column sums of g and of g * d follow from the consumers'
 * weight-gradient products raw_p = G_p^T d [n_p, H] and csum_p = column sums of G_p:
 * sum_i g_ic = sum_p sum_n W_p[n,c] csum_p[n * cs_stride_p], sum_i g_ic d_ic = sum_p sum_n W_p[n,c] raw_p[n,c]. */
int hgv_bn_act_bwd_pre_alg(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* d, int64_t ldd, int64_t N,
                           int H, int relu, float p, const float* gamma, const float* save_mean, const float* save_rstd,
                           int nparts, const float* const* W, const int64_t* ldw, const float* const* raw, const int64_t* ldraw,
                           const float* const* csum, const int64_t* cs_stride, const int* n, float* dy, int64_t lddy,
                           float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, hgv_stream_t stream);
/* The backward of a level (gvae/gvae_skipcon.py:124,130,136 and the head convolution's operand, :117-118) inside the grad-input
 * product of its ONLY consumer.  hgv_bn_bwd_sums_alg: the two column sums of hgv_bn_act_bwd_pre_alg by themselves,
 * sums[2][4 * ceil(H / 4)] = (sum_i g_ic, sum_i g_ic xhat_ic), dgamma / dbeta (nullable) written or added to.
 * hgv_gemm_nt16h_bwd: hgv_gemm_nt16h (no bias) that stores, instead of g = A1 W1^T (+ A2 W2^T), the level's
 *   dy = k * gamma rstd (g - sums[0] / M - xhat sums[1] / M),  xhat = (d - mean) rstd,  k = 1 / (1 - p) where d was kept, else 0
 * (relu != 0: kept = d > 0, else the dropped elements are the negative zeros; gamma nullable = ones), i.e. what
 * hgv_bn_act_bwd_pre_alg makes of g in a pass of its own.  Shapes: hgv_gemm_nt16h_bwd_supported (outputs of 33..144 columns
 * from one source of 97..160 or two sources of 97..128 / 257..288 concatenated columns); lddy <= 144, ldd >= lddy; the pad columns of dy
 * receive zeros.  addend (nullable, [M, N], leading dimension ldadd): g = A1 W1^T (+ A2 W2^T) + addend, for a level with two
 * consumers (a skip connection, gvae/gvae_skipcon.py:110-112 with :128-130) - the first consumer's grad-input product writes its
 * share as usual, the second one's adds it and applies the level; one source of 129..160 or two of 129..144 columns only. */
int hgv_bn_bwd_sums_alg(int H, const float* save_mean, const float* save_rstd, int nparts, const float* const* W, const int64_t* ldw,
                        const float* const* raw, const int64_t* ldraw, const float* const* csum, const int64_t* cs_stride,
                        const int* n, float* sums, float* dgamma, float* dbeta, int accumulate, hgv_stream_t stream);
int hgv_gemm_nt16h_bwd_supported(int64_t M, int N, int K1, int K2, int addend);
int hgv_gemm_nt16h_bwd(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp, int64_t M,
                       int N, float* dy, int64_t lddy, const float* d, int64_t ldd, int relu, float p, const float* gamma,
                       const float* mean, const float* rstd, const float* sums, const float* addend, int64_t ldadd,
                       hgv_stream_t stream);
/* The two-draw forms (the first level of the reference's two encode passes, gvae/gvae_skipcon.py:166-171, shares one
 * pre-activation): hgv_bn_stats_act2 overwrites y with the first draw's d and writes the second draw's to d2 over one read
 * of y (statistics [2][H], col_s / col_t [2][4 * ceil(H / 4)], running statistics updated for the first draw, then the
 * second); hgv_bn_act_bwd2_pre is the backward of both in one pass (dy = sum of the two input gradients) with the column
 * sums from the consumers' products (nparts0 / nparts1 > 0, parts as for hgv_bn_act_bwd_pre_alg) or from the matrices (0 / 0). */
int hgv_bn_stats_act2(float* y, float* d2, int64_t ldy, int64_t N, int H, int relu, float p, uint64_t seed, uint64_t offset,
                      uint64_t offset2, const uint64_t* offset_dev, const float* gamma, const float* beta, float eps,
                      float* running_mean, float* running_var, float momentum, int64_t* num_batches_tracked, float* save_mean,
                      float* save_rstd, float* col_s, float* col_t, void* ws, size_t ws_bytes, hgv_stream_t stream);
int hgv_bn_act_bwd2_pre(const float* g, int64_t ldg, const float* g2, int64_t ldg2, const float* h, int64_t ldh, const float* h2,
                        int64_t ldh2, const float* d1, const float* d2, int64_t ldd, int64_t N, int H, int relu, float p,
                        const float* gamma, const float* save_mean, const float* save_rstd, int nparts0, int nparts1,
                        const float* const* W, const int64_t* ldw, const float* const* raw, const int64_t* ldraw,
                        const float* const* csum, const int64_t* cs_stride, const int* n, float* dy, int64_t lddy, float* dgamma,
                        float* dbeta, int accumulate, void* ws, size_t ws_bytes, hgv_stream_t stream);
int hgv_wgrad_finish(int njobs, float* const* dst, const int64_t* lddst, const float* const* raw, const int64_t* ldraw,
                     const int* N, const int* K, const float* const* s, const float* const* t, const float* const* colsum,
                     const int64_t* cs_stride, int accumulate, hgv_stream_t stream);

/* Range guard of the two-piece fp16 split (no counterpart in the reference: it serves the "auto" projection arithmetic of
 * the lin_l / lin_r / Linear products, gvae/gvae_skipcon.py:61-100).  hgv_row_range_flag sets flag[0] (device int) to 1
 * when some row of [A1 | A2] (A2 nullable; buffer rules of hgv_gemm_nt16h) holds a non-zero entry more than 2^log2_limit
 * below the largest entry of that row, else to 0.  The _gated products are hgv_gemm_nt16h / hgv_gemm_nt16 that do their
 * work only when (*gate != 0) == run_if_set and otherwise leave C untouched, so a caller can enqueue both arithmetics
 * behind one flag without a host round trip: exactly one of them writes C. */
int hgv_row_range_flag(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, int64_t M, int log2_limit,
                       int* flag, hgv_stream_t stream);
int hgv_gemm_nt16h_gated(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                         const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int run_if_set,
                         hgv_stream_t stream);

/* hgv_gemm_nt16h that stores the head of a level's relu -> Dropout -> BatchNorm1d tail itself (gvae/gvae_skipcon.py:105-107):
 * C <- dropout(relu(C)) (relu optional), keep bit of element (row, c) = bit c & 31 of keep_bits[row * ld_bits + (c >> 5)]
 * (hgv_dropout_bits), kept values times scale = 1 / (1 - p); a dropped element is stored as -0.0f when relu == 0 and a
 * kept zero as +0.0f, as hgv_bn_stats_act leaves them.  One-part images of 9 column tiles only (129..144 outputs:
 * hgv_gemm_nt16h_act_supported); hgv_bn_stats_pre takes the statistics of the stored matrix. */
int hgv_gemm_nt16h_act_supported(int N, int K1, int K2);
int hgv_gemm_nt16h_act(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                       const float* bias, int64_t M, int N, float* C, int64_t ldc, const uint32_t* keep_bits, int64_t ld_bits,
                       int relu, float scale, hgv_stream_t stream);

/* hgv_gemm_nt16h_act that also leaves the column statistics of what it stored (the batch statistics of the level's
 * BatchNorm1d, gvae/gvae_skipcon.py:107): stat_partial[parts][2][Hp] (Hp = 4 * ceil(N / 4), parts =
 * hgv_gemm_nt16h_act_parts(M, N, K1, K2)) receives per part p the (mean, M2) over the rows of the 16-row groups p, p + parts, ... ;
 * hgv_bn_stats_merge(partial, parts, 16, M, N, ...) finishes them.  Plain fp32 sums of d and d^2 per part (M2 formed in
 * fp64): meant for levels behind a dropout of rate p >= 0.2, whose zeros bound mean^2 / variance of d by (1 - p) / p. */
int hgv_gemm_nt16h_act_parts(int64_t M, int N, int K1, int K2);
int hgv_gemm_nt16h_act_stats(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                             const float* bias, int64_t M, int N, float* C, int64_t ldc, const uint32_t* keep_bits,
                             int64_t ld_bits, int relu, float scale, float* stat_partial, hgv_stream_t stream);
/* hgv_gemm_nt16h_act_stats for TWO dropout draws of the one product - the first level of the reference's two encode passes
 * shares one pre-activation (gvae/gvae_skipcon.py:166-171): C / stat_partial take the draw of keep_bits, C2 / stat_partial2
 * the draw of keep_bits2 (same leading dimensions; hgv_gemm_nt16h_act2_parts(M) parts each).  Two sources of 97...112 columns
 * each, 129...144 outputs (hgv_gemm_nt16h_act2_supported). */
int hgv_gemm_nt16h_act2_supported(int64_t M, int N, int K1, int K2);
int hgv_gemm_nt16h_act2_parts(int64_t M);
int hgv_gemm_nt16h_act2_stats(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                              const float* bias, int64_t M, int N, float* C, float* C2, int64_t ldc, const uint32_t* keep_bits,
                              const uint32_t* keep_bits2, int64_t ld_bits, int relu, float scale, float* stat_partial,
                              float* stat_partial2, hgv_stream_t stream);
int hgv_gemm_nt16_gated(const float* A1, int64_t lda1, int K1, const float* A2, int64_t lda2, int K2, const void* Wp,
                        const float* bias, int64_t M, int N, float* C, int64_t ldc, const int* gate, int run_if_set,
                        hgv_stream_t stream);

/* Weight gradients of the same projections on the bf16 matrix cores (three-piece split of both operands, six piece
 * products, fp32 accumulation per row slice, fp64 across slices): contract of hgv_gemm_tn2 with A2 / GW2 nullable, plus
 * 16-byte aligned G / A1 / A2 with leading dimensions that are multiples of 4.  hgv_gemm_tn16_supported: 1 for shapes
 * it runs (N <= 288, enough rows, 144 x 144 tiles mostly live); otherwise use hgv_gemm_tn / hgv_gemm_tn2. */
int hgv_gemm_tn16_supported(int64_t M, int N, int K1, int K2);
size_t hgv_gemm_tn16_ws(int64_t M, int N, int K1, int K2);
int hgv_gemm_tn16(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                  const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias, int accumulate,
                  void* ws, size_t ws_bytes, hgv_stream_t stream);

/* hgv_gemm_tn16 from two scaled fp16 pieces per operand and three piece products (one power-of-two scale per workgroup
 * and operand, lowered while the rows stream by); same shapes (hgv_gemm_tn16_supported), workspace (hgv_gemm_tn16_ws) and
 * buffer rules; pad columns of G / A must be zero. */
int hgv_gemm_tn16h(const float* G, int64_t ldg, int N, const float* A1, int64_t lda1, int K1, float* GW1, int64_t ldgw1,
                   const float* A2, int64_t lda2, int K2, float* GW2, int64_t ldgw2, int64_t M, float* gbias, int accumulate,
                   void* ws, size_t ws_bytes, hgv_stream_t stream);

/* ---- clustering export: pairwise distances of the atoms of one structure (scipy.spatial.distance_matrix at
 *      gvae/gvae_skipcon.py:352-354, float64) ----------------------------------------------------------------------
 * coords [n, dim] float64 (dim 1..4, contiguous) -> out [n, n] float64 with leading dimension ld >= n; rowsum [n]
 * (nullable) receives the row sums of the matrix (= its column sums: what the PCA step centres with). */
int hgv_distance_matrix(const double* coords, int64_t n, int dim, double* out, int64_t ld, double* rowsum, hgv_stream_t stream);

/* ---- spectral terms: dense normalised Laplacian L = I - D^-1/2 A D^-1/2 (autoencoder/vae_laplacian.py:24-46, the matrix
 *      whose eigenvalues l_frequency :170-191 and compute_fiedler_value autoencoder/vae_custom_pool.py:50-53 take) -------------
 * edge_index int64 with element (0, e) at edge_index[e * ld_col] and (1, e) at edge_index[ld_row + e * ld_col]; A[r][c] counts
 * the edges r -> c; dis [n] = degree^-1/2 over sources with 0 for isolated nodes (deg.pow(-0.5), inf -> 0).  out [n, ld]
 * float32, ld >= n, element (r, c) = (r == c) - (dis[r] * A[r][c]) * dis[c] in that rounding order (the reference's two
 * products with a diagonal matrix); columns >= n are zero.  status (nullable device int32, zeroed by the caller) gets bit 0
 * when an edge names a node outside [0, n): the reference raises ValueError there, such edges are skipped here.  n <= 65535. */
int hgv_normalized_laplacian(const int64_t* edge_index, int64_t ld_row, int64_t ld_col, int64_t E, int64_t n, const float* dis,
                             float* out, int64_t ld, int32_t* status, hgv_stream_t stream);

/* ---- optimizer step on flat buffers: torch.optim.Adam (gvae/gvae_skipcon.py:294, autoencoder/vae_with_pool.py:270)
 *      and torch.nn.utils.clip_grad_norm_ (autoencoder/vae_custom_pool.py:228-229) ------------------------------
 * hgv_adam_step: torch's Adam arithmetic (bias-corrected, eps added after the sqrt(v)/sqrt(1-b2^t) division,
 * L2 weight decay folded into the gradient) on n contiguous floats; step counts from 1; step_dev (nullable device
 * int64) overrides step so that a captured graph can advance it on the device; gscale_dev (nullable device float)
 * multiplies the gradient first (the clip coefficient).
 * hgv_grad_norm_clip: norm_out[0] = ||g||_2, scale_out[0] = min(1, max_norm / (norm + 1e-6)). */
int hgv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, const int64_t* step_dev, const float* gscale_dev, hgv_stream_t stream);
size_t hgv_grad_norm_ws(void);
int hgv_grad_norm_clip(const float* g, int64_t n, float max_norm, float* norm_out, float* scale_out, void* ws,
                       size_t ws_bytes, hgv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HGVAE_HIP_H */
