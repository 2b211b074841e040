// Mini-batch assembly on the device from a packed shard of graphs.
//
// Replaces the per-item torch.load + Python Batch.from_data_list of the reference's loader
// (gvae/gvae_skipcon.py:236-263: CustomGraphDataset.__getitem__, collate_fn): all graphs of a
// dataset live on the device as ONE disjoint union (features, edge list, and the CSR of the union in
// both directions, built once), and a batch is a gather of the selected graphs with their node and
// edge numbers re-based.  Because graphs are disjoint components stored one after the other, the
// CSR of a batch is the concatenation of the selected graphs' CSR segments with shifted offsets:
// no sort is needed per batch, and the result is bit-identical to building the CSR from the
// batch's edge_index (stable order inside every row is the graph's own edge order in both).
//
// Shard arrays: node_ptr / edge_ptr [G+1] (first node / edge of every graph), x_all [Nt, ldx],
// ei_all [2][Et] (shard-global node numbers), and per direction rowptr_all [Nt+1], col_all [Et]
// (shard-global node numbers), perm_all [Et] (shard-global edge positions).
// Batch arrays: ids [B] (selected graphs, any order, repeats allowed), out_nptr / out_eptr [B+1]
// (prefix sums of the selected graphs' node / edge counts) and the largest node / edge count of a selected graph,
// all computed by the caller.
#include "common.h"

namespace {

struct NodeArgs {
    const float* x_all; int64_t ldx; int F;
    const int64_t* node_ptr; const int64_t* edge_ptr;
    const int32_t* rowptr_dst_all; const int32_t* rowptr_src_all;
    const int64_t* ids; int B; const int64_t* out_nptr; const int64_t* out_eptr;
    int64_t n_out;
    float* x_out; int64_t ldo; int64_t* batch_out; int32_t* rowptr_dst_out; int32_t* rowptr_src_out;
    int ct;                    // threads per row (power of two)
    int vec;                   // rows of x_all and x_out are 16-byte aligned
};

// blockIdx.y = position b of the graph in the batch (its offsets are workgroup-uniform: scalar loads, no
// search per element); blockIdx.x walks its rows.  ct threads copy one node row (16-byte chunks; columns
// [F, ldo) of the output are zeroed); the first of them also writes the node's graph number and CSR row
// pointers.
__global__ __launch_bounds__(256) void k_collate_nodes(const NodeArgs a) {
    const int c = threadIdx.x & (a.ct - 1), slot = threadIdx.x / a.ct, rs = 256 / a.ct;
    const int chunks_out = (int)((a.ldo + 3) >> 2);
    const int b = blockIdx.y;
    const int64_t g = a.ids[b];
    const int64_t p0 = a.out_nptr[b], n_g = a.out_nptr[b + 1] - p0, s0 = a.node_ptr[g];
    const int64_t shift = a.out_eptr[b] - a.edge_ptr[g];
    if (b == a.B - 1 && blockIdx.x == 0 && threadIdx.x == 0) {      // closing row pointers
        const int32_t e_out = (int32_t)a.out_eptr[a.B];
        if (a.rowptr_dst_out) a.rowptr_dst_out[a.n_out] = e_out;
        if (a.rowptr_src_out) a.rowptr_src_out[a.n_out] = e_out;
    }
    for (int64_t r = (int64_t)blockIdx.x * rs + slot; r < n_g; r += (int64_t)gridDim.x * rs) {
        const int64_t p = p0 + r, src = s0 + r;
        if (c == 0) {
            if (a.batch_out) a.batch_out[p] = b;
            if (a.rowptr_dst_out) a.rowptr_dst_out[p] = (int32_t)(a.rowptr_dst_all[src] + shift);
            if (a.rowptr_src_out) a.rowptr_src_out[p] = (int32_t)(a.rowptr_src_all[src] + shift);
        }
        const float* xs = a.x_all + src * a.ldx;
        float* xo = a.x_out + p * a.ldo;
        if (a.vec) {
            for (int ch = c; ch < chunks_out; ch += a.ct) {
                const int c0 = 4 * ch;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 < a.F) {
                    v = *reinterpret_cast<const float4*>(xs + c0);
                    if (c0 + 1 >= a.F) v.y = 0.f;
                    if (c0 + 2 >= a.F) v.z = 0.f;
                    if (c0 + 3 >= a.F) v.w = 0.f;
                }
                *reinterpret_cast<float4*>(xo + c0) = v;
            }
        } else {
            for (int col = c; col < a.ldo; col += a.ct) xo[col] = col < a.F ? xs[col] : 0.f;
        }
    }
}

struct EdgeArgs {
    const int64_t* ei_all; int64_t e_total;        // [2][e_total]
    const int32_t* col_dst_all; const int32_t* perm_dst_all;
    const int32_t* col_src_all; const int32_t* perm_src_all;
    const float* ea_all; int ea_dim;
    const int64_t* node_ptr; const int64_t* edge_ptr;
    const int64_t* ids; int B; const int64_t* out_nptr; const int64_t* out_eptr;
    int64_t e_out;
    int64_t* ei_out;                                // [2][e_out]
    int32_t* col_dst_out; int32_t* perm_dst_out; int32_t* col_src_out; int32_t* perm_src_out;
    float* ea_out;
};

// blockIdx.y = position of the graph in the batch; one thread per edge position of that graph: the edge list
// entry, the two CSR entries that sit at the same position of the graph's segment, and the edge attributes
__global__ __launch_bounds__(256) void k_collate_edges(const EdgeArgs a) {
    const int b = blockIdx.y;
    const int64_t g = a.ids[b];
    const int64_t q0 = a.out_eptr[b], e_g = a.out_eptr[b + 1] - q0, s0 = a.edge_ptr[g];
    const int64_t nshift = a.out_nptr[b] - a.node_ptr[g], eshift = q0 - s0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < e_g; r += (int64_t)gridDim.x * 256) {
        const int64_t q = q0 + r, src = s0 + r;
        if (a.ei_out) {
            a.ei_out[q] = a.ei_all[src] + nshift;
            a.ei_out[a.e_out + q] = a.ei_all[a.e_total + src] + nshift;
        }
        if (a.col_dst_out) {
            a.col_dst_out[q] = (int32_t)(a.col_dst_all[src] + nshift);
            a.perm_dst_out[q] = (int32_t)(a.perm_dst_all[src] + eshift);
        }
        if (a.col_src_out) {
            a.col_src_out[q] = (int32_t)(a.col_src_all[src] + nshift);
            a.perm_src_out[q] = (int32_t)(a.perm_src_all[src] + eshift);
        }
        if (a.ea_out)
            for (int d = 0; d < a.ea_dim; ++d) a.ea_out[q * a.ea_dim + d] = a.ea_all[src * a.ea_dim + d];
    }
}

int grid_cap(int64_t blocks) { return (int)(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks)); }

}  // namespace

extern "C" int hgv_collate_nodes(const float* x_all, int64_t ldx, int F, const int64_t* node_ptr, const int64_t* edge_ptr,
                                 const int32_t* rowptr_dst_all, const int32_t* rowptr_src_all, const int64_t* ids, int B,
                                 const int64_t* out_nptr, const int64_t* out_eptr, int64_t n_out, int64_t max_nodes,
                                 float* x_out, int64_t ldo, int64_t* batch_out, int32_t* rowptr_dst_out,
                                 int32_t* rowptr_src_out, hipStream_t stream) {
    HGV_REQUIRE(B > 0 && B <= 65535 && F > 0 && n_out >= 0 && max_nodes >= 0, HGV_EINVAL, "collate_nodes: bad sizes (1..65535 graphs per batch)");
    HGV_REQUIRE(x_all && node_ptr && edge_ptr && ids && out_nptr && out_eptr && x_out && ldx >= F && ldo >= F, HGV_EINVAL,
                "collate_nodes: null buffer or leading dimension below the width");
    HGV_REQUIRE((rowptr_dst_out == nullptr || rowptr_dst_all != nullptr) && (rowptr_src_out == nullptr || rowptr_src_all != nullptr),
                HGV_EINVAL, "collate_nodes: row pointers requested without the shard's");
    NodeArgs a{x_all, ldx, F, node_ptr, edge_ptr, rowptr_dst_all, rowptr_src_all, ids, B, out_nptr, out_eptr, n_out,
               x_out, ldo, batch_out, rowptr_dst_out, rowptr_src_out, 1, 0};
    a.vec = hgv_aligned(x_all, 16) && hgv_aligned(x_out, 16) && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= (F + 3) / 4 * 4;
    const int64_t units = a.vec ? (ldo + 3) / 4 : ldo;
    while (a.ct < units && a.ct < 64) a.ct <<= 1;
    hipLaunchKernelGGL(k_collate_nodes, dim3(grid_cap(hgv_cdiv(max_nodes, 256 / a.ct)), B), dim3(256), 0, stream, a);
    HGV_CHECK_LAUNCH("collate_nodes");
    return HGV_OK;
}

extern "C" int hgv_collate_edges(const int64_t* ei_all, int64_t e_total, const int32_t* col_dst_all, const int32_t* perm_dst_all,
                                 const int32_t* col_src_all, const int32_t* perm_src_all, const float* ea_all, int ea_dim,
                                 const int64_t* node_ptr, const int64_t* edge_ptr, const int64_t* ids, int B,
                                 const int64_t* out_nptr, const int64_t* out_eptr, int64_t e_out, int64_t max_edges, int64_t* ei_out,
                                 int32_t* col_dst_out, int32_t* perm_dst_out, int32_t* col_src_out, int32_t* perm_src_out,
                                 float* ea_out, hipStream_t stream) {
    HGV_REQUIRE(B > 0 && B <= 65535 && e_out >= 0 && e_total >= 0 && max_edges >= 0, HGV_EINVAL, "collate_edges: bad sizes (1..65535 graphs per batch)");
    if (e_out == 0) return HGV_OK;
    HGV_REQUIRE(node_ptr && edge_ptr && ids && out_nptr && out_eptr, HGV_EINVAL, "collate_edges: null index buffer");
    HGV_REQUIRE(e_out < ((int64_t)1 << 31), HGV_EUNSUPPORTED, "collate_edges: more than 2^31 edges in a batch");
    HGV_REQUIRE(!ei_out || ei_all, HGV_EINVAL, "collate_edges: edge list requested without the shard's");
    HGV_REQUIRE((!col_dst_out || (col_dst_all && perm_dst_all && perm_dst_out)) && (!col_src_out || (col_src_all && perm_src_all && perm_src_out)),
                HGV_EINVAL, "collate_edges: CSR requested without the shard's, or col without perm");
    HGV_REQUIRE(!ea_out || (ea_all && ea_dim > 0), HGV_EINVAL, "collate_edges: edge attributes requested without the shard's");
    EdgeArgs a{ei_all, e_total, col_dst_all, perm_dst_all, col_src_all, perm_src_all, ea_all, ea_dim, node_ptr, edge_ptr, ids, B,
               out_nptr, out_eptr, e_out, ei_out, col_dst_out, perm_dst_out, col_src_out, perm_src_out, ea_out};
    hipLaunchKernelGGL(k_collate_edges, dim3(grid_cap(hgv_cdiv(max_edges, 256)), B), dim3(256), 0, stream, a);
    HGV_CHECK_LAUNCH("collate_edges");
    return HGV_OK;
}
