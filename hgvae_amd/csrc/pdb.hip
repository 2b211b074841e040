// Graph construction from parsed PDB ATOM records, on the device, straight into the packed layout of a shard
// (the disjoint union of all graphs: features [Nt, ldx], edge_index [2, Et] with union-wide node numbers,
// edge attributes [Et]).
//
// Replaces pdb_to_pyg_data of the reference (gvae/create_graphs_for_gvae.py:67-144), which builds one networkx
// graph per file on the host:
//   node features  one-hot(atom name) | one-hot(residue name) | 3 force columns            (:113-127)
//   edges          one directed edge from every ATOM record to the next one of its file       (:99-111, :131-136)
//   edge feature   Euclidean distance of the two atoms: np.linalg.norm(a - b) in float64, stored as float32 (:24-25, :109)
// Integer work (one-hot positions, edge endpoints) is exact; the distance is formed in float64 as
// sqrt((dx*dx + dy*dy) + dz*dz) without fused multiply-adds and rounded to float32 once.
#include "common.h"

namespace {

struct PdbArgs {
    const int32_t* atom_idx; const int32_t* res_idx; const float* forces;   // [Nt], [Nt], [Nt, 3] (nullable: zeros)
    const double* coords;                                                   // [Nt, 3]
    const int64_t* node_ptr; const int64_t* edge_ptr;                       // [G + 1]
    int64_t Nt; int n_atom, n_res, F, ldx;
    float* x; int64_t* ei; int64_t Et; float* ea; int32_t* status;
};

// one thread per (node, 4-column chunk): zero the chunk, set the one-hot ones, place the force columns
__global__ __launch_bounds__(256) void k_pdb_nodes(const PdbArgs a) {
    const int chunks = a.ldx >> 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.Nt * chunks) return;
    const int64_t i = t / chunks;
    const int c0 = (int)(t - i * chunks) * 4;
    int ai = a.atom_idx[i], ri = a.res_idx[i];
    if (ai < 0 || ai >= a.n_atom || ri < 0 || ri >= a.n_res) {
        if (a.status) a.status[0] = 1;          // same value from every faulting thread: a plain store is enough
        ai = min(max(ai, 0), a.n_atom - 1);
        ri = min(max(ri, 0), a.n_res - 1);
    }
    const int rcol = a.n_atom + ri, fcol = a.n_atom + a.n_res;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + j;
        float val = 0.f;
        if (c == ai || c == rcol) val = 1.f;
        else if (c >= fcol && c < a.F && a.forces) val = a.forces[i * 3 + (c - fcol)];
        v[j] = val;
    }
    *reinterpret_cast<float4*>(a.x + i * a.ldx + c0) = make_float4(v[0], v[1], v[2], v[3]);
}

// blockIdx.y = graph; edge e of graph g joins its nodes e and e + 1
__global__ __launch_bounds__(256) void k_pdb_edges(const PdbArgs a) {
    const int g = blockIdx.y;
    const int64_t n0 = a.node_ptr[g], e0 = a.edge_ptr[g];
    const int64_t cnt = a.edge_ptr[g + 1] - e0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = n0 + e, d = s + 1;
        a.ei[e0 + e] = s;
        a.ei[a.Et + e0 + e] = d;
        if (a.ea) {
            const double dx = a.coords[3 * s] - a.coords[3 * d], dy = a.coords[3 * s + 1] - a.coords[3 * d + 1],
                         dz = a.coords[3 * s + 2] - a.coords[3 * d + 2];
            a.ea[e0 + e] = (float)sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)));
        }
    }
}

}  // namespace

// atom_idx / res_idx [Nt] int32 (category numbers), forces [Nt, 3] (nullable), coords [Nt, 3] float64 (nullable when
// ea is null), node_ptr / edge_ptr [G + 1] int64 device arrays with edge_ptr[g+1] - edge_ptr[g] = max(n_g - 1, 0);
// x [Nt, ldx] (ldx a multiple of 4 >= n_atom + n_res + 3, 16-byte aligned; pad columns are zeroed), ei [2][Et],
// ea [Et] (nullable).  status (nullable, caller-zeroed): set to 1 if a category number was out of range.
extern "C" int hgv_pdb_build(const int32_t* atom_idx, const int32_t* res_idx, const float* forces, const double* coords,
                             const int64_t* node_ptr, const int64_t* edge_ptr, int G, int64_t Nt, int64_t Et,
                             int64_t max_edges, int n_atom, int n_res, float* x, int64_t ldx, int64_t* ei, float* ea,
                             int32_t* status, hipStream_t stream) {
    HGV_REQUIRE(G >= 0 && Nt >= 0 && Et >= 0 && n_atom > 0 && n_res > 0, HGV_EINVAL, "pdb_build: bad sizes");
    if (Nt == 0) return HGV_OK;
    const int F = n_atom + n_res + 3;
    HGV_REQUIRE(atom_idx && res_idx && node_ptr && edge_ptr && x, HGV_EINVAL, "pdb_build: null buffer");
    HGV_REQUIRE(ldx % 4 == 0 && ldx >= F && hgv_aligned(x, 16), HGV_EINVAL, "pdb_build: x must be 16-byte aligned with ldx %% 4 == 0, ldx >= %d", F);
    HGV_REQUIRE(Et == 0 || (ei && (!ea || coords)), HGV_EINVAL, "pdb_build: null edge buffer");
    HGV_REQUIRE(G <= 65535, HGV_EUNSUPPORTED, "pdb_build: at most 65535 graphs per call");
    PdbArgs a{atom_idx, res_idx, forces, coords, node_ptr, edge_ptr, Nt, n_atom, n_res, F, (int)ldx, x, ei, Et, ea, status};
    hipLaunchKernelGGL(k_pdb_nodes, dim3((unsigned)hgv_cdiv(Nt * (ldx / 4), 256)), dim3(256), 0, stream, a);
    if (Et > 0 && G > 0) {
        int64_t bx = hgv_cdiv(max_edges > 0 ? max_edges : 1, 256);
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(k_pdb_edges, dim3((unsigned)bx, (unsigned)G), dim3(256), 0, stream, a);
    }
    HGV_CHECK_LAUNCH("pdb_build");
    return HGV_OK;
}
