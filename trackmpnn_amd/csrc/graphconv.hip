// Adjacency pair -> index-form graph with device-side sizes, in ONE launch (batch-1 path, include/tmpnn.h).
//
// What the reference does per call with dense N x N temporaries (models/track_mpnn.py:55-56, models/layers.py:85-88:
// to_dense -> diag -> to_sparse) and what round 1 did with ~100 small torch index ops and two host round trips is
// here one single-workgroup kernel over the COO entries: the graphs of the reference's real call pattern have a few
// hundred to a few thousand rows (N_final ~ 260 .. 1 700), so every intermediate fits the CU's 160 KB LDS
// and the whole conversion is a handful of LDS passes separated by workgroup barriers.  The sizes E / Dn and the
// validation status are left in device memory (tmpnn_dgraph.meta): no host synchronisation.
//
//   1. scatter the entries:   diag[r] += v (r == c)   |   src[r] = c (v > 0), dst[r] = c (v < 0), counts per row
//   2. type mask + validation of the factor-graph invariants (utils/graph.py:151-163, 294-308)
//   3. workgroup scan of the type mask -> pos[], edge_row[], det_row[], per-edge src/dst (rows and det indices)
//   4. det -> incident-edge CSR: degree count (LDS atomics), scan, unordered fill, then a rank placement per det so
//      that every det's incidences are in ascending edge-row order (the order the reductions and the loss rules use)
//   5. optional cross-check of edge_adj (= node_adj^T off the diagonal, complementary diagonal)
#include "common.h"
#include "graphconv_dev.h"

namespace tmpnn {

// (the conversion itself: graphconv_dev.h)
template <bool FROM_ROWS, bool BIG = false>
__global__ __launch_bounds__(GC_THREADS) void k_graph_from_coo(int N, const int64_t* __restrict__ nidx,
                                                               const float* __restrict__ nval, long nnz_n,
                                                               const int64_t* __restrict__ eidx,
                                                               const float* __restrict__ eval_, long nnz_e,
                                                               const uint8_t* __restrict__ r_is_edge,
                                                               const int32_t* __restrict__ r_src,
                                                               const int32_t* __restrict__ r_dst,
                                                               tmpnn_dgraph g, int* __restrict__ scratch) {
    d_graph_from_coo<FROM_ROWS, BIG>(N, nidx, nval, nnz_n, eidx, eval_, nnz_e, r_is_edge, r_src, r_dst, g, scratch);
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

static size_t align4(size_t n) { return (n + 3) & ~(size_t)3; }

size_t tmpnn_dgraph_ints(int cap) {
    if (cap < 0) return 0;
    const size_t c = align4((size_t)cap + 1);
    // meta | is_edge (bytes) | pos src dst src_pos dst_pos edge_row det_row | rowptr | inc
    return align4(TMPNN_DG_META) + align4(((size_t)cap + 3) / 4) + 7 * c + c + 2 * c;
}

int tmpnn_dgraph_bind(void* arena, int cap, int N, tmpnn_dgraph* out) {
    TM_REQUIRE(arena != nullptr && out != nullptr, "dgraph_bind: null pointer");
    TM_REQUIRE(cap >= 0 && N >= 0 && N <= cap, "dgraph_bind: N=%d cap=%d", N, cap);
    TM_REQUIRE(aligned16(arena), "dgraph_bind: the arena must be 16-byte aligned");
    int32_t* p = reinterpret_cast<int32_t*>(arena);
    const size_t c = align4((size_t)cap + 1);
    out->N = N;
    out->cap = cap;
    out->meta = p; p += align4(TMPNN_DG_META);
    out->is_edge = reinterpret_cast<uint8_t*>(p); p += align4(((size_t)cap + 3) / 4);
    out->pos = p; p += c;
    out->src = p; p += c;
    out->dst = p; p += c;
    out->src_pos = p; p += c;
    out->dst_pos = p; p += c;
    out->edge_row = p; p += c;
    out->det_row = p; p += c;
    out->rowptr = p; p += c;
    out->inc = p;
    return TMPNN_OK;
}

int tmpnn_graph_from_coo(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                         const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge, const tmpnn_dgraph* g,
                         tmpnn_stream stream) {
    TM_REQUIRE(g != nullptr && g->meta && g->is_edge && g->pos && g->src && g->dst && g->src_pos && g->dst_pos &&
                   g->edge_row && g->det_row && g->rowptr && g->inc, "graph_from_coo: unbound graph");
    TM_REQUIRE(N >= 0 && N <= TMPNN_DG_MAX_ROWS && N <= g->cap && N == g->N,
               "graph_from_coo: N=%d (limit %d, graph N=%d cap=%d)", N, TMPNN_DG_MAX_ROWS, g->N, g->cap);
    TM_REQUIRE(nnz_node >= 0 && (nnz_node == 0 || (node_idx && node_val)), "graph_from_coo: node_adj entries");
    TM_REQUIRE(nnz_edge >= 0 && (edge_idx == nullptr || nnz_edge == 0 || edge_val), "graph_from_coo: edge_adj entries");
    const size_t shm = sizeof(int) * ((size_t)8 * N + 1);
    TM_SHM_ONCE(k_graph_from_coo<false>, 160 * 1024 - 256);
    hipLaunchKernelGGL(k_graph_from_coo<false>, dim3(1), dim3(GC_THREADS), shm, as_stream(stream), N, node_idx, node_val,
                       (long)nnz_node, edge_idx, edge_val, (long)nnz_edge, (const uint8_t*)nullptr, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, *g, (int*)nullptr);
    return check_launch("graph_from_coo");
}

size_t tmpnn_graph_from_coo_ws_ints(int N) { return N > TMPNN_DG_MAX_ROWS ? (size_t)8 * N + 1 : 0; }

int tmpnn_graph_from_coo_arena_ws(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                                  const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge, void* arena, int cap,
                                  void* ws, size_t ws_ints, tmpnn_stream stream) {
    if (N <= TMPNN_DG_MAX_ROWS)
        return tmpnn_graph_from_coo_arena(N, node_idx, node_val, nnz_node, edge_idx, edge_val, nnz_edge, arena, cap, stream);
    TM_REQUIRE(N <= TMPNN_DG_BIG_ROWS, "graph_from_coo: N=%d exceeds %d rows", N, TMPNN_DG_BIG_ROWS);
    tmpnn_dgraph g;
    const int rc = tmpnn_dgraph_bind(arena, cap, N, &g);
    if (rc) return rc;
    TM_REQUIRE(nnz_node >= 0 && (nnz_node == 0 || (node_idx && node_val)), "graph_from_coo: node_adj entries");
    TM_REQUIRE(nnz_edge >= 0 && (edge_idx == nullptr || nnz_edge == 0 || edge_val), "graph_from_coo: edge_adj entries");
    if (ws == nullptr || ws_ints < tmpnn_graph_from_coo_ws_ints(N))
        return set_error(TMPNN_EWORKSPACE, "graph_from_coo: workspace %zu < %zu ints", ws_ints, tmpnn_graph_from_coo_ws_ints(N));
    const size_t shm_big = sizeof(int) * (GC_THREADS / 64) * GC_RUN;          // 128 KiB: per-wave ranking slices
    TM_SHM_ONCE((k_graph_from_coo<false, true>), shm_big);
    hipLaunchKernelGGL((k_graph_from_coo<false, true>), dim3(1), dim3(GC_THREADS), shm_big, as_stream(stream), N, node_idx, node_val,
                       (long)nnz_node, edge_idx, edge_val, (long)nnz_edge, (const uint8_t*)nullptr, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, g, reinterpret_cast<int*>(ws));
    return check_launch("graph_from_coo (global scratch)");
}

int tmpnn_graph_from_coo_arena(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                               const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge, void* arena, int cap,
                               tmpnn_stream stream) {
    tmpnn_dgraph g;
    const int rc = tmpnn_dgraph_bind(arena, cap, N, &g);
    if (rc) return rc;
    return tmpnn_graph_from_coo(N, node_idx, node_val, nnz_node, edge_idx, edge_val, nnz_edge, &g, stream);
}

int tmpnn_graph_from_rows(int N, const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst,
                          const tmpnn_dgraph* g, tmpnn_stream stream) {
    TM_REQUIRE(g != nullptr && g->meta && g->is_edge && g->pos && g->src && g->dst && g->src_pos && g->dst_pos &&
                   g->edge_row && g->det_row && g->rowptr && g->inc, "graph_from_rows: unbound graph");
    TM_REQUIRE(N >= 0 && N <= TMPNN_DG_MAX_ROWS && N <= g->cap && N == g->N,
               "graph_from_rows: N=%d (limit %d, graph N=%d cap=%d)", N, TMPNN_DG_MAX_ROWS, g->N, g->cap);
    TM_REQUIRE(N == 0 || (is_edge && row_src && row_dst), "graph_from_rows: null row arrays");
    const size_t shm = sizeof(int) * ((size_t)8 * N + 1);
    TM_SHM_ONCE(k_graph_from_coo<true>, 160 * 1024 - 256);
    hipLaunchKernelGGL(k_graph_from_coo<true>, dim3(1), dim3(GC_THREADS), shm, as_stream(stream), N,
                       (const int64_t*)nullptr, (const float*)nullptr, 0L, (const int64_t*)nullptr, (const float*)nullptr, 0L,
                       is_edge, row_src, row_dst, *g, (int*)nullptr);
    return check_launch("graph_from_rows");
}

int tmpnn_graph_from_rows_ws(int N, const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst,
                             const tmpnn_dgraph* g, void* ws, size_t ws_ints, tmpnn_stream stream) {
    if (N <= TMPNN_DG_MAX_ROWS) return tmpnn_graph_from_rows(N, is_edge, row_src, row_dst, g, stream);
    TM_REQUIRE(g != nullptr && g->meta && g->is_edge && g->pos && g->src && g->dst && g->src_pos && g->dst_pos &&
                   g->edge_row && g->det_row && g->rowptr && g->inc, "graph_from_rows: unbound graph");
    TM_REQUIRE(N <= TMPNN_DG_BIG_ROWS && N <= g->cap && N == g->N, "graph_from_rows: N=%d (limit %d, graph N=%d cap=%d)", N,
               TMPNN_DG_BIG_ROWS, g->N, g->cap);
    TM_REQUIRE(is_edge && row_src && row_dst, "graph_from_rows: null row arrays");
    if (ws == nullptr || ws_ints < tmpnn_graph_from_coo_ws_ints(N))
        return set_error(TMPNN_EWORKSPACE, "graph_from_rows: workspace %zu < %zu ints", ws_ints, tmpnn_graph_from_coo_ws_ints(N));
    const size_t shm_big = sizeof(int) * (GC_THREADS / 64) * GC_RUN;
    TM_SHM_ONCE((k_graph_from_coo<true, true>), shm_big);
    hipLaunchKernelGGL((k_graph_from_coo<true, true>), dim3(1), dim3(GC_THREADS), shm_big, as_stream(stream), N,
                       (const int64_t*)nullptr, (const float*)nullptr, 0L, (const int64_t*)nullptr, (const float*)nullptr, 0L,
                       is_edge, row_src, row_dst, *g, reinterpret_cast<int*>(ws));
    return check_launch("graph_from_rows (global scratch)");
}

}  // extern "C"
