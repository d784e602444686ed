/*
 * tmpnn.h -- C ABI of libtmpnn.so: the MI355X (gfx950) implementation of the TrackMPNN
 * message-passing hot path.
 *
 * Every entry point replaces one stage of the reference's per-timestep forward/backward
 * (reference = arangesh/TrackMPNN; file:line cited at each declaration).  The reference has no
 * FFI of its own (it is pure Python on torch ops); the stages below are exactly the ATen calls
 * `TrackMPNN.forward` (models/track_mpnn.py:54-75) and `FactorGraphGRU.forward`
 * (models/layers.py:84-116) dispatch, regrouped by what they compute.
 *
 * Conventions
 *   - all feature tensors are fp32, row-major, with an explicit leading dimension (`ld*`, in
 *     floats) so one [N, G*H] state tensor can be addressed per feature group without repacking;
 *   - all index arrays are int32 and live in device memory;
 *   - the caller owns every buffer (including workspaces); the library never allocates, frees,
 *     or keeps a device pointer past the call;
 *   - work is only ENQUEUED on `stream`; nothing synchronises the device;
 *   - return value: TMPNN_OK (0) or a negative TMPNN_E* code; `tmpnn_last_error()` gives the
 *     thread-local message of the last failure.  Nothing is ever thrown across the boundary.
 *   - supported hidden widths: H in {32, 64, 128, 256}; additionally the multiples of 128 up to 1024 in the entry points of
 *     the paths without attention: tmpnn_gru_{fwd,bwd_data,bwd_weights} (H-generic f32 kernels), tmpnn_wide_*,
 *     tmpnn_input_bn_*, tmpnn_heads_* (<= 1024 columns per call), tmpnn_gather_diff_* / tmpnn_gather_concat_bwd / tmpnn_segsum_*
 *     (served internally in 256-column slices).  tmpnn_att_*, tmpnn_gather_concat_fwd and the fused / tiled H <= 64 forms keep
 *     the four widths.
 *   - no entry point measures, tunes or keeps data between calls; the only process state is the thread-local
 *     error string and idempotent per-kernel function attributes.
 */
#ifndef TMPNN_H
#define TMPNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMPNN_ABI_VERSION 4 /* 2: struct tmpnn_graph gained seg_plan (round 4); 3: win_plan (round 5); 4 (round 6, frozen): the
                               exported set is what a default run can reach (80 entry points: 15 superseded or internal ones
                               left it), tmpnn_input_tf_* take x_rows, + tmpnn_segsum_fwd_live, tmpnn_bce_logits_* */

#define TMPNN_OK 0
#define TMPNN_EINVAL (-1)   /* bad shape / null pointer / unsupported width */
#define TMPNN_ELAUNCH (-2)  /* hipGetLastError() after a launch */
#define TMPNN_EWORKSPACE (-3) /* workspace too small */

typedef void* tmpnn_stream; /* hipStream_t */
typedef void* tmpnn_event;  /* hipEvent_t, created and owned by the caller (the library only records / waits on it) */

/*
 * Index form of the reference's adjacency pair (utils/graph.py:151-163, 294-308): every edge
 * row e of node_adj holds +1 at its earlier det (`src`) and -1 at its later det (`dst`);
 * edge_adj is its transpose, i.e. the det -> incident-edge lists, stored here as a CSR whose
 * entries are edge ROW indices with the sign of edge_adj[d, e] in bit 31 (set = -1 = d is the
 * later det of e).  Built once per call by the host from the adjacency the reference passes.
 */
/* Optional plan of a DENSE graph for the edge -> det segment sum (csrc/agg.hip, k_segsum_tiles): the edge set cut into
 * 8 src x 16 dst tiles over det indices, each summed by the workgroup that streams its rows, so that an edge row is read
 * from HBM once instead of once per endpoint.  Built by the host once per graph (trackmpnn_amd.graph.dense_seg_plan); the
 * caller owns every array including the partial-row buffer `ws` (one segment sum at a time per plan). */
typedef struct tmpnn_seg_plan {
    int32_t T;               /* tiles */
    int32_t I;               /* work items: runs of consecutive tiles that share their 16 dsts */
    int32_t nsplit;          /* = N of the graph: entries of inc2 at or above it address partial rows */
    const int32_t* t_row;    /* [T][8][16] edge row of (src i, dst j) of a tile, -1 where the graph has no such edge */
    const int32_t* items;    /* [I][2] first tile, tile count */
    const int32_t* rowptr2;  /* [Dn+1] CSR of the second pass */
    const int32_t* inc2;     /* per det: nsplit + partial row (tile t, src i: 8t + i; item k, dst j: 8T + 16k + j), or a
                                plain edge row (< nsplit); sign bit as in tmpnn_graph.inc */
    float* ws;               /* [8 T + 16 I][256] partial rows */
    size_t ws_floats;
} tmpnn_seg_plan;

/* Optional plan of a BATCH OF SMALL WINDOWS (block-diagonal: no edge crosses windows) for the edge -> det segment sum at H = 64
 * (csrc/agg.hip, k_segsum_win): a workgroup owns a window at a time and walks its edge rows through the LDS in chunks of 160, once
 * -- each edge row is read from memory ONCE instead of once per endpoint; results equal the CSR kernel's bit for bit.  Incidence i
 * of a det's CSR run belongs to the stream (det, i % 4); stream s of a window's (window-local) dets is added to by lane group
 * s % 128 of the workgroup.  Built by the host once per graph (trackmpnn_amd.graph.build_win_plan) from the window label of every
 * det; the caller owns the arrays. */
typedef struct tmpnn_win_plan {
    int32_t W;                /* windows */
    int32_t nbig;             /* dets of the windows beyond the kernel's capacity (160 dets, 24 chunks, 15 steps a chunk): CSR kernel */
    const int32_t* wrec;      /* [W][8] per window: first entry / count in erow, first visiting position (entry of det / drow) /
                                       count of its dets (capacity + 1: left to the CSR kernel), first step in recs, and the steps of
                                       its chunks, 4 bits each, in three words; 32-byte aligned */
    const int32_t* erow;      /* edge rows, window by window, ascending within a window; a window's list starts at a multiple of 4 */
    const uint16_t* recs;     /* [steps][128] per chunk and step, for each lane group: place of the edge row in the chunk | 0x100
                                       where the det is the edge's later endpoint (the sign bit of tmpnn_graph.inc) | (stream / 128)
                                       << 9 | 0x8000 for "nothing to do"; a lane group's records of one stream are in run order;
                                       256-byte aligned */
    const int32_t* det;       /* [Dn]  det index per visiting position (window-major) */
    const int32_t* drow;      /* [Dn]  its graph row */
    const int32_t* big_order; /* [nbig] det indices of the windows left to the CSR kernel */
} tmpnn_win_plan;

typedef struct tmpnn_graph {
    int32_t N;               /* rows of the state tensor (dets + edges) */
    int32_t E;               /* edge rows */
    int32_t Dn;              /* det rows */
    const int32_t* src;      /* [E]  row of the +1 det of edge e */
    const int32_t* dst;      /* [E]  row of the -1 det of edge e */
    const int32_t* edge_row; /* [E]  row of edge e (ascending) */
    const int32_t* det_row;  /* [Dn] row of det d (ascending) */
    const int32_t* rowptr;   /* [Dn+1] CSR offsets into inc */
    const int32_t* inc;      /* [2E] (edge row) | (sign bit: 0x80000000 when d == dst) */
    const int32_t* det_order; /* [Dn] or NULL: order in which the det -> edge reductions visit the dets (a permutation of
                                0..Dn-1).  Results do not depend on it; a host that batches independent windows lists
                                each window's dets together, so that the two reads of an edge row (one from either
                                endpoint) are issued from the same CU close in time */
    const tmpnn_seg_plan* seg_plan; /* or NULL: dense graphs, H = 256 column blocks -- tmpnn_segsum_fwd and the wide cells'
                                       backward then read every edge row once (see tmpnn_seg_plan) */
    const tmpnn_win_plan* win_plan; /* or NULL: batches of small windows, H = 64 -- the segment sums (tmpnn_segsum_fwd,
                                       tmpnn_gather_diff_bwd) then read every edge row once (see tmpnn_win_plan) */
} tmpnn_graph;

/*
 * Edge tiles: the edge rows of a graph cut into tiles of `rows_per_tile` rows chosen so that a tile touches FEW distinct
 * dets.  The rolling graph's frame blocks are dense [A srcs x D_t dsts] row sets in src-major order
 * (utils/graph.py:141-156, 285-301), so a tile of (8 srcs x 16 dsts) needs 24 rows of a per-det table where 128
 * consecutive rows of a wide block need up to 129, and every gather of h[src] / h[dst] (models/layers.py:90-95) or of
 * their projections becomes a read from an LDS copy of the tile's det list.  Built once per graph
 * (trackmpnn_amd.graph.build_edge_tiles); any tiling that covers every edge row exactly once is valid -- ragged
 * (post-decode) graphs simply get longer det lists.
 */
typedef struct tmpnn_edge_tiles {
    int32_t T;              /* tiles */
    int32_t rows_per_tile;  /* 128 (wide cells) or 32 (H <= 64 cells) */
    const int32_t* t_row;   /* [T * rows_per_tile] graph row of each slot; -1 = padding.  PRECONDITION (device data, not validated by
                               the entry points; trackmpnn_amd.graph.build_edge_tiles guarantees it and tests/test_graph.py checks
                               it): padding occurs in the LAST tile only, BEHIND its valid slots, so slot 0 of every tile is a real
                               row -- the wide forward (k_wide_gru_fwd_pp) has no row guard: a padding slot recomputes and rewrites
                               what its tile's slot 0 writes, byte for byte */
    const int32_t* t_loc;   /* [T * rows_per_tile] (position of the slot's src det in the tile's det list) |
                               (position of its dst det) << 16 */
    const int32_t* t_dptr;  /* [T + 1] offsets into t_dets */
    const int32_t* t_dets;  /* [t_dptr[T]] det INDEX (0..Dn-1, = row of a per-det table), ascending within a tile */
} tmpnn_edge_tiles;

int tmpnn_abi_version(void);
const char* tmpnn_last_error(void);

/* ---- row E: node -> edge message (models/layers.py:90-95) -------------------------------
 * diff  : out[edge_row[e], 0:H]  (=|+=)  in[src[e], 0:H] - in[dst[e], 0:H]
 * concat: out[edge_row[e], 0:2H] (=|+=) [in[src[e], 0:H] | in[dst[e], 0:H]]
 * `accumulate` != 0 adds into out.  The backward of either is tmpnn_segsum_fwd on the gradient
 * (diff: signs +1/-1, concat: column offsets 0/H), exposed as tmpnn_gather_*_bwd. */
int tmpnn_gather_diff_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out,
                          int H, int accumulate, tmpnn_stream stream);
int tmpnn_gather_concat_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out,
                            int H, int accumulate, tmpnn_stream stream);
/* d_in[det_row[d], 0:H] (=|+=) sum_{e: src=d} d_out[row e, 0:H] - sum_{e: dst=d} d_out[row e, 0:H] */
int tmpnn_gather_diff_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din,
                          int H, int accumulate, tmpnn_stream stream);
/* d_in[det_row[d], 0:H] (=|+=) sum_{e: src=d} d_out[row e, 0:H] + sum_{e: dst=d} d_out[row e, H:2H] */
int tmpnn_gather_concat_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din,
                            int H, int accumulate, tmpnn_stream stream);

/* ---- row F: edge -> node signed aggregation, no attention (models/layers.py:103) --------
 * out[det_row[d], 0:H] (=|+=) sum_{e: src=d} in[row e] - sum_{e: dst=d} in[row e]
 * backward: d_in[edge_row[e]] (=|+=) d_out[src[e]] - d_out[dst[e]]  (= tmpnn_gather_diff_fwd). */
int tmpnn_segsum_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out,
                     int H, int accumulate, int compact_out /* out row = d instead of det_row[d] */,
                     tmpnn_stream stream);
/* The same sum where the caller KNOWS that rows >= row_limit of `in` are all-zero -- a forward call's new edge rows, which
 * enter the state as 0 (models/track_mpnn.py:61, utils/graph.py:148,291): those rows are not read (their +-0 changes no bit
 * of a sum).  row_limit >= N reads every row.  Ignored by the dense / window plans (they take the plain form). */
int tmpnn_segsum_fwd_live(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out,
                          int H, int compact_out, int row_limit, tmpnn_stream stream);
int tmpnn_segsum_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din,
                     int H, int accumulate, tmpnn_stream stream);

/* ---- row G: attention-weighted aggregation (models/layers.py:26-43, 105-112) -------------
 * K heads (1..8 per call, all served by the same passes; a model with more runs groups of heads and combines their means: the output of a call is the MEAN over ITS heads).  W_cat [H][K*H]: head k's W_att ([H][H], in x out, as the reference
 * stores it) in columns k*H .. (k+1)*H; a [K][H].
 *   ha    = h[det rows] @ W_cat                       (ha [Dn][K*H]: the heads of a det side by side)
 *   s_e   = LeakyReLU_0.2(|ha_k[src]-ha_k[dst]| . a_k)  (score [2E][K] per CSR POSITION: an edge's score is stored at both
 *                                                      of its positions, so a det reads its run's scores contiguously)
 *   alpha = softmax over each det's incident edges    (alpha [K][2E], CSR order, AFTER dropout)
 *   out[d, 0:H] = 1/K sum_k sum_p sign_p * alpha_kp * h[inc row p]     (COMPACT rows: d, not det_row[d])
 * Also saved for the backward: stats [Dn][K][2] = the softmax's (max, sum exp) per det and head, and esk [K][Dn][H] = the
 * per-head aggregate 1/K sum_p sign_p alpha_kp h[row p] (out = sum_k esk[k]).
 * keep: NULL (eval / no dropout) or uint8 [2E] in CSR order, bit k set = head k keeps the position (kept entries scaled
 *   1/(1-p_drop)): one byte per position serves every head.
 * erec [E][8]: everything an edge-owned pass needs to know about edge e, as one 32-byte record: the DET INDICES of its
 *   src / dst endpoint, its CSR positions in the src det's and in the dst det's run (the inverse of inc) | the graph rows
 *   of src, dst and of the edge itself, 0.
 * One read of h[inc row p] per CSR position serves every head; launches: 1 GEMM + 2 kernels. */
int tmpnn_att_fwd(const tmpnn_graph* g, const int32_t* erec, const float* h, int ld_h, int H, int K,
                  const float* W_cat, const float* a, const uint8_t* keep, float p_drop,
                  float* ha, float* score, float* stats, float* esk, float* alpha, float* out, int ld_out,
                  tmpnn_stream stream);
/* Backward of tmpnn_att_fwd.  d_out [N rows, ld_dout] is read at det rows (row-indexed).
 * Accumulates: d_h (+=, edge rows through the values, det rows through ha), dW_att [K][H][H] (+=, one [H][H] block per
 * head in the reference's layout), da [K][H] (+=).  ws: tmpnn_att_bwd_ws(E, Dn, H, K) floats, 16-byte aligned.
 * inc_other [2E]: det index of the OTHER endpoint of every CSR position (the src det's positions hold dst_pos[e] and vice
 * versa), bit 31 set on dst-side positions.  One edge-owned pass (reads h[row e] once, updates d_h[row e]) + one det-owned
 * pass over the projected det table + two Dn-row GEMMs. */
size_t tmpnn_att_bwd_ws(int E, int Dn, int H, int K);
/* erec [E][8] and inc_other [2E] of the two calls above from the graph's own arrays, in one launch.  pos [N]: row -> index
 * within its type (edge index for edge rows); src_pos / dst_pos [E]: det indices of an edge's endpoints. */
int tmpnn_att_index(const tmpnn_graph* g, const int32_t* pos, const int32_t* src_pos, const int32_t* dst_pos,
                    int32_t* erec, int32_t* inc_other, tmpnn_stream stream);
int tmpnn_att_bwd(const tmpnn_graph* g, const int32_t* erec, const int32_t* inc_other,
                  const float* h, int ld_h, int H, int K,
                  const float* W_cat, const float* a, const uint8_t* keep, float p_drop,
                  const float* ha, const float* score, const float* stats, const float* esk,
                  const float* d_out, int ld_dout, float* ws, size_t ws_floats,
                  float* d_h, int ld_dh, float* dW_att, float* da, tmpnn_stream stream);
/* The same with one output pointer per head (host arrays of K device pointers: dW_heads[k] [H][H] (+=), da_heads[k] [H]
 * (+=)) -- the heads' parameters are separate tensors (models/layers.py:7-24, 67: a ModuleList of GraphAttentionLayer, one per head), so their
 * .grad buffers can be accumulated in place without a stacked temporary. */
int tmpnn_att_bwd_heads(const tmpnn_graph* g, const int32_t* erec, const int32_t* inc_other,
                        const float* h, int ld_h, int H, int K,
                        const float* W_cat, const float* a, const uint8_t* keep, float p_drop,
                        const float* ha, const float* score, const float* stats, const float* esk,
                        const float* d_out, int ld_dout, float* ws, size_t ws_floats,
                        float* d_h, int ld_dh, float* const* dW_heads, float* const* da_heads, tmpnn_stream stream);

/* ---- rows H', I: GRU cells with row indirection + type-masked merge ----------------------
 * (torch.nn.GRUCell as used at models/layers.py:97,114; merge layers.py:116).
 * For every r < R with row = rows[r]:
 *     x      = xmode 0: msg[msg_compact ? r : row, 0:IN]
 *              xmode 1: h[src[r]] - h[dst[r]]          (fused row E, diff;   IN = H)
 *              xmode 2: [h[src[r]] | h[dst[r]]]        (fused row E, concat; IN = 2H)
 *              xmode 3: as xmode 1, but the x-part of the gates is read PRE-PROJECTED: msg = P [Dn][ld_msg >= 3H]
 *                       with P = h[det rows] @ W_ih^T (tmpnn_rows_linear), src/dst = DET INDICES of the two
 *                       endpoints; gi[r] = P[src[r]] - P[dst[r]] (linearity of row E).  H <= 64 only; wih_t unused.
 *     h_out[row] = GRUCell(x, h[row])
 * so running it once with rows = edge_row and once with rows = det_row performs the merge in
 * place.  Weights are passed TRANSPOSED (wih_t [IN][3H], whh_t [H][3H], see tmpnn_transpose);
 * gate order r,z,n.  gates: NULL or 4 planes [4][gate_plane] of row-indexed [N][H] floats
 * (r, z, n, W_hn h + b_hn) saved for the backward.
 * Fused output head (row J): if logit_part != NULL, plane p of logit_part (p < tmpnn_gru_fwd_head_parts(),
 * plane stride part_stride >= N floats) receives at `row` the partial dot product of h_out[row] with
 * w_head[0:H] over the p-th 32-column slice; tmpnn_heads_finish sums the planes of all groups, adds the bias
 * and applies the sigmoid.  tmpnn_gru_fwd_head_parts() == 0 means the fused head is unavailable for that
 * shape (use tmpnn_heads_fwd). */
int tmpnn_gru_fwd_head_parts(int H, int IN, int xmode);
int tmpnn_gru_fwd(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                  const float* msg, int ld_msg, int msg_compact, int IN, const float* h, int ld_h, int H,
                  const float* wih_t, const float* whh_t, const float* b_ih, const float* b_hh,
                  float* h_out, int ld_out, float* gates, size_t gate_plane,
                  const float* w_head, float* logit_part, size_t part_stride, tmpnn_stream stream);
/* Upstream gradient of both backward entry points: dh[row] = d_hout[row] (NULL = 0) + dy[row] * w_head
 * (dy NULL = no head term).  dy [N] is the gradient of the logits (tmpnn_heads_bwd's dy_out) and
 * w_head [H] the slice of the output head (w_node for det rows, w_edge for edge rows) of this group.
 *
 * Data gradient: d_msg[row, 0:IN] = d_gi @ W_ih ; d_h[row] = dh[row]*z + d_gh @ W_hh (both written,
 * not accumulated).  W_ih [3H][IN], W_hh [3H][H] in the reference layout.  If add_msg != NULL the
 * adjoint of row F is fused into the store: d_h[row] += add_msg[add_src[r]] - add_msg[add_dst[r]]
 * (rows = edge_row, add_src/dst = src/dst, add_msg = the d_msg buffer whose DET rows the node cell's
 * call has already written). */
int tmpnn_gru_bwd_data(const int32_t* rows, int R, int IN, const float* h, int ld_h, int H,
                       const float* w_ih, const float* w_hh,
                       const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout,
                       const float* dy, const float* w_head,
                       float* d_msg, int ld_dmsg, float* d_h, int ld_dh,
                       const int32_t* add_src, const int32_t* add_dst, const float* add_msg, int ld_add,
                       tmpnn_stream stream);
/* Weight gradient: dW_ih [3H][IN], dW_hh [3H][H], db_ih [3H], db_hh [3H] are ACCUMULATED (+=).
 * x is re-formed as in tmpnn_gru_fwd (xmode).  ws: tmpnn_gru_bwd_weights_ws(R, IN, H) bytes. */
/* Which H = 64 weight-gradient kernel tmpnn_gru_bwd_weights uses: 1 = bf16x6 split products (default), 0 = f32-input
 * MFMA.  A constant of the process (TMPNN_SPLIT_WEIGHTS=0/1 or TMPNN_SPLIT=0 in the environment when the library is
 * loaded): nothing is measured or decided inside a call, so every run and every rank takes the same kernel and
 * gradients are bitwise reproducible across processes.  tmpnn_gru_bwd_weights_variant takes the form explicitly
 * (variant 0 / 1; -1 = the process default) -- used by bench.py to time both forms. */
size_t tmpnn_gru_bwd_weights_ws(int R, int IN, int H);
int tmpnn_gru_bwd_weights(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                          const float* msg, int ld_msg, int msg_compact, int IN,
                          const float* h, int ld_h, int H,
                          const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout,
                          const float* dy, const float* w_head,
                          float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                          void* ws, size_t ws_bytes, tmpnn_stream stream);

/* Fused backward of one cell: tmpnn_gru_bwd_data + tmpnn_gru_bwd_weights in ONE pass over the gates
 * (arguments as in those two; available when tmpnn_gru_bwd_fused_available(H, IN, xmode) != 0, i.e. H = 64,
 * IN = H, xmode 0 or 1).  With xmode 1 the fused row-F adjoint gathers through the cell's own endpoints:
 * add_src / add_dst must be the src / dst arrays.  ws: tmpnn_gru_bwd_fused_ws(R, IN, H) bytes. */
int tmpnn_gru_bwd_fused_available(int H, int IN, int xmode);
size_t tmpnn_gru_bwd_fused_ws(int R, int IN, int H);
int tmpnn_gru_bwd_fused(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                        const float* msg, int ld_msg, int msg_compact, int IN,
                        const float* h, int ld_h, int H, const float* w_ih, const float* w_hh,
                        const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout,
                        const float* dy, const float* w_head,
                        float* d_msg, int ld_dmsg, float* d_h, int ld_dh,
                        const int32_t* add_src, const int32_t* add_dst, const float* add_msg, int ld_add,
                        float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                        void* ws, size_t ws_bytes, tmpnn_stream stream);

/* tmpnn_gru_fwd's xmode 3 (edge cell over the projected det rows `proj` [Dn][ld_proj >= 3H] of tmpnn_rows_linear) over
 * EDGE TILES (rows_per_tile = 32, covering the graph's R edge rows): the distinct projected rows of a tile are staged in
 * LDS an item ahead instead of gathered per edge row after the matrix phase.  H in {32, 64}; bit-identical to
 * tmpnn_gru_fwd(rows = edge_row, xmode = 3, src / dst = det indices).  layers.py:90-97, rows E + H' + I + J. */
int tmpnn_gru_fwd_tiles(const tmpnn_edge_tiles* tiles, int R, const float* proj, int ld_proj, const float* h, int ld_h, int H,
                        const float* whh_t, const float* b_ih, const float* b_hh, float* h_out, int ld_out, float* gates,
                        size_t gate_plane, const float* w_head, float* logit_part, size_t part_stride, tmpnn_stream stream);

/* out[r, 0:NOUT] = in[rows[r], 0:H] @ wt[H][NOUT]  (compact output rows; H in {32, 64}, NOUT = 3H):
 * the det-row projection P of tmpnn_gru_fwd's xmode 3. */
int tmpnn_rows_linear(const int32_t* rows, int R, const float* in, int ld_in, int H, const float* wt, int NOUT,
                      float* out, int ld_out, tmpnn_stream stream);

/* out [cols][rows] = in[rows][cols]^T (weight re-layout for tmpnn_gru_fwd) */
int tmpnn_transpose(const float* in, int rows, int cols, float* out, tmpnn_stream stream);

/* ---- rows C, D: input transform on the NEW rows (models/track_mpnn.py:45-52, 59-61) -------
 * Lin1 -> BatchNorm1d -> ReLU -> Lin2, evaluated only on the nd new det rows (the new edge
 * rows are all-zero inputs: they enter the BatchNorm statistics as `b1` and are masked to zero
 * afterwards, track_mpnn.py:61).  Statistics are per segment (= per window of a block-diagonal
 * batch; one segment reproduces the reference):
 *   xdet [nd][ld_x] gathered det-row features (group columns start at xdet), F inputs
 *   seg_ptr [S+1] det rows of segment s are [seg_ptr[s], seg_ptr[s+1]); seg_cnt [S] = ALL new
 *   rows of the segment (dets + zero rows)
 *   training: batch stats (biased var, eps 1e-5) written to mean/rstd [S][H] and running stats
 *   updated sequentially over segments with momentum 0.1 (unbiased var); else running stats.
 *   y_save [nd][H] = Lin1 output (saved for backward); out rows written to
 *   h_new[out_row[i], 0:H] (ld_h). */
int tmpnn_input_bn_fwd(const float* xdet, int ld_x, int F, int nd, const int32_t* seg_ptr,
                       const int32_t* seg_cnt, const int32_t* seg_of_det /* [nd] segment of each det row, or NULL */,
                       int S, int H, int training,
                       const float* w1, const float* b1, const float* gamma, const float* beta,
                       float* running_mean, float* running_var,
                       const float* w2, const float* b2,
                       float* y_save, float* mean, float* rstd, float* ws_a /* [nd][H] scratch */,
                       const int32_t* out_row, float* h_new, int ld_h, tmpnn_stream stream);
/* Backward.  d_h rows are read at out_row; mean/rstd/y_save are the forward's outputs (eval
 * mode: one row holding the running statistics).  Accumulates (+=) dw1 [H][F], db1, dgamma,
 * dbeta, dw2 [H][H], db2; writes d_xdet [nd][ld_dx] (may be NULL) and d_xzero [S][F] (may be
 * NULL): the gradient every all-zero row of segment s receives through the batch statistics
 * (0 in eval mode).  ws: tmpnn_input_bn_bwd_ws(nd, S, H, F) floats. */
size_t tmpnn_input_bn_bwd_ws(int nd, int S, int H, int F);
int tmpnn_input_bn_bwd(const float* xdet, int ld_x, int F, int nd, const int32_t* seg_ptr,
                       const int32_t* seg_cnt, const int32_t* seg_of_det, int S, int H, int training,
                       const float* w1, const float* b1, const float* gamma, const float* beta,
                       const float* w2,
                       const float* y_save, const float* mean, const float* rstd,
                       const int32_t* out_row, const float* d_h, int ld_dh,
                       float* d_xdet, int ld_dx, float* d_xzero,
                       float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2,
                       float* ws, size_t ws_floats, tmpnn_stream stream);

/* The same transform in ONE launch per direction for batches of SHORT segments (csrc/intf.hip): a workgroup owns whole
 * segments, so the per-segment statistics, the BatchNorm backward's segment sums and the gradient of the zero rows never
 * leave it.  Arguments and results as tmpnn_input_bn_fwd / _bwd (no activation workspace; the backward's workspace holds
 * one slab of parameter gradients per workgroup, added into dw1 .. db2 in a fixed order) plus max_seg_rows = the longest
 * segment's number of det rows, which the host knows from its plan.  tmpnn_input_tf_supported: H in {32, 64}, F <= 128,
 * max_seg_rows <= 128 -- callers use tmpnn_input_bn_* otherwise (one window of thousands of dets is ONE segment).
 * x_rows (int64 [nd], or NULL): det row i's features are row x_rows[i] of `xdet` -- the caller's x [n, F] with its new det
 * rows listed (the gather of the det rows out of x rides in the kernel's own staging pass); NULL: row i. */
int tmpnn_input_tf_supported(int H, int F, int max_seg_rows);
int tmpnn_input_tf_fwd(const float* xdet, const int64_t* x_rows, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int max_seg_rows, int H, int training, const float* w1, const float* b1,
                       const float* gamma, const float* beta, float* running_mean, float* running_var, const float* w2,
                       const float* b2, float* y_save, float* mean, float* rstd, const int32_t* out_row, float* h_new,
                       int ld_h, tmpnn_stream stream);
size_t tmpnn_input_tf_bwd_ws(int nd, int S, int H, int F, int training);   /* bytes */
int tmpnn_input_tf_bwd(const float* xdet, const int64_t* x_rows, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int max_seg_rows, int H, int training, const float* w1, const float* b1,
                       const float* gamma, const float* beta, const float* w2, const float* y_save, const float* mean,
                       const float* rstd, const int32_t* out_row, const float* d_h, int ld_dh, float* d_xdet, int ld_dx,
                       float* d_xzero, float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, void* ws,
                       size_t ws_bytes, tmpnn_stream stream);

/* ---- row J: masked output heads + sigmoid (models/track_mpnn.py:72-75) -------------------
 * logits[i] = is_edge[i] ? w_e . h[i] + b_e : w_n . h[i] + b_n ; scores = sigmoid(logits).
 * h [N][ld_h], C = G*H columns. */
int tmpnn_heads_fwd(const float* h, int ld_h, int C, int N, const uint8_t* is_edge,
                    const float* w_node, const float* b_node, const float* w_edge, const float* b_edge,
                    float* logits, float* scores, tmpnn_stream stream);
/* logits[i] = sum_p parts[p*part_stride + i] + (is_edge[i] ? b_edge : b_node) ; scores = sigmoid(logits). */
int tmpnn_heads_finish(const float* parts, size_t part_stride, int nparts, int N, const uint8_t* is_edge,
                       const float* b_node, const float* b_edge, float* logits, float* scores, tmpnn_stream stream);
/* dy = d_logits + d_scores * s(1-s) (either may be NULL).  dy_out [N] (may be NULL) receives dy;
 * d_h (may be NULL): d_h[i] (=|+=) dy[i] * w_type(i); dw_node/dw_edge [C], db_node/db_edge [1]
 * accumulated (+=).  ws: tmpnn_heads_bwd_ws(N, C) bytes. */
size_t tmpnn_heads_bwd_ws(int N, int C);
int tmpnn_heads_bwd(const float* h, int ld_h, int C, int N, const uint8_t* is_edge,
                    const float* w_node, const float* w_edge, const float* scores,
                    const float* d_logits, const float* d_scores, float* dy_out,
                    float* d_h, int ld_dh, int accumulate,
                    float* dw_node, float* db_node, float* dw_edge, float* db_edge,
                    void* ws, size_t ws_bytes, tmpnn_stream stream);

/* ---- training targets and losses on the same CSR (models/loss.py) ---------------
 * labels / targets: uint8 [N] (1 = positive).  A det's PAST edges are its CSR entries with the sign bit set,
 * its FUTURE edges the others, both in ascending edge-row order.
 *
 * tmpnn_targets (create_targets, loss.py:8-44): targets[det] = labels[det]; per det the LAST label-positive
 * past edge and the FIRST label-positive future edge get target 1, every other edge 0. */
int tmpnn_targets(const tmpnn_graph* g, const uint8_t* labels, uint8_t* targets, tmpnn_stream stream);
/* CELoss (loss.py:77-115): loss[0] = sum over dets and over their two sets (past, future) that contain a
 * positive target of cross_entropy(logits[set], target) / |set|  (target = last positive of the past set,
 * first positive of the future set).  stats [Dn][2][4] (max, sum exp, target row, set size) is saved for the
 * backward; ws: tmpnn_ce_loss_ws(Dn) floats.  Backward: d_logits[edge rows] += d_loss[0] * d loss / d logit;
 * src_pos/dst_pos [E] = det INDEX of each edge's endpoints. */
size_t tmpnn_ce_loss_ws(int Dn);
int tmpnn_ce_loss_fwd(const tmpnn_graph* g, const float* logits, const uint8_t* targets, float* stats, float* loss,
                      float* ws, size_t ws_floats, tmpnn_stream stream);
int tmpnn_ce_loss_bwd(const tmpnn_graph* g, const int32_t* src_pos, const int32_t* dst_pos, const float* logits,
                      const float* stats, const float* d_loss, float* d_logits, tmpnn_stream stream);
/* FocalLoss (loss.py:47-74) over the R rows listed in `rows`: loss_sum[0] = sum_i -(1-pt_i)^gamma log(pt_i) alpha_t,
 * pt = (t ? s : 1-s) + 1e-10 (the caller divides by R for size_average).  Backward: d_scores[row] += d_loss[0] *
 * scale * d loss_i / d s.  ws: tmpnn_focal_loss_ws(R) floats. */
size_t tmpnn_focal_loss_ws(int R);
int tmpnn_focal_loss_fwd(const int32_t* rows, int R, const float* scores, const uint8_t* targets, float gamma,
                         int use_alpha, float alpha0, float alpha1, float* loss_sum, float* ws, size_t ws_floats,
                         tmpnn_stream stream);
int tmpnn_focal_loss_bwd(const int32_t* rows, int R, const float* scores, const uint8_t* targets, float gamma,
                         int use_alpha, float alpha0, float alpha1, const float* d_loss, float scale, float* d_scores,
                         tmpnn_stream stream);
/* Binary cross-entropy with logits, SUMMED over n elements -- the loss SURVEY 8(d)'s metric is quoted with (BCE on every logit of
 * a call against fixed {0,1} targets; `torch.nn.functional.binary_cross_entropy_with_logits(reduction='sum')`):
 * loss_sum[0] = sum_i max(l_i, 0) - l_i t_i + log(1 + exp(-|l_i|)), fixed summation order (bitwise reproducible);
 * backward d_logits[i] = d_loss[0] * (sigmoid(l_i) - t_i).  targets: float [n].  ws: tmpnn_bce_logits_ws(n) floats. */
size_t tmpnn_bce_logits_ws(long n);
int tmpnn_bce_logits_sum_fwd(const float* logits, const float* targets, long n, float* loss_sum, float* ws, size_t ws_floats,
                             tmpnn_stream stream);
int tmpnn_bce_logits_sum_bwd(const float* logits, const float* targets, long n, const float* d_loss, float* d_logits,
                             tmpnn_stream stream);
/* train.py:70-81 for one forward call of a batch-1 window in ONE launch each way: create_targets, the cross-entropy over the
 * logits and the two focal terms with gamma = 0 and no alpha (edge rows; det rows too with the TP classifier), bit for bit
 * the values tmpnn_targets / tmpnn_ce_loss_* / tmpnn_focal_loss_* produce (same expressions, sums in the same order).
 * out [4]: loss_c, focal sum over the edge rows, focal sum over the det rows, loss_f (= mean + mean, NaN over an empty
 * selection as loss.mean() gives).  targets [N] and stats [Dn][2][4] are kept for the backward.  ws: tmpnn_train_losses_ws
 * floats.  Supported while E, Dn <= 8192 (tmpnn_train_losses_supported); larger graphs take the separate entry points.
 * Backward: d_logits [N] / d_scores [N] (either may be NULL) are WRITTEN for every row (zeros where a row has no term);
 * d_c / d_f: the two seeds (device scalars). */
int tmpnn_train_losses_supported(int E, int Dn);
size_t tmpnn_train_losses_ws(int E, int Dn);
int tmpnn_train_losses_fwd(const tmpnn_graph* g, const float* logits, const float* scores, const uint8_t* labels,
                           int tp_classifier, uint8_t* targets, float* stats, float* out, float* ws, size_t ws_floats,
                           tmpnn_stream stream);
int tmpnn_train_losses_bwd(const tmpnn_graph* g, const int32_t* src_pos, const int32_t* dst_pos, const float* logits,
                           const float* scores, const uint8_t* targets, const float* stats, const float* d_c, const float* d_f,
                           int tp_classifier, float* d_logits, float* d_scores, tmpnn_stream stream);

/* ======================================================================================================
 * Batch-1 path (the reference's real call pattern, train.py:92-107 / infer.py:60-87: ONE small
 * graph per call).  Two things make a call cheap when the graph has a few thousand rows:
 *   (1) the graph's SIZES stay on the device (tmpnn_dgraph): the adjacency -> index conversion is one kernel and the
 *       host never waits for E / Dn, so nothing synchronises between calls;
 *   (2) the whole message-passing iteration is two launches forward (input transform; edge + node cells with the
 *       aggregation, the merge and the output heads fused) and two backward: tmpnn_mp_iter_fwd / _bwd.
 * Limits: N <= TMPNN_DG_BIG_ROWS rows, H in {32, 64}, no attention heads (the staged entry points above cover the
 * rest).  Arithmetic: fp32 throughout (v_mfma_f32_16x16x4_f32 = an fmaf chain), reductions in a fixed order.
 * ====================================================================================================== */
#define TMPNN_DG_MAX_ROWS 4096
#define TMPNN_DG_BIG_ROWS 65535    /* fused iteration and tmpnn_graph_from_coo_arena_ws (work arrays in a global scratch) */
#define TMPNN_TRACK_MAX_ROWS 32768 /* tracker-side operations tmpnn_track_* (row deletion keeps an N-int table in the LDS) */
#define TMPNN_DG_META 8      /* ints in tmpnn_dgraph.meta: [0] E, [1] Dn, [2] status, [3] N, rest reserved */
/* status bits (0 = the adjacency is a TrackMPNN factor graph, SURVEY 8 "graph invariants") */
#define TMPNN_DG_BAD_VALUE 1     /* an off-diagonal entry is not +-1, or an index is out of range */
#define TMPNN_DG_BAD_ROW 2       /* an edge row without exactly one +1 and one -1, or a det row with off-diagonals */
#define TMPNN_DG_BAD_ENDPOINT 4  /* an edge endpoint is not a det row */
#define TMPNN_DG_BAD_ORDER 8     /* not src row < edge row < dst row (utils/graph.py:153-156,298-301) */
#define TMPNN_DG_BAD_EDGE_DIAG 16 /* diag(edge_adj) does not complement diag(node_adj) */
#define TMPNN_DG_BAD_EDGE_ADJ 32 /* edge_adj is not node_adj^T off the diagonal */

/* Index form of one graph with device-side sizes.  All arrays are caller-owned device memory of capacity `cap`
 * rows (one arena of tmpnn_dgraph_ints(cap) int32, carved by tmpnn_dgraph_bind).  When status != 0 the producer
 * stores E = Dn = 0, so that consumers touch nothing; the host reads meta when it chooses to (deferred check). */
typedef struct tmpnn_dgraph {
    int32_t N;          /* rows of the state tensor: host-known (the adjacency's shape) */
    int32_t cap;        /* capacity of the arrays, >= N */
    int32_t* meta;      /* [TMPNN_DG_META] */
    uint8_t* is_edge;   /* [cap]   1 on edge rows */
    int32_t* pos;       /* [cap]   row -> index within its type */
    int32_t* src;       /* [cap]   per edge index: ROW of the +1 det */
    int32_t* dst;       /* [cap]   per edge index: ROW of the -1 det */
    int32_t* src_pos;   /* [cap]   per edge index: det INDEX of src */
    int32_t* dst_pos;   /* [cap]   per edge index: det INDEX of dst */
    int32_t* edge_row;  /* [cap]   row of edge e (ascending) */
    int32_t* det_row;   /* [cap]   row of det d (ascending) */
    int32_t* rowptr;    /* [cap+1] det -> incident edges CSR */
    int32_t* inc;       /* [2 cap] edge row | sign bit, ascending edge row per det (as tmpnn_graph.inc) */
} tmpnn_dgraph;

size_t tmpnn_dgraph_ints(int cap);
/* pure host arithmetic: point `out`'s arrays into `arena` (tmpnn_dgraph_ints(cap) int32, 16-byte aligned) */
int tmpnn_dgraph_bind(void* arena, int cap, int N, tmpnn_dgraph* out);

/* Adjacency pair -> tmpnn_dgraph in ONE launch (replaces the dense round trips of models/track_mpnn.py:55-56 and
 * models/layers.py:85-88, and validates what utils/graph.py:151-163,294-308 build).  COO entries as torch stores
 * them: idx int64 [2][nnz] (rows then columns), val fp32 [nnz]; explicit zeros and UNcoalesced diagonals are fine
 * (duplicate diagonal entries are summed; duplicate off-diagonal entries make the row invalid).  edge_idx may be
 * NULL (no cross-check).  N <= TMPNN_DG_MAX_ROWS. */
int tmpnn_graph_from_coo(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                         const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge,
                         const tmpnn_dgraph* g, tmpnn_stream stream);

/* tmpnn_graph_from_coo on an arena of tmpnn_dgraph_ints(cap) int32 (= tmpnn_dgraph_bind + the conversion, one call) */
int tmpnn_graph_from_coo_arena(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                               const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge, void* arena, int cap,
                               tmpnn_stream stream);
/* The same conversion for graphs of up to TMPNN_DG_BIG_ROWS rows (dense scenes): above TMPNN_DG_MAX_ROWS the work
 * arrays do not fit the LDS and live in `ws` (tmpnn_graph_from_coo_ws_ints(N) ints, 0 for small N: then this is
 * tmpnn_graph_from_coo_arena). */
int tmpnn_graph_from_coo_arena_ws(int N, const int64_t* node_idx, const float* node_val, int64_t nnz_node,
                                  const int64_t* edge_idx, const float* edge_val, int64_t nnz_edge, void* arena, int cap,
                                  void* ws, size_t ws_ints, tmpnn_stream stream);

/* The same index form from the ROW form of a graph (type mask + the two endpoint rows of every edge row): what the
 * tracker-side operations below edit.  Same validation, same status bits. */
/* ... for graphs of up to TMPNN_DG_BIG_ROWS rows (`ws`: tmpnn_graph_from_coo_ws_ints(N) ints, as for tmpnn_graph_from_coo_arena_ws). */
int tmpnn_graph_from_rows_ws(int N, const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst,
                             const tmpnn_dgraph* g, void* ws, size_t ws_ints, tmpnn_stream stream);

/* Parameters of the model as device pointers in the reference's layouts (state_dict keys of SURVEY 8(b)); the same
 * struct with gradient buffers is what tmpnn_mp_iter_bwd accumulates into (+=).  G <= 3 feature groups. */
typedef struct tmpnn_mp_params {
    int32_t G, H, IN_e /* H (diff) or 2H (concat) */, F_total;
    int32_t F[3];                 /* input width of each group's transform */
    float* w1[3]; float* b1[3]; float* gamma[3]; float* beta[3]; float* w2[3]; float* b2[3];   /* input_transforms.g.{0,1,3} */
    float* run_mean[3]; float* run_var[3];                                                      /* BatchNorm buffers (NULL in a gradient struct) */
    int64_t* num_batches_tracked[3];                                                            /* BatchNorm counters, +1 per training call with new rows (may be NULL) */
    float* e_wih[3]; float* e_whh[3]; float* e_bih[3]; float* e_bhh[3];                         /* factor_grus.g.edge_gru */
    float* n_wih[3]; float* n_whh[3]; float* n_bih[3]; float* n_bhh[3];                         /* factor_grus.g.node_gru */
    float* w_node; float* b_node; float* w_edge; float* b_edge;                                 /* output_transform_{node,edge} */
} tmpnn_mp_params;

/* MFMA-operand images of the four GRU weight matrices of every group (forward and backward-data forms): one
 * launch, to be repeated whenever the weights change (once per optimizer step).  prep: tmpnn_mp_iter_prep_floats. */
size_t tmpnn_mp_iter_prep_floats(int G, int H, int IN_e);
int tmpnn_mp_iter_prepare(const tmpnn_mp_params* P, float* prep, tmpnn_stream stream);

/* One TrackMPNN.forward call (models/track_mpnn.py:54-75 + models/layers.py:84-116).
 *   h [N][G*H]: rows [0, N - n_new) hold the carried state on entry; the n_new new rows are written (input
 *   transform on new det rows, zeros on new edge rows) -- this is the h the iteration reads.  x [n_new][ld_x]:
 *   features of the new rows (only det rows are read: edge rows are all-zero by contract, utils/graph.py:148,291).
 *   Outputs: h_out [N][G*H], logits [N], scores [N].  save: tmpnn_mp_iter_save_floats floats kept for
 *   tmpnn_mp_iter_bwd (may be NULL only for an inference call without new rows).  training != 0: batch statistics (one segment) + running-stat update. */
size_t tmpnn_mp_iter_save_floats(int N, int n_new, int G, int H);
int tmpnn_mp_iter_fwd(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new,
                      const float* x, int ld_x, float* h, int training,
                      float* h_out, float* logits, float* scores, float* save, size_t save_floats,
                      tmpnn_stream stream);
/* The same call in PARTS, for models with attention heads (K > 0, models/layers.py:105-112): the attention-weighted
 * aggregate replaces the signed sum between the call's two launches.  parts: bit 0 = skip the input transform launch,
 * bit 1 = skip the iteration launch, bit 2 = the aggregate es [G][N][H] (by det INDEX, the layout of the save buffer at
 * tmpnn_mp_iter_save_es_offset floats) has been written by the caller -- tmpnn_att_fwd(out = save + offset + g N H,
 * ld_out = H) on the state `h` that the input transform launch completed -- and the det tiles read it instead of forming
 * the signed sum.  Sequence: parts = 2, tmpnn_att_fwd per feature group, parts = 1 | 4.  parts = 0 is tmpnn_mp_iter_fwd.
 * bit 3 (round 6; inference only): scores[det rows] = 1 -- a model trained without the TP classifier (the reference README's
 * commands, README.md:52-67), whose detections all count as true positives in the loop (infer.py:53-56, 77-80); logits unchanged. */
size_t tmpnn_mp_iter_save_es_offset(int N, int n_new, int G, int H);
int tmpnn_mp_iter_fwd_parts(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new,
                            const float* x, int ld_x, float* h, int training,
                            float* h_out, float* logits, float* scores, float* save, size_t save_floats, int parts,
                            tmpnn_stream stream);
/* Backward of tmpnn_mp_iter_fwd.  d_scores / d_logits (N values each, element stride st_* >= 0; 0 = one broadcast
 * value, the gradient of a plain sum) and d_hout [N][G*H] may each be NULL.  Writes d_h
 * [N][G*H] (gradient of the h the iteration read: its first N - n_new rows are the gradient of the carried state)
 * and d_x [n_new][F_total] (may be NULL); ACCUMULATES (+=) every parameter gradient into `grads`.
 * ws: tmpnn_mp_iter_bwd_ws bytes. */
size_t tmpnn_mp_iter_bwd_ws(int N, int n_new, int G, int H, int IN_e);
int tmpnn_mp_iter_bwd(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new,
                      const float* x, int ld_x, const float* h, const float* h_out, const float* scores,
                      const float* save, int training,
                      const float* d_scores, int st_dscores, const float* d_logits, int st_dlogits,
                      const float* d_hout, float* d_h, float* d_x, const tmpnn_mp_params* grads,
                      void* ws, size_t ws_bytes, tmpnn_stream stream);
/* ... in PARTS: bit 0 = skip the tile launch (d_h, d_msg = the first N * G * IN_e floats of ws, weight-gradient slabs),
 * bit 1 = skip the finish launch (adjoints of the aggregations on the carried rows, slab reduction, input transform
 * backward), bit 2 = the finish launch does NOT add the adjoint of the edge -> node sum to d_h's edge rows: the caller has
 * -- tmpnn_att_bwd(d_out = ws + g IN_e, ld_dout = G IN_e, d_h + g H, ld_dh = G H) per feature group between the two.
 * Sequence: parts = 2, tmpnn_att_bwd per group, parts = 1 | 4. */
int tmpnn_mp_iter_bwd_parts(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new,
                            const float* x, int ld_x, const float* h, const float* h_out, const float* scores,
                            const float* save, int training,
                            const float* d_scores, int st_dscores, const float* d_logits, int st_dlogits,
                            const float* d_hout, float* d_h, float* d_x, const tmpnn_mp_params* grads,
                            void* ws, size_t ws_bytes, int parts, tmpnn_stream stream);

/* ======================================================================================================
 * Tracker-side graph maintenance on the device (csrc/trackops.hip).  Between two model
 * calls the reference moves a DENSE N x N adjacency and the hidden state to the host and back
 * (utils/graph.py:216-221,326-332 and :420-425,532-537).  Here the graph lives in HBM in ROW form, one entry per
 * state row, all int32 unless noted (capacity <= TMPNN_TRACK_MAX_ROWS rows):
 *     ts, det_id, assoc            y_pred[:, 0..2]: timestep (-1 on edge rows), detection id, associated next detection id
 *     is_edge (uint8), row_src, row_dst   the +1 / -1 det ROW of an edge row (-1 on det rows)
 *     labels (uint8)               ground-truth class (training; may be NULL at inference)
 * and its index form is re-derived with tmpnn_graph_from_rows after every edit.  score: P(positive) per row, fp32
 * (scores[:, 1] of the reference).  Hungarian matching and the walk that finalises tracks stay on the host.
 * ====================================================================================================== */
/* y_pred[:, 2].  mode 0 (training, utils/graph.py:229-245): through the one label-positive future edge; false
 * positives point at themselves; status bit 0 is set if a det has more than one positive future edge.
 * mode 1 (inference, greedy: :251-268 and :437-454): best-scoring future edge (>= 0.5, to a det >= 0.5) of the nearest
 * timestep. */
/* Active set at time t (utils/graph.py:270-278): rows in ascending order into active[], their number into count[0]. */
/* Append the block of timestep t behind row N (utils/graph.py:283-325): A x D edge rows (src-major) then D det rows
 * with ids new_ids[D]; labels from track[det id] (int32 [ND] track of every detection, -1 = false positive; NULL at
 * inference).  The row arrays must have room for N + A*D + D entries. */
/* The rows decode_tracks deletes (utils/graph.py:492-512) as a stream compaction: keep[] = kept rows (ascending),
 * count[0] = their number, count[2] = how many of them are det rows (count: >= 3 ints), o_* = the compacted row form with
 * renumbered endpoints. */
/* out[q][0:W] = in[keep[q]][0:W] for q < count[0] (count read on the device; the launch covers max_rows): the hidden
 * state and the scores follow the deletion without leaving HBM (utils/graph.py:514,519). */
/* decode_tracks' track finalisation (utils/graph.py:456-490) on the device: y_track [ND] = y_out[:, 1] of the sequence
 * (int32, -1 = no track yet; it stays in device memory between calls), updated for the window's dets from the
 * association links assoc[] (det ids, the output of tmpnn_track_associate or of the Hungarian matching), the scores and
 * t_upto exactly as the reference's walk over all detections does.  pos_of_det [ND]: scratch (det id -> position among
 * the graph's dets; entries of dets outside the graph are never read).  ws: tmpnn_track_finalize_ws(N) bytes, only
 * needed when the graph may hold more than 4096 dets (0 otherwise). */
size_t tmpnn_track_finalize_ws(int max_dets);

/* One call per phase of a timestep, as the reference's loops call update_graph / decode_tracks (train.py:102-104,
 * infer.py:70-87): the kernels above enqueued back to back.  At batch 1 a timestep is bound by the number of calls and
 * launches, not by their work.  `small`: int32 [4] in device memory -- [0] a count (active set / kept rows), [1] status bits
 * of the label rule, [2] kept det rows, [3] the NEXT timestep's active-set size (tmpnn_track_retire with next_t >= 0). */
typedef struct tmpnn_track_rows {   /* the row form (see above); labels may be NULL at inference */
    int32_t *ts, *det_id, *assoc;
    uint8_t* is_edge;
    int32_t *src, *dst;
    uint8_t* labels;
} tmpnn_track_rows;
/* initialize_graph (utils/graph.py:96-186): the first block of a sequence from ONE packed upload -- packed [6][N] int32 =
 * ts, det_id, is_edge, src, dst, labels rows; assoc = -1; feats [N][ld_f] = X[det id][0:F] on det rows, zeros on edge rows
 * (NULL: not written); y_track [ND] = -1 (NULL: untouched); then the index form into g (as tmpnn_graph_from_rows_ws). */
int tmpnn_track_load(int N, int ND, const int32_t* packed, const tmpnn_track_rows* rows, const float* X, int ld_x, int F,
                     float* feats, int ld_f, int32_t* y_track, const tmpnn_dgraph* g, void* ws, size_t ws_ints,
                     tmpnn_stream stream);
/* update_graph, first half (utils/graph.py:227-278): the associations (skipped with associate = 0: rows->assoc is current)
 * and the active set of timestep t -> active[], small[0]. */
int tmpnn_track_select(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int mode, int t,
                       int associate, int32_t* active, int32_t* small, tmpnn_stream stream);
/* The same with a scratch for associate = 2: the associations by OPTIMAL ASSIGNMENT per timestep (reference hungarian(),
 * utils/graph.py:33-93 -- scipy's linear_sum_assignment restated on the device, ties included; README.md:67,122 recommends
 * --hungarian).  Inference graphs (mode 1) of <= TMPNN_DG_MAX_ROWS rows; a timestep's problem may have up to
 * tmpnn_track_hungarian_max_dets() rows / columns; cost matrices beyond 4096 entries use `ws` (fp32 [rows x columns]).  A problem
 * that fits neither sets bit 1 (value 2) of small[1] and is left unassociated: the caller then matches on the host instead. */
int tmpnn_track_select_ws(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int mode, int t,
                          int associate, int32_t* active, int32_t* small, void* ws, size_t ws_bytes, tmpnn_stream stream);
int tmpnn_track_hungarian_max_dets(void);
/* update_graph, second half (:283-332): the block of timestep t appended behind row N (tmpnn_track_append), the features
 * of the new rows written (feats [A*D + D][ld_f]: zeros on edge rows, X[new_ids[j]][0:F] on det rows; NULL: not written)
 * and the index form of the grown graph derived into g_new (bound for N + A*D + D rows; ws / ws_ints as
 * tmpnn_graph_from_rows_ws). */
int tmpnn_track_extend(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                       const tmpnn_track_rows* rows, const float* X, int ld_x, int F, float* feats, int ld_f,
                       const tmpnn_dgraph* g_new, void* ws, size_t ws_ints, tmpnn_stream stream);
/* update_graph's second half AND the input transform of the model call that follows it, in ONE launch (round 6; inference,
 * graphs of N + A*D + D <= TMPNN_DG_MAX_ROWS rows, the fused batch-1 path's H in {32, 64}): block 0 appends the block and
 * derives the grown graph's index form into g_new as tmpnn_track_extend does; G further blocks run the transform of the D new
 * dets in eval mode (running statistics) straight from X[new_ids[j]][0:F_total] into h[N + A*D + j][0:G*H] and write zeros to the
 * A*D new edge rows of h -- what the first launch of tmpnn_mp_iter_fwd(training = 0) does with the features tmpnn_track_extend
 * would have written (same code, same arithmetic order: bit-identical).  The caller follows with
 * tmpnn_mp_iter_fwd_parts(parts = 1, x = NULL).  h [N + A*D + D][G*H]: rows [0, N) hold the carried state.  save / save_floats:
 * as tmpnn_mp_iter_fwd for (N + A*D + D, A*D + D) -- the transform's Lin1 outputs and statistics land where that call puts them.
 * counts (or NULL): the `small` words of the tmpnn_track_retire call in front of this one on the stream.  The launch then takes
 * N = counts[0] and A = counts[3] ON THE DEVICE and the arguments N / A (and g_new's binding, h, save) are upper bounds the buffers
 * were sized with: the caller enqueues the timestep's first launch without waiting for the previous decode's counters, reads them
 * while it runs, and re-binds g_new's arena (same capacity) with the exact row count for the calls that follow.
 * gather_src / ld_gather / gather_keep (with counts; or NULL): that decode was called with h_new = NULL, i.e. it left the kept
 * rows' state where it was; further blocks of this launch move it -- h[q][0:G*H] = gather_src[gather_keep[q]][0:G*H] for
 * q < counts[0] -- beside the append and the transform (which touch rows >= N only). */
int tmpnn_track_extend_tf(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                          const tmpnn_track_rows* rows, const float* X, int ld_x, const tmpnn_mp_params* P, float* h,
                          float* save, size_t save_floats, const tmpnn_dgraph* g_new, const int32_t* counts,
                          const float* gather_src, int ld_gather, const int32_t* gather_keep, tmpnn_stream stream);
/* decode_tracks (:431-520): associations from the scores (associate = 1: the greedy rule; 2: optimal assignment per timestep as
 * tmpnn_track_select_ws, graphs of <= TMPNN_DG_MAX_ROWS rows, its cost scratch = fin_ws / fin_ws_bytes, overflow in bit 1 of
 * small[1]; 0: rows->assoc holds them already, e.g. from a matching on the host), track finalisation, row deletion into rows_out, the state rows and scores compacted (h_new [N][ld_hn],
 * s_new [N]; small[0] / small[2] = kept rows / kept det rows; h_new = NULL on graphs of <= TMPNN_DG_MAX_ROWS rows: the kept rows'
 * state is NOT moved -- `keep` lists the rows, the caller moves them, e.g. tmpnn_track_extend_tf's gather_* arguments).  next_t >= 0: also the active set of timestep next_t on
 * the compacted rows by the inference rule -> active[], small[3]; the caller then reads small once per timestep instead of twice.
 * Greedy associations carry over (deletion removes no future edge of a kept det); with associate = 2 the next update_graph would
 * re-derive them by its own assignment sweep over the compacted graph (a det that was assigned and deleted frees its column), so
 * that sweep runs here, over the rows that stay, and rows_out->assoc holds ITS result (round 5).
 * notify (or NULL; round 6): int32 [8] in PINNED HOST memory that the runtime maps into the device's address space at the same
 * address (hipHostMalloc; checked).  The caller clears notify[4]; the launch that produces the counters stores small[0..3] into
 * notify[0..3] and then 1 into notify[4] (release order, system scope).  The host may then poll notify[4] (acquire) instead of
 * copying `small` back behind the call: it has the counts while the kept rows' state is still moving, and no copy is enqueued.
 * The device never waits for the host; a caller that does not see the flag may still synchronise and read `small`. */
int tmpnn_track_retire(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int associate, int t_upto,
                       int ret_win, int32_t* y_track, int ND, int32_t* pos_of_det, void* fin_ws, size_t fin_ws_bytes,
                       int32_t* keep, int32_t* small, const tmpnn_track_rows* rows_out, const float* h, int ld_h, int W,
                       float* h_new, int ld_hn, float* s_new, int next_t, int32_t* active, int32_t* notify,
                       tmpnn_stream stream);


/* ======================================================================================================
 * Wide cells (H = 128 / 256, diff messages; BASELINE.json C5) as LDS-tiled GEMMs on bf16x6 split products
 * (csrc/wide.hip).  Same arithmetic contract as the H <= 64 kernels: fp32 in / out, fp32 accumulate, error no larger
 * than the f32 MFMA chain.  tmpnn_wide_prepare splits the cell's two weight matrices into their MFMA operand images
 * once per optimizer step (prep: tmpnn_wide_prep_bytes bytes, 16-byte aligned).
 * ====================================================================================================== */
int tmpnn_wide_supported(int H, int IN);
size_t tmpnn_wide_prep_bytes(int H, int IN);
int tmpnn_wide_prepare(const float* w_ih /* [3H][IN] */, const float* w_hh /* [3H][H] */, int IN, int H, void* prep,
                       tmpnn_stream stream);
/* Forward of one cell over the rows `rows[R]` with the diff message taken through the projected det rows (rows E + H'
 * + I): P [Dn][3H] = h[det_rows] W_ih^T is written here; gi = P[src_pos[r]] - P[dst_pos[r]] (det INDICES),
 * gh = h[rows[r]] W_hh^T, h_out[rows[r]] = GRUCell; gates: NULL or the 4 planes of tmpnn_gru_fwd. */
int tmpnn_wide_gru_fwd(const void* prep, const int32_t* det_rows, int Dn, const int32_t* rows, int R,
                       const int32_t* src_pos, const int32_t* dst_pos, const float* h, int ld_h, int H,
                       const float* b_ih, const float* b_hh, float* P, float* h_out, int ld_out, float* gates,
                       size_t gate_plane, tmpnn_stream stream);
/* The same forward over edge tiles (rows_per_tile = 128; tiles cover the graph's R edge rows): P rows of a tile's det
 * list are staged in LDS once per tile and hidden chunk instead of gathered per edge row; bit-identical results. */
int tmpnn_wide_gru_fwd_tiled(const void* prep, const int32_t* det_rows, int Dn, const tmpnn_edge_tiles* tiles, int R,
                             const float* h, int ld_h, int H, const float* b_ih, const float* b_hh, float* P,
                             float* h_out, int ld_out, float* gates, size_t gate_plane, tmpnn_stream stream);
/* Data gradient (as tmpnn_gru_bwd_data with IN = H, no fused adjoint): d_msg[rows[r]][0:H] = d_gi W_ih,
 * d_h[rows[r]] = dh z + d_gh W_hh.  ws: tmpnn_wide_gru_bwd_data_ws(R, H) bytes (the materialised d_gi, d_gh). */

/* Weight gradient of the same cell from the gate gradients tmpnn_wide_gru_bwd_data left in ITS workspace (`dg_ws`, read
 * only): dW_ih += d_gi^T (h[src] - h[dst]), dW_hh += d_gh^T h[rows], db_ih / db_hh += column sums (replaces
 * tmpnn_gru_bwd_weights for the wide cells: layers.py:84-116 backward).  ws: tmpnn_wide_gru_bwd_weights_ws(R, H) bytes. */

/* The whole backward of a wide EDGE cell under the diff message x[e] = h[src e] - h[dst e] (models/layers.py:90-95, 107),
 * with both W_ih products taken on the det side by linearity -- the backward twin of the forward's projected det rows:
 *     S[d] = sum_{e: src = d} d_gi[e] - sum_{e: dst = d} d_gi[e]
 *     d_h[det_row[d]] += S[d] W_ih                 (the message adjoint: replaces d_x = d_gi W_ih + tmpnn_gather_diff_bwd)
 *     dW_ih += S^T h[det rows]                     (replaces d_gi^T (h[src] - h[dst]) over the E edge rows)
 * and, over the edge rows, d_h[edge_row[e]] = dh z + d_gh W_hh (plain store), dW_hh += d_gh^T h, db_ih / db_hh += column
 * sums.  The gate gradients are materialised once as [N][4H] = [dr | dz | dn | dn r] indexed by graph row.
 * Replaces tmpnn_wide_gru_bwd_data + tmpnn_wide_gru_bwd_weights + tmpnn_gather_diff_bwd for this cell: two of the four
 * (E x 3H x H) products run over Dn rows.  ws: tmpnn_wide_gru_bwd_diff_ws(N, E, Dn, H) bytes. */
size_t tmpnn_wide_gru_bwd_diff_ws(int N, int E, int Dn, int H);
int tmpnn_wide_gru_bwd_diff(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                            size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                            float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                            size_t ws_bytes, tmpnn_stream stream);
/* The same call with the det-side branch (the signed segment sums of d_gi and the message adjoint) enqueued on a SECOND
 * stream next to the two E-row matrix kernels: forked from `stream` after the gate gradients, joined before the det-side
 * weight gradient, so `stream` order alone still covers every output.  Results are bit-identical to tmpnn_wide_gru_bwd_diff.
 * aux_stream must differ from stream; do not use while `stream` is being captured into a graph.  ev_fork / ev_join: two
 * events of the caller (hipEventDisableTiming is enough) that the call records and waits on -- the library creates, destroys
 * and synchronises nothing; on return (also with an error code) `stream` waits for everything enqueued on aux_stream. */
/* ... and with the adjoint of row F (models/layers.py:103: d_h[e] += add_msg[src[e]] - add_msg[dst[e]], what
 * tmpnn_gather_diff_fwd(g, add_msg, ld_add, d_h, ld_dh, H, accumulate = 1) would add afterwards) taken in the epilogue of the
 * E-row product: one read-modify-write pass over d_h's edge rows less.  add_msg: the table whose det rows hold d_es (>= H
 * columns); aux_stream may be NULL (one stream; the events are then unused).  Bit-identical to the two calls. */
int tmpnn_wide_gru_bwd_diff_fused(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                                  size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                                  float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                                  size_t ws_bytes, const float* add_msg, int ld_add, tmpnn_stream stream,
                                  tmpnn_stream aux_stream, tmpnn_event ev_fork, tmpnn_event ev_join);

/* ---- comparison builds only (-DTMPNN_KEEP_VARIANTS; tools/build_variant.sh) ------------------------------------------------
 * Superseded forms kept for A/B runs: the two-kernel backward of the wide cells (the shipped path takes both W_ih products on
 * the det side, tmpnn_wide_gru_bwd_diff*), its two-stream form, the explicit choice of the H = 64 weight-gradient kernel, the
 * block append without features.  The shipped libtmpnn.so does NOT export them. */
#ifdef TMPNN_KEEP_VARIANTS
int tmpnn_track_append(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                       int32_t* ts, int32_t* det_id, int32_t* assoc, uint8_t* is_edge, int32_t* row_src,
                       int32_t* row_dst, uint8_t* labels, tmpnn_stream stream);
int tmpnn_gru_bwd_weights_variant(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                                  const float* msg, int ld_msg, int msg_compact, int IN,
                                  const float* h, int ld_h, int H,
                                  const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout,
                                  const float* dy, const float* w_head,
                                  float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                                  void* ws, size_t ws_bytes, int variant, tmpnn_stream stream);
int tmpnn_gru_bwd_weights_choice(void);
int tmpnn_wide_gru_bwd_diff_aux(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                            size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                            float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                            size_t ws_bytes, tmpnn_stream stream,
                                tmpnn_stream aux_stream, tmpnn_event ev_fork, tmpnn_event ev_join);
size_t tmpnn_wide_gru_bwd_data_ws(int R, int H);
int tmpnn_wide_gru_bwd_data(const void* prep, const int32_t* rows, int R, const float* h, int ld_h, int H,
                            const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy,
                            const float* w_head, float* d_msg, int ld_dmsg, float* d_h, int ld_dh, void* ws,
                            size_t ws_bytes, tmpnn_stream stream);
size_t tmpnn_wide_gru_bwd_weights_ws(int R, int H);
int tmpnn_wide_gru_bwd_weights(const void* dg_ws, const int32_t* rows, int R, const int32_t* src, const int32_t* dst,
                               const float* h, int ld_h, int H, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                               void* ws, size_t ws_bytes, tmpnn_stream stream);
#endif

#ifdef __cplusplus
}
#endif
#endif /* TMPNN_H */
