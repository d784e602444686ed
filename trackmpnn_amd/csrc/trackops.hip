// Tracker-side graph maintenance on the device: what the reference does between two
// model calls on the host with a DENSE N x N numpy adjacency and two PCIe crossings per timestep
// (utils/graph.py:189-334 update_graph, :392-539 decode_tracks) is here a handful of small kernels over the ROW form
// of the graph, which stays in HBM next to the hidden state:
//
//     ts[N]       timestep of a det row, -1 on edge rows            (y_pred[:, 0])
//     det_id[N]   index of the detection in the sequence, -1 edges  (y_pred[:, 1])
//     assoc[N]    det_id of the associated next detection, or -1    (y_pred[:, 2])
//     is_edge[N], row_src[N], row_dst[N]                            (node_adj: +1 / -1 column of an edge row)
//     labels[N]   ground-truth class of the row (training)
//
//   tmpnn_track_associate   utils/graph.py:227-268 / :431-454  y_pred[:, 2] from the labels (train) or the scores (greedy)
//   tmpnn_track_active      :270-278                           the dets offered for association at time t (compacted)
//   tmpnn_track_append      :283-325                           A x D new edge rows (src-major) + D new det rows, labels
//   tmpnn_track_delete      :492-520                           which rows decode_tracks drops; compacted row form
//   tmpnn_track_gather      :514-519                           stream compaction of the state rows (h, scores) by the kept rows
//
//   tmpnn_track_finalize    :456-490                           y_out[:, 1]: the walk along the association links
//
// The index form (CSR etc.) is re-derived from the rows by tmpnn_graph_from_rows (csrc/graphconv.hip).  The Hungarian
// matching runs on the device too since round 5 (d_track_hungarian: scipy's algorithm restated).  Integer work,
// HBM/latency bound, graphs of <= TMPNN_TRACK_MAX_ROWS rows: single-workgroup kernels with LDS scans.
#include <algorithm>

#include "common.h"
#include "graphconv_dev.h"
#include "small_bn_dev.h"

namespace tmpnn {

static constexpr int TK_THREADS = 1024;

__device__ __forceinline__ int tk_block_scan(int v, int* s_wave, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < TK_THREADS / 64; ++w) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[TK_THREADS / 64] = run;
    }
    __syncthreads();
    const int res = s_wave[wave] + inc - v;
    *total = s_wave[TK_THREADS / 64];
    __syncthreads();
    return res;
}

// ---- y_pred[:, 2] by optimal assignment (reference hungarian(), utils/graph.py:33-93; README: --hungarian) -----------------
// Swept in ascending timestep t over the timesteps that edges lead into.  Rows of the problem: the dets with an edge into t
// that are still unassociated (ascending graph row -- np.unique), columns: EVERY det of timestep t (ascending row), cost
// 1 - score of the edge (fp32), 100 where there is none; an assignment is kept where its cost is <= 0.5.  The solver is scipy's
// linear_sum_assignment (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp: shortest augmenting paths in fp64, the matrix
// transposed when it has more rows than columns) restated step for step, INCLUDING how it breaks ties, because which of several
// optimal assignments comes out decides the tracks: the scan over the `remaining` columns keeps the first column of minimal
// reduced cost unless a later one of equal cost is unassigned (then the last such), and `remaining` is filled in reverse and
// compacted by moving its last entry into the hole.  One wave runs a problem: lane l owns columns l, l + 64, ... (reduced costs,
// duals, path, position in `remaining` in registers), the row state lives in LDS; the scan's sequential rule is two wave
// reductions over positions.  Problems of up to HG_MAX rows / columns; cost matrices of up to HG_LDS_COST entries in LDS, larger
// ones in the caller's scratch.
static constexpr int HG_MAX = 256, HG_K = HG_MAX / 64, HG_LDS_COST = 4096;
struct HgShared {
    double u[HG_MAX], spc[HG_MAX];
    int col4row[HG_MAX], path[HG_MAX], rowlist[HG_MAX];
    unsigned char SR[HG_MAX];
    float cost[HG_LDS_COST];
    unsigned char flag[4096];
    short ridx[4096];
    short ad[4096];                                       // per det index: det index of its association, -1 none
    int wave[TK_THREADS / 64 + 1];
    int d0, d1, tnext, nr, nrows;
};
// A retire launch runs the sweep twice (decode_tracks on the grown graph, then the next update_graph's over the rows that stay).
// The second sweep's problem of a timestep is the first sweep's whenever its rows are the same dets (its columns -- every det of
// a timestep at or after t_upto -- and its costs -- the scores -- are; an edge that goes away starts at a deleted det, which was a
// row of the first problem or took no part in it): the first sweep leaves rows and outcome of each problem here and the second
// takes the outcome over instead of solving again (round 6; in steady state only the oldest timestep's problem differs).
static constexpr int HG_MEMO_T = 16, HG_MEMO_R = 64;
struct HgMemo {
    int n;
    int t[HG_MEMO_T], nr[HG_MEMO_T], nc[HG_MEMO_T], d0[HG_MEMO_T];
    short rows[HG_MEMO_T][HG_MEMO_R], assign[HG_MEMO_T][HG_MEMO_R];    // det index of a row; det index it was associated with, -1 none
};
__device__ __forceinline__ void hg_wave_sync() {          // LDS writes of this wave visible to its other lanes (one wave only)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Wave-wide minima without the LDS crossbar (__shfl_xor is a ds_bpermute: ~120 cycles a step, five reductions per scan made a
// scan ~3600 cycles): four DPP exchange steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), then the four rows through v_readlane and the scalar unit.  Keys are unsigned: a double goes through the usual
// order-preserving map of its bits.
template <int CTRL>
__device__ __forceinline__ uint32_t hg_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint64_t hg_min_step64(uint64_t x) {
    const uint64_t y = ((uint64_t)hg_dpp<CTRL>((uint32_t)(x >> 32)) << 32) | hg_dpp<CTRL>((uint32_t)x);
    return y < x ? y : x;
}
// ROWS: rows of 16 lanes that can hold a live key (problems of <= 16 columns: the first row alone -- three pairs of v_readlane less)
template <int ROWS = 4>
__device__ __forceinline__ uint64_t hg_wave_min64(uint64_t x) {
    x = hg_min_step64<0xB1>(x);          // quad_perm [1,0,3,2]
    x = hg_min_step64<0x4E>(x);          // quad_perm [2,3,0,1]
    x = hg_min_step64<0x141>(x);         // row_half_mirror
    x = hg_min_step64<0x140>(x);         // row_mirror: every lane of a row of 16 holds the row's minimum
    uint64_t m = ~0ull;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t y = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(x >> 32), 16 * r) << 32) | (uint32_t)__builtin_amdgcn_readlane((uint32_t)x, 16 * r);
        m = y < m ? y : m;
    }
    return m;
}
template <int CTRL>
__device__ __forceinline__ uint32_t hg_min_step32(uint32_t x) { const uint32_t y = hg_dpp<CTRL>(x); return y < x ? y : x; }
template <int ROWS = 4>
__device__ __forceinline__ uint32_t hg_wave_min32(uint32_t x) {
    x = hg_min_step32<0xB1>(x);
    x = hg_min_step32<0x4E>(x);
    x = hg_min_step32<0x141>(x);
    x = hg_min_step32<0x140>(x);
    uint32_t m = ~0u;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { const uint32_t y = (uint32_t)__builtin_amdgcn_readlane(x, 16 * r); m = y < m ? y : m; }
    return m;
}
// minima over a ROW of 16 lanes (the four DPP steps alone: every lane of the row ends up with its row's minimum)
__device__ __forceinline__ uint32_t hg_row_min32(uint32_t x) {
    x = hg_min_step32<0xB1>(x);
    x = hg_min_step32<0x4E>(x);
    x = hg_min_step32<0x141>(x);
    return hg_min_step32<0x140>(x);
}
__device__ __forceinline__ uint64_t hg_row_min64(uint64_t x) {
    x = hg_min_step64<0xB1>(x);
    x = hg_min_step64<0x4E>(x);
    x = hg_min_step64<0x141>(x);
    return hg_min_step64<0x140>(x);
}
__device__ __forceinline__ uint64_t hg_key(double x) {        // order-preserving: a < b  <=>  key(a) < key(b)   (no NaNs here)
    const uint64_t b = (uint64_t)__double_as_longlong(x + 0.0);           // (-0.0 -> +0.0: scipy compares with ==)
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ int hg_wave_min_i(int x) {          // (block-level helper of the sweep: ints >= 0 or INT_MAX)
    return (int)hg_wave_min32((uint32_t)x);
}
// ---- y_pred[:, 2] ------------------------------------------------------------------------------------------
// mode 0 (train, utils/graph.py:229-245): a true-positive det is associated through its ONE label-positive future
//   edge (more than one: status bit 1); a false positive is "associated" with itself so that it stays inactive.
// mode 1 (inference, greedy, :251-268 / :437-454): a det scored >= 0.5 looks at its future edges scored >= 0.5 that
//   lead to a det scored >= 0.5, keeps those of the NEAREST timestep (rows before the first det row after the first
//   such edge) and takes the highest score (first of equals).
// mode 1 runs a group of 16 LANES per det, a lane per incident edge (round 6): a thread per det walked its incidences through a
// chain of four dependent loads each (inc -> pos -> dst -> score: ~50 L2 round trips for a det of a KITTI window, and only Dn
// threads at work); a group issues every incidence's chain side by side and picks the first / the best edge by DPP row reductions.
// (i0, stride): a 16-aligned thread index and thread count (a block's threadIdx.x / size, or a 256-thread grid's global index).
__device__ __forceinline__ void d_track_associate(tmpnn_dgraph g, const int32_t* __restrict__ det_id,
                                                         const uint8_t* __restrict__ labels,
                                                         const float* __restrict__ score, int mode,
                                                         int32_t* __restrict__ assoc, int32_t* __restrict__ status, int i0, int stride) {
    const int N = g.N, Dn = g.meta[1];
    for (int r = i0; r < N; r += stride)
        if (g.is_edge[r]) assoc[r] = -1;
    if (mode != 0) {
        // a group of 16 lanes (one DPP row) per det: a KITTI / BDD det has ~6-20 incidences, and all dets of a window go in one round
        const int lane = i0 & 15, grp = i0 >> 4, ng = stride >> 4;
        for (int d = grp; d < Dn; d += ng) {                           // (group-uniform control flow throughout)
            const int row = g.det_row[d];
            int out = -1;
            if (score[row] >= 0.5f) {
                const int p0 = g.rowptr[d], p1 = g.rowptr[d + 1];
                // the first qualifying future edge: incidences are in ascending edge-row order, so it is the smallest row of
                // the first chunk of 16 that holds one
                int first = 0x7fffffff, first_pos = 0;
                for (int base = p0; base < p1 && first == 0x7fffffff; base += 16) {
                    const int p = base + lane;
                    int cand = 0x7fffffff, cpos = 0;
                    if (p < p1) {
                        const int key = g.inc[p];
                        if (key >= 0) {
                            cpos = g.pos[key];
                            if (score[key] >= 0.5f && score[g.dst[cpos]] >= 0.5f) cand = key;
                        }
                    }
                    first = (int)hg_row_min32((uint32_t)cand);
                    // (its index among the edge rows rides along: the lane that holds `first` hands it to the group)
                    first_pos = (int)hg_row_min32(cand == first ? (uint32_t)cpos : 0xFFFFFFFFu);
                }
                if (first != 0x7fffffff) {
                    // first det row after `first`: `first` has first - pos[first] det rows in front of it (pos = its index among
                    // the edge rows), and det_row is ascending
                    const int lo = first - first_pos;
                    const int limit = lo < Dn ? g.det_row[lo] : N;
                    // highest score among the qualifying edges in [first, limit), the first of equals: the maximum of
                    // (score bits, ~row) -- scores here are >= 0.5, their bit patterns order like the values
                    uint64_t bestk = 0;
                    for (int base = p0; base < p1; base += 16) {
                        const int p = base + lane;
                        uint64_t k = 0;
                        if (p < p1) {
                            const int key = g.inc[p];
                            if (key >= first && key < limit) {
                                const float sc = score[key];
                                if (sc >= 0.5f && score[g.dst[g.pos[key]]] >= 0.5f)
                                    k = ((uint64_t)__float_as_uint(sc) << 32) | (uint32_t)(0x7fffffff - key);
                            }
                        }
                        k = ~hg_row_min64(~k);
                        bestk = k > bestk ? k : bestk;
                    }
                    if (bestk) out = det_id[g.dst[g.pos[0x7fffffff - (int)(uint32_t)bestk]]];
                }
            }
            if (lane == 0) assoc[row] = out;
        }
        return;
    }
    for (int d = i0; d < Dn; d += stride) {
        const int row = g.det_row[d];
        const int p0 = g.rowptr[d], p1 = g.rowptr[d + 1];
        int out = -1;
        if (labels[row]) {
            int cnt = 0;
            for (int p = p0; p < p1; ++p) {
                const int key = g.inc[p];
                if (key < 0) continue;                        // past edge (this det is its later endpoint)
                if (labels[key]) { ++cnt; out = det_id[g.dst[g.pos[key]]]; }
            }
            if (cnt > 1) { atomicOr(status, 1); }
            if (cnt != 1) out = cnt == 0 ? -1 : out;
        } else {
            out = det_id[row];
        }
        assoc[row] = out;
    }
}
__global__ __launch_bounds__(256) void k_track_associate(tmpnn_dgraph g, const int32_t* __restrict__ det_id, const uint8_t* __restrict__ labels, const float* __restrict__ score, int mode, int32_t* __restrict__ assoc, int32_t* __restrict__ status) {
    d_track_associate(g, det_id, labels, score, mode, assoc, status, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}

// rows i < nr <= nc columns; cost(i, j) = C[i * sr + j * sc]; result in S.col4row[0..nr)
__device__ void hg_wave_solve(HgShared& S, const float* C, int sr, int sc, int nr, int nc, int lane) {
    double v[HG_K];
    int r4c[HG_K];
#pragma unroll
    for (int k = 0; k < HG_K; ++k) { v[k] = 0.0; r4c[k] = -1; }
    for (int i = lane; i < nr; i += 64) { S.u[i] = 0.0; S.col4row[i] = -1; }
    hg_wave_sync();
    const int KU = (nc + 63) >> 6;                        // column groups in use
    for (int cur = 0; cur < nr; ++cur) {
        double spc[HG_K];
        int path[HG_K], pos[HG_K];
        bool alive[HG_K], scj[HG_K];
#pragma unroll
        for (int k = 0; k < HG_K; ++k) {
            const int j = lane + 64 * k;
            spc[k] = __builtin_huge_val(); path[k] = -1; pos[k] = nc - 1 - j; alive[k] = j < nc; scj[k] = false;
        }
        for (int i = lane; i < nr; i += 64) S.SR[i] = 0;
        hg_wave_sync();
        int i = cur, sink = -1, num_rem = nc;
        double min_val = 0.0;
        while (sink < 0 && num_rem > 0) {
            if (lane == 0) S.SR[i] = 1;
            const double ui = S.u[i];
            double m = __builtin_huge_val();
#pragma unroll
            for (int k = 0; k < HG_K; ++k) {
                if (k < KU && alive[k]) {
                    const double c = (double)C[(size_t)i * sr + (size_t)(lane + 64 * k) * sc];
                    const double r = min_val + c - ui - v[k];
                    if (r < spc[k]) { path[k] = i; spc[k] = r; }
                    m = spc[k] < m ? spc[k] : m;
                }
            }
            // the minimum, then the scan's choice among the columns that attain it: an unassigned one if there is any (the one at
            // the HIGHEST position of `remaining`), else the one at the lowest position -- one key: unassigned first
            const uint64_t mk = hg_wave_min64(hg_key(m));
            uint32_t sel = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < HG_K; ++k)
                if (k < KU && alive[k] && hg_key(spc[k]) == mk) {
                    const uint32_t q = r4c[k] == -1 ? (uint32_t)(HG_MAX - 1 - pos[k]) : (uint32_t)(HG_MAX + pos[k]);
                    sel = q < sel ? q : sel;
                }
            sel = hg_wave_min32(sel);
            const int ipos = sel < (uint32_t)HG_MAX ? HG_MAX - 1 - (int)sel : (int)sel - HG_MAX;
            int jsel = -1, r4sel = -2;
#pragma unroll
            for (int k = 0; k < HG_K; ++k) {
                const bool mine = k < KU && alive[k] && pos[k] == ipos;
                const unsigned long long bal = __ballot(mine);
                if (bal) {                                       // (exactly one column sits at a position)
                    const int src = __ffsll((long long)bal) - 1;
                    jsel = src + 64 * k;
                    r4sel = __builtin_amdgcn_readlane(r4c[k], src);
                    m = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(spc[k]) >> 32), src) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(spc[k]), src));
                }
            }
            min_val = m;
            if (r4sel == -1) sink = jsel; else i = r4sel;
#pragma unroll
            for (int k = 0; k < HG_K; ++k) {
                if (lane + 64 * k == jsel) { scj[k] = true; alive[k] = false; }
                else if (alive[k] && pos[k] == num_rem - 1) pos[k] = ipos;
            }
            --num_rem;
        }
        if (sink < 0) { if (lane == 0) S.nr = -1; return; }        // (cannot happen with nr <= nc and finite costs: never spin)
        // dual variables (with col4row as it was BEFORE the augmentation), then the augmentation along `path`
#pragma unroll
        for (int k = 0; k < HG_K; ++k)
            if (lane + 64 * k < nc) { S.spc[lane + 64 * k] = spc[k]; S.path[lane + 64 * k] = path[k]; }
        hg_wave_sync();
        for (int i2 = lane; i2 < nr; i2 += 64)
            if (S.SR[i2]) S.u[i2] += (i2 == cur) ? min_val : (min_val - S.spc[S.col4row[i2]]);
#pragma unroll
        for (int k = 0; k < HG_K; ++k)
            if (scj[k]) v[k] -= min_val - spc[k];
        hg_wave_sync();
        int j = sink;
        for (;;) {
            const int ip = S.path[j];
#pragma unroll
            for (int k = 0; k < HG_K; ++k)
                if (lane + 64 * k == j) r4c[k] = ip;
            const int old = S.col4row[ip];
            hg_wave_sync();
            if (lane == 0) S.col4row[ip] = j;
            hg_wave_sync();
            j = old;
            if (ip == cur) break;
        }
    }
}

// The same solver for problems of <= 64 columns (every KITTI / BDD frame): lane l is column l AND row l, all state in registers, the
// only LDS traffic is the cost row of a scan; rows and columns are addressed with v_readlane / ballots instead of LDS arrays.
__device__ __forceinline__ double hg_readlane_f64(double x, int l) {
    const long long b = __double_as_longlong(x);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)b, l));
}
template <int ROWS>
__device__ void hg_wave_solve64(HgShared& S, const float* C, int sr, int sc, int nr, int nc, int lane) {
    double v = 0.0, u = 0.0;
    int r4c = -1, c4r = -1;
    for (int cur = 0; cur < nr; ++cur) {
        double spc = __builtin_huge_val();
        int path = -1, pos = nc - 1 - lane;
        bool alive = lane < nc, scj = false;
        unsigned long long sr_mask = 0;
        int i = cur, sink = -1, num_rem = nc;
        double min_val = 0.0;
        while (sink < 0 && num_rem > 0) {
            i = __builtin_amdgcn_readfirstlane(i);
            sr_mask |= 1ull << i;
            const double ui = hg_readlane_f64(u, i);
            if (alive) {
                const double c = (double)C[(size_t)i * sr + (size_t)lane * sc];
                const double r = min_val + c - ui - v;
                if (r < spc) { path = i; spc = r; }
            }
            const uint64_t mk = hg_wave_min64<ROWS>(alive ? hg_key(spc) : ~0ull);
            uint32_t sel = 0xFFFFFFFFu;
            if (alive && hg_key(spc) == mk) sel = r4c == -1 ? (uint32_t)(HG_MAX - 1 - pos) : (uint32_t)(HG_MAX + pos);
            sel = hg_wave_min32<ROWS>(sel);
            const int ipos = sel < (uint32_t)HG_MAX ? HG_MAX - 1 - (int)sel : (int)sel - HG_MAX;
            const unsigned long long bal = __ballot(alive && pos == ipos);
            const int jsel = __ffsll((long long)bal) - 1;           // (exactly one column sits at a position)
            const int r4sel = __builtin_amdgcn_readlane(r4c, jsel);
            min_val = hg_readlane_f64(spc, jsel);
            if (r4sel == -1) sink = jsel; else i = r4sel;
            if (lane == jsel) { scj = true; alive = false; }
            else if (alive && pos == num_rem - 1) pos = ipos;
            --num_rem;
        }
        if (sink < 0) { if (lane == 0) S.nr = -1; return; }          // (cannot happen with nr <= nc and finite costs: never spin)
        // dual variables (with the assignment as it was BEFORE the augmentation): row l needs the reduced cost of ITS column
        const double spc_mine = __shfl(spc, c4r < 0 ? 0 : c4r);
        if ((sr_mask >> lane) & 1ull) u += (lane == cur) ? min_val : (min_val - spc_mine);
        if (scj) v -= min_val - spc;
        int j = sink;
        for (;;) {
            const int ip = __builtin_amdgcn_readlane(path, j);
            if (lane == j) r4c = ip;
            const int old = __builtin_amdgcn_readlane(c4r, ip);
            if (lane == ip) c4r = j;
            j = old;
            if (ip == cur) break;
        }
    }
    if (lane < nr) S.col4row[lane] = c4r;
}

// the whole sweep, one block of TK_THREADS threads, graphs of <= FIN_LDS_DETS rows with <= HG_MAX dets per problem; status bit 2:
// a problem exceeded HG_MAX or the scratch (the associations are then incomplete: the host falls back to its own matching).
// What a timestep's passes need of an edge (the timestep it leads into, its endpoints' det indices, its cost) and of a det (its
// timestep) is read ONCE into registers (four edges / dets per thread): a pass is then LDS and register work, not a chain of
// dependent global loads per timestep; the result is kept per det index in LDS and written out once at the end.
#define HG_STAMP(i) do { } while (0)
__device__ void d_track_hungarian(tmpnn_dgraph g, const int32_t* __restrict__ ts, const int32_t* __restrict__ det_id,
                                  const float* __restrict__ score, int32_t* __restrict__ assoc, int32_t* __restrict__ status,
                                  float* __restrict__ cost_ws, int cost_ws_floats, const int* remap = nullptr, int n_out = 0,
                                  HgMemo* memo = nullptr, int memo_mode = 0 /* 1: record, 2: look up */) {
    // remap (k_track_retire, after the deletion): new index of a row that stays, -1 for a deleted one -- the sweep then runs over
    // the rows that stay, as the reference's NEXT update_graph runs it on the compacted graph (deleted dets and the edges that go
    // with them take no part; det ids and the order of rows are what they will be), and writes assoc [n_out] in the new numbering
    __shared__ HgShared S;
    constexpr int PER = 4096 / TK_THREADS;
    const int tid = threadIdx.x, N = g.N, E = g.meta[0], Dn = g.meta[1];
    for (int r = tid; r < (remap ? n_out : N); r += TK_THREADS) assoc[r] = -1;
    int e_t[PER], e_sp[PER], e_dp[PER], d_t[PER];
    float e_c[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = tid + TK_THREADS * k, d = tid + TK_THREADS * k;
        e_t[k] = 0x7fffffff; e_sp[k] = 0; e_dp[k] = 0; e_c[k] = 0.f; d_t[k] = -0x7fffffff;
        if (e < E) {
            const int er = g.edge_row[e];
            e_t[k] = ts[g.dst[e]]; e_sp[k] = g.src_pos[e]; e_dp[k] = g.dst_pos[e]; e_c[k] = 1.0f - score[er];
            if (remap && remap[er] < 0) e_t[k] = 0x7fffffff;
        }
        if (d < Dn) {
            const int dr = g.det_row[d];
            d_t[k] = ts[dr]; S.ad[d] = -1;
            if (remap && remap[dr] < 0) d_t[k] = -0x7fffffff;
        }
    }
    __syncthreads();
    if (E == 0 || Dn == 0) return;
    int t_done = -0x7fffffff;
    // (an iteration starts with no timestep chosen, no det range and no row flags: set here, and again inside the loop once
    //  everyone has read them -- one barrier less than doing it at the top)
    if (tid == 0) { S.tnext = 0x7fffffff; S.d0 = 0x7fffffff; S.d1 = -1; if (memo_mode == 1) memo->n = 0; }
    for (int d = tid; d < Dn; d += TK_THREADS) S.flag[d] = 0;
    __syncthreads();
    for (;;) {
        HG_STAMP(8);
        // the next timestep that edges lead into; its dets are a contiguous run of the det list (rows are in time order)
        {
            int m = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < PER; ++k) if (e_t[k] > t_done && e_t[k] < m) m = e_t[k];
            m = hg_wave_min_i(m);
            if ((tid & 63) == 0 && m != 0x7fffffff) atomicMin(&S.tnext, m);
        }
        __syncthreads();
        const int t = S.tnext;
        if (t == 0x7fffffff) break;
        t_done = t;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (d_t[k] == t) { atomicMin(&S.d0, tid + TK_THREADS * k); atomicMax(&S.d1, tid + TK_THREADS * k); }
            if (e_t[k] == t && S.ad[e_sp[k]] < 0) S.flag[e_sp[k]] = 1;     // rows: unassociated src dets of the edges into t
        }
        __syncthreads();
        const int d0 = S.d0, nc = S.d1 - S.d0 + 1;
        // the rows in ascending det index: ONE wave ranks the flagged dets by ballots (a block-wide scan is three barriers a pass)
        if (tid < 64) {
            int run = 0;
            for (int base = 0; base < Dn; base += 64) {
                const int d = base + tid;
                const bool f = d < Dn && S.flag[d];
                const unsigned long long bal = __ballot(f);
                if (f) {
                    const int k = run + __popcll(bal & ((1ull << tid) - 1ull));
                    S.ridx[d] = (short)min(k, 32767);
                    if (k < HG_MAX) S.rowlist[k] = d;
                }
                run += __popcll(bal);
            }
            if (tid == 0) S.nrows = run;
        }
        __syncthreads();
        const int nr = S.nrows;
        if (tid == 0) { S.tnext = 0x7fffffff; S.d0 = 0x7fffffff; S.d1 = -1; S.nr = nr; }
        for (int d = tid; d < Dn; d += TK_THREADS) S.flag[d] = 0;
        const bool lds_cost = (long)nr * nc <= HG_LDS_COST;
        if (nr == 0 || nr > HG_MAX || nc > HG_MAX || (!lds_cost && (cost_ws == nullptr || (long)nr * nc > cost_ws_floats))) {
            if (nr != 0 && tid == 0) atomicOr(status, 2);
            __syncthreads();
            continue;
        }
        if (memo_mode == 2) {                                 // (block-uniform throughout)
            int hit = -1;
            for (int q = 0; q < memo->n; ++q) if (memo->t[q] == t) { hit = q; break; }
            if (hit >= 0 && memo->nr[hit] == nr && memo->nc[hit] == nc && memo->d0[hit] == d0) {
                int diff = 0;
                for (int i = tid; i < nr; i += TK_THREADS) diff |= memo->rows[hit][i] != (short)S.rowlist[i];
                if (!__syncthreads_or(diff)) {
                    for (int i = tid; i < nr; i += TK_THREADS) {
                        const int a = memo->assign[hit][i];
                        if (a >= 0) S.ad[S.rowlist[i]] = (short)a;
                    }
                    __syncthreads();
                    continue;
                }
            }
        }
        const int rec = (memo_mode == 1 && memo->n < HG_MEMO_T && nr <= HG_MEMO_R) ? memo->n : -1;
        if (rec >= 0)
            for (int i = tid; i < nr; i += TK_THREADS) { memo->rows[rec][i] = (short)S.rowlist[i]; memo->assign[rec][i] = -1; }
        float* C = lds_cost ? S.cost : cost_ws;
        for (int x = tid; x < nr * nc; x += TK_THREADS) C[x] = 100.0f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (e_t[k] == t && S.ad[e_sp[k]] < 0) C[(int)S.ridx[e_sp[k]] * nc + (e_dp[k] - d0)] = e_c[k];
        __threadfence_block();
        __syncthreads();
        const bool tr = nc < nr;                              // (scipy transposes a tall matrix)
        HG_STAMP(5);
        if (tid < 64) {
            if (max(nr, nc) <= 64) {
                // (the solver's columns: the longer side; lanes beyond them never hold a live key)
                if (max(nr, nc) <= 16) {
                    if (tr) hg_wave_solve64<1>(S, C, 1, nc, nc, nr, tid);
                    else hg_wave_solve64<1>(S, C, nc, 1, nr, nc, tid);
                } else {
                    if (tr) hg_wave_solve64<4>(S, C, 1, nc, nc, nr, tid);
                    else hg_wave_solve64<4>(S, C, nc, 1, nr, nc, tid);
                }
            } else {
                if (tr) hg_wave_solve(S, C, 1, nc, nc, nr, tid);
                else hg_wave_solve(S, C, nc, 1, nr, nc, tid);
            }
        }
        HG_STAMP(6);
        __threadfence_block();
        __syncthreads();
        HG_STAMP(7);
        if (S.nr < 0) { if (tid == 0) atomicOr(status, 2); __syncthreads(); continue; }
        const int na = tr ? nc : nr;
        for (int i = tid; i < na; i += TK_THREADS) {
            const int j = S.col4row[i];
            const int pr = tr ? j : i, cu = tr ? i : j;          // (row of the problem = prev det, column = det of timestep t)
            if (C[pr * nc + cu] <= 0.5f) {
                S.ad[S.rowlist[pr]] = (short)(d0 + cu);
                if (rec >= 0) memo->assign[rec][pr] = (short)(d0 + cu);
            }
        }
        if (rec >= 0 && tid == 0) { memo->t[rec] = t; memo->nr[rec] = nr; memo->nc[rec] = nc; memo->d0[rec] = d0; memo->n = rec + 1; }
        __syncthreads();
    }
    // y_pred[:, 2] of the associated dets: the det id of the partner
    for (int d = tid; d < Dn; d += TK_THREADS) {
        const int pd = S.ad[d];
        if (pd >= 0) { const int row = g.det_row[d]; assoc[remap ? remap[row] : row] = det_id[g.det_row[pd]]; }
    }
}

// ---- active set (utils/graph.py:270-278), ascending rows ----------------------------------------------------------
// train: det rows not yet associated, or of the last timestep before t.   inference: unassociated dets scored >= 0.5.
// n_dev (or NULL): the row count read on the device instead (the rows a deletion just compacted: the host has not seen it yet)
__device__ __forceinline__ void d_track_active(int N, const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ ts,
                                                             const int32_t* __restrict__ assoc,
                                                             const float* __restrict__ score, int mode, int t,
                                                             int32_t* __restrict__ active, int32_t* __restrict__ count) {
    __shared__ int s_wave[TK_THREADS / 64 + 1];
    if (n_dev) N = n_dev[0];
    __shared__ int s_tprev;
    const int tid = threadIdx.x;
    if (tid == 0) s_tprev = -2147483647;
    __syncthreads();
    if (mode == 0) {
        int m = -2147483647;
        for (int r = tid; r < N; r += TK_THREADS) { const int v = ts[r]; if (v < t && v > m) m = v; }
        atomicMax(&s_tprev, m);
        __syncthreads();
    }
    const int tprev = s_tprev;
    const int IT = (N + TK_THREADS - 1) / TK_THREADS;
    const int r0 = tid * IT, r1 = min(N, r0 + IT);
    // (a row's three operands are requested together: behind short-circuit tests they were a chain of dependent loads; the
    //  verdicts are kept in a mask -- IT <= 32 at TMPNN_TRACK_MAX_ROWS -- instead of being derived twice)
    int mine = 0;
    uint32_t amask = 0;
    for (int r = r0; r < r1; ++r) {
        const int v = ts[r], a = assoc[r];
        const float sc = (mode == 0) ? 1.0f : score[r];
        const bool on = v != -1 && (mode == 0 ? (a == -1 || v == tprev) : (a == -1 && sc >= 0.5f));
        amask |= (on ? 1u : 0u) << (r - r0);
        mine += on ? 1 : 0;
    }
    int total;
    int p = tk_block_scan(mine, s_wave, &total);
    for (int r = r0; r < r1; ++r)
        if ((amask >> (r - r0)) & 1u) active[p++] = r;
    if (tid == 0) count[0] = total;
}
__global__ __launch_bounds__(TK_THREADS) void k_track_active(int N, const int32_t* __restrict__ n_dev, const int32_t* __restrict__ ts, const int32_t* __restrict__ assoc, const float* __restrict__ score, int mode, int t, int32_t* __restrict__ active, int32_t* __restrict__ count) {
    d_track_active(N, n_dev, ts, assoc, score, mode, t, active, count);
}

// ---- append the block of timestep t (utils/graph.py:283-325) ------------------------------------------------------
// rows [N, N + A*D): edge (a, j) at N + a*D + j with src = active[a], dst = N + A*D + j ; rows [N + A*D, N + A*D + D): dets.
// labels (training): det j is positive iff its track id >= 0; edge (a, j) iff both belong to the same track.
__device__ __forceinline__ void d_track_append(int N, int A, int D, const int32_t* __restrict__ active,
                                               const int32_t* __restrict__ new_ids, int t,
                                               const int32_t* __restrict__ track /* [ND] or NULL */,
                                               int32_t* __restrict__ ts, int32_t* __restrict__ det_id,
                                               int32_t* __restrict__ assoc, uint8_t* __restrict__ is_edge,
                                               int32_t* __restrict__ row_src, int32_t* __restrict__ row_dst,
                                               uint8_t* __restrict__ labels,
                                               const float* __restrict__ X /* [ND][ld_x] or NULL */, int ld_x, int F,
                                               float* __restrict__ feats /* [n][ld_f]: zeros on edge rows */, int ld_f,
                                               const long i0, const long stride) {
    const int n = A * D + D;
    if (feats) {          // the new rows' features (utils/graph.py:291-293, 318): zeros on the edge rows, X[id] on the det rows
        const long total = (long)n * F;
        for (long i = i0; i < total; i += stride) {
            const int r = (int)(i / F), c = (int)(i % F);
            feats[(size_t)r * ld_f + c] = r < A * D ? 0.f : X[(size_t)new_ids[r - A * D] * ld_x + c];
        }
    }
    for (int i = (int)i0; i < n; i += (int)stride) {
        const int r = N + i;
        assoc[r] = -1;
        if (i < A * D) {
            const int a = i / D, j = i % D;
            const int s = active[a];
            ts[r] = -1; det_id[r] = -1; is_edge[r] = 1;
            row_src[r] = s; row_dst[r] = N + A * D + j;
            if (labels) {
                const int ta = track ? track[det_id[s]] : -1, tj = track ? track[new_ids[j]] : -1;
                labels[r] = (tj != -1 && ta == tj) ? 1 : 0;
            }
        } else {
            const int j = i - A * D;
            ts[r] = t; det_id[r] = new_ids[j]; is_edge[r] = 0;
            row_src[r] = -1; row_dst[r] = -1;
            if (labels) labels[r] = (track && track[new_ids[j]] >= 0) ? 1 : 0;
        }
    }
}
__global__ __launch_bounds__(256) void k_track_append(int N, int A, int D, const int32_t* __restrict__ active, const int32_t* __restrict__ new_ids, int t, const int32_t* __restrict__ track, int32_t* __restrict__ ts, int32_t* __restrict__ det_id, int32_t* __restrict__ assoc, uint8_t* __restrict__ is_edge, int32_t* __restrict__ row_src, int32_t* __restrict__ row_dst, uint8_t* __restrict__ labels, const float* __restrict__ X, int ld_x, int F, float* __restrict__ feats, int ld_f) {
    d_track_append(N, A, D, active, new_ids, t, track, ts, det_id, assoc, is_edge, row_src, row_dst, labels, X, ld_x, F, feats, ld_f,
                   (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}

// ---- update_graph's second half AND the model call's input transform in ONE launch (inference, LDS-sized graphs) ----------------
// block 0: the block of timestep t appended behind row N, then the grown graph's index form (graphconv_dev.h), as
// tmpnn_track_extend does in two launches; blocks 1 .. G: the input transform of the D new dets (small_bn_dev.h; eval mode:
// running statistics) straight from X[new_ids[j]] into h[N + A*D + j], zeros on the A*D new edge rows -- what the first launch
// of tmpnn_mp_iter_fwd would do behind the append.  Neither half reads what the other writes: they run side by side, and a greedy
// timestep has two launches (~6 us of launch gaps) and the transform's ~14 us less on its critical path.
struct ExtendArgs {
    int N, A, D, t;
    const int32_t* active; const int32_t* new_ids; const int32_t* track;
    tmpnn_track_rows r;
};
// counts (or NULL): the `small` words of the decode launch in front of this one on the stream -- N = counts[0] (the rows it kept)
// and A = counts[3] (the active set it derived) are then read HERE, and e.N / e.A / the graph's N / the transform's row counts
// are only what the host sized the buffers with (upper bounds): the host enqueues this launch without waiting for that decode.
// gat_* (with counts): the state rows the previous decode kept have NOT been moved yet (tmpnn_track_retire with h_new = NULL): the
// blocks behind the transform's do it here -- h[q][0:G*H] = gat_src[gat_keep[q]][0:G*H] for q < counts[0] -- beside the append and
// the transform, neither of which reads or writes those rows (one launch and its ~5 us less per timestep).
template <int H>
__global__ __launch_bounds__(GC_THREADS) void k_track_extend_tf(ExtendArgs e, tmpnn_dgraph g, BnFwdArgs b, BnSrcBlock src,
                                                                const int32_t* __restrict__ counts,
                                                                const float* __restrict__ gat_src, int gat_ld,
                                                                const int32_t* __restrict__ gat_keep) {
    const int N = counts ? counts[0] : e.N, A = counts ? counts[3] : e.A;
    const int n = A * e.D + e.D, Nt = N + n;
    if ((int)blockIdx.x > b.P.G) {
        const int GHc = b.P.G * H, lpr = GHc / 4;
        const long total = (long)N * lpr, stride = (long)(gridDim.x - 1 - b.P.G) * GC_THREADS;
        for (long i = (long)(blockIdx.x - 1 - b.P.G) * GC_THREADS + threadIdx.x; i < total; i += stride) {
            const int q = (int)(i / lpr), c = (int)(i % lpr) * 4;
            *reinterpret_cast<float4*>(b.h + (size_t)q * GHc + c) = *reinterpret_cast<const float4*>(gat_src + (size_t)gat_keep[q] * gat_ld + c);
        }
        return;
    }
    if (blockIdx.x == 0) {
        d_track_append(N, A, e.D, e.active, e.new_ids, e.t, e.track, e.r.ts, e.r.det_id, e.r.assoc, e.r.is_edge, e.r.src,
                       e.r.dst, e.r.labels, nullptr, 0, 0, nullptr, 0, (long)threadIdx.x, (long)GC_THREADS);
        __threadfence_block();
        __syncthreads();
        d_graph_from_coo<true, false>(Nt, nullptr, nullptr, 0L, nullptr, nullptr, 0L, e.r.is_edge, e.r.src, e.r.dst, g, nullptr);
    } else {
        const int gi = (int)blockIdx.x - 1;
        if (threadIdx.x >= 256) {
            // the transform is written for 256 threads; the other 768 write the zeros of the A*D new edge rows of this group's
            // columns (track_mpnn.py:61; ~50 stores per thread of the transform at a BDD-sized block) and leave: ended waves drop
            // out of the block's barriers
            const int GH = b.P.G * H, lpr = H / 4;
            const long total = (long)A * e.D * lpr;
            for (long i = threadIdx.x - 256; i < total; i += GC_THREADS - 256)
                *reinterpret_cast<float4*>(b.h + (size_t)(N + i / lpr) * GH + gi * H + 4 * (i % lpr)) = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        d_small_bn_fwd<H>(b, gi, BnSrcBlock{A * e.D, src.ids, src.X, src.ld_x}, Nt, n);
    }
}

// ---- rows decode_tracks deletes (utils/graph.py:492-512) + the compacted row form ------------------------------
// max_id = 1 + the last det row before t_upto.  Deleted: every row < max_id except the dets RETAINED for later
// association (unassociated, scored >= 0.5, not older than t_upto - ret_win); and every edge row >= max_id that
// starts at a deleted det.  Kept rows are renumbered in order; row_src / row_dst follow.
__device__ __forceinline__ void d_track_delete(int N, const int32_t* __restrict__ ts,
                                                             const int32_t* __restrict__ det_id,
                                                             const int32_t* __restrict__ assoc,
                                                             const float* __restrict__ score,
                                                             const uint8_t* __restrict__ is_edge,
                                                             const int32_t* __restrict__ row_src,
                                                             const int32_t* __restrict__ row_dst,
                                                             const uint8_t* __restrict__ labels, int t_upto, int ret_win,
                                                             int32_t* __restrict__ keep /* [N]: kept rows, ascending */,
                                                             int32_t* __restrict__ count,
                                                             int32_t* __restrict__ o_ts, int32_t* __restrict__ o_det_id,
                                                             int32_t* __restrict__ o_assoc, uint8_t* __restrict__ o_is_edge,
                                                             int32_t* __restrict__ o_src, int32_t* __restrict__ o_dst,
                                                             uint8_t* __restrict__ o_labels) {
    extern __shared__ int s_new[];                         // [N] new index of a kept row, -1 if deleted
    __shared__ int s_wave[TK_THREADS / 64 + 1];
    __shared__ int s_max, s_dets;
    const int tid = threadIdx.x;
    if (tid == 0) { s_max = 0; s_dets = 0; }
    __syncthreads();
    if (N <= TK_THREADS) {
        // a row per thread (the windows of the reference's loops): everything the phase reads of its row is requested up front and
        // kept -- the general form below reads the rows three times (max_id, the verdicts, the copy), a round trip each (round 6)
        const int r = tid;
        const bool in = r < N;
        const int rc = in ? r : 0;
        const int t = ts[rc], di = det_id[rc], a = assoc[rc], sr = row_src[rc], dr = row_dst[rc];
        const float sc = score[rc];
        const bool e = is_edge[rc] != 0;
        const uint8_t lb = labels ? labels[rc] : (uint8_t)0;
        atomicMax(&s_max, (in && t != -1 && t < t_upto) ? r + 1 : 0);
        __syncthreads();
        const int max_id = s_max;
        bool k = false;
        if (in) {
            if (r < max_id) k = !e && a == -1 && sc >= 0.5f && t >= t_upto - ret_win;
            else if (!e || sr >= max_id) k = true;
            else {                                              // an edge at or after max_id: dropped with its start det
                const int a2 = assoc[sr], t2 = ts[sr];
                const float s2 = score[sr];
                k = a2 == -1 && s2 >= 0.5f && t2 >= t_upto - ret_win;
            }
        }
        if (k && !e) atomicAdd(&s_dets, 1);
        int total;
        const int p = tk_block_scan(k ? 1 : 0, s_wave, &total);
        if (in) s_new[r] = k ? p : -1;
        if (k) keep[p] = r;
        __syncthreads();
        if (k) {
            o_ts[p] = t; o_det_id[p] = di; o_assoc[p] = a; o_is_edge[p] = e ? 1 : 0;
            if (o_labels) o_labels[p] = lb;
            o_src[p] = e ? s_new[sr] : -1;
            o_dst[p] = e ? s_new[dr] : -1;
        }
        if (tid == 0) { count[0] = total; count[2] = s_dets; }
        return;
    }
    int m = 0;
    for (int r = tid; r < N; r += TK_THREADS) { const int v = ts[r]; if (v != -1 && v < t_upto) m = max(m, r + 1); }
    atomicMax(&s_max, m);
    __syncthreads();
    const int max_id = s_max;
    // kept: below max_id the RETAINED dets only (unassociated, scored >= 0.5, not older than t_upto - ret_win); at or after it every
    // det, and every edge whose start det is not a dropped one
    const int IT = (N + TK_THREADS - 1) / TK_THREADS;
    const int r0 = tid * IT, r1 = min(N, r0 + IT);
    int mine = 0, mine_dets = 0;
    uint32_t kmask = 0;                                     // the verdicts of this thread's rows (IT <= 32 at TMPNN_TRACK_MAX_ROWS):
    for (int r = r0; r < r1; ++r) {                         // evaluated once, a row's operands requested together: behind the
        const bool e = is_edge[r] != 0;                     // short-circuit tests of kept() they were a chain of dependent loads
        const int a = assoc[r], t = ts[r], sr = row_src[r];
        const float sc = score[r];
        bool k;
        if (r < max_id) k = !e && a == -1 && sc >= 0.5f && t >= t_upto - ret_win;
        else if (!e) k = true;
        else if (sr >= max_id) k = true;
        else {                                              // an edge at or after max_id: dropped with its start det
            const int a2 = assoc[sr], t2 = ts[sr];
            const float s2 = score[sr];
            k = a2 == -1 && s2 >= 0.5f && t2 >= t_upto - ret_win;
        }
        kmask |= (k ? 1u : 0u) << (r - r0);
        mine += k ? 1 : 0;
        mine_dets += (k && !e) ? 1 : 0;
    }
    if (mine_dets) atomicAdd(&s_dets, mine_dets);
    int total;
    int p = tk_block_scan(mine, s_wave, &total);
    for (int r = r0; r < r1; ++r) {
        if ((kmask >> (r - r0)) & 1u) { s_new[r] = p; keep[p] = r; ++p; }
        else s_new[r] = -1;
    }
    __syncthreads();
    for (int r = tid; r < N; r += TK_THREADS) {
        const int q = s_new[r];
        if (q < 0) continue;
        o_ts[q] = ts[r]; o_det_id[q] = det_id[r]; o_assoc[q] = assoc[r]; o_is_edge[q] = is_edge[r];
        if (o_labels) o_labels[q] = labels ? labels[r] : 0;
        if (is_edge[r]) { o_src[q] = s_new[row_src[r]]; o_dst[q] = s_new[row_dst[r]]; }
        else { o_src[q] = -1; o_dst[q] = -1; }
    }
    if (tid == 0) { count[0] = total; count[2] = s_dets; }        // kept rows; kept DET rows (the host's E / Dn bookkeeping)
}
__global__ __launch_bounds__(TK_THREADS) void k_track_delete(int N, const int32_t* __restrict__ ts, const int32_t* __restrict__ det_id, const int32_t* __restrict__ assoc, const float* __restrict__ score, const uint8_t* __restrict__ is_edge, const int32_t* __restrict__ row_src, const int32_t* __restrict__ row_dst, const uint8_t* __restrict__ labels, int t_upto, int ret_win, int32_t* __restrict__ keep /* [N]: kept rows, ascending */, int32_t* __restrict__ count, int32_t* __restrict__ o_ts, int32_t* __restrict__ o_det_id, int32_t* __restrict__ o_assoc, uint8_t* __restrict__ o_is_edge, int32_t* __restrict__ o_src, int32_t* __restrict__ o_dst, uint8_t* __restrict__ o_labels) {
    d_track_delete(N, ts, det_id, assoc, score, is_edge, row_src, row_dst, labels, t_upto, ret_win, keep, count, o_ts, o_det_id, o_assoc, o_is_edge, o_src, o_dst, o_labels);
}

// out[q, :] = in[keep[q], :] for q < count (count read on the device: the launch is sized for the worst case)
__global__ __launch_bounds__(256) void k_track_gather(const float* __restrict__ in, int ld_in, int W,
                                                      const int32_t* __restrict__ keep, const int32_t* __restrict__ count,
                                                      float* __restrict__ out, int ld_out) {
    const int n = count[0];
    const int lpr = (W + 3) / 4;
    const long total = (long)n * lpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i / lpr), c = (int)(i % lpr) * 4;
        const float* src = in + (size_t)keep[q] * ld_in + c;
        float* dst = out + (size_t)q * ld_out + c;
        if (c + 4 <= W && ((ld_in | ld_out) & 3) == 0) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else for (int k = 0; k < 4 && c + k < W; ++k) dst[k] = src[k];
    }
}


// the state rows AND the scores in one launch: out_h[q, :] = h[keep[q], :], out_s[q] = score[keep[q]]
__device__ __forceinline__ void d_track_gather2(const float* __restrict__ h, int ld_h, int W,
                                                       const float* __restrict__ score, const int32_t* __restrict__ keep,
                                                       const int32_t* __restrict__ count, float* __restrict__ out_h,
                                                       int ld_out, float* __restrict__ out_s, long i0, long stride) {
    const int n = count[0];
    const int lpr = (W + 3) / 4 + 1;                      // the last slot of a row moves its score
    const long total = (long)n * lpr;
    for (long i = i0; i < total; i += stride) {
        const int q = (int)(i / lpr), k = (int)(i % lpr);
        const int r = keep[q];
        if (k == lpr - 1) { out_s[q] = score[r]; continue; }
        const int c = k * 4;
        const float* src = h + (size_t)r * ld_h + c;
        float* dst = out_h + (size_t)q * ld_out + c;
        if (c + 4 <= W && ((ld_h | ld_out) & 3) == 0) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else for (int j = 0; j < 4 && c + j < W; ++j) dst[j] = src[j];
    }
}
__global__ __launch_bounds__(256) void k_track_gather2(const float* __restrict__ h, int ld_h, int W, const float* __restrict__ score, const int32_t* __restrict__ keep, const int32_t* __restrict__ count, float* __restrict__ out_h, int ld_out, float* __restrict__ out_s) {
    d_track_gather2(h, ld_h, W, score, keep, count, out_h, ld_out, out_s, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
}


// ---- track finalisation (utils/graph.py:456-490): y_out[:, 1] from the association links, on the device ------------------
// The reference walks, for every detection of the sequence in id order, the linked list det -> y_pred[det, 2] -> ... from
// each not-yet-visited det before t_upto scored >= 0.5, writing the walk's track id (the start's existing id, else the
// next free one) over EVERYTHING on its path -- also over dets an earlier walk of the same call has labelled: the last
// walk through a det wins.  Links lead to later detections, and the graph's det rows are in time order, so one forward
// pass over the window's dets computes the same thing: a det is "reached" when an on-path predecessor's walk continues
// into it, it starts a walk when it is eligible and not reached, and its label is that of the LATEST start among the
// walks through it.  Dn is tens to hundreds: the pass is one thread over LDS-resident arrays (all other steps -- the
// id -> position table, the scan for the next free id, the label write-back -- are parallel), and the host is not
// involved: y_out stays on the device for the whole sequence.
// Where the window's det ids do NOT ascend with its rows (a sequence whose detections are listed in no particular order) the
// order of the starts is the det-id order, a later start may begin in the middle of an earlier walk's path, and the numbering
// depends on it: the kernel notices (one parallel comparison of neighbours) and runs the reference's loop literally -- the
// window's dets ranked by id (a parallel count), then one thread starting the walks in that order over the same LDS arrays.
static constexpr int FIN_LDS_DETS = 4096;

__device__ __forceinline__ void d_track_finalize(tmpnn_dgraph g, const int32_t* __restrict__ ts,
                                                               const int32_t* __restrict__ det_id,
                                                               const int32_t* __restrict__ assoc,
                                                               const float* __restrict__ score, int t_upto,
                                                               int32_t* __restrict__ y_track, int ND,
                                                               int32_t* __restrict__ pos_of_det, int32_t* __restrict__ ws) {
    __shared__ int s_next[FIN_LDS_DETS], s_best[FIN_LDS_DETS], s_tid[FIN_LDS_DETS], s_old[FIN_LDS_DETS];
    __shared__ unsigned char s_flag[FIN_LDS_DETS];         // bit 0 eligible start, bit 1 reached, bit 2 on a path
    __shared__ int s_order[FIN_LDS_DETS];                  // (unsorted windows only) position of the det with the k-th smallest id
    __shared__ int s_max, s_unsorted;
    const int tid = threadIdx.x;
    const int Dn = g.meta[1];
    // arrays of the pass: LDS up to FIN_LDS_DETS dets, the caller's scratch beyond (same code through generic pointers)
    int* nextk = Dn <= FIN_LDS_DETS ? s_next : ws;
    int* best = Dn <= FIN_LDS_DETS ? s_best : ws + Dn;
    int* tidv = Dn <= FIN_LDS_DETS ? s_tid : ws + 2 * (size_t)Dn;
    int* oldv = Dn <= FIN_LDS_DETS ? s_old : ws + 3 * (size_t)Dn;
    unsigned char* flag = Dn <= FIN_LDS_DETS ? s_flag : reinterpret_cast<unsigned char*>(ws + 4 * (size_t)Dn);
    int* order = Dn <= FIN_LDS_DETS ? s_order : ws + 5 * (size_t)Dn;
    if (tid == 0) { s_max = -1; s_unsorted = 0; }
    __syncthreads();
    {   // next free track id = max(y_out[:, 1]) + 1 (utils/graph.py:457)
        int m = -1;
        for (int i = tid; i < ND; i += TK_THREADS) m = max(m, y_track[i]);
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
        if ((tid & 63) == 0) atomicMax(&s_max, m);
    }
    for (int k = tid; k < Dn; k += TK_THREADS) pos_of_det[det_id[g.det_row[k]]] = k;
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < Dn; k += TK_THREADS) {
        const int r = g.det_row[k];
        const int a = assoc[r];
        int nk = -1;
        if (a >= 0) {
            const int ka = pos_of_det[a];
            // the walk stops between two detections that are both at or after t_upto (utils/graph.py:486-487)
            if (!(ts[r] >= t_upto && ts[g.det_row[ka]] >= t_upto)) nk = ka;
        }
        nextk[k] = nk;
        best[k] = -1;
        oldv[k] = y_track[det_id[r]];
        flag[k] = (ts[r] < t_upto && score[r] >= 0.5f) ? 1 : 0;
        if (k + 1 < Dn && det_id[r] > det_id[g.det_row[k + 1]]) s_unsorted = 1;
    }
    __syncthreads();
    if (s_unsorted) {                                           // block-uniform
        // the reference's loop as written (utils/graph.py:458-490): starts in det-id order, labels read and written live
        for (int k = tid; k < Dn; k += TK_THREADS) {
            const int id = det_id[g.det_row[k]];
            int rank = 0;
            for (int j = 0; j < Dn; ++j) rank += det_id[g.det_row[j]] < id ? 1 : 0;
            order[rank] = k;
        }
        __syncthreads();
        if (tid == 0) {
            int next_id = s_max + 1;
            for (int i = 0; i < Dn; ++i) {
                int k = order[i];
                const unsigned char f = flag[k];
                if (!(f & 1)) { flag[k] = f | 2; continue; }    // not a start (at / after t_upto, or a false positive): visited
                if (f & 2) continue;                            // an earlier walk came through
                const int cur = oldv[k] != -1 ? oldv[k] : next_id++;
                for (int steps = 0; steps < Dn; ++steps) {      // (a det links to a LATER det: no cycles; bounded anyway)
                    flag[k] |= 2 | 4;
                    oldv[k] = cur;
                    const int nk = nextk[k];
                    if (nk < 0) break;
                    k = nk;
                }
            }
        }
        __syncthreads();
        for (int k = tid; k < Dn; k += TK_THREADS)
            if (flag[k] & 4) y_track[det_id[g.det_row[k]]] = oldv[k];
        return;
    }
    if (Dn <= FIN_LDS_DETS) {
        // The same pass WITHOUT its serial loop (round 6: one thread over LDS arrays took 0.15 us per det -- 5 of the 8 us of this
        // function for a KITTI window, 9 of 12 for a BDD one).  A det's link leads to a later det, so the pass is three closures
        // along the links, each by pointer doubling in ONE wave (lanes over the dets in chunks of 64, no workgroup barrier; a
        // round's reads may or may not see what the same round wrote -- every mark is monotone, so both are right):
        //   on a path  = eligible, or linked from a det on a path          (bit 2 of the flag, value 4)
        //   reached    = linked from a det on a path (bit 1)  ->  a START is an eligible det that is not reached
        //   label      = that of the LATEST start (largest position) whose walk comes through: max along the links
        // and the new ids are handed out in the order of the starts: a count over the positions before.
        if (tid < 64) {
            const int lane = tid;
            int* const jump = s_order;                                  // (free here: the det-id order is the other branch's)
            for (int k = lane; k < Dn; k += 64) {
                const int nk = s_next[k];
                jump[k] = nk > k ? nk : -1;
                if (s_flag[k] & 1) s_flag[k] |= 4;
            }
            hg_wave_sync();
            for (;;) {
                int more = 0;
                for (int k = lane; k < Dn; k += 64) {
                    const int j = jump[k];
                    if (j >= 0) {
                        if (s_flag[k] & 4) s_flag[j] |= 4;
                        const int jj = jump[j];
                        jump[k] = jj;
                        more |= jj >= 0 ? 1 : 0;
                    }
                }
                hg_wave_sync();
                if (__ballot(more != 0) == 0) break;
            }
            for (int k = lane; k < Dn; k += 64) {
                const int nk = s_next[k];
                if (nk > k && (s_flag[k] & 4)) s_flag[nk] |= 2;
            }
            hg_wave_sync();
            int run = s_max + 1;                                        // next free track id
            for (int base = 0; base < Dn; base += 64) {
                const int k = base + lane;
                const unsigned char f = k < Dn ? s_flag[k] : (unsigned char)0;
                const bool start = (f & 1) && !(f & 2);
                const bool fresh = start && s_old[k] == -1;
                const unsigned long long bal = __ballot(fresh);
                if (k < Dn) {
                    s_best[k] = start ? k : -1;
                    if (start) s_tid[k] = fresh ? run + __popcll(bal & ((1ull << lane) - 1ull)) : s_old[k];
                    const int nk = s_next[k];
                    jump[k] = nk > k ? nk : -1;
                }
                run += __popcll(bal);
            }
            hg_wave_sync();
            for (;;) {
                int more = 0;
                for (int k = lane; k < Dn; k += 64) {
                    const int j = jump[k];
                    if (j >= 0) {
                        const int b = s_best[k];
                        if (b >= 0) atomicMax(&s_best[j], b);
                        const int jj = jump[j];
                        jump[k] = jj;
                        more |= jj >= 0 ? 1 : 0;
                    }
                }
                hg_wave_sync();
                if (__ballot(more != 0) == 0) break;
            }
        }
    } else if (tid == 0) {
        int next_id = s_max + 1;
        for (int k = 0; k < Dn; ++k) {
            unsigned char f = flag[k];
            if (!(f & 2) && (f & 1)) {                          // starts a walk: an existing track continues, or a new id
                best[k] = k;
                tidv[k] = oldv[k] != -1 ? oldv[k] : next_id++;
                f |= 4;
            } else if (f & 2) {
                f |= 4;
            }
            flag[k] = f;
            if (f & 4) {
                const int nk = nextk[k];
                if (nk > k) {                                   // (links lead forward in time; anything else ends the walk)
                    flag[nk] |= 2;
                    if (best[k] > best[nk]) best[nk] = best[k];
                }
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < Dn; k += TK_THREADS)
        if (flag[k] & 4) y_track[det_id[g.det_row[k]]] = tidv[best[k]];
}
__global__ __launch_bounds__(TK_THREADS) void k_track_finalize(tmpnn_dgraph g, const int32_t* __restrict__ ts, const int32_t* __restrict__ det_id, const int32_t* __restrict__ assoc, const float* __restrict__ score, int t_upto, int32_t* __restrict__ y_track, int ND, int32_t* __restrict__ pos_of_det, int32_t* __restrict__ ws) {
    d_track_finalize(g, ts, det_id, assoc, score, t_upto, y_track, ND, pos_of_det, ws);
}


// ---- the counters of a retire call mirrored into host memory the device can write (notify: int32 [8], pinned + mapped) --------
// [0..3] = small[0..3], then [4] = 1 with release order at system scope: the host polls [4] instead of enqueueing a device -> host
// copy behind the launch (a blit kernel of ~4 us plus the copy's own synchronisation, per timestep of the batch-1 loops), and
// has the counts while the state rows of the same call are still moving.  The device never waits for the host.
__device__ __forceinline__ void d_track_publish(const int32_t* __restrict__ small, int32_t* __restrict__ notify) {
    const int c0 = __hip_atomic_load(small + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c1 = __hip_atomic_load(small + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c2 = __hip_atomic_load(small + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c3 = __hip_atomic_load(small + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(notify + 0, c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(notify + 1, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(notify + 2, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(notify + 3, c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(notify + 4, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_track_publish(const int32_t* __restrict__ small, int32_t* __restrict__ notify) {
    if (threadIdx.x == 0 && blockIdx.x == 0) d_track_publish(small, notify);
}

// ---- decode_tracks for LDS-sized graphs: the decisions as phases of one 1024-thread block, the state rows behind it ----------
// (a greedy timestep's GPU time is ~10 dependent launches of 2-8 us kernels; each launch saved is ~2 us of device gap and ~2 us
//  of host time)
#define TK_STAMP(i) do { } while (0)
__global__ __launch_bounds__(TK_THREADS) void k_track_retire(tmpnn_dgraph g, tmpnn_track_rows r, const float* __restrict__ score,
                                                             int associate, int t_upto, int ret_win,
                                                             int32_t* __restrict__ y_track, int ND,
                                                             int32_t* __restrict__ pos_of_det, int32_t* __restrict__ keep,
                                                             int32_t* __restrict__ small, tmpnn_track_rows o,
                                                             float* __restrict__ s_new, int next_t, int32_t* __restrict__ active,
                                                             int32_t* __restrict__ fin_ws /* unused by the finalisation at this
                                                             size (a kernel argument because a literal null in its LDS /
                                                             global pointer select crashes hipcc) */, int hung_floats,
                                                             int32_t* __restrict__ notify) {
    __shared__ HgMemo memo;                     // (associate = 2: what the first sweep leaves for the second, see HgMemo)
    if (associate == 2) {                       // optimal assignment per timestep (--hungarian); its cost scratch rides in fin_ws
        if (threadIdx.x == 0) small[1] = 0;
        __syncthreads();
        d_track_hungarian(g, r.ts, r.det_id, score, r.assoc, small + 1, reinterpret_cast<float*>(fin_ws), hung_floats, nullptr, 0,
                          &memo, next_t >= 0 ? 1 : 0);
        __syncthreads();
    } else if (associate) {
        d_track_associate(g, r.det_id, nullptr, score, 1, r.assoc, small + 1, (int)threadIdx.x, TK_THREADS);
        __syncthreads();
    }
    TK_STAMP(0);
    // (the finalisation walk runs LAST: nothing below reads what it writes -- y_track, pos_of_det -- and it reads the rows as the
    //  association left them; the counters the host waits for are then published before it, see `notify`)
    d_track_delete(g.N, r.ts, r.det_id, r.assoc, score, r.is_edge, r.src, r.dst, r.labels, t_upto, ret_win, keep, small, o.ts,
                   o.det_id, o.assoc, o.is_edge, o.src, o.dst, o.labels);
    __syncthreads();
    TK_STAMP(2);
    // the kept rows' scores here (the next active set reads them); their state rows move in a launch of many blocks behind this
    // one (tmpnn_track_retire): one block takes 14 us for a KITTI window's 96 KB and 84 us for a 12-frame window's
    for (int q = threadIdx.x, nk = small[0]; q < nk; q += TK_THREADS) s_new[q] = score[keep[q]];
    TK_STAMP(3);
    if (next_t >= 0) {
        __syncthreads();
        if (associate == 2) {
            // --hungarian: the next update_graph re-derives the associations by its own sweep over the compacted graph (a det that
            // was assigned and deleted frees its column there); that sweep here, over the rows that stay, into the new rows
            extern __shared__ int tk_new_index[];             // (d_track_delete's: new index of a kept row, -1 deleted)
            d_track_hungarian(g, r.ts, r.det_id, score, o.assoc, small + 1, reinterpret_cast<float*>(fin_ws), hung_floats,
                              tk_new_index, small[0], &memo, 2);
            __threadfence_block();
            __syncthreads();
        }
        d_track_active(0, small, o.ts, o.assoc, s_new, 1, next_t, active, small + 3);
    }
    TK_STAMP(4);
    __syncthreads();
    // the host may go on (it sizes the next timestep's launches from these) while the walk below and the state rows' launch run
    if (notify && threadIdx.x == 0) d_track_publish(small, notify);
    d_track_finalize(g, r.ts, r.det_id, r.assoc, score, t_upto, y_track, ND, pos_of_det, fin_ws);
    TK_STAMP(1);
}

// update_graph's first half in one launch (LDS-sized graphs): status word cleared, associations, active set
__global__ __launch_bounds__(TK_THREADS) void k_track_select(tmpnn_dgraph g, tmpnn_track_rows r, const float* __restrict__ score,
                                                             int mode, int t, int associate, int32_t* __restrict__ active,
                                                             int32_t* __restrict__ small, float* __restrict__ hung_ws, int hung_floats) {
    if (associate == 2) {
        if (threadIdx.x == 0) small[1] = 0;
        __syncthreads();
        d_track_hungarian(g, r.ts, r.det_id, score, r.assoc, small + 1, hung_ws, hung_floats);
        __syncthreads();
    } else if (associate) {
        if (mode == 0 && threadIdx.x == 0) small[1] = 0;
        __syncthreads();
        d_track_associate(g, r.det_id, mode == 0 ? r.labels : nullptr, mode == 0 ? nullptr : score, mode, r.assoc, small + 1,
                          (int)threadIdx.x, TK_THREADS);
        __syncthreads();
    }
    d_track_active(g.N, nullptr, r.ts, r.assoc, score, mode, t, active, small);
}

// ---- the first block of a sequence (initialize_graph, utils/graph.py:96-186), uploaded as ONE packed int32 array ----------------
// packed [6][N]: ts, det_id, is_edge, src, dst, labels.  Also: assoc = -1, the features of the block (X[det id] on det rows, zeros
// on edge rows) and y_out[:, 1] = -1 for the whole sequence.
__global__ __launch_bounds__(256) void k_track_load(int N, int ND, const int32_t* __restrict__ packed, int32_t* __restrict__ ts,
                                                    int32_t* __restrict__ det_id, int32_t* __restrict__ assoc,
                                                    uint8_t* __restrict__ is_edge, int32_t* __restrict__ row_src,
                                                    int32_t* __restrict__ row_dst, uint8_t* __restrict__ labels,
                                                    const float* __restrict__ X, int ld_x, int F, float* __restrict__ feats,
                                                    int ld_f, int32_t* __restrict__ y_track) {
    const long stride = (long)gridDim.x * 256;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (long r = i0; r < N; r += stride) {
        ts[r] = packed[r]; det_id[r] = packed[(size_t)N + r]; is_edge[r] = (uint8_t)packed[(size_t)2 * N + r];
        row_src[r] = packed[(size_t)3 * N + r]; row_dst[r] = packed[(size_t)4 * N + r];
        if (labels) labels[r] = (uint8_t)packed[(size_t)5 * N + r];
        assoc[r] = -1;
    }
    if (feats)
        for (long i = i0; i < (long)N * F; i += stride) {
            const int r = (int)(i / F), c = (int)(i % F);
            const int id = packed[(size_t)N + r];
            feats[(size_t)r * ld_f + c] = id >= 0 ? X[(size_t)id * ld_x + c] : 0.f;
        }
    if (y_track)
        for (long i = i0; i < ND; i += stride) y_track[i] = -1;
}

}  // namespace tmpnn

using namespace tmpnn;

// a notify buffer is written by a kernel through the pointer the host reads it by: it must be host memory the runtime mapped into
// the device's address space at that very address.  Asked of the runtime once per buffer (the loops reuse one per sequence).
static bool notify_ok(const int32_t* notify) {
    if (!notify) return true;
    static thread_local const int32_t* verified = nullptr;
    if (notify == verified) return true;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, notify) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (at.type != hipMemoryTypeHost || at.devicePointer != (void*)notify) return false;
    verified = notify;
    return true;
}

extern "C" {


int tmpnn_track_associate(const tmpnn_dgraph* g, const int32_t* det_id, const uint8_t* labels, const float* score,
                          int mode, int32_t* assoc, int32_t* status, tmpnn_stream stream) {
    TM_REQUIRE(g && g->meta && det_id && assoc && status, "track_associate: null pointer");
    TM_REQUIRE(mode == 0 ? labels != nullptr : (mode == 1 && score != nullptr), "track_associate: mode %d needs %s", mode,
               mode == 0 ? "labels" : "scores");
    if (g->N == 0) return TMPNN_OK;
    hipLaunchKernelGGL(k_track_associate, dim3(ceil_div(g->N, 256)), dim3(256), 0, as_stream(stream), *g, det_id, labels,
                       score, mode, assoc, status);
    return check_launch("track_associate");
}

int tmpnn_track_active(int N, const int32_t* ts, const int32_t* assoc, const float* score, int mode, int t,
                       int32_t* active, int32_t* count, tmpnn_stream stream) {
    TM_REQUIRE(N >= 0 && N <= TMPNN_TRACK_MAX_ROWS, "track_active: N=%d (limit %d)", N, TMPNN_TRACK_MAX_ROWS);
    TM_REQUIRE(ts && assoc && active && count && (mode == 0 || score), "track_active: null pointer");
    hipLaunchKernelGGL(k_track_active, dim3(1), dim3(TK_THREADS), 0, as_stream(stream), N, (const int32_t*)nullptr, ts, assoc,
                       score, mode, t, active, count);
    return check_launch("track_active");
}

#ifdef TMPNN_KEEP_VARIANTS
int tmpnn_track_append(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                       int32_t* ts, int32_t* det_id, int32_t* assoc, uint8_t* is_edge, int32_t* row_src,
                       int32_t* row_dst, uint8_t* labels, tmpnn_stream stream) {
    TM_REQUIRE(N >= 0 && A >= 0 && D >= 0 && (long)N + (long)A * D + D <= TMPNN_TRACK_MAX_ROWS,
               "track_append: N=%d A=%d D=%d exceeds %d rows", N, A, D, TMPNN_TRACK_MAX_ROWS);
    if (D == 0) return TMPNN_OK;
    TM_REQUIRE((A == 0 || active) && new_ids && ts && det_id && assoc && is_edge && row_src && row_dst,
               "track_append: null pointer");
    const int n = A * D + D;
    hipLaunchKernelGGL(k_track_append, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), N, A, D, active, new_ids,
                       t, track, ts, det_id, assoc, is_edge, row_src, row_dst, labels, (const float*)nullptr, 0, 0,
                       (float*)nullptr, 0);
    return check_launch("track_append");
}
#endif  // TMPNN_KEEP_VARIANTS

int tmpnn_track_delete(int N, const int32_t* ts, const int32_t* det_id, const int32_t* assoc, const float* score,
                       const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst, const uint8_t* labels,
                       int t_upto, int ret_win, int32_t* keep, int32_t* count, int32_t* o_ts, int32_t* o_det_id,
                       int32_t* o_assoc, uint8_t* o_is_edge, int32_t* o_src, int32_t* o_dst, uint8_t* o_labels,
                       tmpnn_stream stream) {
    TM_REQUIRE(N >= 0 && N <= TMPNN_TRACK_MAX_ROWS, "track_delete: N=%d (limit %d)", N, TMPNN_TRACK_MAX_ROWS);
    TM_REQUIRE(ts && det_id && assoc && score && is_edge && row_src && row_dst && keep && count && o_ts && o_det_id &&
                   o_assoc && o_is_edge && o_src && o_dst, "track_delete: null pointer");
    TM_SHM_ONCE(k_track_delete, sizeof(int) * TMPNN_TRACK_MAX_ROWS);          // up to 128 KiB: the renumbering table
    hipLaunchKernelGGL(k_track_delete, dim3(1), dim3(TK_THREADS), sizeof(int) * (size_t)(N > 0 ? N : 1), as_stream(stream),
                       N, ts, det_id, assoc, score, is_edge, row_src, row_dst, labels, t_upto, ret_win, keep, count, o_ts,
                       o_det_id, o_assoc, o_is_edge, o_src, o_dst, o_labels);
    return check_launch("track_delete");
}

int tmpnn_track_gather(const float* in, int ld_in, int W, int max_rows, const int32_t* keep, const int32_t* count,
                       float* out, int ld_out, tmpnn_stream stream) {
    TM_REQUIRE(W > 0 && ld_in >= W && ld_out >= W && max_rows >= 0, "track_gather: W=%d ld_in=%d ld_out=%d", W, ld_in, ld_out);
    if (max_rows == 0) return TMPNN_OK;
    TM_REQUIRE(in && keep && count && out, "track_gather: null pointer");
    TM_REQUIRE(aligned16(in) && aligned16(out), "track_gather: 16-byte alignment");
    long blocks = ((long)max_rows * ((W + 3) / 4) + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_track_gather, dim3((int)blocks), dim3(256), 0, as_stream(stream), in, ld_in, W, keep, count, out,
                       ld_out);
    return check_launch("track_gather");
}

size_t tmpnn_track_finalize_ws(int max_dets) { return max_dets > FIN_LDS_DETS ? sizeof(int32_t) * 6 * (size_t)max_dets : 0; }

int tmpnn_track_finalize(const tmpnn_dgraph* g, const int32_t* ts, const int32_t* det_id, const int32_t* assoc,
                         const float* score, int t_upto, int32_t* y_track, int ND, int32_t* pos_of_det, void* ws,
                         size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(g && g->meta && g->det_row && ts && det_id && assoc && score && y_track && pos_of_det && ND > 0,
               "track_finalize: null pointer / empty sequence");
    TM_REQUIRE(g->N >= 0 && g->N <= TMPNN_TRACK_MAX_ROWS, "track_finalize: N=%d (limit %d)", g->N, TMPNN_TRACK_MAX_ROWS);
    if (g->N == 0) return TMPNN_OK;
    // (the det count lives on the device; beyond FIN_LDS_DETS dets the pass needs 6 ints per det of scratch)
    TM_REQUIRE(g->N <= FIN_LDS_DETS || (ws != nullptr && ws_bytes >= tmpnn_track_finalize_ws(g->N)),
               "track_finalize: %d rows may hold more than %d dets: workspace of tmpnn_track_finalize_ws(N) bytes needed", g->N,
               FIN_LDS_DETS);
    hipLaunchKernelGGL(k_track_finalize, dim3(1), dim3(TK_THREADS), 0, as_stream(stream), *g, ts, det_id, assoc, score, t_upto,
                       y_track, ND, pos_of_det, reinterpret_cast<int32_t*>(ws));
    return check_launch("track_finalize");
}

// ---- one call per phase of a timestep (the reference's update_graph / decode_tracks as the loops call them): the same kernels
// as the entry points above, enqueued back to back -- a timestep is host-bound at batch 1 (a launch through the FFI costs more
// than the kernel runs), so what matters is the number of calls and launches, not their work.
static int rows_ok(const tmpnn_track_rows* r) {
    return r && r->ts && r->det_id && r->assoc && r->is_edge && r->src && r->dst;
}

int tmpnn_track_hungarian_max_dets(void) { return HG_MAX; }

int tmpnn_track_select(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int mode, int t,
                       int associate, int32_t* active, int32_t* small, tmpnn_stream stream) {
    return tmpnn_track_select_ws(g, rows, score, mode, t, associate, active, small, nullptr, 0, stream);
}

int tmpnn_track_select_ws(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int mode, int t,
                          int associate, int32_t* active, int32_t* small, void* ws, size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(g && small, "track_select: null pointer");
    if (g->N == 0) {            // an emptied graph (cur_win_size = 1 and a timestep without detections): no active det, status 0;
        if (hipMemsetAsync(small, 0, 2 * sizeof(int32_t), as_stream(stream)) != hipSuccess)     // zero-size tensors have null data
            return set_error(TMPNN_ELAUNCH, "track_select: clearing the counters failed");
        return TMPNN_OK;
    }
    TM_REQUIRE(rows_ok(rows) && active, "track_select: null pointer");
    TM_REQUIRE(mode == 0 ? rows->labels != nullptr : (mode == 1 && score != nullptr), "track_select: mode %d needs %s", mode,
               mode == 0 ? "labels" : "scores");
    TM_REQUIRE(associate != 2 || (mode == 1 && g->N <= FIN_LDS_DETS),
               "track_select: the optimal assignment (associate = 2) serves inference graphs of <= %d rows", FIN_LDS_DETS);
    int rc;
    if (g->N > 0 && g->N <= FIN_LDS_DETS) {
        hipLaunchKernelGGL(k_track_select, dim3(1), dim3(TK_THREADS), 0, as_stream(stream), *g, *rows, score, mode, t,
                           associate, active, small, reinterpret_cast<float*>(ws), (int)std::min<size_t>(ws_bytes / 4, 1u << 30));
        return check_launch("track_select");
    }
    if (associate) {
        if (mode == 0 && hipMemsetAsync(small + 1, 0, sizeof(int32_t), as_stream(stream)) != hipSuccess)
            return set_error(TMPNN_ELAUNCH, "track_select: clearing the status word failed");
        if ((rc = tmpnn_track_associate(g, rows->det_id, mode == 0 ? rows->labels : nullptr, mode == 0 ? nullptr : score, mode,
                                        rows->assoc, small + 1, stream))) return rc;
    }
    return tmpnn_track_active(g->N, rows->ts, rows->assoc, score, mode, t, active, small, stream);
}

int tmpnn_track_extend(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                       const tmpnn_track_rows* rows, const float* X, int ld_x, int F, float* feats, int ld_f,
                       const tmpnn_dgraph* g_new, void* ws, size_t ws_ints, tmpnn_stream stream) {
    TM_REQUIRE(N >= 0 && A >= 0 && D > 0 && (long)N + (long)A * D + D <= TMPNN_TRACK_MAX_ROWS,
               "track_extend: N=%d A=%d D=%d exceeds %d rows", N, A, D, TMPNN_TRACK_MAX_ROWS);
    TM_REQUIRE(rows_ok(rows) && (A == 0 || active) && new_ids && g_new, "track_extend: null pointer");
    TM_REQUIRE(feats == nullptr || (X && F > 0 && ld_x >= F && ld_f >= F), "track_extend: feature arguments (F=%d ld_x=%d ld_f=%d)",
               F, ld_x, ld_f);
    const int n = A * D + D;
    const long work = feats ? (long)n * F : (long)n;
    long blocks = (work + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_track_append, dim3((int)blocks), dim3(256), 0, as_stream(stream), N, A, D, active, new_ids, t, track,
                       rows->ts, rows->det_id, rows->assoc, rows->is_edge, rows->src, rows->dst, rows->labels, X, ld_x, F,
                       feats, ld_f);
    int rc = check_launch("track_extend");
    if (rc) return rc;
    return tmpnn_graph_from_rows_ws(N + n, rows->is_edge, rows->src, rows->dst, g_new, ws, ws_ints, stream);
}

int tmpnn_track_extend_tf(int N, int A, int D, const int32_t* active, const int32_t* new_ids, int t, const int32_t* track,
                          const tmpnn_track_rows* rows, const float* X, int ld_x, const tmpnn_mp_params* P, float* h,
                          float* save, size_t save_floats, const tmpnn_dgraph* g_new, const int32_t* counts,
                          const float* gather_src, int ld_gather, const int32_t* gather_keep, tmpnn_stream stream) {
    TM_REQUIRE(gather_src == nullptr || (counts != nullptr && gather_keep != nullptr && P != nullptr && ld_gather >= P->G * P->H &&
                                         aligned16(gather_src) && (ld_gather & 3) == 0),
               "track_extend_tf: the deferred state gather needs counts, the kept-row list and an aligned source of >= G*H columns");
    TM_REQUIRE(N >= 0 && A >= 0 && D > 0 && (long)N + (long)A * D + D <= TMPNN_DG_MAX_ROWS,
               "track_extend_tf: N=%d A=%d D=%d exceeds the one-launch form's %d rows", N, A, D, TMPNN_DG_MAX_ROWS);
    TM_REQUIRE(rows_ok(rows) && (A == 0 || active) && new_ids && g_new && P && X && h && save, "track_extend_tf: null pointer");
    const int n = A * D + D, Nt = N + n, G = P->G, H = P->H;
    TM_REQUIRE((H == 32 || H == 64) && G >= 1 && G <= 3, "track_extend_tf: H=%d G=%d (the fused batch-1 path)", H, G);
    TM_REQUIRE(ld_x >= P->F_total, "track_extend_tf: X [ND][ld %d] for F_total=%d", ld_x, P->F_total);
    for (int q = 0; q < G; ++q)
        TM_REQUIRE(P->F[q] > 0 && P->w1[q] && P->b1[q] && P->gamma[q] && P->beta[q] && P->w2[q] && P->b2[q] && P->run_mean[q] &&
                       P->run_var[q], "track_extend_tf: null input-transform pointer in group %d", q);
    TM_REQUIRE(g_new->meta && g_new->is_edge && g_new->pos && g_new->src && g_new->dst && g_new->src_pos && g_new->dst_pos &&
                   g_new->edge_row && g_new->det_row && g_new->rowptr && g_new->inc && (counts ? g_new->N <= Nt : g_new->N == Nt) &&
                   Nt <= g_new->cap,
               "track_extend_tf: g_new must be bound for N + A*D + D = %d rows", Nt);
    TM_REQUIRE(aligned16(h) && aligned16(save), "track_extend_tf: h / save must be 16-byte aligned");
    if (save_floats < tmpnn_mp_iter_save_floats(Nt, n, G, H))
        return set_error(TMPNN_EWORKSPACE, "track_extend_tf: save buffer %zu < %zu floats", save_floats,
                         tmpnn_mp_iter_save_floats(Nt, n, G, H));
    const SaveLayout SL = save_layout(Nt, n, G, H);
    ExtendArgs e{N, A, D, t, active, new_ids, track, *rows};
    BnFwdArgs b{*P, *g_new, n, 0, nullptr, 0, h, save + SL.ysave, save + SL.mean, save + SL.rstd,
                reinterpret_cast<int*>(save + SL.total)};
    BnSrcBlock src{A * D, new_ids, X, ld_x};
    const size_t shm = std::max(sizeof(int) * ((size_t)8 * Nt + 1), sizeof(float) * 64 * (size_t)(H + 1));
    // (gather blocks: one per 1024 float4 of the rows there may be, at most 32)
    const int gb = gather_src ? std::min(32, std::max(1, ceil_div((long)N * (G * H / 4), GC_THREADS))) : 0;
    if (H == 64) {
        TM_SHM_ONCE(k_track_extend_tf<64>, sizeof(int) * (8 * TMPNN_DG_MAX_ROWS + 1));
        hipLaunchKernelGGL(k_track_extend_tf<64>, dim3(1 + G + gb), dim3(GC_THREADS), shm, as_stream(stream), e, *g_new, b, src, counts,
                           gather_src, ld_gather, gather_keep);
    } else {
        TM_SHM_ONCE(k_track_extend_tf<32>, sizeof(int) * (8 * TMPNN_DG_MAX_ROWS + 1));
        hipLaunchKernelGGL(k_track_extend_tf<32>, dim3(1 + G + gb), dim3(GC_THREADS), shm, as_stream(stream), e, *g_new, b, src, counts,
                           gather_src, ld_gather, gather_keep);
    }
    return check_launch("track_extend_tf");
}

int tmpnn_track_retire(const tmpnn_dgraph* g, const tmpnn_track_rows* rows, const float* score, int associate, int t_upto,
                       int ret_win, int32_t* y_track, int ND, int32_t* pos_of_det, void* fin_ws, size_t fin_ws_bytes,
                       int32_t* keep, int32_t* small, const tmpnn_track_rows* rows_out, const float* h, int ld_h, int W,
                       float* h_new, int ld_hn, float* s_new, int next_t, int32_t* active, int32_t* notify,
                       tmpnn_stream stream) {
    TM_REQUIRE(g && small, "track_retire: null pointer");
    TM_REQUIRE(notify_ok(notify), "track_retire: notify must be pinned host memory mapped into the device's address space at the "
               "same address (hipHostMalloc), 32 bytes");
    if (g->N == 0) {            // nothing to decode (the old per-phase entry points returned early too): kept rows = kept dets =
        if (hipMemsetAsync(small, 0, 4 * sizeof(int32_t), as_stream(stream)) != hipSuccess)     // next active set = 0
            return set_error(TMPNN_ELAUNCH, "track_retire: clearing the counters failed");
        if (notify) {
            hipLaunchKernelGGL(k_track_publish, dim3(1), dim3(64), 0, as_stream(stream), small, notify);
            return check_launch("track_retire (publish)");
        }
        return TMPNN_OK;
    }
    TM_REQUIRE(rows_ok(rows) && rows_ok(rows_out) && score && keep && s_new && (h_new == nullptr || h), "track_retire: null pointer");
    TM_REQUIRE(h_new == nullptr || (W > 0 && ld_h >= W && ld_hn >= W && aligned16(h) && aligned16(h_new)),
               "track_retire: W=%d ld_h=%d ld_hn=%d", W, ld_h, ld_hn);
    TM_REQUIRE(h_new != nullptr || (g->N > 0 && g->N <= FIN_LDS_DETS), "track_retire: the state gather can be left to the caller "
               "(h_new = NULL) on graphs of <= %d rows only", FIN_LDS_DETS);
    TM_REQUIRE(next_t < 0 || active, "track_retire: the next timestep's active set needs its buffer");
    const int N = g->N;
    int rc;
    if (N > 0 && N <= FIN_LDS_DETS) {
        // LDS-sized graphs (the reference's windows: a few hundred rows): all five steps as phases of one block
        TM_REQUIRE(y_track && pos_of_det && ND > 0, "track_retire: null pointer / empty sequence");
        TM_SHM_ONCE(k_track_retire, sizeof(int) * FIN_LDS_DETS);
        hipLaunchKernelGGL(k_track_retire, dim3(1), dim3(TK_THREADS), sizeof(int) * (size_t)N, as_stream(stream), *g, *rows, score,
                           associate, t_upto, ret_win, y_track, ND, pos_of_det, keep, small, *rows_out, s_new, next_t, active,
                           reinterpret_cast<int32_t*>(fin_ws),
                           associate == 2 ? (int)std::min<size_t>(fin_ws_bytes / 4, 1u << 30) : 0, notify);
        if ((rc = check_launch("track_retire"))) return rc;
        if (h_new == nullptr) return TMPNN_OK;      // (the caller moves the kept rows' state itself: tmpnn_track_extend_tf)
        // the kept rows' state: sized for every row kept, the count read on the device (blocks beyond it leave at once)
        long blocks = ((long)N * ((W + 3) / 4) + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_track_gather, dim3((int)blocks), dim3(256), 0, as_stream(stream), h, ld_h, W, keep, small, h_new, ld_hn);
        return check_launch("track_retire (state rows)");
    }
    TM_REQUIRE(associate != 2, "track_retire: the optimal assignment (associate = 2) serves graphs of <= %d rows", FIN_LDS_DETS);
    if (associate &&
        (rc = tmpnn_track_associate(g, rows->det_id, nullptr, score, 1, rows->assoc, small + 1, stream))) return rc;
    if ((rc = tmpnn_track_finalize(g, rows->ts, rows->det_id, rows->assoc, score, t_upto, y_track, ND, pos_of_det, fin_ws,
                                   fin_ws_bytes, stream))) return rc;
    if ((rc = tmpnn_track_delete(N, rows->ts, rows->det_id, rows->assoc, score, rows->is_edge, rows->src, rows->dst,
                                 rows->labels, t_upto, ret_win, keep, small, rows_out->ts, rows_out->det_id, rows_out->assoc,
                                 rows_out->is_edge, rows_out->src, rows_out->dst, rows_out->labels, stream))) return rc;
    if (N > 0) {
        long blocks = ((long)N * ((W + 3) / 4 + 1) + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_track_gather2, dim3((int)blocks), dim3(256), 0, as_stream(stream), h, ld_h, W, score, keep, small,
                           h_new, ld_hn, s_new);
        if ((rc = check_launch("track_retire (gather)"))) return rc;
    }
    if (next_t >= 0) {
        // the active set of the next timestep on the compacted rows (inference rule): their number is on the device only
        hipLaunchKernelGGL(k_track_active, dim3(1), dim3(TK_THREADS), 0, as_stream(stream), 0, (const int32_t*)small,
                           (const int32_t*)rows_out->ts, (const int32_t*)rows_out->assoc, (const float*)s_new, 1, next_t, active,
                           small + 3);
        if ((rc = check_launch("track_retire (next active set)"))) return rc;
    }
    if (notify) {
        hipLaunchKernelGGL(k_track_publish, dim3(1), dim3(64), 0, as_stream(stream), small, notify);
        return check_launch("track_retire (publish)");
    }
    return TMPNN_OK;
}

int tmpnn_track_load(int N, int ND, const int32_t* packed, const tmpnn_track_rows* rows, const float* X, int ld_x, int F,
                     float* feats, int ld_f, int32_t* y_track, const tmpnn_dgraph* g, void* ws, size_t ws_ints,
                     tmpnn_stream stream) {
    TM_REQUIRE(N >= 0 && N <= TMPNN_TRACK_MAX_ROWS && ND >= 0, "track_load: N=%d (limit %d) ND=%d", N, TMPNN_TRACK_MAX_ROWS, ND);
    TM_REQUIRE(rows_ok(rows) && (N == 0 || packed) && g, "track_load: null pointer");
    TM_REQUIRE(feats == nullptr || (X && F > 0 && ld_x >= F && ld_f >= F), "track_load: feature arguments (F=%d ld_x=%d ld_f=%d)", F,
               ld_x, ld_f);
    const long work = std::max<long>(std::max<long>((long)N * (feats ? F : 1), ND), 1);
    long blocks = (work + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_track_load, dim3((int)blocks), dim3(256), 0, as_stream(stream), N, ND, packed, rows->ts, rows->det_id,
                       rows->assoc, rows->is_edge, rows->src, rows->dst, rows->labels, X, ld_x, F, feats, ld_f, y_track);
    int rc = check_launch("track_load");
    if (rc) return rc;
    return tmpnn_graph_from_rows_ws(N, rows->is_edge, rows->src, rows->dst, g, ws, ws_ints, stream);
}

}  // extern "C"
