// The adjacency / row form -> index form conversion as a device function of ONE workgroup of GC_THREADS threads
// (csrc/graphconv.hip launches it as k_graph_from_coo; csrc/trackops.hip runs it as a phase of the tracker's one-launch
// block append, tmpnn_track_extend_tf).  See graphconv.hip for the algorithm.
#pragma once
#include "common.h"

namespace tmpnn {

static constexpr int GC_THREADS = 1024;
static constexpr int GC_RUN = 2048;         // BIG mode: incidences of one det ranked out of a per-wave LDS slice (16 x 8 KiB)

// exclusive scan of one int per thread over the workgroup (16 waves); returns the prefix, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave /* [17] */, int* total) {
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
        for (int w = 0; w < GC_THREADS / 64; ++w) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[GC_THREADS / 64] = run;
    }
    __syncthreads();
    const int res = s_wave[wave] + inc - v;
    *total = s_wave[GC_THREADS / 64];
    __syncthreads();            // s_wave may be reused by the caller's next scan
    return res;
}

// FROM_ROWS: the input is the row form (type mask + per-row endpoints) instead of the COO entries: the tracker-side
// operations (csrc/trackops.hip: append / delete) edit the graph row-wise and re-derive the index form with this.
// BIG: the work arrays (8 N + 1 ints) live in a caller-provided global scratch instead of the LDS (N > TMPNN_DG_MAX_ROWS:
// dense scenes, up to TMPNN_DG_BIG_ROWS rows); same code, global atomics, still one workgroup.
template <bool FROM_ROWS, bool BIG = false>
__device__ __forceinline__ void d_graph_from_coo(int N, const int64_t* __restrict__ nidx,
                                                 const float* __restrict__ nval, long nnz_n,
                                                 const int64_t* __restrict__ eidx,
                                                 const float* __restrict__ eval_, long nnz_e,
                                                 const uint8_t* __restrict__ r_is_edge,
                                                 const int32_t* __restrict__ r_src,
                                                 const int32_t* __restrict__ r_dst,
                                                 const tmpnn_dgraph& g, int* __restrict__ scratch) {
    extern __shared__ int lds_dyn[];
    int* const lds = BIG ? scratch : lds_dyn;
    float* s_diag = reinterpret_cast<float*>(lds);   // [N]  sum of the diagonal entries of node_adj
    int* s_src = lds + N;                            // [N]  by ROW: column of the +1 entry
    int* s_dst = s_src + N;                          // [N]  by ROW: column of the -1 entry
    int* s_cnt = s_dst + N;                          // [N]  (#+1) | (#-1) << 16 ; later: index of the row within its type
    int* s_deg = s_cnt + N;                          // [N]  by det index: degree, then fill cursor
    int* s_ptr = s_deg + N;                          // [N + 1] rowptr
    int* s_inc = s_ptr + N + 1;                      // [2N] unordered incidences ; later: diag of edge_adj
    __shared__ int s_wave[GC_THREADS / 64 + 1];
    __shared__ int s_flags, s_off;
    const int tid = threadIdx.x;

    for (int r = tid; r < N; r += GC_THREADS) { s_diag[r] = 0.f; s_src[r] = -1; s_dst[r] = -1; s_cnt[r] = 0; s_deg[r] = 0; }
    if (tid == 0) { s_flags = 0; s_off = 0; }
    __syncthreads();

    // 1. scatter node_adj (or take the rows as given)
    int flags = 0;
    if (FROM_ROWS) {
        for (int r = tid; r < N; r += GC_THREADS) {
            const bool e = r_is_edge[r] != 0;
            s_diag[r] = e ? 0.f : 1.f;
            s_cnt[r] = e ? 0x10001 : 0;
            const int s = e ? r_src[r] : -1, d = e ? r_dst[r] : -1;
            const bool ok = s >= 0 && s < N && d >= 0 && d < N;
            if (e && !ok) flags |= TMPNN_DG_BAD_VALUE;
            s_src[r] = ok ? s : -1;
            s_dst[r] = ok ? d : -1;
        }
    }
    for (long i = tid; !FROM_ROWS && i < nnz_n; i += GC_THREADS) {
        const float v = nval[i];
        if (v == 0.f) continue;                                    // explicit zeros (I_node = eye - I_edge)
        const long r = nidx[i], c = nidx[nnz_n + i];
        if (r < 0 || r >= N || c < 0 || c >= N) { flags |= TMPNN_DG_BAD_VALUE; continue; }
        if (r == c) { atomicAdd(&s_diag[r], v); continue; }
        if (fabsf(v) != 1.0f) flags |= TMPNN_DG_BAD_VALUE;
        if (v > 0.f) { atomicAdd(&s_cnt[r], 1); s_src[r] = (int)c; }
        else { atomicAdd(&s_cnt[r], 0x10000); s_dst[r] = (int)c; }
    }
    __syncthreads();

    // 2. + 3. type mask, per-row checks, scan.  Thread t owns rows [t*IT, (t+1)*IT).
    const int IT = (N + GC_THREADS - 1) / GC_THREADS;
    const int r0 = tid * IT, r1 = min(N, r0 + IT);
    int my_edges = 0;
    for (int r = r0; r < r1; ++r) {
        const bool is_det = s_diag[r] != 0.f;
        const int cnt = s_cnt[r];
        if (is_det ? cnt != 0 : cnt != 0x10001) flags |= TMPNN_DG_BAD_ROW;
        my_edges += is_det ? 0 : 1;
    }
    int E_total;
    int e_idx = block_excl_scan(my_edges, s_wave, &E_total);
    for (int r = r0; r < r1; ++r) {
        const bool is_det = s_diag[r] != 0.f;
        s_cnt[r] = is_det ? (r - e_idx) : e_idx;                   // index within its type
        e_idx += is_det ? 0 : 1;
    }
    const int E = E_total, Dn = N - E_total;
    // BIG: the degree counters / fill cursors take hundreds of atomics per det (a det of a dense scene has ~300 incident
    // edges): in global memory those serialise at L2 (600 k of the 1.4 M cycles of a 10.9 k-row conversion); the LDS,
    // idle until the ranking phase, holds them instead whenever the dets fit
    const bool deg_lds = BIG && Dn <= (GC_THREADS / 64) * GC_RUN;
    if (deg_lds)
        for (int d = tid; d < Dn; d += GC_THREADS) lds_dyn[d] = 0;
    __syncthreads();
    for (int r = tid; r < N; r += GC_THREADS) {
        const bool is_det = s_diag[r] != 0.f;
        const int p = s_cnt[r];
        g.is_edge[r] = is_det ? 0 : 1;
        g.pos[r] = p;
        if (is_det) { g.det_row[p] = r; continue; }
        const int s = s_src[r], d = s_dst[r];
        const bool ends_ok = s >= 0 && d >= 0 && s_diag[s] != 0.f && s_diag[d] != 0.f;
        if (!ends_ok) { flags |= TMPNN_DG_BAD_ENDPOINT; s_src[r] = s_dst[r] = -1; continue; }
        if (!(s < r && r < d)) flags |= TMPNN_DG_BAD_ORDER;
        g.edge_row[p] = r;
        g.src[p] = s;
        g.dst[p] = d;
        g.src_pos[p] = s_cnt[s];
        g.dst_pos[p] = s_cnt[d];
        if (deg_lds) { atomicAdd(&lds_dyn[s_cnt[s]], 1); atomicAdd(&lds_dyn[s_cnt[d]], 1); }
        else { atomicAdd(&s_deg[s_cnt[s]], 1); atomicAdd(&s_deg[s_cnt[d]], 1); }
    }
    __syncthreads();

    // 4. CSR: scan of the degrees (thread t owns dets [t*IT, (t+1)*IT)), unordered fill, rank placement
    {
        const int d0 = tid * IT, d1 = min(Dn, d0 + IT);
        int mine = 0;
        for (int d = d0; d < d1; ++d) mine += deg_lds ? lds_dyn[d] : s_deg[d];
        int total;
        int run = block_excl_scan(mine, s_wave, &total);
        for (int d = d0; d < d1; ++d) {
            const int t = deg_lds ? lds_dyn[d] : s_deg[d];
            s_ptr[d] = run;
            if (deg_lds) lds_dyn[d] = run; else s_deg[d] = run;
            run += t;
        }
        if (tid == 0) s_ptr[Dn] = total;
    }
    __syncthreads();
    for (int r = tid; r < N; r += GC_THREADS) {
        if (s_diag[r] != 0.f) continue;
        const int s = s_src[r], d = s_dst[r];
        if (s < 0) continue;
        if (deg_lds) {
            s_inc[atomicAdd(&lds_dyn[s_cnt[s]], 1)] = r;
            s_inc[atomicAdd(&lds_dyn[s_cnt[d]], 1)] = r | (int)0x80000000u;
        } else {
            s_inc[atomicAdd(&s_deg[s_cnt[s]], 1)] = r;                          // + : d is the earlier det
            s_inc[atomicAdd(&s_deg[s_cnt[d]], 1)] = r | (int)0x80000000u;       // - : d is the later det
        }
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int d = wave; d < Dn; d += GC_THREADS / 64) {
            const int base = s_ptr[d], L = s_ptr[d + 1] - base;
            if (BIG && L + 3 <= GC_RUN) {
                // the work arrays are in global memory here, but the LDS is free: the det's run is ranked from this wave's
                // LDS slice (the L^2 comparisons of a 300-edge det out of L2 were 45 % of a 10 k-row conversion)
                int* const w_keys = lds_dyn + wave * GC_RUN;
                const int L4 = (L + 3) & ~3;                    // padded with rows no key is smaller than
                for (int i = lane; i < L4; i += 64) w_keys[i] = i < L ? (s_inc[base + i] & 0x7fffffff) : 0x7fffffff;
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < L; i += 64) {
                    const int key = s_inc[base + i];
                    const int row = key & 0x7fffffff;
                    int rank = 0;
                    for (int j = 0; j < L4; j += 4) {          // four keys per (broadcast) 16-byte LDS read
                        const int4 k4 = *reinterpret_cast<const int4*>(w_keys + j);
                        rank += (k4.x < row) + (k4.y < row) + (k4.z < row) + (k4.w < row);
                    }
                    g.inc[base + rank] = key;
                }
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // (the run's keys four per 16-byte read where the run allows it: every lane reads the same words, and one read per key
            //  made this placement -- L reads for each of L lanes -- the longest phase of the conversion for BDD-sized dets)
            const int* const run = s_inc + base;
            const int head = min(L, (int)((16u - ((unsigned)(size_t)run & 15u)) & 15u) >> 2);
            for (int i = lane; i < L; i += 64) {
                const int key = run[i];
                const int row = key & 0x7fffffff;
                int rank = 0, j = 0;
                for (; j < head; ++j) rank += ((run[j] & 0x7fffffff) < row) ? 1 : 0;
                for (; j + 4 <= L; j += 4) {
                    const int4 k4 = *reinterpret_cast<const int4*>(run + j);
                    rank += ((k4.x & 0x7fffffff) < row) + ((k4.y & 0x7fffffff) < row) + ((k4.z & 0x7fffffff) < row) + ((k4.w & 0x7fffffff) < row);
                }
                for (; j < L; ++j) rank += ((run[j] & 0x7fffffff) < row) ? 1 : 0;
                g.inc[base + rank] = key;
            }
        }
        for (int d = tid; d <= Dn; d += GC_THREADS) g.rowptr[d] = s_ptr[d];
    }
    __syncthreads();

    // 5. edge_adj must be node_adj^T off the diagonal (its entries carry the signs) with the complementary diagonal
    if (eidx != nullptr) {
        float* s_diag2 = reinterpret_cast<float*>(s_inc);
        for (int r = tid; r < N; r += GC_THREADS) s_diag2[r] = 0.f;
        __syncthreads();
        int off = 0;
        for (long i = tid; i < nnz_e; i += GC_THREADS) {
            const float v = eval_[i];
            if (v == 0.f) continue;
            const long r = eidx[i], c = eidx[nnz_e + i];
            if (r < 0 || r >= N || c < 0 || c >= N) { flags |= TMPNN_DG_BAD_VALUE; continue; }
            if (r == c) { atomicAdd(&s_diag2[r], v); continue; }
            const bool ok = fabsf(v) == 1.0f && s_diag[c] == 0.f && (v > 0.f ? s_src[c] == (int)r : s_dst[c] == (int)r);
            if (!ok) flags |= TMPNN_DG_BAD_EDGE_ADJ;
            ++off;
        }
        atomicAdd(&s_off, off);
        __syncthreads();
        for (int r = tid; r < N; r += GC_THREADS)
            if ((s_diag2[r] != 0.f) != (s_diag[r] == 0.f)) flags |= TMPNN_DG_BAD_EDGE_DIAG;
        if (tid == 0 && s_off != 2 * E) flags |= TMPNN_DG_BAD_EDGE_ADJ;
    }
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    if (tid == 0) {
        const int f = s_flags;
        g.meta[0] = f ? 0 : E;          // an invalid graph is presented as EMPTY: no consumer follows a bad index
        g.meta[1] = f ? 0 : Dn;
        g.meta[2] = f;
        g.meta[3] = N;
        g.meta[4] = E;
        g.meta[5] = Dn;
    }
}

}  // namespace tmpnn
