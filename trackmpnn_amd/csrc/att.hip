// Attention-weighted edge -> node aggregation (models/layers.py:26-43,105-112).
//
// The reference builds K dense N x N softmax matrices; the informative part is one scalar per
// (det, incident edge).  Here everything is per CSR position, all K heads in the same pass:
//   ha       = h[dets] [W_0 | .. | W_{K-1}]                     (one GEMM, Dn rows of K*H floats)
//   s_k[e]   = LeakyReLU_0.2( |ha_k[src] - ha_k[dst]| . a_k )   one scalar per EDGE and head, shared by both ends
//   alpha    = per-det softmax over its CSR run (wave per det: max / exp-sum by xor-shuffles), dropout
//   es[d]    = 1/K sum_k sum_p sign_p alpha'_kp h[row_p]        (values are h, not ha: layers.py:38)
// Round 4 form (the round-1 kernels took nine launches per call and read h[row_p] five times):
//   forward  = k_att_score (edge-owned, rows of ha out of L1/L2; the score lands at BOTH CSR positions of the edge) +
//              k_att_fwd (det-owned: ONE read of h[row_p] serves every head; the dependent chain visiting order -> rowptr ->
//              incidences / scores -> rows is software-pipelined three dets deep, the idiom of k_segsum_pipe);
//   backward = k_att_bwd_det (det-owned: t_p = <d_es[d], h[row_p]>, the softmax adjoint per position -- its
//              sum_q alpha_q dalpha_q term is <d_es[d], es_k[d]> with the per-head aggregate the forward saved, so the run
//              is walked ONCE -- one record per position) -> k_att_bwd_edge (edge-owned: both endpoints' records meet,
//              d_h[e] += w_s d_es[src] - w_d d_es[dst], dpre = dse leaky' back to both positions) -> k_att_bwd_dha
//              (det-owned: d_ha[d] = a o sum_p dpre_p sgn(ha[d] - ha[other_p]) and the da partials) -> two Dn-row GEMMs.
// Every reduction has a fixed order (no float atomics): results are bitwise reproducible.
#include "common.h"

namespace tmpnn {

static constexpr float LEAKY = 0.2f;
static constexpr int KMAX = 8;
static constexpr int NONE = 0x7fffffff;       // "no incidence" (a real entry never has all 31 row bits set)
static constexpr int ATT_CHUNK = 32;          // consecutive positions of the visiting order per block visit

__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float group_sum(float v, int lpr) {
    for (int off = lpr >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float4 groups_reduce(float4 acc, int lpr) {
    for (int off = lpr; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    return acc;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0.f) - (x < 0.f)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

struct AttArgs {
    int N, E, Dn, H, K;
    const int32_t* det_row;
    const int32_t* rowptr; const int32_t* inc; const int32_t* det_order;
    const int32_t* erec;        // [E][8] (src det, dst det, src position, dst position | src row, dst row, edge row, 0)
    const int32_t* inc_other;   // [2E] det index of the OTHER endpoint of each CSR position | bit 31 = dst side (backward)
    const float* h; int ld_h;
    const float* a;             // [K][H]
    const uint8_t* keep;        // [2E] bit k = head k keeps the position, or null
    float scale;                // 1/(1-p) when keep != null
    const float* ha;            // [Dn][K*H]
    float* score;               // [2E][K] per CSR POSITION (an edge's score sits at both of its positions)
    float* stats;               // [Dn][K][2] softmax (max, sum exp)
    float* esk;                 // [K][Dn][H] per-head aggregate
    float* alpha;               // [K][2E]
};

// Who scatters: every per-position array (score, the backward's records, dpre) is WRITTEN by an edge-owned kernel at the
// edge's two positions (fire-and-forget 4 K-byte stores) or by the det that owns the run (contiguous), and READ by the
// det-owned kernels contiguously, at the same index as inc[p] -- so a det-owned pass has the dependent chain of the plain
// segment sum (visiting order -> rowptr -> incidences -> rows) and not one level more.

// ------------------------------------------------------------------------------------------
// score[p][k] = leaky(|ha_k[src]-ha_k[dst]| . a_k) at both positions p of edge e  (edge-owned; models/layers.py:27-33)
// consecutive edge rows share their src and cycle through the same D_t dsts: the rows of ha come out of L1
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void k_att_score(AttArgs A) {
    const int H = A.H, KH = KT * H;
    const int lpr = H >> 2, rpb = 256 / lpr;
    const int c4 = (threadIdx.x % lpr) * 4, slot = threadIdx.x / lpr;
    constexpr int U = KT <= 2 ? 4 : (KT <= 4 ? 2 : 1);
    float4 av[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) av[k] = ld4(A.a + (size_t)k * H + c4);
    const long chunk = (long)rpb * U;
    long e0 = (long)blockIdx.x * chunk + slot;
    const long step = (long)gridDim.x * chunk;
    const int lane = threadIdx.x & 63, lg = threadIdx.x % lpr, gbase = lane - lg;
    int4 idn[U];                         // (src det, dst det, src position, dst position), held by lane 0 of the group
    auto load_ids = [&](long base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long e = base + (long)u * rpb;
            const long ec = e < A.E ? e : (A.E - 1);
            idn[u] = *reinterpret_cast<const int4*>(A.erec + 8 * ec);
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) idn[u] = make_int4(0, 0, 0, 0);
    if (e0 < A.E) load_ids(e0);
    for (; e0 < A.E; e0 += step) {
        float4 x[U][KT], y[U][KT];
        int2 pq[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = e0 + (long)u * rpb < A.E;
            const int sdet = __shfl(idn[u].x, gbase), ddet = __shfl(idn[u].y, gbase);
            pq[u] = make_int2(__shfl(idn[u].z, gbase), __shfl(idn[u].w, gbase));
            const float* ps = A.ha + (size_t)sdet * KH + c4;
            const float* pd = A.ha + (size_t)ddet * KH + c4;
#pragma unroll
            for (int k = 0; k < KT; ++k) { x[u][k] = ld4(ps + k * H); y[u][k] = ld4(pd + k * H); }
        }
        if (e0 + step < A.E) load_ids(e0 + step);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sc[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float4 p = x[u][k], q = y[u][k], w = av[k];
                float t = fabsf(p.x - q.x) * w.x + fabsf(p.y - q.y) * w.y + fabsf(p.z - q.z) * w.z + fabsf(p.w - q.w) * w.w;
                t = group_sum(t, lpr);
                sc[k] = t > 0.f ? t : LEAKY * t;
            }
            // lane 0 of the group writes the src-side copy, lane 1 the dst-side copy
            if (ok[u] && lg < 2) {
                float* o = A.score + (size_t)(lg == 0 ? pq[u].x : pq[u].y) * KT;
#pragma unroll
                for (int k = 0; k < KT; ++k) o[k] = sc[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The det-owned passes: ONE DET PER LANE GROUP (the H/4 lanes that move a row), 64/(H/4) dets per wave side by side.
// A wave that owns one det at a time pays one exposed memory latency per det (its rows cannot be requested before its
// softmax statistics, which wait for its scores): measured 3.9 us per det and wave, rows or no rows (-DATT_EXP_NOROWS saved
// a quarter), i.e. the pass was bound by dets in flight, not by bytes.  With a det per group the statistics are 16-lane
// reductions of four dets at once, a lane IS a position of its group's run (no redistribution of per-position scalars),
// nothing is combined across groups, and a wave keeps 4 x U rows in flight.  The loops run to the longest run of the wave's
// groups (dets that are neighbours in the visiting order: one window, similar degrees).
// A block visits the order in chunks of ATT_CHUNK positions; per ROUND each of its 4 * ngrp groups takes one position.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float grp_max(float v, int lpr) {
    for (int off = lpr >> 1; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int across_groups_max(int v, int lpr) {
    for (int off = lpr; off < 64; off <<= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

#ifndef ATT_U
#define ATT_U 4
#endif
#define ATT_DET_ROUNDS()                                                                                      \
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;                                                 \
    const int lpr = A.H >> 2, ngrp = 64 / lpr, grp = lane / lpr, gl = lane % lpr, c4 = gl * 4, gbase = lane - gl; \
    const int dpr = 4 * ngrp, rpc = ATT_CHUNK / dpr, slot = wv * ngrp + grp;                                  \
    const long nchunk = (A.Dn + ATT_CHUNK - 1) / ATT_CHUNK;                                                   \
    long kA = 0;                                                                                              \
    bool moreA = false, moreB = false, moreC = false;                                                         \
    int dA = -1, dB = -1, p0B = 0, p1B = 0, dC = -1, p0C = 0, p1C = 0;                                        \
    auto fetchA = [&]() {                                                                                     \
        const long ch = (long)blockIdx.x + (kA / rpc) * gridDim.x;                                            \
        moreA = ch < nchunk;                                                                                  \
        dA = -1;                                                                                              \
        if (moreA) {                                                                                          \
            const long i = ch * ATT_CHUNK + (kA % rpc) * dpr + slot;                                          \
            if (i < A.Dn) dA = A.det_order ? A.det_order[i] : (int)i;                                         \
        }                                                                                                     \
        ++kA;                                                                                                 \
    }

// ------------------------------------------------------------------------------------------
// forward, det-owned: softmax over the det's run, dropout, alpha out, es[d] and es_k[d] from ONE read of h[row_p]
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void k_att_fwd(AttArgs A, float* __restrict__ out, int ld_out) {
    ATT_DET_ROUNDS();
    constexpr int U = ATT_U;                       // rows in flight per group
    const float invK = 1.0f / (float)KT;
    const size_t E2 = (size_t)2 * A.E;
    // pipeline: A det id | B CSR range | C the run's first two chunks of lpr positions: incidences, keep bits, scores | (current)
    int vC0 = NONE, vC1 = NONE;
    unsigned kbC0 = 0, kbC1 = 0;
    float sC0[KT], sC1[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) sC0[k] = sC1[k] = -INFINITY;
    auto advanceB = [&]() {
        moreB = moreA; dB = dA; p0B = p1B = 0;
        if (dB >= 0) { p0B = A.rowptr[dB]; p1B = A.rowptr[dB + 1]; }
    };
    const uint8_t* kbase = A.keep ? A.keep : reinterpret_cast<const uint8_t*>(A.inc);     // (any readable 2E bytes when there is no mask)
    auto load_pos = [&](int p, int p1, int& v, unsigned& kb, float (&s)[KT]) {
        // (clamped index + select instead of a conditional load: that is a branch, and hipcc drains the request queue at its join)
        const bool ok = p < p1;
        const int pc = ok ? p : 0;
        const int vl = A.inc[pc];
        const unsigned kl = kbase[pc];
        const float* sp = A.score + (size_t)pc * KT;
        float sl[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) sl[k] = sp[k];
        v = ok ? vl : NONE;
        kb = (ok && A.keep) ? kl : 0u;
#pragma unroll
        for (int k = 0; k < KT; ++k) s[k] = ok ? sl[k] : -INFINITY;
    };
    auto advanceC = [&]() {
        moreC = moreB; dC = dB; p0C = p0B; p1C = p1B;
        load_pos(p0C + gl, p1C, vC0, kbC0, sC0);
        load_pos(p0C + lpr + gl, p1C, vC1, kbC1, sC1);
    };
    auto advance = [&]() { advanceC(); advanceB(); fetchA(); };
    fetchA(); advanceB(); fetchA(); advance();

    while (moreC) {
        const int d = dC, p0 = p0C, L = p1C - p0C;
        const int Lmax = across_groups_max(L, lpr);
        const int nch = (Lmax + lpr - 1) / lpr;            // wave-uniform
        const int v0 = vC0, v1 = vC1;
        const unsigned kb0 = kbC0, kb1 = kbC1;
        float s0[KT], s1[KT], m[KT], z[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) { s0[k] = sC0[k]; s1[k] = sC1[k]; }
        bool advanced = false;
        if (nch > 2) { advance(); advanced = true; }        // long runs (dense scenes): the chunks beyond two are not prefetched
        // softmax statistics, two passes as the reference (max, then sum of exp); dead lanes hold -inf
#pragma unroll
        for (int k = 0; k < KT; ++k) m[k] = fmaxf(s0[k], s1[k]);
        for (int c = 2; c < nch; ++c) {
            int vv; unsigned kk; float ss[KT];
            load_pos(p0 + c * lpr + gl, p0 + L, vv, kk, ss);
#pragma unroll
            for (int k = 0; k < KT; ++k) m[k] = fmaxf(m[k], ss[k]);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            m[k] = grp_max(m[k], lpr);
            z[k] = (v0 != NONE ? expf(s0[k] - m[k]) : 0.f) + (v1 != NONE ? expf(s1[k] - m[k]) : 0.f);
        }
        for (int c = 2; c < nch; ++c) {
            int vv; unsigned kk; float ss[KT];
            load_pos(p0 + c * lpr + gl, p0 + L, vv, kk, ss);
#pragma unroll
            for (int k = 0; k < KT; ++k) z[k] += vv != NONE ? expf(ss[k] - m[k]) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) z[k] = group_sum(z[k], lpr);
        float4 acc[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] = zero4();
        auto chunk = [&](int c, int vv, unsigned kk, const float (&ss)[KT]) {
            const int Lc = min(lpr, max(L - c * lpr, 0));            // this group's positions in the chunk
            const int Lcm = min(lpr, Lmax - c * lpr);                // the wave's longest
            const bool live = vv != NONE;
            float w[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                float al = live ? expf(ss[k] - m[k]) / z[k] : 0.f;
                if (A.keep) al *= ((kk >> k) & 1u) ? A.scale : 0.f;
                if (live) A.alpha[(size_t)k * E2 + p0 + c * lpr + gl] = al;
                w[k] = (vv < 0 ? -invK : invK) * al;
            }
            const int npass = (Lcm + U - 1) / U;
            for (int j = 0; j < npass; ++j) {
                float4 x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = j * U + u;
                    const int vq = __shfl(vv, gbase + q);
                    const bool okq = q < Lc;
                    const float4 xl = ld4(A.h + (size_t)(okq ? (vq & 0x7fffffff) : 0) * A.ld_h + c4);
                    x[u] = okq ? xl : zero4();
                }
                if (!advanced) { advance(); advanced = true; }      // the next round's index loads travel behind the rows
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = j * U + u;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float t = __shfl(w[k], gbase + q);
                        const float wq = q < Lc ? t : 0.f;
                        acc[k].x += wq * x[u].x; acc[k].y += wq * x[u].y; acc[k].z += wq * x[u].z; acc[k].w += wq * x[u].w;
                    }
                }
            }
        };
        for (int c = 0; c < nch; ++c) {                     // (one copy of the body: chunks 0 and 1 come from the prefetch)
            int vv = c == 0 ? v0 : v1; unsigned kk = c == 0 ? kb0 : kb1; float ss[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) ss[k] = c == 0 ? s0[k] : s1[k];
            if (c >= 2) load_pos(p0 + c * lpr + gl, p0 + L, vv, kk, ss);
            chunk(c, vv, kk, ss);
        }
        if (!advanced) advance();
        if (d >= 0) {
            float4 tot = zero4();
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                tot.x += acc[k].x; tot.y += acc[k].y; tot.z += acc[k].z; tot.w += acc[k].w;
                *reinterpret_cast<float4*>(A.esk + ((size_t)k * A.Dn + d) * A.H + c4) = acc[k];
            }
            *reinterpret_cast<float4*>(out + (size_t)d * ld_out + c4) = tot;
            if (gl < KT) {
                float mm = m[0], zz = z[0];
#pragma unroll
                for (int k = 1; k < KT; ++k) if (gl == k) { mm = m[k]; zz = z[k]; }
                *reinterpret_cast<float2*>(A.stats + ((size_t)d * KT + gl) * 2) = make_float2(L > 0 ? mm : 0.f, L > 0 ? zz : 1.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward 1 (tiny, det-level): detrec[d][k] = (max_k, sumexp_k, <d_es[d], es_k[d]>, 0) -- what the edge pass needs to know
// about an endpoint, as ONE 16-byte load per head.  <d_es[d], es_k[d]> = sum_q alpha_kq dalpha_kq of the softmax adjoint.
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void k_att_dotk(AttArgs A, const float* __restrict__ d_out, int ld_dout, float* __restrict__ detrec) {
    const int lpr = A.H >> 2, rpb = 256 / lpr;
    const int gl = threadIdx.x % lpr, c4 = gl * 4, slot = threadIdx.x / lpr;
    for (long d = (long)blockIdx.x * rpb + slot; d < A.Dn; d += (long)gridDim.x * rpb) {
        const float4 g = ld4(d_out + (size_t)A.det_row[d] * ld_dout + c4);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float t = group_sum(dot4(g, ld4(A.esk + ((size_t)k * A.Dn + d) * A.H + c4)), lpr);
            if (gl == 0) {
                const float2 st = *reinterpret_cast<const float2*>(A.stats + ((size_t)d * KT + k) * 2);
                *reinterpret_cast<float4*>(detrec + ((size_t)d * KT + k) * 4) = make_float4(st.x, st.y, t, 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward 2, EDGE-owned: everything of the adjoint that involves the edge row h[e] -- read ONCE.  With g = d_es:
//   t_s = <g[src], h[e]>, t_d = <g[dst], h[e]>                       (the det-owned form read h[e] once per endpoint)
//   per side: alpha_k = exp(s_k - max_k[det]) / sum_k[det], keep, dalpha_k = keep sign t / K,
//             ds_k = alpha_k (dalpha_k - dotk[det][k]),  w = 1/K sum_k alpha'_k
//   d_h[row e] += w_s g[src] - w_d g[dst]
//   dpre_k = (ds_k,s + ds_k,d) leaky'(s_k)  -> both CSR positions of e (read contiguously by k_att_bwd_dha)
//   da_k partial += dpre_k |ha_k[src] - ha_k[dst]|
// The pass is bound by vector-memory INSTRUCTIONS (the first form issued ~40 per edge slot, 1.67 ms per 6 M edges): the
// seven indices of an edge are one 32-byte record (2 loads), what an endpoint contributes is one 16-byte record per head,
// and even lanes take the src side, odd lanes the dst side of the per-edge scalar work, so one load serves both.
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void k_att_bwd_edge(AttArgs A, const float* __restrict__ d_out, int ld_dout,
                                                      const float* __restrict__ detrec, float* __restrict__ dpre,
                                                      float* __restrict__ d_h, int ld_dh, int edges_per_block,
                                                      float* __restrict__ part) {
    extern __shared__ float sm[];        // [slots][H]
    const int H = A.H, KH = KT * H;
    const int lpr = H >> 2, slots = 256 / lpr;
    const int lg = threadIdx.x % lpr, c4 = lg * 4, slot = threadIdx.x / lpr;
    const int lane = threadIdx.x & 63, gbase = lane - lg;
    constexpr int U = KT <= 2 ? 2 : 1;
    const float invK = 1.0f / (float)KT;
    const uint8_t* kbase = A.keep ? A.keep : reinterpret_cast<const uint8_t*>(A.erec);
    const int e_lo = blockIdx.x * edges_per_block;
    const int e_hi = min(A.E, e_lo + edges_per_block);
    float4 dacc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) dacc[k] = zero4();
    // erec of the next iteration's edges: even lanes of a group hold (src det, dst det, src position, dst position), odd lanes
    // (src row, dst row, edge row, -): one 16-byte load per lane instead of two
    int4 idn[U];
    auto load_ids = [&](int base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = base + u * slots;
            const int ec = e < e_hi ? e : (e_hi - 1);
            idn[u] = *reinterpret_cast<const int4*>(A.erec + 8 * (size_t)ec + 4 * (lg & 1));
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) idn[u] = make_int4(0, 0, 0, 0);
    int e0 = e_lo + slot;
    if (e_lo < e_hi) load_ids(e0);
    for (; e0 - slot < e_hi; e0 += slots * U) {           // (block-uniform trip count: every thread reaches the barriers below)
        float4 x[U], gs[U], gd[U], cur[U], hs[U][KT], hd[U][KT], R[U][KT];
        float sc[U][KT];
        unsigned kb[U];
        int rr[U], pmine[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = e0 + u * slots < e_hi;
            const int sdet = __shfl(idn[u].x, gbase), ddet = __shfl(idn[u].y, gbase);
            const int spos = __shfl(idn[u].z, gbase), dpos = __shfl(idn[u].w, gbase);
            const int srow = __shfl(idn[u].x, gbase + 1), drow = __shfl(idn[u].y, gbase + 1), erow = __shfl(idn[u].z, gbase + 1);
            rr[u] = erow;
            x[u] = ld4(A.h + (size_t)erow * A.ld_h + c4);
            cur[u] = ld4(d_h + (size_t)erow * ld_dh + c4);
            gs[u] = ld4(d_out + (size_t)srow * ld_dout + c4);
            gd[u] = ld4(d_out + (size_t)drow * ld_dout + c4);
            const float* hps = A.ha + (size_t)sdet * KH + c4;
            const float* hpd = A.ha + (size_t)ddet * KH + c4;
#pragma unroll
            for (int k = 0; k < KT; ++k) { hs[u][k] = ld4(hps + k * H); hd[u][k] = ld4(hpd + k * H); }
            // the per-edge scalar work: lane 0 takes the src side, lane 1 the dst side
            const int det = (lg & 1) ? ddet : sdet;
            pmine[u] = (lg & 1) ? dpos : spos;
            kb[u] = A.keep ? kbase[pmine[u]] : 0u;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                R[u][k] = ld4(detrec + ((size_t)det * KT + k) * 4);
                sc[u][k] = A.score[(size_t)spos * KT + k];
            }
        }
        if (e0 - slot + slots * U < e_hi) load_ids(e0 + slots * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float ts = group_sum(dot4(gs[u], x[u]), lpr);
            const float td = group_sum(dot4(gd[u], x[u]), lpr);
            float wsum = 0.f, ds[KT], dp_[KT];
            {
                const float sgt = (lg & 1) ? -invK * td : invK * ts;       // sign_p t_p / K of this lane's side
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const float al = expf(sc[u][k] - R[u][k].x) / R[u][k].y;
                    const float mk = A.keep ? (((kb[u] >> k) & 1u) ? A.scale : 0.f) : 1.0f;
                    wsum += al * mk;
                    ds[k] = al * (mk * sgt - R[u][k].z) * (sc[u][k] > 0.f ? 1.0f : LEAKY);
                }
                wsum *= invK;
            }
            const float ws_ = __shfl(wsum, gbase), wd_ = __shfl(wsum, gbase + 1);
#pragma unroll
            for (int k = 0; k < KT; ++k) dp_[k] = __shfl(ds[k], gbase) + __shfl(ds[k], gbase + 1);
            if (!ok[u]) continue;
            float4 c = cur[u];
            c.x += ws_ * gs[u].x - wd_ * gd[u].x; c.y += ws_ * gs[u].y - wd_ * gd[u].y;
            c.z += ws_ * gs[u].z - wd_ * gd[u].z; c.w += ws_ * gs[u].w - wd_ * gd[u].w;
            *reinterpret_cast<float4*>(d_h + (size_t)rr[u] * ld_dh + c4) = c;
            if (lg < 2) {
                float* o = dpre + (size_t)pmine[u] * KT;
#pragma unroll
                for (int k = 0; k < KT; ++k) o[k] = dp_[k];
            }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float4 p = hs[u][k], q = hd[u][k];
                dacc[k].x += dp_[k] * fabsf(p.x - q.x); dacc[k].y += dp_[k] * fabsf(p.y - q.y);
                dacc[k].z += dp_[k] * fabsf(p.z - q.z); dacc[k].w += dp_[k] * fabsf(p.w - q.w);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        __syncthreads();
        *reinterpret_cast<float4*>(sm + (size_t)slot * H + c4) = dacc[k];
        __syncthreads();
        for (int j = threadIdx.x; j < H; j += 256) {
            float s = 0.f;
            for (int q = 0; q < slots; ++q) s += sm[(size_t)q * H + j];
            part[((size_t)blockIdx.x * KT + k) * H + j] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward 3, det-owned: d_ha[d][k] = a_k o sum_p dpre_k[p] sgn(ha_k[d] - ha_k[other_p])
//   (src side: sgn(ha[src] - ha[dst]); dst side: -sgn(ha[src] - ha[dst]) = sgn(ha[d] - ha[other]) as well)
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void k_att_bwd_dha(AttArgs A, const float* __restrict__ dpre, float* __restrict__ d_ha) {
    ATT_DET_ROUNDS();
    const int H = A.H, KH = KT * H;
    constexpr int U = KT <= 1 ? 4 : (KT <= 4 ? 2 : 1);
    float4 av[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) av[k] = ld4(A.a + (size_t)k * H + c4);
    // pipeline: A det id | B CSR range | C two chunks of other endpoints (+ side bit) and dpre, own row of ha | (current)
    int oC0 = 0, oC1 = 0;
    bool okC0 = false, okC1 = false;
    float4 ownC[KT];
    float fC0[KT], fC1[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) { ownC[k] = zero4(); fC0[k] = fC1[k] = 0.f; }
    auto advanceB = [&]() {
        moreB = moreA; dB = dA; p0B = p1B = 0;
        if (dB >= 0) { p0B = A.rowptr[dB]; p1B = A.rowptr[dB + 1]; }
    };
    auto load_pos = [&](int p, int p1, int& o, bool& ok, float (&f)[KT]) {
        ok = p < p1;
        const int pc = ok ? p : 0;
        const int ol = A.inc_other[pc];
        const float* fp = dpre + (size_t)pc * KT;
        float fl[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) fl[k] = fp[k];
        o = ok ? ol : 0;
#pragma unroll
        for (int k = 0; k < KT; ++k) f[k] = ok ? fl[k] : 0.f;
    };
    auto advanceC = [&]() {
        moreC = moreB; dC = dB; p0C = p0B; p1C = p1B;
        load_pos(p0C + gl, p1C, oC0, okC0, fC0);
        load_pos(p0C + lpr + gl, p1C, oC1, okC1, fC1);
        if (dC >= 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k) ownC[k] = ld4(A.ha + (size_t)dC * KH + k * H + c4);
        }
    };
    auto advance = [&]() { advanceC(); advanceB(); fetchA(); };
    fetchA(); advanceB(); fetchA(); advance();

    while (moreC) {
        const int d = dC, p0 = p0C, L = p1C - p0C;
        const int Lmax = across_groups_max(L, lpr);
        const int nch = (Lmax + lpr - 1) / lpr;
        const int o0 = oC0, o1 = oC1;
        float f0[KT], f1[KT];
        float4 own[KT], acc[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) { f0[k] = fC0[k]; f1[k] = fC1[k]; own[k] = ownC[k]; acc[k] = zero4(); }
        bool advanced = false;
        if (nch > 2) { advance(); advanced = true; }
        auto chunk = [&](int c, int oo, const float (&ff)[KT]) {
            const int Lc = min(lpr, max(L - c * lpr, 0));
            const int Lcm = min(lpr, Lmax - c * lpr);
            const int npass = (Lcm + U - 1) / U;
            for (int j = 0; j < npass; ++j) {
                float4 x[U][KT];
                int oq[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = j * U + u;
                    oq[u] = __shfl(oo, gbase + q);
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float4 xl = ld4(A.ha + (size_t)(q < Lc ? (oq[u] & 0x7fffffff) : 0) * KH + k * H + c4);
                        x[u][k] = q < Lc ? xl : zero4();
                    }
                }
                if (!advanced) { advance(); advanced = true; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = j * U + u;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float t = __shfl(ff[k], gbase + q);
                        const float fk = q < Lc ? t : 0.f;
                        const float dx = own[k].x - x[u][k].x, dy = own[k].y - x[u][k].y;
                        const float dz = own[k].z - x[u][k].z, dw = own[k].w - x[u][k].w;
                        acc[k].x += fk * sgnf(dx); acc[k].y += fk * sgnf(dy); acc[k].z += fk * sgnf(dz); acc[k].w += fk * sgnf(dw);
                    }
                }
            }
        };
        for (int c = 0; c < nch; ++c) {
            int oo = c == 0 ? o0 : o1; bool ok; float ff[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) ff[k] = c == 0 ? f0[k] : f1[k];
            if (c >= 2) load_pos(p0 + c * lpr + gl, p0 + L, oo, ok, ff);
            chunk(c, oo, ff);
        }
        if (!advanced) advance();
        if (d >= 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k)
                *reinterpret_cast<float4*>(d_ha + (size_t)d * KH + k * H + c4) =
                    make_float4(acc[k].x * av[k].x, acc[k].y * av[k].y, acc[k].z * av[k].z, acc[k].w * av[k].w);
        }
    }
}

// the index arrays of the attention passes in one launch (what FrameGraph.att_index() would take a dozen torch index ops for):
// thread t < E fills the per-edge half of erec[t], thread t < 2E the position it owns (erec[e][2 + side] = t, inc_other[t])
__global__ __launch_bounds__(256) void k_att_index(int E, const int32_t* __restrict__ inc, const int32_t* __restrict__ pos,
                                                   const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                   const int32_t* __restrict__ src_pos, const int32_t* __restrict__ dst_pos,
                                                   const int32_t* __restrict__ edge_row, int32_t* __restrict__ erec,
                                                   int32_t* __restrict__ inc_other) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < E) {
        int32_t* r = erec + 8 * t;
        r[0] = src_pos[t]; r[1] = dst_pos[t]; r[4] = src[t]; r[5] = dst[t]; r[6] = edge_row[t]; r[7] = 0;
    }
    if (t < 2L * E) {
        const int v = inc[t];
        const int e = pos[v & 0x7fffffff];
        const int side = v < 0 ? 1 : 0;
        erec[8 * (size_t)e + 2 + side] = (int)t;
        inc_other[t] = side ? (src_pos[e] | (int)0x80000000) : dst_pos[e];
    }
}

// the stacked temporaries into the heads' own gradient buffers: dW_k[i][j] += dWcat[i][k H + j], da_k[j] += da[k][j]
struct HeadPtrs { float* w[KMAX]; float* a[KMAX]; };
__global__ void k_att_heads_add(const float* __restrict__ dWcat, const float* __restrict__ da, int H, int K, HeadPtrs P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = H * H + H;
    if (i >= K * per) return;
    const int k = i / per, j = i % per;
    if (j < H * H) P.w[k][j] += dWcat[(size_t)(j / H) * K * H + k * H + (j % H)];
    else P.a[k][j - H * H] += da[(size_t)k * H + (j - H * H)];
}

static int att_grid(long dets) {
    long b = (dets + ATT_CHUNK - 1) / ATT_CHUNK;
    if (b > 256L * 64) b = 256L * 64;
    if (b < 1) b = 1;
    return (int)b;
}

static int check_att(const tmpnn_graph* g, const float* h, int ld_h, int H, int K) {
    TM_REQUIRE(g && h, "att: null pointer");
    TM_REQUIRE(supported_H(H), "att: unsupported H=%d", H);
    TM_REQUIRE(K >= 1 && K <= KMAX, "att: K=%d (1..%d supported)", K, KMAX);
    TM_REQUIRE(g->N >= 0 && g->E >= 0 && g->Dn >= 0 && (long)g->E + g->Dn == g->N, "att: graph sizes inconsistent");
    if (g->E > 0) TM_REQUIRE(g->inc, "att: graph edge arrays are null");
    if (g->Dn > 0) TM_REQUIRE(g->det_row && g->rowptr, "att: graph det arrays are null");
    TM_REQUIRE(ld_h >= H && (ld_h & 3) == 0 && aligned16(h), "att: state rows must be 16-byte aligned");
    return TMPNN_OK;
}

static int att_edge_blocks(int E, int* per) {
    int nb = (E + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    *per = (E + nb - 1) / nb;
    return (E + *per - 1) / (*per > 0 ? *per : 1);
}

// workspace of the backward, in floats: detrec [Dn][K][4] | dpre [2E][K] | d_ha [Dn][K H] | dWcat [H][K H] | da [K][H] |
// max(da partials [blocks][K][H] (+ their fold), split-K slabs of the dW product)
struct BwdWs { size_t dotk, dpre, dha, dw, da, tail, total; };
static BwdWs att_bwd_layout(int E, int Dn, int H, int K) {
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    BwdWs w;
    const size_t E2 = 2 * (size_t)(E > 0 ? E : 1), D = (size_t)(Dn > 0 ? Dn : 1);
    int per;
    const int nb = att_edge_blocks(E > 0 ? E : 1, &per);
    size_t o = 0;
    w.dotk = o; o += up(D * K * 4);
    w.dpre = o; o += up(E2 * K);
    w.dha = o; o += up(D * K * H);
    w.dw = o; o += up((size_t)H * K * H);
    w.da = o; o += up((size_t)K * H);
    w.tail = o;
    const size_t a = (size_t)nb * K * H + reduce_slabs_ws_floats(nb, (size_t)K * H);
    const size_t c = rows_outer_ws_floats(H, K * H, (int)D);
    o += up(a > c ? a : c);
    w.total = o;
    return w;
}

#define ATT_DISPATCH(K_, CALL)                                                          \
    switch (K_) {                                                                       \
        case 1: { constexpr int KT = 1; CALL; } break;                                  \
        case 2: { constexpr int KT = 2; CALL; } break;                                  \
        case 3: { constexpr int KT = 3; CALL; } break;                                  \
        case 4: { constexpr int KT = 4; CALL; } break;                                  \
        case 5: { constexpr int KT = 5; CALL; } break;                                  \
        case 6: { constexpr int KT = 6; CALL; } break;                                  \
        case 7: { constexpr int KT = 7; CALL; } break;                                  \
        default: { constexpr int KT = 8; CALL; } break;                                 \
    }

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

size_t tmpnn_att_bwd_ws(int E, int Dn, int H, int K) { return att_bwd_layout(E, Dn, H, K).total; }

int tmpnn_att_index(const tmpnn_graph* g, const int32_t* pos, const int32_t* src_pos, const int32_t* dst_pos, int32_t* erec,
                    int32_t* inc_other, tmpnn_stream stream) {
    TM_REQUIRE(g && g->E >= 0, "att_index: graph");
    if (g->E == 0) return TMPNN_OK;
    TM_REQUIRE(pos && src_pos && dst_pos && erec && inc_other && g->inc && g->src && g->dst && g->edge_row && aligned16(erec),
               "att_index: null pointer (erec 16-byte aligned)");
    hipLaunchKernelGGL(k_att_index, dim3(ceil_div(2L * g->E, 256)), dim3(256), 0, as_stream(stream), g->E, g->inc, pos, g->src,
                       g->dst, src_pos, dst_pos, g->edge_row, erec, inc_other);
    return check_launch("att_index");
}

int tmpnn_att_fwd(const tmpnn_graph* g, const int32_t* erec, const float* h,
                  int ld_h, int H, int K, const float* W_cat, const float* a, const uint8_t* keep, float p_drop, float* ha, float* score,
                  float* stats, float* esk, float* alpha, float* out, int ld_out, tmpnn_stream stream) {
    int rc = check_att(g, h, ld_h, H, K);
    if (rc) return rc;
    TM_REQUIRE(W_cat && a && ha && score && stats && esk && alpha && out && (g->E == 0 || erec), "att_fwd: null pointer");
    TM_REQUIRE(ld_out >= H && (ld_out & 3) == 0 && aligned16(out) && aligned16(ha) && aligned16(a) && aligned16(esk) &&
                   aligned16(stats), "att_fwd: rows must be 16-byte aligned");
    TM_REQUIRE(keep == nullptr || (p_drop >= 0.f && p_drop < 1.f), "att_fwd: p_drop=%f", p_drop);
    if (g->Dn == 0) return TMPNN_OK;
    hipStream_t st = as_stream(stream);
    // ha = h[det rows] @ [W_0 | .. | W_{K-1}]: on the matrix pipe (bf16x6) in column chunks of <= 192 where the row kernel
    // serves the width, else the generic strided product
    if (rows_gemm_supported(H, H) && (ld_h & 3) == 0) {
        for (int k0 = 0; k0 < K;) {                            // <= 3 heads per launch: NOUT / 32 in {2, 4, 6} (H = 64), {1, 2, 3} (H = 32)
            const int nh = K - k0 < 3 ? K - k0 : 3;
            if ((rc = launch_rows_gemm(g->det_row, g->Dn, h, ld_h, H, W_cat + (size_t)k0 * H, K * H, 0, nh * H,
                                       ha + (size_t)k0 * H, K * H, nullptr, 0, st))) return rc;
            k0 += nh;
        }
    } else {
        GemmArgs ga{h, ld_h, 1, g->det_row, nullptr, W_cat, (long)K * H, 1, nullptr, ha, (long)K * H, nullptr, g->Dn, K * H, H, 0};
        if ((rc = launch_gemm(ga, st))) return rc;
    }
    AttArgs A{g->N, g->E, g->Dn, H, K, g->det_row, g->rowptr, g->inc,
              g->det_order, erec, nullptr, h, ld_h, a, keep, keep ? 1.0f / (1.0f - p_drop) : 1.0f, ha, score, stats, esk, alpha};
    if (g->E > 0) {
        const int rpb = 256 / (H >> 2);
        const int U = K <= 2 ? 4 : (K <= 4 ? 2 : 1);
        long nb = ((long)g->E + (long)rpb * U - 1) / ((long)rpb * U);
        if (nb > 256L * 64) nb = 256L * 64;
        ATT_DISPATCH(K, hipLaunchKernelGGL((k_att_score<KT>), dim3((int)nb), dim3(256), 0, st, A));
        if ((rc = check_launch("att_score"))) return rc;
    }
    ATT_DISPATCH(K, hipLaunchKernelGGL((k_att_fwd<KT>), dim3(att_grid(g->Dn)), dim3(256), 0, st, A, out, ld_out));
    return check_launch("att_fwd");
}

}  // extern "C"

static int att_bwd_impl(const tmpnn_graph* g, const int32_t* erec, const int32_t* inc_other,
                        const float* h, int ld_h, int H, int K, const float* W_cat, const float* a, const uint8_t* keep,
                        float p_drop, const float* ha, const float* score, const float* stats, const float* esk,
                        const float* d_out, int ld_dout, float* ws, size_t ws_floats, float* d_h, int ld_dh,
                        const HeadPtrs& heads, tmpnn_stream stream) {
    int rc = check_att(g, h, ld_h, H, K);
    if (rc) return rc;
    TM_REQUIRE(W_cat && a && ha && score && stats && esk && d_out && ws && d_h, "att_bwd: null pointer");
    TM_REQUIRE(g->E == 0 || (inc_other && erec), "att_bwd: inc_other / erec is null");
    TM_REQUIRE((ld_dout & 3) == 0 && (ld_dh & 3) == 0 && aligned16(d_out) && aligned16(d_h) && aligned16(ws) && aligned16(ha) &&
                   aligned16(esk) && aligned16(stats) && aligned16(a), "att_bwd: rows must be 16-byte aligned");
    TM_REQUIRE(keep == nullptr || (p_drop >= 0.f && p_drop < 1.f), "att_bwd: p_drop=%f", p_drop);
    if (g->Dn == 0 || g->E == 0) return TMPNN_OK;
    const BwdWs L = att_bwd_layout(g->E, g->Dn, H, K);
    if (ws_floats < L.total) return set_error(TMPNN_EWORKSPACE, "att_bwd: workspace %zu < %zu floats", ws_floats, L.total);
    hipStream_t st = as_stream(stream);
    AttArgs A{g->N, g->E, g->Dn, H, K, g->det_row, g->rowptr, g->inc,
              g->det_order, erec, inc_other, h, ld_h, a, keep, keep ? 1.0f / (1.0f - p_drop) : 1.0f, ha,
              const_cast<float*>(score), const_cast<float*>(stats), const_cast<float*>(esk), nullptr};
    float* dotk = ws + L.dotk;
    float* dpre = ws + L.dpre;
    float* dha = ws + L.dha;
    float* dwc = ws + L.dw;
    float* dat = ws + L.da;
    float* tail = ws + L.tail;
    const dim3 gd(att_grid(g->Dn)), blk(256);
    {
        const int rpb = 256 / (H >> 2);
        long nb = ((long)g->Dn + rpb - 1) / rpb;
        if (nb > 256L * 16) nb = 256L * 16;
        ATT_DISPATCH(K, hipLaunchKernelGGL((k_att_dotk<KT>), dim3((int)nb), blk, 0, st, A, d_out, ld_dout, dotk));
        if ((rc = check_launch("att_dotk"))) return rc;
    }
    {
        int per;
        const int nb = att_edge_blocks(g->E, &per);
        const int slots = 256 / (H >> 2);
        ATT_DISPATCH(K, hipLaunchKernelGGL((k_att_bwd_edge<KT>), dim3(nb), blk, sizeof(float) * slots * H, st, A, d_out, ld_dout,
                                           dotk, dpre, d_h, ld_dh, per, tail));
        if ((rc = check_launch("att_bwd_edge"))) return rc;
        float* ws2 = tail + (size_t)nb * K * H;
        if ((rc = launch_reduce_slabs(tail, (size_t)K * H, nb, dat, (size_t)K * H, 0, st, ws2))) return rc;
    }
    ATT_DISPATCH(K, hipLaunchKernelGGL((k_att_bwd_dha<KT>), gd, blk, 0, st, A, dpre, dha));
    if ((rc = check_launch("att_bwd_dha"))) return rc;
    // dWcat[i][n] = sum_d h[det d][i] * d_ha[d][n]            (the da partials in `tail` are dead: reused for the slabs)
    if ((rc = launch_rows_outer(h, ld_h, g->det_row, dha, K * H, g->Dn, H, K * H, dwc, K * H, 0, tail, L.total - L.tail, st)))
        return rc;
    // d_h[det rows] += d_ha @ Wcat^T, head by head on the matrix pipe (W_k used transposed in place)
    if (rows_gemm_supported(K * H, H)) {
        // all heads in one product where K H <= 128 (one read-modify-write pass over d_h's det rows instead of K)
        if ((rc = launch_rows_gemm(nullptr, g->Dn, dha, K * H, K * H, W_cat, K * H, 1, H, d_h, ld_dh, g->det_row, 1, st))) return rc;
    } else if (rows_gemm_supported(H, H)) {
        for (int k = 0; k < K; ++k)
            if ((rc = launch_rows_gemm(nullptr, g->Dn, dha + (size_t)k * H, K * H, H, W_cat + (size_t)k * H, K * H, 1, H, d_h, ld_dh,
                                       g->det_row, 1, st))) return rc;
    } else {
        GemmArgs gh{dha, (long)K * H, 1, nullptr, nullptr, W_cat, 1, (long)K * H, nullptr, d_h, ld_dh, g->det_row, g->Dn, H, K * H, 1};
        if ((rc = launch_gemm(gh, st))) return rc;
    }
    const int n = K * (H * H + H);
    hipLaunchKernelGGL(k_att_heads_add, dim3(ceil_div(n, 256)), dim3(256), 0, st, dwc, dat, H, K, heads);
    return check_launch("att_heads_add");
}

extern "C" {

int tmpnn_att_bwd(const tmpnn_graph* g, const int32_t* erec, const int32_t* inc_other, const float* h,
                  int ld_h, int H, int K, const float* W_cat, const float* a, const uint8_t* keep, float p_drop, const float* ha,
                  const float* score, const float* stats, const float* esk, const float* d_out, int ld_dout, float* ws,
                  size_t ws_floats, float* d_h, int ld_dh, float* dW_att, float* da, tmpnn_stream stream) {
    TM_REQUIRE(dW_att && da, "att_bwd: null gradient pointer");
    TM_REQUIRE(K >= 1 && K <= KMAX, "att_bwd: K=%d", K);
    HeadPtrs P{};
    for (int k = 0; k < K; ++k) { P.w[k] = dW_att + (size_t)k * H * H; P.a[k] = da + (size_t)k * H; }
    return att_bwd_impl(g, erec, inc_other, h, ld_h, H, K, W_cat, a, keep, p_drop, ha, score, stats, esk, d_out,
                        ld_dout, ws, ws_floats, d_h, ld_dh, P, stream);
}

int tmpnn_att_bwd_heads(const tmpnn_graph* g, const int32_t* erec, const int32_t* inc_other,
                        const float* h, int ld_h, int H, int K, const float* W_cat, const float* a, const uint8_t* keep,
                        float p_drop, const float* ha, const float* score, const float* stats, const float* esk,
                        const float* d_out, int ld_dout, float* ws, size_t ws_floats, float* d_h, int ld_dh,
                        float* const* dW_heads, float* const* da_heads, tmpnn_stream stream) {
    TM_REQUIRE(dW_heads && da_heads, "att_bwd_heads: null pointer");
    TM_REQUIRE(K >= 1 && K <= KMAX, "att_bwd_heads: K=%d", K);
    HeadPtrs P{};
    for (int k = 0; k < K; ++k) {
        TM_REQUIRE(dW_heads[k] && da_heads[k], "att_bwd_heads: null gradient pointer of head %d", k);
        P.w[k] = dW_heads[k]; P.a[k] = da_heads[k];
    }
    return att_bwd_impl(g, erec, inc_other, h, ld_h, H, K, W_cat, a, keep, p_drop, ha, score, stats, esk, d_out,
                        ld_dout, ws, ws_floats, d_h, ld_dh, P, stream);
}

}  // extern "C"
