// Attention-weighted edge -> node aggregation (models/layers.py:26-43,105-112).
//
// The reference builds K dense N x N softmax matrices; the informative part is one scalar per
// (det, incident edge).  Here everything is per CSR position:
//   ha_k   = h[dets] W_k                                  (small GEMM, Dn rows)
//   s_k[e] = LeakyReLU_0.2( |ha_k[src] - ha_k[dst]| . a_k )   one scalar per EDGE, shared by both ends
//   alpha  = per-det softmax over its CSR run (wave per det: max / exp-sum by xor-shuffles)
//   es[d]  = 1/K sum_k sum_p sign_p alpha'_kp h[row_p]     (values are h, not ha: layers.py:38)
// Backward is written in gather form wherever a det owns the reduction and as two race-free
// scatter passes (src side, then dst side) where an edge row receives from its two dets, so it is
// deterministic without float atomics.
#include "common.h"

namespace tmpnn {

static constexpr float LEAKY = 0.2f;
static constexpr int KMAX = 8;

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

struct AttArgs {
    int N, E, Dn, H, K;
    const int32_t* src; const int32_t* dst; const int32_t* edge_row; const int32_t* det_row;
    const int32_t* rowptr; const int32_t* inc; const int32_t* pos;
    const float* h; int ld_h;
    const float* a;            // [K][H]
    const uint8_t* keep;       // [K][2E] or null
    float scale;               // 1/(1-p) when keep != null
    const float* ha;           // [K][Dn][H]
    float* score;              // [K][N] (edge rows)
    float* alpha;              // [K][2E]
};

// score[k][row e] = leaky(|ha_k[src]-ha_k[dst]| . a_k)
__global__ __launch_bounds__(256) void k_att_score(AttArgs A) {
    const int lpr = A.H >> 2;
    const int rpb = 256 / lpr;
    const int c4 = (threadIdx.x % lpr) * 4;
    const int slot = threadIdx.x / lpr;
    const long total = (long)A.E * A.K;
    for (long it = (long)blockIdx.x * rpb + slot; it < total; it += (long)gridDim.x * rpb) {
        const int k = (int)(it / A.E), e = (int)(it % A.E);
        const float* hak = A.ha + (size_t)k * A.Dn * A.H;
        const float4 u = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.src[e]] * A.H + c4);
        const float4 v = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.dst[e]] * A.H + c4);
        const float4 w = *reinterpret_cast<const float4*>(A.a + (size_t)k * A.H + c4);
        float s = fabsf(u.x - v.x) * w.x + fabsf(u.y - v.y) * w.y + fabsf(u.z - v.z) * w.z + fabsf(u.w - v.w) * w.w;
        s = group_sum(s, lpr);
        if (c4 == 0) A.score[(size_t)k * A.N + A.edge_row[e]] = s > 0.f ? s : LEAKY * s;
    }
}

// per-det softmax statistics of head k over the CSR run [p0, p1): (max, sum exp)
__device__ __forceinline__ void softmax_stats(const AttArgs& A, int k, int p0, int p1, int lane, float* m, float* z) {
    const float* sk = A.score + (size_t)k * A.N;
    float mx = -INFINITY;
    for (int p = p0 + lane; p < p1; p += 64) mx = fmaxf(mx, sk[A.inc[p] & 0x7fffffff]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int p = p0 + lane; p < p1; p += 64) sm += expf(sk[A.inc[p] & 0x7fffffff] - mx);
    *m = mx;
    *z = wave_sum(sm);
}

// forward: alpha (post-dropout) per position and es[d] (compact rows)
__global__ __launch_bounds__(256) void k_att_fwd_agg(AttArgs A, float* __restrict__ out, int ld_out) {
    const int lane = threadIdx.x & 63;
    const int lpr = A.H >> 2, ngrp = 64 / lpr, grp = lane / lpr, c4 = (lane % lpr) * 4;
    const long nwaves = (long)gridDim.x * 4;
    const float invK = 1.0f / (float)A.K;
    for (long d = (long)blockIdx.x * 4 + (threadIdx.x >> 6); d < A.Dn; d += nwaves) {
        const int p0 = A.rowptr[d], p1 = A.rowptr[d + 1];
        float m[KMAX], z[KMAX];
        for (int k = 0; k < A.K; ++k) softmax_stats(A, k, p0, p1, lane, &m[k], &z[k]);
        for (int p = p0 + lane; p < p1; p += 64) {
            const int row = A.inc[p] & 0x7fffffff;
            for (int k = 0; k < A.K; ++k) {
                float al = expf(A.score[(size_t)k * A.N + row] - m[k]) / z[k];
                if (A.keep) al *= A.keep[(size_t)k * 2 * A.E + p] ? A.scale : 0.f;
                A.alpha[(size_t)k * 2 * A.E + p] = al;
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = p0 + grp; p < p1; p += ngrp) {
            const int v = A.inc[p];
            const int row = v & 0x7fffffff;
            float w = 0.f;
            for (int k = 0; k < A.K; ++k) {
                float al = expf(A.score[(size_t)k * A.N + row] - m[k]) / z[k];
                if (A.keep) al *= A.keep[(size_t)k * 2 * A.E + p] ? A.scale : 0.f;
                w += al;
            }
            w *= (v < 0 ? -invK : invK);
            const float4 x = *reinterpret_cast<const float4*>(A.h + (size_t)row * A.ld_h + c4);
            acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
        }
        acc = groups_reduce(acc, lpr);
        if (grp == 0) *reinterpret_cast<float4*>(out + (size_t)d * ld_out + c4) = acc;
    }
}

// backward 1: t[p] = < d_es[d], h[row_p] >
__global__ __launch_bounds__(256) void k_att_bwd_t(AttArgs A, const float* __restrict__ d_out, int ld_dout,
                                                   float* __restrict__ t) {
    const int lane = threadIdx.x & 63;
    const int lpr = A.H >> 2, ngrp = 64 / lpr, grp = lane / lpr, c4 = (lane % lpr) * 4;
    const long nwaves = (long)gridDim.x * 4;
    for (long d = (long)blockIdx.x * 4 + (threadIdx.x >> 6); d < A.Dn; d += nwaves) {
        const int p0 = A.rowptr[d], p1 = A.rowptr[d + 1];
        const float4 g = *reinterpret_cast<const float4*>(d_out + (size_t)A.det_row[d] * ld_dout + c4);
        // every group must take part in the shuffles: iterate to a common trip count
        const int trips = (p1 - p0 + ngrp - 1) / ngrp;
        for (int i = 0; i < trips; ++i) {
            const int p = p0 + i * ngrp + grp;
            float s = 0.f;
            if (p < p1) {
                const int row = A.inc[p] & 0x7fffffff;
                const float4 x = *reinterpret_cast<const float4*>(A.h + (size_t)row * A.ld_h + c4);
                s = g.x * x.x + g.y * x.y + g.z * x.z + g.w * x.w;
            }
            s = group_sum(s, lpr);
            if (p < p1 && c4 == 0) t[p] = s;
        }
    }
}

// backward 2: ds[k][p] = alpha_kp (dalpha_kp - sum_q alpha_kq dalpha_kq),  dalpha = sign/K * t * keepscale
__global__ __launch_bounds__(256) void k_att_bwd_ds(AttArgs A, const float* __restrict__ t, float* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4;
    const float invK = 1.0f / (float)A.K;
    for (long d = (long)blockIdx.x * 4 + (threadIdx.x >> 6); d < A.Dn; d += nwaves) {
        const int p0 = A.rowptr[d], p1 = A.rowptr[d + 1];
        for (int k = 0; k < A.K; ++k) {
            float m, z;
            softmax_stats(A, k, p0, p1, lane, &m, &z);
            const float* sk = A.score + (size_t)k * A.N;
            float dot = 0.f;
            for (int p = p0 + lane; p < p1; p += 64) {
                const int v = A.inc[p];
                const float al = expf(sk[v & 0x7fffffff] - m) / z;
                float da = (v < 0 ? -invK : invK) * t[p];
                if (A.keep) da *= A.keep[(size_t)k * 2 * A.E + p] ? A.scale : 0.f;
                dot += al * da;
            }
            dot = wave_sum(dot);
            for (int p = p0 + lane; p < p1; p += 64) {
                const int v = A.inc[p];
                const float al = expf(sk[v & 0x7fffffff] - m) / z;
                float da = (v < 0 ? -invK : invK) * t[p];
                if (A.keep) da *= A.keep[(size_t)k * 2 * A.E + p] ? A.scale : 0.f;
                ds[(size_t)k * 2 * A.E + p] = al * (da - dot);
            }
        }
    }
}

// backward 3 (run twice: NEG = 0 handles the src-side incidences, NEG = 1 the dst side; within one
// pass every edge row is owned by exactly one det, so the updates are race free):
//   d_h[row_p] += sign/K * (sum_k alpha'_kp) * d_es[d]     ;   dse[k][row_p] (=|+=) ds[k][p]
template <int NEG>
__global__ __launch_bounds__(256) void k_att_bwd_scatter(AttArgs A, const float* __restrict__ d_out, int ld_dout,
                                                         const float* __restrict__ ds, float* __restrict__ dse,
                                                         float* __restrict__ d_h, int ld_dh) {
    const int lane = threadIdx.x & 63;
    const int lpr = A.H >> 2, ngrp = 64 / lpr, grp = lane / lpr, c4 = (lane % lpr) * 4;
    const long nwaves = (long)gridDim.x * 4;
    const float invK = 1.0f / (float)A.K;
    for (long d = (long)blockIdx.x * 4 + (threadIdx.x >> 6); d < A.Dn; d += nwaves) {
        const int p0 = A.rowptr[d], p1 = A.rowptr[d + 1];
        const float4 g = *reinterpret_cast<const float4*>(d_out + (size_t)A.det_row[d] * ld_dout + c4);
        for (int p = p0 + grp; p < p1; p += ngrp) {
            const int v = A.inc[p];
            if ((v < 0) != (NEG != 0)) continue;
            const int row = v & 0x7fffffff;
            float w = 0.f;
            for (int k = 0; k < A.K; ++k) w += A.alpha[(size_t)k * 2 * A.E + p];
            w *= NEG ? -invK : invK;
            float* o = d_h + (size_t)row * ld_dh + c4;
            float4 cur = *reinterpret_cast<const float4*>(o);
            cur.x += w * g.x; cur.y += w * g.y; cur.z += w * g.z; cur.w += w * g.w;
            *reinterpret_cast<float4*>(o) = cur;
            if (c4 == 0)
                for (int k = 0; k < A.K; ++k) {
                    float* q = dse + (size_t)k * A.N + row;
                    const float x = ds[(size_t)k * 2 * A.E + p];
                    *q = NEG ? *q + x : x;
                }
        }
    }
}

// backward 4: dpre[k][row e] = dse * leaky'(pre)  (in place) ; da partial per block: [K][H]
__global__ __launch_bounds__(256) void k_att_bwd_edge(AttArgs A, float* __restrict__ dse, int edges_per_block,
                                                      float* __restrict__ part) {
    extern __shared__ float sm[];        // [slots][H]
    const int lpr = A.H >> 2;
    const int slots = 256 / lpr;
    const int c4 = (threadIdx.x % lpr) * 4;
    const int slot = threadIdx.x / lpr;
    const int e0 = blockIdx.x * edges_per_block;
    const int e1 = min(A.E, e0 + edges_per_block);
    for (int k = 0; k < A.K; ++k) {
        const float* hak = A.ha + (size_t)k * A.Dn * A.H;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = e0 + slot; e < e1; e += slots) {
            const int row = A.edge_row[e];
            const size_t si = (size_t)k * A.N + row;
            const float dpre = dse[si] * (A.score[si] > 0.f ? 1.0f : LEAKY);
            if (c4 == 0) dse[si] = dpre;
            const float4 u = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.src[e]] * A.H + c4);
            const float4 v = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.dst[e]] * A.H + c4);
            acc.x += dpre * fabsf(u.x - v.x); acc.y += dpre * fabsf(u.y - v.y);
            acc.z += dpre * fabsf(u.z - v.z); acc.w += dpre * fabsf(u.w - v.w);
        }
        __syncthreads();
        *reinterpret_cast<float4*>(sm + (size_t)slot * A.H + c4) = acc;
        __syncthreads();
        for (int j = threadIdx.x; j < A.H; j += 256) {
            float s = 0.f;
            for (int q = 0; q < slots; ++q) s += sm[(size_t)q * A.H + j];
            part[((size_t)blockIdx.x * A.K + k) * A.H + j] = s;
        }
    }
}

__device__ __forceinline__ float sgnf(float x) { return (x > 0.f) - (x < 0.f); }

// backward 5: d_ha[k][d] = sum_p sign_p * dpre[k][row_p] * a_k o sgn(ha_k[src] - ha_k[dst])
__global__ __launch_bounds__(256) void k_att_bwd_dha(AttArgs A, const float* __restrict__ dpre,
                                                     float* __restrict__ d_ha) {
    const int lane = threadIdx.x & 63;
    const int lpr = A.H >> 2, ngrp = 64 / lpr, grp = lane / lpr, c4 = (lane % lpr) * 4;
    const long nwaves = (long)gridDim.x * 4;
    const long total = (long)A.Dn * A.K;
    for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < total; it += nwaves) {
        const int k = (int)(it / A.Dn), d = (int)(it % A.Dn);
        const float* hak = A.ha + (size_t)k * A.Dn * A.H;
        const float4 w = *reinterpret_cast<const float4*>(A.a + (size_t)k * A.H + c4);
        const int p0 = A.rowptr[d], p1 = A.rowptr[d + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = p0 + grp; p < p1; p += ngrp) {
            const int v = A.inc[p];
            const int row = v & 0x7fffffff;
            const int e = A.pos[row];
            const float4 u = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.src[e]] * A.H + c4);
            const float4 x = *reinterpret_cast<const float4*>(hak + (size_t)A.pos[A.dst[e]] * A.H + c4);
            const float f = (v < 0 ? -1.0f : 1.0f) * dpre[(size_t)k * A.N + row];
            acc.x += f * w.x * sgnf(u.x - x.x); acc.y += f * w.y * sgnf(u.y - x.y);
            acc.z += f * w.z * sgnf(u.z - x.z); acc.w += f * w.w * sgnf(u.w - x.w);
        }
        acc = groups_reduce(acc, lpr);
        if (grp == 0) *reinterpret_cast<float4*>(d_ha + ((size_t)k * A.Dn + d) * A.H + c4) = acc;
    }
}

static int att_grid(long waves) {
    long b = (waves + 3) / 4;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

static int att_edge_blocks(int E, int* per) {
    int nb = (E + 255) / 256;
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    *per = (E + nb - 1) / nb;
    return (E + *per - 1) / (*per > 0 ? *per : 1);
}

static int check_att(const tmpnn_graph* g, const int32_t* pos, const float* h, int ld_h, int H, int K) {
    TM_REQUIRE(g && pos && h, "att: null pointer");
    TM_REQUIRE(supported_H(H), "att: unsupported H=%d", H);
    TM_REQUIRE(K >= 1 && K <= KMAX, "att: K=%d (1..%d supported)", K, KMAX);
    TM_REQUIRE((long)g->E + g->Dn == g->N, "att: graph sizes inconsistent");
    TM_REQUIRE(ld_h >= H && (ld_h & 3) == 0 && aligned16(h), "att: state rows must be 16-byte aligned");
    return TMPNN_OK;
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

size_t tmpnn_att_bwd_ws(int E, int Dn, int H, int K) {
    // floats: ds [K][2E] + t [2E] | da partials [nblk][K][H] | split-K GEMM slabs for dW (reused)
    int per;
    const int nb = att_edge_blocks(E > 0 ? E : 1, &per);
    size_t a = (size_t)(K + 1) * 2 * (E > 0 ? E : 1);
    size_t b = (size_t)nb * K * H;
    size_t c = gemm_splitk_ws_floats(H, H, Dn > 0 ? Dn : 1);
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}

int tmpnn_att_fwd(const tmpnn_graph* g, const int32_t* pos, const float* h, int ld_h, int H, int K,
                  const float* W_att, const float* a, const uint8_t* keep, float p_drop, float* ws_ha, float* score,
                  float* alpha, float* out, int ld_out, tmpnn_stream stream) {
    int rc = check_att(g, pos, h, ld_h, H, K);
    if (rc) return rc;
    TM_REQUIRE(W_att && a && ws_ha && score && alpha && out, "att_fwd: null pointer");
    TM_REQUIRE(ld_out >= H && (ld_out & 3) == 0 && aligned16(out) && aligned16(ws_ha) && aligned16(a),
               "att_fwd: rows must be 16-byte aligned");
    TM_REQUIRE(keep == nullptr || (p_drop >= 0.f && p_drop < 1.f), "att_fwd: p_drop=%f", p_drop);
    if (g->Dn == 0) return TMPNN_OK;
    hipStream_t st = as_stream(stream);
    for (int k = 0; k < K; ++k) {   // ha_k = h[det rows] @ W_k
        GemmArgs ga{h, ld_h, 1, g->det_row, nullptr, W_att + (size_t)k * H * H, H, 1, nullptr,
                    ws_ha + (size_t)k * g->Dn * H, H, nullptr, g->Dn, H, H, 0};
        if ((rc = launch_gemm(ga, st))) return rc;
    }
    AttArgs A{g->N, g->E, g->Dn, H, K, g->src, g->dst, g->edge_row, g->det_row, g->rowptr, g->inc, pos, h, ld_h,
              a, keep, keep ? 1.0f / (1.0f - p_drop) : 1.0f, ws_ha, score, alpha};
    if (g->E > 0) {
        const int rpb = 256 / (H >> 2);
        long nb = ((long)g->E * K + rpb - 1) / rpb;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_att_score, dim3((int)nb), dim3(256), 0, st, A);
        if ((rc = check_launch("att_score"))) return rc;
    }
    hipLaunchKernelGGL(k_att_fwd_agg, dim3(att_grid(g->Dn)), dim3(256), 0, st, A, out, ld_out);
    return check_launch("att_fwd_agg");
}

}  // extern "C"

// dW_att / da: the stacked outputs [K][H][H] / [K][H], or (NULL) one pointer per head in dW_heads / da_heads (host arrays)
static int att_bwd_impl(const tmpnn_graph* g, const int32_t* pos, const float* h, int ld_h, int H, int K,
                        const float* W_att, const float* a, const uint8_t* keep, float p_drop, const float* ws_ha,
                        const float* score, const float* alpha, const float* d_out, int ld_dout, float* ws, size_t ws_floats,
                        float* ws_dha, float* ws_edge, float* d_h, int ld_dh, float* dW_att, float* da,
                        float* const* dW_heads, float* const* da_heads, tmpnn_stream stream) {
    int rc = check_att(g, pos, h, ld_h, H, K);
    if (rc) return rc;
    TM_REQUIRE(W_att && a && ws_ha && score && alpha && d_out && ws && ws_dha && ws_edge && d_h &&
                   ((dW_att && da) || (dW_heads && da_heads)), "att_bwd: null pointer");
    if (!dW_att)
        for (int k = 0; k < K; ++k) TM_REQUIRE(dW_heads[k] && da_heads[k], "att_bwd_heads: null gradient pointer of head %d", k);
    TM_REQUIRE((ld_dout & 3) == 0 && (ld_dh & 3) == 0 && aligned16(d_out) && aligned16(d_h) && aligned16(ws_dha),
               "att_bwd: rows must be 16-byte aligned");
    if (g->Dn == 0 || g->E == 0) return TMPNN_OK;
    const size_t need = tmpnn_att_bwd_ws(g->E, g->Dn, H, K);
    if (ws_floats < need) return set_error(TMPNN_EWORKSPACE, "att_bwd: workspace %zu < %zu floats", ws_floats, need);
    hipStream_t st = as_stream(stream);
    AttArgs A{g->N, g->E, g->Dn, H, K, g->src, g->dst, g->edge_row, g->det_row, g->rowptr, g->inc, pos, h, ld_h,
              a, keep, keep ? 1.0f / (1.0f - p_drop) : 1.0f, ws_ha, const_cast<float*>(score),
              const_cast<float*>(alpha)};
    float* ds = ws;                               // [K][2E]
    float* t = ws + (size_t)K * 2 * g->E;         // [2E]
    const dim3 gd(att_grid(g->Dn)), blk(256);
    hipLaunchKernelGGL(k_att_bwd_t, gd, blk, 0, st, A, d_out, ld_dout, t);
    if ((rc = check_launch("att_bwd_t"))) return rc;
    hipLaunchKernelGGL(k_att_bwd_ds, gd, blk, 0, st, A, t, ds);
    if ((rc = check_launch("att_bwd_ds"))) return rc;
    hipLaunchKernelGGL((k_att_bwd_scatter<0>), gd, blk, 0, st, A, d_out, ld_dout, ds, ws_edge, d_h, ld_dh);
    if ((rc = check_launch("att_bwd_scatter0"))) return rc;
    hipLaunchKernelGGL((k_att_bwd_scatter<1>), gd, blk, 0, st, A, d_out, ld_dout, ds, ws_edge, d_h, ld_dh);
    if ((rc = check_launch("att_bwd_scatter1"))) return rc;
    // ds/t are dead from here on: reuse ws for the da partials
    int per;
    const int nb = att_edge_blocks(g->E, &per);
    const int slots = 256 / (H >> 2);
    hipLaunchKernelGGL(k_att_bwd_edge, dim3(nb), blk, sizeof(float) * slots * H, st, A, ws_edge, per, ws);
    if ((rc = check_launch("att_bwd_edge"))) return rc;
    if (da) {
        if ((rc = launch_reduce_slabs(ws, (size_t)K * H, nb, da, (size_t)K * H, 1, st))) return rc;
    } else {
        for (int k = 0; k < K; ++k)
            if ((rc = launch_reduce_slabs(ws + (size_t)k * H, (size_t)K * H, nb, da_heads[k], (size_t)H, 1, st))) return rc;
    }
    hipLaunchKernelGGL(k_att_bwd_dha, dim3(att_grid((long)g->Dn * K)), blk, 0, st, A, ws_edge, ws_dha);
    if ((rc = check_launch("att_bwd_dha"))) return rc;
    for (int k = 0; k < K; ++k) {
        const float* dha = ws_dha + (size_t)k * g->Dn * H;
        // dW_k[i][j] += sum_d h[det d][i] * d_ha_k[d][j]
        GemmArgs gw{h, 1, ld_h, nullptr, g->det_row, dha, H, 1, nullptr, dW_att ? dW_att + (size_t)k * H * H : dW_heads[k], H,
                    nullptr, H, H, g->Dn, 1};
        if ((rc = launch_gemm_splitk(gw, ws, ws_floats, st))) return rc;
        // d_h[det rows] += d_ha_k @ W_k^T
        GemmArgs gh{dha, H, 1, nullptr, nullptr, W_att + (size_t)k * H * H, 1, H, nullptr, d_h, ld_dh, g->det_row,
                    g->Dn, H, H, 1};
        if ((rc = launch_gemm(gh, st))) return rc;
    }
    return TMPNN_OK;
}

extern "C" {

int tmpnn_att_bwd(const tmpnn_graph* g, const int32_t* pos, const float* h, int ld_h, int H, int K,
                  const float* W_att, const float* a, const uint8_t* keep, float p_drop, const float* ws_ha,
                  const float* score, const float* alpha, const float* d_out, int ld_dout, float* ws, size_t ws_floats,
                  float* ws_dha, float* ws_edge, float* d_h, int ld_dh, float* dW_att, float* da,
                  tmpnn_stream stream) {
    TM_REQUIRE(dW_att && da, "att_bwd: null pointer");
    return att_bwd_impl(g, pos, h, ld_h, H, K, W_att, a, keep, p_drop, ws_ha, score, alpha, d_out, ld_dout, ws, ws_floats, ws_dha,
                        ws_edge, d_h, ld_dh, dW_att, da, nullptr, nullptr, stream);
}

int tmpnn_att_bwd_heads(const tmpnn_graph* g, const int32_t* pos, const float* h, int ld_h, int H, int K,
                        const float* W_att, const float* a, const uint8_t* keep, float p_drop, const float* ws_ha,
                        const float* score, const float* alpha, const float* d_out, int ld_dout, float* ws, size_t ws_floats,
                        float* ws_dha, float* ws_edge, float* d_h, int ld_dh, float* const* dW_heads, float* const* da_heads,
                        tmpnn_stream stream) {
    TM_REQUIRE(dW_heads && da_heads, "att_bwd_heads: null pointer");
    return att_bwd_impl(g, pos, h, ld_h, H, K, W_att, a, keep, p_drop, ws_ha, score, alpha, d_out, ld_dout, ws, ws_floats, ws_dha,
                        ws_edge, d_h, ld_dh, nullptr, nullptr, dW_heads, da_heads, stream);
}

}  // extern "C"
