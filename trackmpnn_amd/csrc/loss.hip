// Training losses over the same det -> incident-edge CSR as the message-passing path
// (reference models/loss.py:8-44 create_targets, :47-74 FocalLoss, :77-115 CELoss).
//
// The reference walks every det in a Python loop over a dense N x N adjacency with .cpu() round trips
// (139 ms against 84 ms of model time at C1).  Here a det's PAST edges are the CSR entries with the sign bit
// set (the det is the later endpoint) and its FUTURE edges the others; entries are in ascending edge-row
// order, which is the order loss.py's "last positive" / "first positive" rules refer to.  One thread per det
// (runs are ~16 entries), one thread per edge for the adjoint; every reduction has a fixed order.
#include "common.h"

namespace tmpnn {

// targets[det] = labels[det]; per det: the LAST positive past edge and the FIRST positive future edge get 1
__global__ void k_targets(int Dn, const int32_t* __restrict__ det_row, const int32_t* __restrict__ rowptr,
                          const int32_t* __restrict__ inc, const uint8_t* __restrict__ labels,
                          uint8_t* __restrict__ targets) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= Dn) return;
    const int drow = det_row[d];
    targets[drow] = labels[drow];
    int last_past = -1, first_future = -1;
    for (int p = rowptr[d]; p < rowptr[d + 1]; ++p) {
        const int v = inc[p];
        const int row = v & 0x7fffffff;
        if (!labels[row]) continue;
        if (v < 0) last_past = row;
        else if (first_future < 0) first_future = row;
    }
    if (last_past >= 0) targets[last_past] = 1;
    if (first_future >= 0) targets[first_future] = 1;
}

// per det and per set s (0 = past, 1 = future): stats[d][s] = (max, sum exp, target row or -1, set size);
// loss_det[d] = sum_s [ (log Z + max - logit[target]) / size ]   (sets without a positive target contribute 0)
__global__ void k_ce_fwd(int Dn, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ inc,
                         const float* __restrict__ logits, const uint8_t* __restrict__ targets,
                         float* __restrict__ stats, float* __restrict__ loss_det) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= Dn) return;
    const int p0 = rowptr[d], p1 = rowptr[d + 1];
    float loss = 0.f;
    for (int s = 0; s < 2; ++s) {
        const bool want_neg = (s == 0);
        float mx = -INFINITY;
        int n = 0, trow = -1;
        for (int p = p0; p < p1; ++p) {
            const int v = inc[p];
            if ((v < 0) != want_neg) continue;
            const int row = v & 0x7fffffff;
            mx = fmaxf(mx, logits[row]);
            ++n;
            if (targets[row]) {
                if (want_neg) trow = row;              // past: the last positive
                else if (trow < 0) trow = row;         // future: the first positive
            }
        }
        float z = 0.f;
        if (trow >= 0) {
            for (int p = p0; p < p1; ++p) {
                const int v = inc[p];
                if ((v < 0) != want_neg) continue;
                z += expf(logits[v & 0x7fffffff] - mx);
            }
            loss += (logf(z) + mx - logits[trow]) / (float)n;
        }
        float* st = stats + ((size_t)d * 2 + s) * 4;
        st[0] = mx; st[1] = z; st[2] = (float)trow; st[3] = (float)n;
    }
    loss_det[d] = loss;
}

// d_logits[edge row e] += g * [ softmax term of src's future set + softmax term of dst's past set ]
__global__ void k_ce_bwd(int E, const int32_t* __restrict__ edge_row, const int32_t* __restrict__ src_pos,
                         const int32_t* __restrict__ dst_pos, const float* __restrict__ logits,
                         const float* __restrict__ stats, const float* __restrict__ d_loss,
                         float* __restrict__ d_logits) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int row = edge_row[e];
    const float l = logits[row];
    const float g = d_loss[0];
    float acc = 0.f;
    {   // future set of the src det
        const float* st = stats + ((size_t)src_pos[e] * 2 + 1) * 4;
        const int trow = (int)st[2];
        if (trow >= 0) acc += (expf(l - st[0]) / st[1] - (trow == row ? 1.f : 0.f)) / st[3];
    }
    {   // past set of the dst det
        const float* st = stats + ((size_t)dst_pos[e] * 2 + 0) * 4;
        const int trow = (int)st[2];
        if (trow >= 0) acc += (expf(l - st[0]) / st[1] - (trow == row ? 1.f : 0.f)) / st[3];
    }
    d_logits[row] += g * acc;
}

// FocalLoss (loss.py:47-74) on the rows of one type: p_t = t ? s : 1-s ; logpt = log(p_t + 1e-10) ;
// loss_i = -(1 - exp(logpt))^gamma * logpt * alpha_t.  rows: list of R row ids.  Per-row losses are written to
// out[r] (summed/averaged by the caller's ordered reduction).
__global__ void k_focal_fwd(const int32_t* __restrict__ rows, int R, const float* __restrict__ scores,
                            const uint8_t* __restrict__ targets, float gamma, int use_alpha, float alpha0,
                            float alpha1, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int row = rows[r];
    const bool t = targets[row] != 0;
    const float s = scores[row];
    const float logpt = logf((t ? s : 1.0f - s) + 1e-10f);
    const float pt = expf(logpt);
    const float at = use_alpha ? (t ? alpha1 : alpha0) : 1.0f;
    const float w = gamma == 0.f ? 1.0f : powf(1.0f - pt, gamma);
    out[r] = -w * logpt * at;
}

// d_scores[row] += g * d loss_i / d s   (g already holds 1/R for the mean)
__global__ void k_focal_bwd(const int32_t* __restrict__ rows, int R, const float* __restrict__ scores,
                            const uint8_t* __restrict__ targets, float gamma, int use_alpha, float alpha0,
                            float alpha1, const float* __restrict__ d_loss, float scale,
                            float* __restrict__ d_scores) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int row = rows[r];
    const bool t = targets[row] != 0;
    const float s = scores[row];
    const float q = (t ? s : 1.0f - s) + 1e-10f;          // p_t + eps ; pt = exp(log q) = q
    const float logpt = logf(q);
    const float at = use_alpha ? (t ? alpha1 : alpha0) : 1.0f;
    // loss = -(1-q)^gamma * log q * at ; d/dq = gamma (1-q)^(gamma-1) log q * at - (1-q)^gamma / q * at
    float dq;
    if (gamma == 0.f) dq = -at / q;
    else dq = at * (gamma * powf(1.0f - q, gamma - 1.0f) * logpt - powf(1.0f - q, gamma) / q);
    d_scores[row] += d_loss[0] * scale * dq * (t ? 1.0f : -1.0f);
}


// ---- train.py:70-81 for one forward call of a batch-1 window in ONE launch (and one for its adjoint) ---------------------------
// create_targets, the cross-entropy over the logits and the two focal terms (gamma = 0, no alpha: what train.py constructs), with
// their sums -- one block, phases separated by barriers.  At batch 1 a call's loss section was ~14 launches forward and ~8
// backward of 1-5 us kernels, i.e. bound by the launches.  Every value is formed exactly as the separate kernels above form it
// and the three sums are taken in launch_colsum's order (128-row chunks, four interleaved running sums per chunk, the chunks in
// sequence), so the results equal the separate entry points bit for bit.  Limits: E, Dn <= TL_MAX_ROWS (64 chunks: no fold).
static constexpr int TL_THREADS = 1024;
static constexpr int TL_MAX_ROWS = 64 * 128;

__device__ float tl_ordered_sum(const float* __restrict__ v, int R, float* s_part /* [64][4] */, float* s_chunk /* [64] */) {
    const int tid = threadIdx.x;
    const int nch = (R + 127) / 128;
    if (tid < nch * 4) {
        const int j = tid >> 2, slot = tid & 3;
        const int r1 = min(R, j * 128 + 128);
        float acc = 0.f;
        for (int r = j * 128 + slot; r < r1; r += 4) acc += v[r];
        s_part[tid] = acc;
    }
    __syncthreads();
    if (tid < nch) s_chunk[tid] = s_part[4 * tid] + s_part[4 * tid + 1] + s_part[4 * tid + 2] + s_part[4 * tid + 3];
    __syncthreads();
    float total = 0.f;
    if (tid == 0)
        for (int k = 0; k < nch; ++k) total += s_chunk[k];
    __syncthreads();
    return total;                                            // valid on thread 0
}

__global__ __launch_bounds__(TL_THREADS) void k_train_losses_fwd(tmpnn_graph g, const float* __restrict__ logits,
                                                                 const float* __restrict__ scores,
                                                                 const uint8_t* __restrict__ labels, int tp, float inv_e,
                                                                 float inv_d, uint8_t* __restrict__ targets,
                                                                 float* __restrict__ stats, float* __restrict__ out,
                                                                 float* __restrict__ ws) {
    __shared__ float s_part[256], s_chunk[64];
    const int tid = threadIdx.x;
    const int N = g.N, E = g.E, Dn = g.Dn;
    float* loss_det = ws;
    float* fe = ws + (Dn > 0 ? Dn : 1);
    float* fd = fe + (E > 0 ? E : 1);
    for (int r = tid; r < N; r += TL_THREADS) targets[r] = 0;
    __syncthreads();
    // create_targets (k_targets)
    for (int d = tid; d < Dn; d += TL_THREADS) {
        const int drow = g.det_row[d];
        targets[drow] = labels[drow];
        int last_past = -1, first_future = -1;
        for (int p = g.rowptr[d]; p < g.rowptr[d + 1]; ++p) {
            const int v = g.inc[p];
            const int row = v & 0x7fffffff;
            if (!labels[row]) continue;
            if (v < 0) last_past = row;
            else if (first_future < 0) first_future = row;
        }
        if (last_past >= 0) targets[last_past] = 1;
        if (first_future >= 0) targets[first_future] = 1;
    }
    __syncthreads();
    // cross-entropy per det (k_ce_fwd)
    for (int d = tid; d < Dn; d += TL_THREADS) {
        const int p0 = g.rowptr[d], p1 = g.rowptr[d + 1];
        float loss = 0.f;
        for (int s = 0; s < 2; ++s) {
            const bool want_neg = (s == 0);
            float mx = -INFINITY;
            int n = 0, trow = -1;
            for (int p = p0; p < p1; ++p) {
                const int v = g.inc[p];
                if ((v < 0) != want_neg) continue;
                const int row = v & 0x7fffffff;
                mx = fmaxf(mx, logits[row]);
                ++n;
                if (targets[row]) {
                    if (want_neg) trow = row;
                    else if (trow < 0) trow = row;
                }
            }
            float z = 0.f;
            if (trow >= 0) {
                for (int p = p0; p < p1; ++p) {
                    const int v = g.inc[p];
                    if ((v < 0) != want_neg) continue;
                    z += expf(logits[v & 0x7fffffff] - mx);
                }
                loss += (logf(z) + mx - logits[trow]) / (float)n;
            }
            float* st = stats + ((size_t)d * 2 + s) * 4;
            st[0] = mx; st[1] = z; st[2] = (float)trow; st[3] = (float)n;
        }
        loss_det[d] = loss;
    }
    // focal terms per row (k_focal_fwd with gamma = 0, no alpha: w = 1, at = 1)
    for (int i = tid; i < E + (tp ? Dn : 0); i += TL_THREADS) {
        const int row = i < E ? g.edge_row[i] : g.det_row[i - E];
        const bool t = targets[row] != 0;
        const float sc = scores[row];
        const float logpt = logf((t ? sc : 1.0f - sc) + 1e-10f);
        const float v = -1.0f * logpt * 1.0f;
        if (i < E) fe[i] = v; else fd[i - E] = v;
    }
    __syncthreads();
    const float lc = tl_ordered_sum(loss_det, Dn, s_part, s_chunk);
    const float se = tl_ordered_sum(fe, E, s_part, s_chunk);
    const float sd = tp ? tl_ordered_sum(fd, Dn, s_part, s_chunk) : 0.f;
    if (tid == 0) {
        out[0] = lc; out[1] = se; out[2] = sd;
        // loss_f = focal_node(...) + focal_edge(...) (train.py:81), each a mean: sum * (1 / R), NaN over an empty selection
        const float me = E > 0 ? se * inv_e : __builtin_nanf("");
        const float md = Dn > 0 ? sd * inv_d : __builtin_nanf("");
        out[3] = tp ? md + me : me;
    }
}

// the adjoint of both losses, every row written once (no zero fill, no read-modify-write): items [0, E) are the edge rows,
// [E, E + Dn) the det rows
__global__ __launch_bounds__(256) void k_train_losses_bwd(tmpnn_graph g, const int32_t* __restrict__ src_pos,
                                                          const int32_t* __restrict__ dst_pos,
                                                          const float* __restrict__ logits, const float* __restrict__ scores,
                                                          const uint8_t* __restrict__ targets, const float* __restrict__ stats,
                                                          const float* __restrict__ d_c, const float* __restrict__ d_f, int tp,
                                                          float inv_e, float inv_d, float* __restrict__ d_logits,
                                                          float* __restrict__ d_scores) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int E = g.E, Dn = g.Dn;
    if (i >= E + Dn) return;
    const bool is_e = i < E;
    const int row = is_e ? g.edge_row[i] : g.det_row[i - E];
    if (d_logits) {
        float v = 0.f;
        if (is_e) {                                          // k_ce_bwd
            const float l = logits[row];
            float acc = 0.f;
            {
                const float* st = stats + ((size_t)src_pos[i] * 2 + 1) * 4;
                const int trow = (int)st[2];
                if (trow >= 0) acc += (expf(l - st[0]) / st[1] - (trow == row ? 1.f : 0.f)) / st[3];
            }
            {
                const float* st = stats + ((size_t)dst_pos[i] * 2 + 0) * 4;
                const int trow = (int)st[2];
                if (trow >= 0) acc += (expf(l - st[0]) / st[1] - (trow == row ? 1.f : 0.f)) / st[3];
            }
            v = 0.f + d_c[0] * acc;
        }
        d_logits[row] = v;
    }
    if (d_scores) {
        float v = 0.f;
        if (is_e || tp) {                                    // k_focal_bwd, gamma = 0: dq = -1 / q
            const bool t = targets[row] != 0;
            const float sc = scores[row];
            const float q = (t ? sc : 1.0f - sc) + 1e-10f;
            const float dq = -1.0f / q;
            v = 0.f + d_f[0] * (is_e ? inv_e : inv_d) * dq * (t ? 1.0f : -1.0f);
        }
        d_scores[row] = v;
    }
}

// ---- binary cross-entropy with logits, summed (the loss of SURVEY 8(d)'s metric: BCE on ALL logits of a call against fixed
// {0,1} targets) in one pass each way instead of torch's six element-wise launches + reduction per call ----------------------
// loss_i = max(l, 0) - l t + log(1 + exp(-|l|)) ; d loss_i / d l = sigmoid(l) - t.  Thread -> contiguous run of 16 elements,
// block -> 4096 elements, fixed-order reductions (lane tree, waves in sequence, then the blocks' partials in sequence by one
// wave): bitwise reproducible.
static constexpr int BCE_PER_BLOCK = 4096;
__global__ __launch_bounds__(256) void k_bce_fwd(const float* __restrict__ logits, const float* __restrict__ targets, long n,
                                                 float* __restrict__ part) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * BCE_PER_BLOCK + threadIdx.x * 4;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = base + (long)j * 1024;
        if (i + 3 < n) {
            const float4 l = *reinterpret_cast<const float4*>(logits + i);
            const float4 t = *reinterpret_cast<const float4*>(targets + i);
            const float lv[4] = {l.x, l.y, l.z, l.w}, tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) s += fmaxf(lv[e], 0.f) - lv[e] * tv[e] + log1pf(__expf(-fabsf(lv[e])));
        } else {
            for (long k = i; k < n && k < i + 4; ++k) {
                const float lv = logits[k], tv = targets[k];
                s += fmaxf(lv, 0.f) - lv * tv + log1pf(__expf(-fabsf(lv)));
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(64) void k_bce_finish(const float* __restrict__ part, int nb, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) loss[0] = s;
}
__global__ __launch_bounds__(256) void k_bce_bwd(const float* __restrict__ logits, const float* __restrict__ targets, long n,
                                                 const float* __restrict__ d_loss, float* __restrict__ d_logits) {
    const float g = d_loss[0];
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 l = *reinterpret_cast<const float4*>(logits + i);
        const float4 t = *reinterpret_cast<const float4*>(targets + i);
        float4 d;
        d.x = g * (__builtin_amdgcn_rcpf(1.0f + __expf(-l.x)) - t.x);
        d.y = g * (__builtin_amdgcn_rcpf(1.0f + __expf(-l.y)) - t.y);
        d.z = g * (__builtin_amdgcn_rcpf(1.0f + __expf(-l.z)) - t.z);
        d.w = g * (__builtin_amdgcn_rcpf(1.0f + __expf(-l.w)) - t.w);
        *reinterpret_cast<float4*>(d_logits + i) = d;
    } else {
        for (long k = i; k < n; ++k) d_logits[k] = g * (__builtin_amdgcn_rcpf(1.0f + __expf(-logits[k])) - targets[k]);
    }
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_targets(const tmpnn_graph* g, const uint8_t* labels, uint8_t* targets, tmpnn_stream stream) {
    TM_REQUIRE(g && labels && targets, "targets: null pointer");
    hipStream_t st = as_stream(stream);
    if (g->N > 0) {
        hipError_t e = hipMemsetAsync(targets, 0, (size_t)g->N, st);
        if (e != hipSuccess) return set_error(TMPNN_ELAUNCH, "targets: memset: %s", hipGetErrorString(e));
    }
    if (g->Dn == 0) return TMPNN_OK;
    hipLaunchKernelGGL(k_targets, dim3(ceil_div(g->Dn, 256)), dim3(256), 0, st, g->Dn, g->det_row, g->rowptr, g->inc,
                       labels, targets);
    return check_launch("targets");
}

size_t tmpnn_ce_loss_ws(int Dn) { return (size_t)(Dn > 0 ? Dn : 1) + colsum_ws_floats(Dn > 0 ? Dn : 1, 1); }

int tmpnn_ce_loss_fwd(const tmpnn_graph* g, const float* logits, const uint8_t* targets, float* stats, float* loss,
                      float* ws, size_t ws_floats, tmpnn_stream stream) {
    TM_REQUIRE(g && logits && targets && stats && loss && ws, "ce_loss_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (g->Dn == 0) {
        hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), st);
        return e == hipSuccess ? TMPNN_OK : set_error(TMPNN_ELAUNCH, "ce_loss_fwd: memset");
    }
    if (ws_floats < tmpnn_ce_loss_ws(g->Dn)) return set_error(TMPNN_EWORKSPACE, "ce_loss_fwd: workspace too small");
    float* loss_det = ws;
    hipLaunchKernelGGL(k_ce_fwd, dim3(ceil_div(g->Dn, 256)), dim3(256), 0, st, g->Dn, g->rowptr, g->inc, logits, targets,
                       stats, loss_det);
    int rc = check_launch("ce_fwd");
    if (rc) return rc;
    return launch_colsum(loss_det, 1, nullptr, 0, g->Dn, 1, loss, 0, ws + g->Dn, ws_floats - g->Dn, st);
}

int tmpnn_ce_loss_bwd(const tmpnn_graph* g, const int32_t* src_pos, const int32_t* dst_pos, const float* logits,
                      const float* stats, const float* d_loss, float* d_logits, tmpnn_stream stream) {
    TM_REQUIRE(g && logits && stats && d_loss && d_logits, "ce_loss_bwd: null pointer");
    if (g->E == 0) return TMPNN_OK;
    TM_REQUIRE(src_pos && dst_pos, "ce_loss_bwd: det indices of the edge endpoints are required");
    hipLaunchKernelGGL(k_ce_bwd, dim3(ceil_div(g->E, 256)), dim3(256), 0, as_stream(stream), g->E, g->edge_row, src_pos,
                       dst_pos, logits, stats, d_loss, d_logits);
    return check_launch("ce_bwd");
}

size_t tmpnn_focal_loss_ws(int R) { return (size_t)(R > 0 ? R : 1) + colsum_ws_floats(R > 0 ? R : 1, 1); }

int tmpnn_focal_loss_fwd(const int32_t* rows, int R, const float* scores, const uint8_t* targets, float gamma,
                         int use_alpha, float alpha0, float alpha1, float* loss_sum, float* ws, size_t ws_floats,
                         tmpnn_stream stream) {
    TM_REQUIRE(scores && targets && loss_sum && ws && R >= 0, "focal_loss_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (R == 0) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(float), st);
        return e == hipSuccess ? TMPNN_OK : set_error(TMPNN_ELAUNCH, "focal_loss_fwd: memset");
    }
    TM_REQUIRE(rows != nullptr, "focal_loss_fwd: rows is null");
    if (ws_floats < tmpnn_focal_loss_ws(R)) return set_error(TMPNN_EWORKSPACE, "focal_loss_fwd: workspace too small");
    hipLaunchKernelGGL(k_focal_fwd, dim3(ceil_div(R, 256)), dim3(256), 0, st, rows, R, scores, targets, gamma, use_alpha,
                       alpha0, alpha1, ws);
    int rc = check_launch("focal_fwd");
    if (rc) return rc;
    return launch_colsum(ws, 1, nullptr, 0, R, 1, loss_sum, 0, ws + R, ws_floats - R, st);
}

int tmpnn_focal_loss_bwd(const int32_t* rows, int R, const float* scores, const uint8_t* targets, float gamma,
                         int use_alpha, float alpha0, float alpha1, const float* d_loss, float scale, float* d_scores,
                         tmpnn_stream stream) {
    TM_REQUIRE(scores && targets && d_loss && d_scores && R >= 0, "focal_loss_bwd: null pointer");
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(rows != nullptr, "focal_loss_bwd: rows is null");
    hipLaunchKernelGGL(k_focal_bwd, dim3(ceil_div(R, 256)), dim3(256), 0, as_stream(stream), rows, R, scores, targets,
                       gamma, use_alpha, alpha0, alpha1, d_loss, scale, d_scores);
    return check_launch("focal_bwd");
}

size_t tmpnn_bce_logits_ws(long n) { return (size_t)((n > 0 ? n : 1) + BCE_PER_BLOCK - 1) / BCE_PER_BLOCK; }

int tmpnn_bce_logits_sum_fwd(const float* logits, const float* targets, long n, float* loss_sum, float* ws, size_t ws_floats,
                             tmpnn_stream stream) {
    TM_REQUIRE(n >= 0 && loss_sum && (n == 0 || (logits && targets && ws)), "bce_logits_sum_fwd: null pointer");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(float), st);
        return e == hipSuccess ? TMPNN_OK : set_error(TMPNN_ELAUNCH, "bce_logits_sum_fwd: memset");
    }
    TM_REQUIRE(aligned16(logits) && aligned16(targets), "bce_logits_sum_fwd: logits / targets must be 16-byte aligned");
    const size_t nb = tmpnn_bce_logits_ws(n);
    if (ws_floats < nb) return set_error(TMPNN_EWORKSPACE, "bce_logits_sum_fwd: workspace holds %zu floats, %zu needed", ws_floats, nb);
    hipLaunchKernelGGL(k_bce_fwd, dim3((unsigned)nb), dim3(256), 0, st, logits, targets, n, ws);
    int rc = check_launch("bce_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(k_bce_finish, dim3(1), dim3(64), 0, st, ws, (int)nb, loss_sum);
    return check_launch("bce_finish");
}

int tmpnn_bce_logits_sum_bwd(const float* logits, const float* targets, long n, const float* d_loss, float* d_logits,
                             tmpnn_stream stream) {
    TM_REQUIRE(n >= 0 && (n == 0 || (logits && targets && d_loss && d_logits)), "bce_logits_sum_bwd: null pointer");
    if (n == 0) return TMPNN_OK;
    TM_REQUIRE(aligned16(logits) && aligned16(targets) && aligned16(d_logits), "bce_logits_sum_bwd: 16-byte alignment");
    hipLaunchKernelGGL(k_bce_bwd, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, as_stream(stream), logits, targets, n, d_loss,
                       d_logits);
    return check_launch("bce_bwd");
}

int tmpnn_train_losses_supported(int E, int Dn) { return (E >= 0 && Dn >= 0 && E <= TL_MAX_ROWS && Dn <= TL_MAX_ROWS) ? 1 : 0; }
size_t tmpnn_train_losses_ws(int E, int Dn) { return (size_t)2 * (Dn > 0 ? Dn : 1) + (size_t)(E > 0 ? E : 1); }

int tmpnn_train_losses_fwd(const tmpnn_graph* g, const float* logits, const float* scores, const uint8_t* labels,
                           int tp_classifier, uint8_t* targets, float* stats, float* out, float* ws, size_t ws_floats,
                           tmpnn_stream stream) {
    TM_REQUIRE(g && logits && scores && labels && targets && stats && out && ws, "train_losses_fwd: null pointer");
    TM_REQUIRE(tmpnn_train_losses_supported(g->E, g->Dn), "train_losses_fwd: E=%d Dn=%d (one-launch form: <= %d each)", g->E, g->Dn,
               TL_MAX_ROWS);
    if (ws_floats < tmpnn_train_losses_ws(g->E, g->Dn)) return set_error(TMPNN_EWORKSPACE, "train_losses_fwd: workspace too small");
    const float inv_e = g->E > 0 ? (float)(1.0 / g->E) : 0.f, inv_d = g->Dn > 0 ? (float)(1.0 / g->Dn) : 0.f;
    hipLaunchKernelGGL(k_train_losses_fwd, dim3(1), dim3(TL_THREADS), 0, as_stream(stream), *g, logits, scores, labels,
                       tp_classifier ? 1 : 0, inv_e, inv_d, targets, stats, out, ws);
    return check_launch("train_losses_fwd");
}

int tmpnn_train_losses_bwd(const tmpnn_graph* g, const int32_t* src_pos, const int32_t* dst_pos, const float* logits,
                           const float* scores, const uint8_t* targets, const float* stats, const float* d_c, const float* d_f,
                           int tp_classifier, float* d_logits, float* d_scores, tmpnn_stream stream) {
    TM_REQUIRE(g && logits && scores && targets && stats, "train_losses_bwd: null pointer");
    TM_REQUIRE((d_logits == nullptr || d_c) && (d_scores == nullptr || d_f), "train_losses_bwd: a gradient without its seed");
    TM_REQUIRE(g->E == 0 || (src_pos && dst_pos), "train_losses_bwd: det indices of the edge endpoints are required");
    if (g->N == 0 || (!d_logits && !d_scores)) return TMPNN_OK;
    const float inv_e = g->E > 0 ? (float)(1.0 / g->E) : 0.f, inv_d = g->Dn > 0 ? (float)(1.0 / g->Dn) : 0.f;
    hipLaunchKernelGGL(k_train_losses_bwd, dim3(ceil_div(g->N, 256)), dim3(256), 0, as_stream(stream), *g, src_pos, dst_pos,
                       logits, scores, targets, stats, d_c, d_f, tp_classifier ? 1 : 0, inv_e, inv_d, d_logits, d_scores);
    return check_launch("train_losses_bwd");
}

}  // extern "C"
