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

}  // extern "C"
