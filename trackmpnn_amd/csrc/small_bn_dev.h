// The input transform of the batch-1 path (Lin-BN-ReLU-Lin on the new det rows, zeros on the new edge rows) as a device
// function: csrc/small.hip launches it as k_small_bn_fwd (one block per feature group), csrc/trackops.hip runs it in further
// blocks of the tracker's one-launch block append (tmpnn_track_extend_tf).  Same code, same arithmetic order in both.
#pragma once
#include "common.h"

namespace tmpnn {

static constexpr float BN_EPS_S = 1e-5f;
static constexpr float BN_MOM_S = 0.1f;

// ------------------------------------------------------------------------------------------------------------
// saved-for-backward layout (floats)
// ------------------------------------------------------------------------------------------------------------
struct SaveLayout { size_t gates, es, ysave, mean, rstd, total; };
__host__ __device__ inline SaveLayout save_layout(int N, int n, int G, int H) {
    SaveLayout L;
    L.gates = 0;                                   // [G][4][N][H]   r, z, n, W_hn h + b_hn   (row-indexed)
    L.es = L.gates + (size_t)G * 4 * N * H;         // [G][N][H]      edge -> node sums, by det INDEX
    L.ysave = L.es + (size_t)G * N * H;             // [G][n][H]      Lin1 output of the new det rows (by new-det index)
    L.mean = L.ysave + (size_t)G * (n > 0 ? n : 1) * H;   // [G][H]
    L.rstd = L.mean + (size_t)G * H;                // [G][H]
    L.total = L.rstd + (size_t)G * H;
    return L;
}

// ------------------------------------------------------------------------------------------------------------
// input transform, forward: one block per feature group
// ------------------------------------------------------------------------------------------------------------
struct BnFwdArgs {
    tmpnn_mp_params P;
    tmpnn_dgraph g;
    int n_new, training;
    const float* x; int ld_x;
    float* h;                  // [N][G*H]
    float* ysave; float* mean; float* rstd;     // may be scratch when nothing is saved
    int* newdet;               // [n_new + 1] scratch: local indices of the new det rows, count at [n_new]
};

// Where a new row's type and features come from.  BnSrcGraph: the graph's type mask and the caller's x [n_new][ld_x] (the model
// call as the reference makes it).  BnSrcBlock: the tracker's block of one timestep before it exists in memory (rows [0, ne) are
// edge rows -- zero features by construction --, rows [ne, n) the new dets with features X[ids[i - ne]]): tmpnn_track_extend_tf
// runs the transform in the launch that appends the block.
struct BnSrcGraph {
    const uint8_t* is_edge; int N_old; const float* x; int ld_x;
    static constexpr bool kCheckEdgeRows = true;
    static constexpr bool kBlockLayout = false;
    __device__ __forceinline__ bool edge(int i) const { return is_edge[N_old + i] != 0; }
    __device__ __forceinline__ const float* row(int i) const { return x + (size_t)i * ld_x; }
};
struct BnSrcBlock {
    int ne; const int32_t* ids; const float* X; int ld_x;
    static constexpr bool kCheckEdgeRows = false;
    static constexpr bool kBlockLayout = true;      // edge rows first, then the dets: no scan for the det list, and the caller zeroes
                                                    // the edge rows of h itself (k_track_extend_tf: its otherwise idle threads)
    __device__ __forceinline__ bool edge(int i) const { return i < ne; }
    __device__ __forceinline__ const float* row(int i) const { return X + (size_t)ids[i - ne] * ld_x; }
};

// one workgroup (its first 256 threads) per feature group gi; dynamic LDS: 64 * (H + 1) floats
// (rows_total / rows_new >= 0: the graph's row count and the number of new rows where they are known on the device only -- the
//  struct's fields are then what the host sized the buffers with; the struct itself is never written: a kernel argument that is
//  modified is copied to scratch memory, and every pointer of it is then read from there)
template <int H, class SRC>
__device__ __forceinline__ void d_small_bn_fwd(const BnFwdArgs& a, const int gi, const SRC src, const int rows_total = -1,
                                               const int rows_new = -1) {
    const int G = a.P.G, GH = G * H;
    const int N = rows_total >= 0 ? rows_total : a.g.N, n = rows_new >= 0 ? rows_new : a.n_new, N_old = N - n;
    const int F = a.P.F[gi];
    int f0 = 0;
    for (int q = 0; q < gi; ++q) f0 += a.P.F[q];
    const int tid = threadIdx.x;
    __shared__ int s_wsum[5];
    __shared__ float s_mean[H], s_rstd[H], s_w2t[H * (H + 1)];
    extern __shared__ float s_a[];                 // [CH][H + 1] activation chunk
    int* newdet = a.newdet + (size_t)gi * (n + 1);

    // ---- compact list of the new det rows (ascending); every group's block builds its own copy
    if constexpr (SRC::kBlockLayout) {
        const int nd_ = n - src.ne;
        for (int i = tid; i < nd_; i += 256) newdet[i] = src.ne + i;
        if (tid == 0) s_wsum[4] = nd_;
        __syncthreads();
    } else {
        const int IT = (n + 255) / 256;
        const int i0 = tid * IT, i1 = min(n, i0 + IT);
        int cnt = 0;
        for (int i = i0; i < i1; ++i) cnt += src.edge(i) ? 0 : 1;
        const int lane = tid & 63, wave = tid >> 6;
        int inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        if (tid == 0) { int run = 0; for (int w = 0; w < 4; ++w) { const int t = s_wsum[w]; s_wsum[w] = run; run += t; } s_wsum[4] = run; }
        __syncthreads();
        int p = s_wsum[wave] + inc - cnt;
        for (int i = i0; i < i1; ++i)
            if (!src.edge(i)) newdet[p++] = i;
    }
    const int nd = s_wsum[4];
    if (tid == 0) newdet[n] = nd;
    // The new EDGE rows of x must be all-zero (utils/graph.py:148-149, 291-292 always builds them so): only det rows are
    // read here, while the reference would run whatever an edge row holds through Lin1 and into the batch statistics.
    // A non-zero edge row therefore marks the call invalid (status bit 64: NaN outputs, ValueError at the next check)
    // instead of diverging silently.
    if (SRC::kCheckEdgeRows) {
        bool bad = false;
        for (int idx = tid; idx < n * F; idx += 256) {
            const int i = idx / F, f = idx - i * F;
            if (src.edge(i) && src.row(i)[f0 + f] != 0.f) bad = true;
        }
        if (bad) atomicOr(&a.g.meta[2], 64);
    }
    // new edge rows start at zero (track_mpnn.py:61); new det rows are written below
    for (int idx = tid; !SRC::kBlockLayout && idx < n * (H / 4); idx += 256) {
        const int i = idx / (H / 4), c4 = idx % (H / 4);
        if (src.edge(i))
            *reinterpret_cast<float4*>(a.h + (size_t)(N_old + i) * GH + gi * H + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float* W1 = a.P.w1[gi];
    const float* b1 = a.P.b1[gi];
    float* ysave = a.ysave + (size_t)gi * (n > 0 ? n : 1) * H;
    const int c = tid % H, sub = tid / H;
    constexpr int NSUB = 256 / H;
    // ---- Lin1 on the det rows: y1[i][c] = b1[c] + sum_f x[i][f] W1[c][f]
    for (int i = sub; i < nd; i += NSUB) {
        const float* xr = src.row(newdet[i]) + f0;
        float acc = b1[c];
        for (int f = 0; f < F; ++f) acc = fmaf(xr[f], W1[c * F + f], acc);
        ysave[(size_t)i * H + c] = acc;
    }
    __syncthreads();
    // ---- statistics over ALL n new rows: the n - nd zero rows contribute Lin1(0) = b1 (models/track_mpnn.py:59)
    if (a.training) {
        const float cnt = (float)n, nz = (float)(n - nd);
        if (tid < H) {
            const float b = b1[tid];
            float sum = nz * b;
            for (int i = 0; i < nd; ++i) sum += ysave[(size_t)i * H + tid];
            const float m = sum / cnt;
            float sq = nz * (b - m) * (b - m);
            for (int i = 0; i < nd; ++i) { const float d = ysave[(size_t)i * H + tid] - m; sq += d * d; }
            const float var = sq / cnt;
            s_mean[tid] = m;
            s_rstd[tid] = rsqrtf(var + BN_EPS_S);
            float* rm = a.P.run_mean[gi];
            float* rv = a.P.run_var[gi];
            rm[tid] = (1.0f - BN_MOM_S) * rm[tid] + BN_MOM_S * m;
            rv[tid] = (1.0f - BN_MOM_S) * rv[tid] + BN_MOM_S * (var * (cnt / (cnt - 1.0f)));
            if (tid == 0 && a.P.num_batches_tracked[gi]) a.P.num_batches_tracked[gi][0] += 1;
        }
    } else if (tid < H) {
        s_mean[tid] = a.P.run_mean[gi][tid];
        s_rstd[tid] = rsqrtf(a.P.run_var[gi][tid] + BN_EPS_S);
    }
    // W2 transposed into LDS: s_w2t[k][c] = W2[c][k]
    const float* W2 = a.P.w2[gi];
    for (int idx = tid; idx < H * H; idx += 256) { const int cc = idx / H, k = idx % H; s_w2t[k * (H + 1) + cc] = W2[idx]; }
    __syncthreads();
    if (tid < H) {
        a.mean[(size_t)gi * H + tid] = s_mean[tid];
        a.rstd[(size_t)gi * H + tid] = s_rstd[tid];
    }
    // ---- a = relu(gamma yhat + beta) ; out = a W2^T + b2 -> h[new det rows], in chunks of CH rows
    constexpr int CH = 64;
    const float gam = a.P.gamma[gi][c], bet = a.P.beta[gi][c], b2 = a.P.b2[gi][c];
    for (int i0 = 0; i0 < nd; i0 += CH) {
        const int rows = min(CH, nd - i0);
        for (int i = sub; i < rows; i += NSUB) {
            const float yh = (ysave[(size_t)(i0 + i) * H + c] - s_mean[c]) * s_rstd[c];
            s_a[i * (H + 1) + c] = fmaxf(yh * gam + bet, 0.f);
        }
        __syncthreads();
        for (int i = sub; i < rows; i += NSUB) {
            float acc = b2;
#pragma unroll 8
            for (int k = 0; k < H; ++k) acc = fmaf(s_a[i * (H + 1) + k], s_w2t[k * (H + 1) + c], acc);
            a.h[(size_t)(N_old + newdet[i0 + i]) * GH + gi * H + c] = acc;
        }
        __syncthreads();
    }
}

}  // namespace tmpnn
