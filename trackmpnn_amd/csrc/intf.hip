// Input transform of the NEW det rows (models/track_mpnn.py:45-52, 59-61: Linear -> BatchNorm1d -> ReLU -> Linear with one
// BatchNorm segment per tracking window) in ONE launch per direction, for batches of SHORT segments.
//
// The staged form (csrc/dense.hip) runs the transform as a dozen generic launches forward and backward (two small
// GEMMs, per-segment statistics, element-wise passes, column sums, split-K weight gradients): ~23 launches per forward
// call and its backward, 3 ms of a 31 ms C2 step for 25 MB of data (round 2).  A window contributes a handful of new
// det rows per call (KITTI: ~6), so every per-segment quantity -- batch statistics with the analytic zero-row terms, the
// BatchNorm backward's two segment sums, the gradient of the all-zero edge rows -- is LOCAL to a workgroup that owns whole
// segments.  Here a workgroup owns IT_SPB consecutive segments and walks them in chunks of whole segments (<= IT_CH rows):
//
//   forward   Lin1 into LDS (+ y_save) -> per-segment mean / rstd -> normalise + ReLU -> Lin2 (64 x 64 weights resident in
//             LDS, four rows per thread in registers) -> h[new det rows]
//   backward  d_out gathered, a / yhat recomputed from y_save -> dW2, db2 (registers, whole block) -> d_out W2 -> ReLU mask,
//             dgamma / dbeta -> the two segment sums -> dy, dy0 -> dW1, db1, d_x (det rows), d_x (zero rows, per segment)
//             -> ONE slab of parameter gradients per workgroup; k_it_reduce adds the slabs into the six gradient buffers in
//             a fixed order (no float atomics).
//
// Entry points tmpnn_input_tf_fwd / _bwd: same arguments and results as tmpnn_input_bn_fwd / _bwd plus `max_seg_rows`
// (the host knows the longest segment of its plan); callers fall back to the staged form for long segments (one window
// of thousands of dets is ONE segment: BASELINE C5) or wide inputs.
#include "common.h"
#include <stdlib.h>

namespace tmpnn {

static constexpr float IT_EPS = 1e-5f;
static constexpr int IT_CH = 128;       // rows per chunk (whole segments)
static constexpr int IT_SPB = 32;       // segments per group of a persistent workgroup (KITTI-shaped: ~190 det rows = two chunks)
static constexpr int IT_FMAX = 128;     // widest input group (vis: 128 columns)
static constexpr int IT_NT = 1024;      // threads per workgroup: the phases are chains of LDS round trips, hidden only by
                                        // other waves (measured: 256 threads, one wave per SIMD, 316 us per backward launch)

struct ItFwdArgs {
    const float* x; const int64_t* x_rows; int ld_x; int F; int nd;
    const int32_t* seg_ptr; const int32_t* seg_cnt; const int32_t* seg_of_det; int S; int training;
    const float* w1; const float* b1; const float* gamma; const float* beta;
    const float* run_mean; const float* run_var; const float* w2; const float* b2;
    float* y_save; float* mean; float* rstd; const int32_t* out_row; float* h_new; int ld_h;
};

// the chunk [sa, sb) of whole segments starting at sa with at most IT_CH det rows (at least one segment), from the
// workgroup's LDS copy of its seg_ptr entries (s_sp[j] = seg_ptr[s0 + j]: a walk over global memory costs a dependent
// round trip per segment)
__device__ __forceinline__ int it_chunk_end(const int* s_sp, int s0, int sa, int s1) {
    const int base = s_sp[sa - s0];
    int sb = sa + 1;
    while (sb < s1 && s_sp[sb + 1 - s0] - base <= IT_CH) ++sb;
    return sb;
}

template <int H>
__global__ __launch_bounds__(IT_NT) void k_it_fwd(ItFwdArgs a) {
    constexpr int NSUB = IT_NT / H, LD = H + 4;
    extern __shared__ __attribute__((aligned(16))) float it_lds[];
    float* s_w2 = it_lds;                          // [H][LD]   W2[c][k]
    float* s_y = s_w2 + H * LD;                    // [IT_CH][LD]
    float* s_a = s_y + IT_CH * LD;                 // [IT_CH][LD]
    float* s_mean = s_a + IT_CH * LD;              // [IT_SPB][H]
    float* s_rstd = s_mean + IT_SPB * H;           // [IT_SPB][H]
    float* s_w1 = s_rstd + IT_SPB * H;             // [H][F + 1]  (F > 16)  |  x chunk [IT_CH][16] (F <= 16)
    float* s_x = s_w1;
    const int tid = threadIdx.x, c = tid % H, sub = tid / H;
    const int F = a.F;
    for (int i = tid; i < H * H; i += IT_NT) s_w2[(i / H) * LD + (i % H)] = a.w2[i];
    float w1r[16];
    if (F <= 16) {
#pragma unroll
        for (int f = 0; f < 16; ++f) w1r[f] = f < F ? a.w1[c * F + f] : 0.f;
    } else {
        for (int i = tid; i < H * F; i += IT_NT) s_w1[(i / F) * (F + 1) + (i % F)] = a.w1[i];
    }
    const float b1 = a.b1[c], gam = a.gamma[c], bet = a.beta[c], b2 = a.b2[c];
    __shared__ int s_sp[IT_SPB + 1];
    if (!a.training) {
        if (sub == 0) {
            const float m = a.run_mean[c], r = rsqrtf(a.run_var[c] + IT_EPS);
            s_mean[c] = m; s_rstd[c] = r;
            if (blockIdx.x == 0) { a.mean[c] = m; a.rstd[c] = r; }
        }
    }
    // a workgroup is persistent: it walks groups of IT_SPB segments (eval: of IT_CH rows), the weights loaded once
    const int ngroups = a.training ? (a.S + IT_SPB - 1) / IT_SPB : (a.nd + IT_CH - 1) / IT_CH;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    int s0, s1;
    if (a.training) { s0 = grp * IT_SPB; s1 = min(a.S, s0 + IT_SPB); }
    else { s0 = grp; s1 = s0 + 1; }                   // eval: "segment" = IT_CH consecutive rows, running statistics
    __syncthreads();                                  // (the previous group's readers of s_sp are done)
    if (a.training && tid <= s1 - s0) s_sp[tid] = a.seg_ptr[s0 + tid];
    __syncthreads();
    for (int sa = s0; sa < s1;) {
        int sb, ra, rb;
        if (a.training) { sb = it_chunk_end(s_sp, s0, sa, s1); ra = s_sp[sa - s0]; rb = s_sp[sb - s0]; }
        else { sb = s1; ra = sa * IT_CH; rb = min(a.nd, ra + IT_CH); }
        const int nr = rb - ra;
        // ---- Lin1 (narrow groups: the chunk's x rows go through LDS -- one coalesced pass instead of a dependent global
        //      load per row and thread)
        if (F <= 16) {
            for (int t = tid; t < nr * 16; t += IT_NT) {
                const int i = t >> 4, f = t & 15;
                const size_t xrow = a.x_rows ? (size_t)a.x_rows[ra + i] : (size_t)(ra + i);
                s_x[t] = f < F ? a.x[xrow * a.ld_x + f] : 0.f;
            }
            __syncthreads();
        }
        for (int i = sub; i < nr; i += NSUB) {
            const float* xr = a.x + ((F > 16 && a.x_rows) ? (size_t)a.x_rows[ra + i] : (size_t)(ra + i)) * a.ld_x;
            float acc = b1;
            if (F <= 16) {
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4) {
                    const float4 xv = *reinterpret_cast<const float4*>(s_x + i * 16 + 4 * f4);
                    acc = fmaf(xv.x, w1r[4 * f4], acc); acc = fmaf(xv.y, w1r[4 * f4 + 1], acc);
                    acc = fmaf(xv.z, w1r[4 * f4 + 2], acc); acc = fmaf(xv.w, w1r[4 * f4 + 3], acc);
                }
            } else {
                for (int f = 0; f < F; ++f) acc = fmaf(xr[f], s_w1[c * (F + 1) + f], acc);
            }
            s_y[i * LD + c] = acc;
            a.y_save[(size_t)(ra + i) * H + c] = acc;
        }
        __syncthreads();
        // ---- batch statistics per segment over ALL its new rows: the zero (edge) rows contribute Lin1(0) = b1
        if (a.training) {
            for (int s = sa + sub; s < sb; s += NSUB) {
                const int p0 = s_sp[s - s0] - ra, p1 = s_sp[s + 1 - s0] - ra;
                const float cnt = (float)a.seg_cnt[s], nz = cnt - (float)(p1 - p0);
                float sum = nz * b1;
                for (int i = p0; i < p1; ++i) sum += s_y[i * LD + c];
                const float m = sum / cnt;
                float sq = nz * (b1 - m) * (b1 - m);
                for (int i = p0; i < p1; ++i) { const float d = s_y[i * LD + c] - m; sq += d * d; }
                const float r = rsqrtf(sq / cnt + IT_EPS);
                s_mean[(s - sa) * H + c] = m; s_rstd[(s - sa) * H + c] = r;
                a.mean[(size_t)s * H + c] = m; a.rstd[(size_t)s * H + c] = r;
            }
            __syncthreads();
        }
        // ---- normalise + ReLU (a row's segment from the LDS copy of seg_ptr: rows per segment are few)
        for (int i = sub; i < nr; i += NSUB) {
            int sl = 0;
            if (a.training) { while (s_sp[sa - s0 + sl + 1] - ra <= i) ++sl; }
            const float yh = (s_y[i * LD + c] - s_mean[sl * H + c]) * s_rstd[sl * H + c];
            s_a[i * LD + c] = fmaxf(yh * gam + bet, 0.f);
        }
        __syncthreads();
        // ---- Lin2: out[i][c] = b2[c] + sum_k a[i][k] W2[c][k], four rows per thread
        for (int i0 = sub; i0 < nr; i0 += 4 * NSUB) {
            float acc[4] = {b2, b2, b2, b2};
            int rr[4], orow[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) rr[t] = min(i0 + t * NSUB, nr - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) orow[t] = a.out_row[ra + rr[t]];        // (requested before the products, used after)
#pragma unroll 4
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const float4 w = *reinterpret_cast<const float4*>(s_w2 + c * LD + 4 * k4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 av = *reinterpret_cast<const float4*>(s_a + rr[t] * LD + 4 * k4);
                    acc[t] = fmaf(av.x, w.x, acc[t]); acc[t] = fmaf(av.y, w.y, acc[t]);
                    acc[t] = fmaf(av.z, w.z, acc[t]); acc[t] = fmaf(av.w, w.w, acc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + t * NSUB;
                if (i < nr) a.h_new[(size_t)orow[t] * a.ld_h + c] = acc[t];
            }
        }
        __syncthreads();
        sa = sb;
    }
    }
}

struct ItBwdArgs {
    const float* x; const int64_t* x_rows; int ld_x; int F; int nd;
    const int32_t* seg_ptr; const int32_t* seg_cnt; const int32_t* seg_of_det; int S; int training;
    const float* w1; const float* b1; const float* gamma; const float* beta; const float* w2;
    const float* y_save; const float* mean; const float* rstd; const int32_t* out_row;
    const float* d_h; int ld_dh; float* d_xdet; int ld_dx; float* d_xzero;
    float* slabs; int slab_floats;          // per workgroup: dW2 [H][H] | dW1 [H][F] | db2 | dgamma | dbeta | db1
};

// FPT: dW1 columns per thread (f = sub, sub + NSUB, ...): 1 covers the 2d / temp groups (F <= 16 <= NSUB), IT_FMAX / NSUB the
// vis group
template <int H, int FPT>
__global__ __launch_bounds__(IT_NT) void k_it_bwd(ItBwdArgs a) {
    constexpr int NSUB = IT_NT / H, LD = H + 4, KPT = H / NSUB;       // KPT: dW2 columns per thread
    extern __shared__ __attribute__((aligned(16))) float it_lds[];
    float* s_w2 = it_lds;                          // [H][LD]   W2[c][k]   (da = d W2: thread k walks c -> column reads)
    float* s_d = s_w2 + H * LD;                    // [IT_CH][LD]  d_out
    float* s_a = s_d + IT_CH * LD;                 // [IT_CH][LD]  a, then d_yhat, then dy
    float* s_yh = s_a + IT_CH * LD;                // [IT_CH][LD]  yhat
    float* s_dy0 = s_yh + IT_CH * LD;              // [IT_SPB][H]  gradient of a segment's zero rows (pre-Lin1)
    float* s_red = s_dy0 + IT_SPB * H;             // [4][IT_NT]   end-of-block combine of the vector gradients
    float* s_x = s_red + 4 * IT_NT;                  // [IT_CH][16]  x chunk (F <= 16)
    float* s_w1 = s_x + IT_CH * 16;                // [H][17]      W1 (F <= 16)
    const int tid = threadIdx.x, c = tid % H, sub = tid / H;
    const int F = a.F;
    // W2 stored TRANSPOSED for the d W2 product: s_w2[k][c] = W2[c][k]
    for (int i = tid; i < H * H; i += IT_NT) s_w2[(i % H) * LD + (i / H)] = a.w2[i];
    if (F <= 16)
        for (int i = tid; i < H * 16; i += IT_NT) s_w1[(i >> 4) * 17 + (i & 15)] = (i & 15) < F ? a.w1[(i >> 4) * F + (i & 15)] : 0.f;
    const float b1 = a.b1[c], gam = a.gamma[c], bet = a.beta[c];
    float acc_w2[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) acc_w2[j] = 0.f;
    float acc_w1[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) acc_w1[j] = 0.f;
    float v_db2 = 0.f, v_dg = 0.f, v_dbt = 0.f, v_db1 = 0.f;
    __shared__ int s_sp[IT_SPB + 1];
    // persistent: groups of IT_SPB segments (eval: of IT_CH rows); the weight-gradient accumulators live in registers over
    // all of them and leave as ONE slab per workgroup
    const int ngroups = a.training ? (a.S + IT_SPB - 1) / IT_SPB : (a.nd + IT_CH - 1) / IT_CH;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    int s0, s1;
    if (a.training) { s0 = grp * IT_SPB; s1 = min(a.S, s0 + IT_SPB); }
    else { s0 = grp; s1 = s0 + 1; }
    __syncthreads();
    if (a.training && tid <= s1 - s0) s_sp[tid] = a.seg_ptr[s0 + tid];
    __syncthreads();
    for (int sa = s0; sa < s1;) {
        int sb, ra, rb;
        if (a.training) { sb = it_chunk_end(s_sp, s0, sa, s1); ra = s_sp[sa - s0]; rb = s_sp[sb - s0]; }
        else { sb = s1; ra = sa * IT_CH; rb = min(a.nd, ra + IT_CH); }
        const int nr = rb - ra;
        // ---- d_out (gathered), yhat and a (recomputed).  Every load of a thread's <= IT_CH / NSUB rows is requested
        //      before the first is used: the row id -> d_out chain is two dependent round trips, taken once, not per row
        {
            constexpr int RPT = IT_CH / NSUB;
            int orow[RPT], sg[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = min(sub + j * NSUB, nr - 1);
                orow[j] = a.out_row[ra + i];
                int sl = 0;
                if (a.training) { while (s_sp[sa - s0 + sl + 1] - ra <= i) ++sl; }
                sg[j] = a.training ? sa + sl : 0;
            }
            float dv[RPT], yv[RPT], mv[RPT], rv[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = min(sub + j * NSUB, nr - 1);
                dv[j] = a.d_h[(size_t)orow[j] * a.ld_dh + c];
                yv[j] = a.y_save[(size_t)(ra + i) * H + c];
                mv[j] = a.mean[(size_t)sg[j] * H + c];
                rv[j] = a.rstd[(size_t)sg[j] * H + c];
            }
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int i = sub + j * NSUB;
                if (i < nr) {
                    const float yh = (yv[j] - mv[j]) * rv[j];
                    s_yh[i * LD + c] = yh;
                    s_a[i * LD + c] = fmaxf(yh * gam + bet, 0.f);
                    s_d[i * LD + c] = dv[j];
                }
            }
        }
        if (F <= 16) {
            for (int t = tid; t < nr * 16; t += IT_NT) {
                const int i = t >> 4, f = t & 15;
                const size_t xrow = a.x_rows ? (size_t)a.x_rows[ra + i] : (size_t)(ra + i);
                s_x[t] = f < F ? a.x[xrow * a.ld_x + f] : 0.f;
            }
        }
        __syncthreads();
        // ---- dW2[c][k] += sum_i d_out[i][c] a[i][k]  (k = KPT sub .. + KPT) ; db2[c] += sum_i d_out[i][c]
#pragma unroll 8
        for (int i = 0; i < nr; ++i) {
            const float d = s_d[i * LD + c];
            if (sub == 0) v_db2 += d;
            if constexpr (KPT % 4 == 0) {
#pragma unroll
                for (int j4 = 0; j4 < KPT / 4; ++j4) {
                    const float4 av = *reinterpret_cast<const float4*>(s_a + i * LD + KPT * sub + 4 * j4);
                    acc_w2[4 * j4 + 0] = fmaf(d, av.x, acc_w2[4 * j4 + 0]); acc_w2[4 * j4 + 1] = fmaf(d, av.y, acc_w2[4 * j4 + 1]);
                    acc_w2[4 * j4 + 2] = fmaf(d, av.z, acc_w2[4 * j4 + 2]); acc_w2[4 * j4 + 3] = fmaf(d, av.w, acc_w2[4 * j4 + 3]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < KPT; ++j) acc_w2[j] = fmaf(d, s_a[i * LD + KPT * sub + j], acc_w2[j]);
            }
        }
        __syncthreads();
        // ---- da[i][k] = sum_c d_out[i][c] W2[c][k] (thread k = c-index), ReLU mask, dgamma / dbeta, d_yhat -> s_a
        for (int i0 = sub; i0 < nr; i0 += 4 * NSUB) {
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            int rr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) rr[t] = min(i0 + t * NSUB, nr - 1);
#pragma unroll 4
            for (int c4 = 0; c4 < H / 4; ++c4) {
                const float4 w = *reinterpret_cast<const float4*>(s_w2 + c * LD + 4 * c4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 dv = *reinterpret_cast<const float4*>(s_d + rr[t] * LD + 4 * c4);
                    da[t] = fmaf(dv.x, w.x, da[t]); da[t] = fmaf(dv.y, w.y, da[t]);
                    da[t] = fmaf(dv.z, w.z, da[t]); da[t] = fmaf(dv.w, w.w, da[t]);
                }
            }
            float dyh[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + t * NSUB;
                dyh[t] = 0.f;
                if (i < nr) {
                    const float dz = s_a[i * LD + c] > 0.f ? da[t] : 0.f;
                    v_dg = fmaf(dz, s_yh[i * LD + c], v_dg);
                    v_dbt += dz;
                    dyh[t] = dz * gam;
                }
            }
            // (this thread is the only reader of column c of its own rows from here on: the a values are overwritten)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = i0 + t * NSUB;
                if (i < nr) s_a[i * LD + c] = dyh[t];
            }
        }
        __syncthreads();
        // ---- BatchNorm backward per segment: dy = rstd (d_yhat - s1 / n - yhat s2 / n) on det rows; the segment's zero
        //      rows (yhat0 = (b1 - mean) rstd, d_yhat0 = 0) receive dy0 = rstd (- s1 / n - yhat0 s2 / n) each
        if (a.training) {
            for (int s = sa + sub; s < sb; s += NSUB) {
                const int p0 = s_sp[s - s0] - ra, p1 = s_sp[s + 1 - s0] - ra;
                const float cnt = (float)a.seg_cnt[s], nz = cnt - (float)(p1 - p0);
                const float m = a.mean[(size_t)s * H + c], r = a.rstd[(size_t)s * H + c];
                float q1 = 0.f, q2 = 0.f;
                for (int i = p0; i < p1; ++i) { const float g = s_a[i * LD + c]; q1 += g; q2 = fmaf(g, s_yh[i * LD + c], q2); }
                const float inv = 1.0f / cnt;
                for (int i = p0; i < p1; ++i) {
                    const float dy = r * (s_a[i * LD + c] - q1 * inv - s_yh[i * LD + c] * q2 * inv);
                    s_a[i * LD + c] = dy;
                    v_db1 += dy;
                }
                const float yh0 = (b1 - m) * r;
                const float dy0 = r * (-q1 * inv - yh0 * q2 * inv);
                v_db1 = fmaf(nz, dy0, v_db1);
                s_dy0[(s - sa) * H + c] = dy0;
            }
        } else {
            const float r = a.rstd[c];
            for (int i = sub; i < nr; i += NSUB) { const float dy = r * s_a[i * LD + c]; s_a[i * LD + c] = dy; v_db1 += dy; }
        }
        __syncthreads();
        // ---- dW1[k][f] += sum_i dy[i][k] x[i][f]   (thread k = c-index; f = sub, sub + NSUB, ...)
        if (F <= 16) {
#pragma unroll 4
            for (int i = 0; i < nr; ++i) {
                const float dy = s_a[i * LD + c];
#pragma unroll
                for (int j = 0; j < (FPT < 4 ? FPT : 4); ++j) {
                    const int f = sub + j * NSUB;
                    if (f < 16) acc_w1[j] = fmaf(dy, s_x[i * 16 + f], acc_w1[j]);     // (columns >= F hold zeros)
                }
            }
        } else {
            for (int i = 0; i < nr; ++i) {
                const float dy = s_a[i * LD + c];
                const float* xr = a.x + (a.x_rows ? (size_t)a.x_rows[ra + i] : (size_t)(ra + i)) * a.ld_x;
#pragma unroll
                for (int j = 0; j < FPT; ++j) {
                    const int f = sub + j * NSUB;
                    if (f < F) acc_w1[j] = fmaf(dy, xr[f], acc_w1[j]);
                }
            }
        }
        // ---- d_x of the det rows: d_xdet[i][f] = sum_k dy[i][k] W1[k][f] ; of the zero rows, per segment: dy0 W1
        if (a.d_xdet) {
            for (int t = tid; t < nr * F; t += IT_NT) {
                const int i = t / F, f = t - i * F;
                float acc = 0.f;
                if (F <= 16) {
#pragma unroll 8
                    for (int k = 0; k < H; ++k) acc = fmaf(s_a[i * LD + k], s_w1[k * 17 + f], acc);
                } else {
                    for (int k = 0; k < H; ++k) acc = fmaf(s_a[i * LD + k], a.w1[k * F + f], acc);
                }
                a.d_xdet[(size_t)(ra + i) * a.ld_dx + f] = acc;
            }
        }
        if (a.d_xzero && a.training) {
            const int ns = sb - sa;
            for (int t = tid; t < ns * F; t += IT_NT) {
                const int s = t / F, f = t - s * F;
                float acc = 0.f;
                if (F <= 16) {
#pragma unroll 8
                    for (int k = 0; k < H; ++k) acc = fmaf(s_dy0[s * H + k], s_w1[k * 17 + f], acc);
                } else {
                    for (int k = 0; k < H; ++k) acc = fmaf(s_dy0[s * H + k], a.w1[k * F + f], acc);
                }
                a.d_xzero[(size_t)(sa + s) * F + f] = acc;
            }
        }
        __syncthreads();
        sa = sb;
    }
    }
    // ---- this workgroup's slab
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_floats;
#pragma unroll
    for (int j = 0; j < KPT; ++j) slab[c * H + KPT * sub + j] = acc_w2[j];
    float* sl_w1 = slab + H * H;
#pragma unroll
    for (int j = 0; j < FPT; ++j) {
        const int f = sub + j * NSUB;
        if (f < F) sl_w1[c * F + f] = acc_w1[j];
    }
    s_red[0 * IT_NT + tid] = v_db2; s_red[1 * IT_NT + tid] = v_dg; s_red[2 * IT_NT + tid] = v_dbt; s_red[3 * IT_NT + tid] = v_db1;
    __syncthreads();
    if (sub == 0) {
        float* sl_v = sl_w1 + H * F;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = 0.f;
            for (int t = 0; t < NSUB; ++t) v += s_red[q * IT_NT + t * H + c];
            sl_v[q * H + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the same transform with TILES OWNED BY WAVES (training mode, H = 64, F <= 16, segments of <= ITW_R det rows).
// The workgroup form above walks a group's chunks one after another, every phase between two block-wide barriers, and its
// three 64 x 64 products run as scalar FMAs against LDS operands: 85-100 us forward and 150 us backward per call for ~100 k det
// rows (1.5 ms of a 28-ms C2 step).  Here a wave owns a tile -- consecutive whole segments of at most ITW_R det rows -- from its
// first load to its last store: lane = hidden column, so every per-column quantity of the transform (Lin1, the batch
// statistics with their analytic zero-row terms, normalise + ReLU, dgamma / dbeta, the BatchNorm backward's two segment sums)
// is lane-private, nothing waits at a barrier, and the eight waves of a workgroup are eight independent chains.  What crosses
// lanes goes through a wave-private LDS tile read with broadcast ds_read_b128: Lin2 (the lane keeps row c of W2 in 64 registers),
// and in the backward dW2^T (64 accumulators: the lane's column k of a against a broadcast row of d_out) fused with
// da = d_out W2 (the lane keeps column k of W2) -- one pass over the d_out rows for both.  Results equal the workgroup form's to
// rounding (same formulas; the products sum in another order).  Gradients with respect to x (d_xdet / d_xzero) are not formed
// here: callers that ask for them take the workgroup form.
static constexpr int ITW_R = 32;        // det rows of a tile
static constexpr int ITW_MS = 8;        // segments of a tile (their statistics ride in registers)
static constexpr int ITW_NW = 8;        // waves of a workgroup
static constexpr int ITW_SPB = 64;      // segments of a workgroup's group: one tile boundary search per lane
static constexpr int ITW_FWD_WAVE_FLOATS = ITW_R * 64 + ITW_R * 16;              // a tile | x rows
static constexpr int ITW_BWD_WAVE_FLOATS = 2 * ITW_R * 64 + ITW_R * 16;          // yhat | d_out -> d_yhat -> dy | x rows

#define ITW_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// the tile [sa, sb) of whole segments starting at sa: at most ITW_R det rows and ITW_MS segments (at least one segment)
__device__ __forceinline__ int itw_chunk_end(const int* s_sp, int s0, int sa, int s1) {
    const int base = s_sp[sa - s0];
    int sb = sa + 1;
    while (sb < s1 && sb - sa < ITW_MS && s_sp[sb + 1 - s0] - base <= ITW_R) ++sb;
    return sb;
}

// Every global round trip of a tile is taken ONCE, for all its rows together: a loop that loads, waits and stores row by row
// pays the memory latency per row (the first form of these kernels: 80 / 110 us per call, most of it in such loops).
// x rows [ra, ra + nr) of the tile -> s_x[i][16] (columns >= F zero): nr * 16 <= 512 elements = 8 per lane; the row ids
// (x_rows) of all eight are requested, then the eight values, then the LDS writes
__device__ __forceinline__ void itw_stage_x(const float* __restrict__ x, const int64_t* __restrict__ x_rows, int ld_x, int F,
                                            int ra, int nr, int lane, float* s_x) {
    size_t xrow[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = lane + 64 * q, i = min(t >> 4, nr - 1);
        xrow[q] = x_rows ? (size_t)x_rows[ra + i] : (size_t)(ra + i);
    }
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int f = lane & 15;
        v[q] = x[xrow[q] * ld_x + min(f, F - 1)];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = lane + 64 * q;
        if (t < nr * 16) s_x[t] = (lane & 15) < F ? v[q] : 0.f;
    }
}

// sum over rows [p0, p1) of the lane's column of a [.][64] tile, eight reads in flight
__device__ __forceinline__ float itw_col_sum(const float* col, int p0, int p1) {
    float sum = 0.f;
    for (int b = p0; b < p1; b += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = col[min(b + t, p1 - 1) * 64];
#pragma unroll
        for (int t = 0; t < 8; ++t) sum += (b + t < p1) ? v[t] : 0.f;
    }
    return sum;
}

__global__ __launch_bounds__(ITW_NW * 64) void k_itw_fwd(ItFwdArgs a) {
    constexpr int H = 64;
    extern __shared__ __attribute__((aligned(16))) float it_lds[];
    __shared__ int s_sp[ITW_SPB + 1];
    const int tid = threadIdx.x, c = tid & 63, wave = tid >> 6;
    float* s_a = it_lds + wave * ITW_FWD_WAVE_FLOATS;       // [ITW_R][64]: y, then a
    float* s_x = s_a + ITW_R * 64;                          // [ITW_R][16]
    float* col = s_a + c;                                   // the lane's column
    const int F = a.F;
    float w1r[16], w2r[64];
#pragma unroll
    for (int f = 0; f < 16; ++f) w1r[f] = f < F ? a.w1[c * F + f] : 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const float4 w = *reinterpret_cast<const float4*>(a.w2 + c * H + 4 * k4);
        w2r[4 * k4] = w.x; w2r[4 * k4 + 1] = w.y; w2r[4 * k4 + 2] = w.z; w2r[4 * k4 + 3] = w.w;
    }
    const float b1 = a.b1[c], gam = a.gamma[c], bet = a.beta[c], b2 = a.b2[c];
    const int ngroups = (a.S + ITW_SPB - 1) / ITW_SPB;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int s0 = grp * ITW_SPB, s1 = min(a.S, s0 + ITW_SPB);
        __syncthreads();                                  // (the previous group's readers of s_sp are done)
        if (tid <= s1 - s0) s_sp[tid] = a.seg_ptr[s0 + tid];
        __syncthreads();
        // lane l: where a tile that starts at segment s0 + l ends; the wave then follows the chain from segment s0 (the tiles
        // are dealt to the waves round robin: every wave finds the same chain)
        const int nxt_l = (s0 + c < s1) ? itw_chunk_end(s_sp, s0, s0 + c, s1) : s1;
        int j = 0;
        for (int sa = s0; sa < s1; ++j) {
            const int sb = __shfl(nxt_l, sa - s0, 64);
            if ((j & (ITW_NW - 1)) == wave) {
                const int ra = s_sp[sa - s0], nr = s_sp[sb - s0] - ra;
                int orow_l = 0, cnt_l = 1;
                if (c < nr) orow_l = a.out_row[ra + c];              // (requested first, used last)
                if (sa + c < sb) cnt_l = a.seg_cnt[sa + c];
                if (nr > 0) {
                    itw_stage_x(a.x, a.x_rows, a.ld_x, F, ra, nr, c, s_x);
                    ITW_LDS_SYNC();
                    // ---- Lin1 -> y (own column of the tile) + y_save, two rows per pass
                    for (int i = 0; i < nr; i += 2) {
                        const int i1 = min(i + 1, nr - 1);
                        float acc0 = b1, acc1 = b1;
#pragma unroll
                        for (int f4 = 0; f4 < 4; ++f4) {
                            const float4 x0 = *reinterpret_cast<const float4*>(s_x + i * 16 + 4 * f4);
                            const float4 x1 = *reinterpret_cast<const float4*>(s_x + i1 * 16 + 4 * f4);
                            acc0 = fmaf(x0.x, w1r[4 * f4], acc0); acc0 = fmaf(x0.y, w1r[4 * f4 + 1], acc0);
                            acc0 = fmaf(x0.z, w1r[4 * f4 + 2], acc0); acc0 = fmaf(x0.w, w1r[4 * f4 + 3], acc0);
                            acc1 = fmaf(x1.x, w1r[4 * f4], acc1); acc1 = fmaf(x1.y, w1r[4 * f4 + 1], acc1);
                            acc1 = fmaf(x1.z, w1r[4 * f4 + 2], acc1); acc1 = fmaf(x1.w, w1r[4 * f4 + 3], acc1);
                        }
                        col[i * 64] = acc0;
                        a.y_save[(size_t)(ra + i) * H + c] = acc0;
                        if (i + 1 < nr) { col[i1 * 64] = acc1; a.y_save[(size_t)(ra + i1) * H + c] = acc1; }
                    }
                }
                // ---- per segment: batch statistics over ALL its new rows (the zero rows contribute Lin1(0) = b1), then
                //      normalise + ReLU in place (lane-private column: no synchronisation)
                for (int s = sa; s < sb; ++s) {
                    const int p0 = s_sp[s - s0] - ra, p1 = s_sp[s + 1 - s0] - ra;
                    const float cnt = (float)__shfl(cnt_l, s - sa, 64), nz = cnt - (float)(p1 - p0);
                    const float m = (nz * b1 + itw_col_sum(col, p0, p1)) / cnt;
                    float sq = nz * (b1 - m) * (b1 - m);
                    for (int b = p0; b < p1; b += 8) {
                        float v[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = col[min(b + t, p1 - 1) * 64];
#pragma unroll
                        for (int t = 0; t < 8; ++t) { const float d = v[t] - m; sq += (b + t < p1) ? d * d : 0.f; }
                    }
                    const float r = rsqrtf(sq / cnt + IT_EPS);
                    a.mean[(size_t)s * H + c] = m; a.rstd[(size_t)s * H + c] = r;
                    for (int b = p0; b < p1; b += 8) {
                        float v[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = col[min(b + t, p1 - 1) * 64];
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (b + t < p1) col[(b + t) * 64] = fmaxf((v[t] - m) * r * gam + bet, 0.f);
                    }
                }
                ITW_LDS_SYNC();
                // ---- Lin2: out[i][c] = b2[c] + sum_k a[i][k] W2[c][k]; two rows per pass (independent chains; four rows made
                //      hipcc hoist 64 operand reads and spill)
                for (int i = 0; i < nr; i += 2) {
                    const int i1 = min(i + 1, nr - 1);
                    float acc0 = b2, acc1 = b2;
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(s_a + i * 64 + 4 * k4);
                        const float4 a1 = *reinterpret_cast<const float4*>(s_a + i1 * 64 + 4 * k4);
                        acc0 = fmaf(a0.x, w2r[4 * k4], acc0); acc0 = fmaf(a0.y, w2r[4 * k4 + 1], acc0);
                        acc0 = fmaf(a0.z, w2r[4 * k4 + 2], acc0); acc0 = fmaf(a0.w, w2r[4 * k4 + 3], acc0);
                        acc1 = fmaf(a1.x, w2r[4 * k4], acc1); acc1 = fmaf(a1.y, w2r[4 * k4 + 1], acc1);
                        acc1 = fmaf(a1.z, w2r[4 * k4 + 2], acc1); acc1 = fmaf(a1.w, w2r[4 * k4 + 3], acc1);
                    }
                    const int o0 = __shfl(orow_l, i, 64), o1 = __shfl(orow_l, i1, 64);
                    a.h_new[(size_t)o0 * a.ld_h + c] = acc0;
                    if (i + 1 < nr) a.h_new[(size_t)o1 * a.ld_h + c] = acc1;
                }
                ITW_LDS_SYNC();                                      // (the tile is free for the wave's next one)
            }
            sa = sb;
        }
    }
}

__global__ __launch_bounds__(ITW_NW * 64) void k_itw_bwd(ItBwdArgs a) {
    constexpr int H = 64;
    extern __shared__ __attribute__((aligned(16))) float it_lds[];
    __shared__ int s_sp[ITW_SPB + 1];
    const int tid = threadIdx.x, c = tid & 63, wave = tid >> 6;       // c: the lane's hidden column (k of a / yhat / dy)
    float* s_yh = it_lds + wave * ITW_BWD_WAVE_FLOATS;      // [ITW_R][64] yhat (lane-private columns)
    float* s_d = s_yh + ITW_R * 64;                         // [ITW_R][64] d_out rows (broadcast reads), then d_yhat, then dy
    float* s_x = s_d + ITW_R * 64;                          // [ITW_R][16]
    float* cyh = s_yh + c;
    float* cd = s_d + c;
    const int F = a.F;
    float w2c[64];                                          // column c of W2: w2c[j] = W2[j][c]
#pragma unroll
    for (int j = 0; j < 64; ++j) w2c[j] = a.w2[j * H + c];
    const float b1 = a.b1[c], gam = a.gamma[c], bet = a.beta[c];
    float acc_w2t[64];                                      // acc_w2t[j] = dW2[j][c]
#pragma unroll
    for (int j = 0; j < 64; ++j) acc_w2t[j] = 0.f;
    float acc_w1[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc_w1[f] = 0.f;
    float v_dg = 0.f, v_dbt = 0.f, v_db1 = 0.f;
    float v_db2 = 0.f;                                      // (lane c sums d_out[.][c])
    const int ngroups = (a.S + ITW_SPB - 1) / ITW_SPB;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int s0 = grp * ITW_SPB, s1 = min(a.S, s0 + ITW_SPB);
        __syncthreads();
        if (tid <= s1 - s0) s_sp[tid] = a.seg_ptr[s0 + tid];
        __syncthreads();
        const int nxt_l = (s0 + c < s1) ? itw_chunk_end(s_sp, s0, s0 + c, s1) : s1;      // (see k_itw_fwd)
        int j = 0;
        for (int sa = s0; sa < s1; ++j) {
            const int sb = __shfl(nxt_l, sa - s0, 64);
            if ((j & (ITW_NW - 1)) == wave) {
                const int ra = s_sp[sa - s0], nr = s_sp[sb - s0] - ra;
                int orow_l = 0, cnt_l = 1;
                if (c < nr) orow_l = a.out_row[ra + c];
                if (sa + c < sb) cnt_l = a.seg_cnt[sa + c];
                // the statistics of the tile's (<= ITW_MS) segments: requested together, kept for both uses below
                float mseg[ITW_MS], rseg[ITW_MS];
#pragma unroll
                for (int t = 0; t < ITW_MS; ++t) {
                    const int s = min(sa + t, sb - 1);
                    mseg[t] = a.mean[(size_t)s * H + c];
                    rseg[t] = a.rstd[(size_t)s * H + c];
                }
                if (nr > 0) itw_stage_x(a.x, a.x_rows, a.ld_x, F, ra, nr, c, s_x);
                // ---- d_out rows (gathered) -> s_d, y -> s_yh: eight rows' loads in flight per pass
                for (int b = 0; b < nr; b += 8) {
                    float dv[8], yv[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int i = min(b + t, nr - 1);
                        const int orow = __shfl(orow_l, i, 64);
                        dv[t] = a.d_h[(size_t)orow * a.ld_dh + c];
                        yv[t] = a.y_save[(size_t)(ra + i) * H + c];
                    }
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (b + t < nr) { cd[(b + t) * 64] = dv[t]; cyh[(b + t) * 64] = yv[t]; v_db2 += dv[t]; }
                    }
                }
                // ---- yhat = (y - mean) rstd of the row's segment, in place
#pragma unroll
                for (int t = 0; t < ITW_MS; ++t) {
                    if (sa + t < sb) {
                        const int p0 = s_sp[sa + t - s0] - ra, p1 = s_sp[sa + t + 1 - s0] - ra;
                        for (int b = p0; b < p1; b += 8) {
                            float v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) v[u] = cyh[min(b + u, p1 - 1) * 64];
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (b + u < p1) cyh[(b + u) * 64] = (v[u] - mseg[t]) * rseg[t];
                        }
                    }
                }
                ITW_LDS_SYNC();
                // ---- one pass over the d_out rows: dW2[j][c] += d_out[i][j] a[i][c] and da[i][c] = sum_j d_out[i][j] W2[j][c];
                //      then the ReLU mask, dgamma / dbeta and d_yhat (into the lane's own column of s_d: row i of s_d has been
                //      read by every lane of the wave when its iteration ends)
                for (int i = 0; i < nr; ++i) {
                    const float yh = cyh[i * 64];
                    const float av = fmaxf(yh * gam + bet, 0.f);
                    float da0 = 0.f, da1 = 0.f;
#pragma unroll
                    for (int j4 = 0; j4 < 16; ++j4) {
                        const float4 dv = *reinterpret_cast<const float4*>(s_d + i * 64 + 4 * j4);
                        acc_w2t[4 * j4] = fmaf(dv.x, av, acc_w2t[4 * j4]); acc_w2t[4 * j4 + 1] = fmaf(dv.y, av, acc_w2t[4 * j4 + 1]);
                        acc_w2t[4 * j4 + 2] = fmaf(dv.z, av, acc_w2t[4 * j4 + 2]); acc_w2t[4 * j4 + 3] = fmaf(dv.w, av, acc_w2t[4 * j4 + 3]);
                        da0 = fmaf(dv.x, w2c[4 * j4], da0); da1 = fmaf(dv.y, w2c[4 * j4 + 1], da1);
                        da0 = fmaf(dv.z, w2c[4 * j4 + 2], da0); da1 = fmaf(dv.w, w2c[4 * j4 + 3], da1);
                    }
                    const float dz = av > 0.f ? da0 + da1 : 0.f;
                    v_dg = fmaf(dz, yh, v_dg);
                    v_dbt += dz;
                    ITW_LDS_SYNC();                                  // (every lane's reads of row i are back)
                    cd[i * 64] = dz * gam;
                }
                // ---- BatchNorm backward per segment: dy = rstd (d_yhat - q1 / n - yhat q2 / n) on the det rows; each of the
                //      segment's zero rows (yhat0 = (b1 - mean) rstd, d_yhat0 = 0) receives dy0 = rstd (- q1 / n - yhat0 q2 / n)
#pragma unroll
                for (int t = 0; t < ITW_MS; ++t) {
                    if (sa + t < sb) {
                        const int p0 = s_sp[sa + t - s0] - ra, p1 = s_sp[sa + t + 1 - s0] - ra;
                        const float cnt = (float)__shfl(cnt_l, t, 64), nz = cnt - (float)(p1 - p0);
                        const float m = mseg[t], r = rseg[t];
                        float q1 = 0.f, q2 = 0.f;
                        for (int b = p0; b < p1; b += 8) {
                            float g[8], y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const int i = min(b + u, p1 - 1); g[u] = cd[i * 64]; y[u] = cyh[i * 64]; }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (b + u < p1) { q1 += g[u]; q2 = fmaf(g[u], y[u], q2); }
                        }
                        const float inv = 1.0f / cnt;
                        for (int b = p0; b < p1; b += 8) {
                            float g[8], y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const int i = min(b + u, p1 - 1); g[u] = cd[i * 64]; y[u] = cyh[i * 64]; }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                if (b + u < p1) {
                                    const float dy = r * (g[u] - q1 * inv - y[u] * q2 * inv);
                                    cd[(b + u) * 64] = dy;
                                    v_db1 += dy;
                                }
                            }
                        }
                        const float yh0 = (b1 - m) * r;
                        const float dy0 = r * (-q1 * inv - yh0 * q2 * inv);
                        v_db1 = fmaf(nz, dy0, v_db1);
                    }
                }
                // ---- dW1[c][f] += sum_i dy[i][c] x[i][f]   (columns >= F of s_x hold zeros), two rows per pass
                for (int i = 0; i < nr; i += 2) {
                    const int i1 = min(i + 1, nr - 1);
                    const float dy0 = cd[i * 64], dy1 = (i + 1 < nr) ? cd[i1 * 64] : 0.f;
#pragma unroll
                    for (int f4 = 0; f4 < 4; ++f4) {
                        const float4 x0 = *reinterpret_cast<const float4*>(s_x + i * 16 + 4 * f4);
                        const float4 x1 = *reinterpret_cast<const float4*>(s_x + i1 * 16 + 4 * f4);
                        acc_w1[4 * f4] = fmaf(dy0, x0.x, acc_w1[4 * f4]); acc_w1[4 * f4 + 1] = fmaf(dy0, x0.y, acc_w1[4 * f4 + 1]);
                        acc_w1[4 * f4 + 2] = fmaf(dy0, x0.z, acc_w1[4 * f4 + 2]); acc_w1[4 * f4 + 3] = fmaf(dy0, x0.w, acc_w1[4 * f4 + 3]);
                        acc_w1[4 * f4] = fmaf(dy1, x1.x, acc_w1[4 * f4]); acc_w1[4 * f4 + 1] = fmaf(dy1, x1.y, acc_w1[4 * f4 + 1]);
                        acc_w1[4 * f4 + 2] = fmaf(dy1, x1.z, acc_w1[4 * f4 + 2]); acc_w1[4 * f4 + 3] = fmaf(dy1, x1.w, acc_w1[4 * f4 + 3]);
                    }
                }
                ITW_LDS_SYNC();
            }
            sa = sb;
        }
    }
    // ---- this workgroup's slab: the eight waves' accumulators combined through LDS in wave order (deterministic)
    __syncthreads();
    float* s_all = it_lds;                                  // per wave an area of ITW_BWD_WAVE_FLOATS >= 64 * 64 floats
    float* mine = s_all + wave * ITW_BWD_WAVE_FLOATS;
#pragma unroll
    for (int j = 0; j < 64; ++j) mine[j * 64 + c] = acc_w2t[j];
    __syncthreads();
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_floats;
    for (int e = tid; e < H * H; e += ITW_NW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ITW_NW; ++w) v += s_all[w * ITW_BWD_WAVE_FLOATS + e];
        slab[e] = v;                                        // e = j * 64 + c = dW2[j][c]
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 16; ++f) mine[c * 16 + f] = acc_w1[f];
    mine[1024 + c] = v_db2; mine[1088 + c] = v_dg; mine[1152 + c] = v_dbt; mine[1216 + c] = v_db1;
    __syncthreads();
    float* sl_w1 = slab + H * H;
    for (int e = tid; e < H * F; e += ITW_NW * 64) {
        const int k = e / F, f = e - k * F;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ITW_NW; ++w) v += s_all[w * ITW_BWD_WAVE_FLOATS + k * 16 + f];
        sl_w1[e] = v;
    }
    float* sl_v = sl_w1 + H * F;
    for (int e = tid; e < 4 * H; e += ITW_NW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < ITW_NW; ++w) v += s_all[w * ITW_BWD_WAVE_FLOATS + 1024 + e];
        sl_v[e] = v;                                        // [db2 | dgamma | dbeta | db1]
    }
}

// the wave-owned form serves training calls at H = 64 with narrow inputs and segments of at most ITW_R det rows
// (TMPNN_IT_WAVE=0 keeps the workgroup form: A/B runs)
static bool itw_serves(int H, int F, int max_seg_rows, int training) {
    static const int off = [] { const char* e = getenv("TMPNN_IT_WAVE"); return (e && e[0] == '0') ? 1 : 0; }();
    return !off && training && H == 64 && F <= 16 && max_seg_rows <= ITW_R;
}
static int itw_blocks(int S) {
    const int groups = ceil_div(S, ITW_SPB);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return groups < cus ? groups : cus;
}

// dst (+)= sum over slabs, fixed order: element e of [dW2 | dW1 | db2 | dgamma | dbeta | db1].  A block takes 32 elements
// x 8 slices of the slab list; the slices are combined through LDS in slice order (deterministic).
__global__ __launch_bounds__(256) void k_it_reduce(const float* __restrict__ slabs, int nslab, int slab_floats, int H, int F,
                                                   float* dw2, float* dw1, float* db2, float* dgamma, float* dbeta, float* db1) {
    __shared__ float s_part[8][32];
    const int el = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int n_w2 = H * H, n_w1 = H * F, total = n_w2 + n_w1 + 4 * H;
    const int per = (nslab + 7) / 8;
    const int sl0 = slice * per, sl1 = min(nslab, sl0 + per);
    float v = 0.f;
    if (e < total)
        for (int s = sl0; s < sl1; ++s) v += slabs[(size_t)s * slab_floats + e];
    s_part[slice][el] = v;
    __syncthreads();
    if (slice != 0 || e >= total) return;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += s_part[q][el];
    if (e < n_w2) dw2[e] += sum;
    else if (e < n_w2 + n_w1) dw1[e - n_w2] += sum;
    else {
        const int q = (e - n_w2 - n_w1) / H, j = (e - n_w2 - n_w1) % H;
        (q == 0 ? db2 : q == 1 ? dgamma : q == 2 ? dbeta : db1)[j] += sum;
    }
}

// running statistics after S segments (momentum 0.1, unbiased variance), the recurrence of BatchNorm1d applied once per
// segment in order: a segment L places from the end carries 0.1 * 0.9^L (csrc/dense.hip k_bn_running)
__global__ __launch_bounds__(64) void k_it_running(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const int32_t* __restrict__ seg_cnt, int S, int H,
                                                   float* __restrict__ rm, float* __restrict__ rv) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int s0 = S > 320 ? S - 320 : 0;
    float m = 0.f, v = 0.f;
    for (int s = s0 + lane; s < S; s += 64) {
        const float w = 0.1f * powf(0.9f, (float)(S - 1 - s));
        const float cnt = (float)seg_cnt[s];
        const float r = rstd[(size_t)s * H + j];
        const float var = 1.0f / (r * r) - IT_EPS;
        m += w * mean[(size_t)s * H + j];
        v += w * var * (cnt / (cnt - 1.0f));
    }
    for (int off = 32; off >= 1; off >>= 1) { m += __shfl_xor(m, off); v += __shfl_xor(v, off); }
    if (lane == 0) {
        const float keep = s0 > 0 ? 0.f : powf(0.9f, (float)S);
        rm[j] = keep * rm[j] + m;
        rv[j] = keep * rv[j] + v;
    }
}

static size_t it_fwd_shm(int H, int F) {
    return sizeof(float) * ((size_t)H * (H + 4) + 2 * (size_t)IT_CH * (H + 4) + 2 * (size_t)IT_SPB * H +
                            (F > 16 ? (size_t)H * (F + 1) : (size_t)IT_CH * 16));
}
static size_t it_bwd_shm(int H) {
    return sizeof(float) * ((size_t)H * (H + 4) + 3 * (size_t)IT_CH * (H + 4) + (size_t)IT_SPB * H + 4 * IT_NT + (size_t)IT_CH * 16 +
                            (size_t)H * 17);
}
// persistent workgroups: one per CU (their LDS footprint admits no second one), fewer when there is less work
static int it_blocks(int nd, int S, int training) {
    const int groups = training ? ceil_div(S, IT_SPB) : ceil_div(nd, IT_CH);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return groups < cus ? groups : cus;
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_input_tf_supported(int H, int F, int max_seg_rows) {
    return ((H == 32 || H == 64) && F > 0 && F <= IT_FMAX && max_seg_rows >= 0 && max_seg_rows <= IT_CH) ? 1 : 0;
}

int tmpnn_input_tf_fwd(const float* xdet, const int64_t* x_rows, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int max_seg_rows, int H, int training, const float* w1, const float* b1,
                       const float* gamma, const float* beta, float* running_mean, float* running_var, const float* w2,
                       const float* b2, float* y_save, float* mean, float* rstd, const int32_t* out_row, float* h_new,
                       int ld_h, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_input_tf_supported(H, F, max_seg_rows), "input_tf_fwd: H=%d F=%d max_seg_rows=%d (H in {32, 64}, F <= %d, "
               "segments of at most %d det rows; use tmpnn_input_bn_fwd otherwise)", H, F, max_seg_rows, IT_FMAX, IT_CH);
    TM_REQUIRE(nd >= 0 && S >= 0, "input_tf_fwd: nd=%d S=%d", nd, S);
    TM_REQUIRE(w1 && b1 && gamma && beta && running_mean && running_var && w2 && b2 && mean && rstd, "input_tf_fwd: null parameter pointer");
    TM_REQUIRE(!training || (seg_ptr && seg_cnt && seg_of_det && S > 0), "input_tf_fwd: training needs segments");
    hipStream_t st = as_stream(stream);
    if (nd == 0 && !training) return TMPNN_OK;
    TM_REQUIRE(nd == 0 || (xdet && y_save && out_row && h_new && ld_x >= F && ld_h >= H), "input_tf_fwd: null/short buffers");
    ItFwdArgs a{xdet, x_rows, ld_x, F, nd, seg_ptr, seg_cnt, seg_of_det, S, training, w1, b1, gamma, beta, running_mean, running_var,
                w2, b2, y_save, mean, rstd, out_row, h_new, ld_h};
    if (nd > 0 && itw_serves(H, F, max_seg_rows, training)) {
        const size_t shmw = sizeof(float) * (size_t)ITW_NW * ITW_FWD_WAVE_FLOATS;
        TM_SHM_ONCE(k_itw_fwd, shmw);
        hipLaunchKernelGGL(k_itw_fwd, dim3(itw_blocks(S)), dim3(ITW_NW * 64), shmw, st, a);
        int rc = check_launch("itw_fwd");
        if (rc) return rc;
        hipLaunchKernelGGL(k_it_running, dim3(H), dim3(64), 0, st, mean, rstd, seg_cnt, S, H, running_mean, running_var);
        return check_launch("it_running");
    }
    const int nb = it_blocks(nd, S, training);
    const size_t shm = it_fwd_shm(H, F);
    if (nb > 0) {
        if (H == 64) { TM_SHM_ONCE(k_it_fwd<64>, it_fwd_shm(64, IT_FMAX)); hipLaunchKernelGGL(k_it_fwd<64>, dim3(nb), dim3(IT_NT), shm, st, a); }
        else { TM_SHM_ONCE(k_it_fwd<32>, it_fwd_shm(32, IT_FMAX)); hipLaunchKernelGGL(k_it_fwd<32>, dim3(nb), dim3(IT_NT), shm, st, a); }
        int rc = check_launch("it_fwd");
        if (rc) return rc;
    }
    if (training) {
        hipLaunchKernelGGL(k_it_running, dim3(H), dim3(64), 0, st, mean, rstd, seg_cnt, S, H, running_mean, running_var);
        return check_launch("it_running");
    }
    return TMPNN_OK;
}

size_t tmpnn_input_tf_bwd_ws(int nd, int S, int H, int F, int training) {
    int nb = it_blocks(nd, S, training);
    if (training && itw_blocks(S) > nb) nb = itw_blocks(S);          // (whichever form the launch picks: one slab per workgroup)
    return sizeof(float) * (size_t)(nb > 0 ? nb : 1) * ((size_t)H * H + (size_t)H * F + 4 * H);
}

int tmpnn_input_tf_bwd(const float* xdet, const int64_t* x_rows, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int max_seg_rows, int H, int training, const float* w1, const float* b1,
                       const float* gamma, const float* beta, const float* w2, const float* y_save, const float* mean,
                       const float* rstd, const int32_t* out_row, const float* d_h, int ld_dh, float* d_xdet, int ld_dx,
                       float* d_xzero, float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, void* ws,
                       size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_input_tf_supported(H, F, max_seg_rows), "input_tf_bwd: H=%d F=%d max_seg_rows=%d", H, F, max_seg_rows);
    TM_REQUIRE(nd >= 0 && S >= 0, "input_tf_bwd: nd=%d S=%d", nd, S);
    TM_REQUIRE(w1 && b1 && gamma && beta && w2 && mean && rstd && dw1 && db1 && dgamma && dbeta && dw2 && db2, "input_tf_bwd: null parameter pointer");
    hipStream_t st = as_stream(stream);
    if (nd == 0) {
        if (d_xzero && S > 0) (void)hipMemsetAsync(d_xzero, 0, sizeof(float) * (size_t)S * F, st);
        return TMPNN_OK;       // no det rows: nothing reaches the loss through this transform
    }
    TM_REQUIRE(xdet && y_save && out_row && d_h && ws, "input_tf_bwd: null buffers");
    TM_REQUIRE(!training || (seg_ptr && seg_cnt && seg_of_det && S > 0), "input_tf_bwd: training needs segments");
    const size_t need = tmpnn_input_tf_bwd_ws(nd, S, H, F, training);
    if (ws_bytes < need) return set_error(TMPNN_EWORKSPACE, "input_tf_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    if (!training && d_xzero && S > 0) (void)hipMemsetAsync(d_xzero, 0, sizeof(float) * (size_t)S * F, st);
    const bool wave_form = itw_serves(H, F, max_seg_rows, training) && d_xdet == nullptr && d_xzero == nullptr;
    const int nb = wave_form ? itw_blocks(S) : it_blocks(nd, S, training);
    const int slab_floats = H * H + H * F + 4 * H;
    ItBwdArgs a{xdet, x_rows, ld_x, F, nd, seg_ptr, seg_cnt, seg_of_det, S, training, w1, b1, gamma, beta, w2, y_save, mean, rstd, out_row,
                d_h, ld_dh, d_xdet, ld_dx, d_xzero, reinterpret_cast<float*>(ws), slab_floats};
    if (wave_form) {
        const size_t shmw = sizeof(float) * (size_t)ITW_NW * ITW_BWD_WAVE_FLOATS;
        TM_SHM_ONCE(k_itw_bwd, shmw);
        hipLaunchKernelGGL(k_itw_bwd, dim3(nb), dim3(ITW_NW * 64), shmw, st, a);
        int rc = check_launch("itw_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(k_it_reduce, dim3(ceil_div(slab_floats, 32)), dim3(256), 0, st, reinterpret_cast<const float*>(ws), nb,
                           slab_floats, H, F, dw2, dw1, db2, dgamma, dbeta, db1);
        return check_launch("it_reduce");
    }
#define IT_BWD(HH, FF)                                                                                   \
    do {                                                                                                 \
        TM_SHM_ONCE((k_it_bwd<HH, FF>), it_bwd_shm(HH));                                                 \
        hipLaunchKernelGGL((k_it_bwd<HH, FF>), dim3(nb), dim3(IT_NT), it_bwd_shm(HH), st, a);              \
    } while (0)
    if (H == 64) { if (F <= 16) IT_BWD(64, 1); else IT_BWD(64, IT_FMAX / 16); }
    else { if (F <= 16) IT_BWD(32, 1); else IT_BWD(32, IT_FMAX / 32); }
#undef IT_BWD
    int rc = check_launch("it_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(k_it_reduce, dim3(ceil_div(slab_floats, 32)), dim3(256), 0, st, reinterpret_cast<const float*>(ws), nb,
                       slab_floats, H, F, dw2, dw1, db2, dgamma, dbeta, db1);
    return check_launch("it_reduce");
}

}  // extern "C"
