// GRU cells of the factor-graph update (SURVEY 8(a) rows H', I; models/layers.py:97,114,116),
// forward + backward, on the fp32 matrix cores of gfx950.
//
// Shape of the problem: rows x [IN|H] times a SMALL weight matrix ([IN|H] x 3H, <= 0.8 MB), rows
// in the millions.  So the kernels are organised around rows, not around a GEMM grid:
//   * one wave owns 32 state rows and keeps their A-operand slices in registers
//     (v_mfma_f32_32x32x2_f32: A[i = lane&31][k = lane>>5], one VGPR per operand);
//   * the k index is enumerated as k = 32*kt + 16*(lane>>5) + s, so a lane reads 16 CONTIGUOUS
//     floats (4 x dwordx4) of its row per k-tile -- no LDS staging, no transposes;
//   * the weight (B) operand is a coalesced 128-byte read per half-wave from L1/L2 -- at the
//     f32 MFMA rate (64 cycles per instruction) one such read per MFMA is far below L2 bandwidth;
//   * the node->edge message (row E) is formed in the A-operand load (h[src]-h[dst]) and never
//     touches HBM; the type-masked merge (row I) is the row indirection of the store.
// Exact fp32 (MFMA f32 == fmaf chain), within 1e-6 of torch.nn.GRUCell.
#include "common.h"

namespace tmpnn {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row (within the wave's 32) held by accumulator register `reg` of lane-half `half`
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ void load16(const float* __restrict__ p, float* v) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = q[i];
        v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}

struct GruFwdArgs {
    const int32_t* rows; int R;
    const int32_t* src; const int32_t* dst;
    const float* msg; int ld_msg; int IN; int msg_compact;
    const float* h; int ld_h; int H;
    const float* wih_t; const float* whh_t; const float* b_ih; const float* b_hh;
    float* h_out; int ld_out;
    float* gates; size_t gate_plane;
};

// 16 floats of x for list position li at feature offset f0 (multiple of 16)
template <int XMODE>
__device__ __forceinline__ void load_x16(const GruFwdArgs& a, int li, int row, int f0, float* v) {
    if (XMODE == 0) {
        load16(a.msg + (size_t)(a.msg_compact ? li : row) * a.ld_msg + f0, v);
    } else if (XMODE == 1) {
        float u[16], w[16];
        load16(a.h + (size_t)a.src[li] * a.ld_h + f0, u);
        load16(a.h + (size_t)a.dst[li] * a.ld_h + f0, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = u[i] - w[i];
    } else {
        if (f0 < a.H) load16(a.h + (size_t)a.src[li] * a.ld_h + f0, v);
        else          load16(a.h + (size_t)a.dst[li] * a.ld_h + (f0 - a.H), v);
    }
}

// grid: (ceil(R/128), H/(32*CT)); block 256 = 4 waves x 32 rows; each block computes 32*CT output
// features of all three gates for its rows.
template <int CT, int XMODE>
__global__ __launch_bounds__(256, 2) void k_gru_fwd(GruFwdArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= a.R) return;
    const int H = a.H, H3 = 3 * a.H;
    const int col0 = blockIdx.y * (32 * CT);
    const int li = min(r0 + c, a.R - 1);
    const int row = a.rows[li];

    f32x16 acc_r[CT], acc_z[CT], acc_in[CT], acc_hn[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_r[t][i] = 0.f; acc_z[t][i] = 0.f; acc_in[t][i] = 0.f; acc_hn[t][i] = 0.f; }

    // ---- x part:  gi = x @ W_ih^T
    for (int kt = 0; kt < a.IN / 32; ++kt) {
        const int f0 = kt * 32 + half * 16;
        float av[16];
        load_x16<XMODE>(a, li, row, f0, av);
        const float* __restrict__ b0 = a.wih_t + (size_t)f0 * H3 + col0 + c;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* __restrict__ b = b0 + (size_t)s * H3;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                acc_r[t] = mfma32(av[s], b[t * 32], acc_r[t]);
                acc_z[t] = mfma32(av[s], b[H + t * 32], acc_z[t]);
                acc_in[t] = mfma32(av[s], b[2 * H + t * 32], acc_in[t]);
            }
        }
    }
    // ---- h part:  gh = h @ W_hh^T
    for (int kt = 0; kt < H / 32; ++kt) {
        const int f0 = kt * 32 + half * 16;
        float av[16];
        load16(a.h + (size_t)row * a.ld_h + f0, av);
        const float* __restrict__ b0 = a.whh_t + (size_t)f0 * H3 + col0 + c;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* __restrict__ b = b0 + (size_t)s * H3;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                acc_r[t] = mfma32(av[s], b[t * 32], acc_r[t]);
                acc_z[t] = mfma32(av[s], b[H + t * 32], acc_z[t]);
                acc_hn[t] = mfma32(av[s], b[2 * H + t * 32], acc_hn[t]);
            }
        }
    }
    // ---- gate epilogue, merge-by-row store
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int col = col0 + t * 32 + c;
        const float br = a.b_ih[col] + a.b_hh[col];
        const float bz = a.b_ih[H + col] + a.b_hh[H + col];
        const float bin = a.b_ih[2 * H + col];
        const float bhn = a.b_hh[2 * H + col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lpos = r0 + acc_row(reg, half);
            if (lpos < a.R) {
                const int orow = a.rows[lpos];
                const float r = sigmoidf_(acc_r[t][reg] + br);
                const float z = sigmoidf_(acc_z[t][reg] + bz);
                const float hn = acc_hn[t][reg] + bhn;
                const float n = tanhf(acc_in[t][reg] + bin + r * hn);
                const float hp = a.h[(size_t)orow * a.ld_h + col];
                a.h_out[(size_t)orow * a.ld_out + col] = (1.0f - z) * n + z * hp;
                if (a.gates) {
                    float* gp = a.gates + (size_t)orow * H + col;
                    gp[0] = r;
                    gp[a.gate_plane] = z;
                    gp[2 * a.gate_plane] = n;
                    gp[3 * a.gate_plane] = hn;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, data path
// ------------------------------------------------------------------------------------------
struct GruBwdDataArgs {
    const int32_t* rows; int R; int IN;
    const float* h; int ld_h; int H;
    const float* w_ih; const float* w_hh;
    const float* gates; size_t gate_plane;
    const float* d_hout; int ld_dhout;
    float* d_msg; int ld_dmsg;
    float* d_h; int ld_dh;
};

// grid: (ceil(R/128), (IN+H)/(32*NT)); a block's NT*32 output columns lie entirely in d_msg
// (virtual column < IN) or in d_h.
template <int NT>
__global__ __launch_bounds__(256) void k_gru_bwd_data(GruBwdDataArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= a.R) return;
    const int H = a.H;
    const int vcol0 = blockIdx.y * (32 * NT);
    const bool is_dx = vcol0 < a.IN;
    const int n0 = is_dx ? vcol0 : vcol0 - a.IN;
    const float* __restrict__ W = is_dx ? a.w_ih : a.w_hh;
    const int ldw = is_dx ? a.IN : H;
    const int li = min(r0 + c, a.R - 1);
    const int row = a.rows[li];
    const size_t gp = a.gate_plane;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int fb = 0; fb < H / 32; ++fb) {
        const int f0 = fb * 32 + half * 16;
        float dh[16], r[16], z[16], n[16], hn[16], hp[16];
        load16(a.d_hout + (size_t)row * a.ld_dhout + f0, dh);
        const float* g0 = a.gates + (size_t)row * H + f0;
        load16(g0, r);
        load16(g0 + gp, z);
        load16(g0 + 2 * gp, n);
        load16(g0 + 3 * gp, hn);
        load16(a.h + (size_t)row * a.ld_h + f0, hp);
        float ar[16], az[16], an[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float dn = dh[i] * (1.0f - z[i]) * (1.0f - n[i] * n[i]);
            ar[i] = dn * hn[i] * r[i] * (1.0f - r[i]);
            az[i] = dh[i] * (hp[i] - n[i]) * z[i] * (1.0f - z[i]);
            an[i] = is_dx ? dn : dn * r[i];
        }
        const float* __restrict__ wr = W + (size_t)f0 * ldw + n0 + c;
        const float* __restrict__ wz = W + (size_t)(H + f0) * ldw + n0 + c;
        const float* __restrict__ wn = W + (size_t)(2 * H + f0) * ldw + n0 + c;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = mfma32(ar[s], wr[(size_t)s * ldw + t * 32], acc[t]);
                acc[t] = mfma32(az[s], wz[(size_t)s * ldw + t * 32], acc[t]);
                acc[t] = mfma32(an[s], wn[(size_t)s * ldw + t * 32], acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = n0 + t * 32 + c;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lpos = r0 + acc_row(reg, half);
            if (lpos < a.R) {
                const int orow = a.rows[lpos];
                if (is_dx) {
                    a.d_msg[(size_t)orow * a.ld_dmsg + col] = acc[t][reg];
                } else {
                    const float zz = a.gates[gp + (size_t)orow * H + col];
                    a.d_h[(size_t)orow * a.ld_dh + col] =
                        acc[t][reg] + a.d_hout[(size_t)orow * a.ld_dhout + col] * zz;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, weight path:  dW_ih = d_gi^T x,  dW_hh = d_gh^T h,  db = column sums
// One WAVE = one worker (row slab rs, 32 gate features q, up to 4 x 32 columns of [x | h]).
// Partial products go to slabs, reduced afterwards in a fixed order (no float atomics).
// ------------------------------------------------------------------------------------------
struct GruBwdWArgs {
    const int32_t* rows; int R; const int32_t* src; const int32_t* dst;
    const float* msg; int ld_msg; int IN; int msg_compact;
    const float* h; int ld_h; int H;
    const float* gates; size_t gate_plane;
    const float* d_hout; int ld_dhout;
    float* slab_w;      // [n_rs][3H][IN+H]
    float* slab_b;      // [n_rs][2][3H]
    int n_rs, RS, NQ, NCH;
};

template <int XMODE>
__device__ __forceinline__ float load_x1(const GruBwdWArgs& a, int lpos, int orow, int col) {
    if (XMODE == 0) return a.msg[(size_t)(a.msg_compact ? lpos : orow) * a.ld_msg + col];
    if (XMODE == 1) return a.h[(size_t)a.src[lpos] * a.ld_h + col] - a.h[(size_t)a.dst[lpos] * a.ld_h + col];
    return col < a.H ? a.h[(size_t)a.src[lpos] * a.ld_h + col] : a.h[(size_t)a.dst[lpos] * a.ld_h + col - a.H];
}

template <int XMODE>
__global__ __launch_bounds__(256) void k_gru_bwd_weights(GruBwdWArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nworkers = (long)a.n_rs * a.NQ * a.NCH;
    if (w >= nworkers) return;
    const int q = (int)(w % a.NQ);
    const int ch = (int)((w / a.NQ) % a.NCH);
    const int rs = (int)(w / ((long)a.NQ * a.NCH));
    const int H = a.H, XH = a.IN + a.H;
    const int ntw = min(4, XH / 32 - ch * 4);
    const int g = (q * 32) / H;               // gate of this worker's 32 features
    const int f = (q * 32) % H + c;           // hidden feature of this lane
    const size_t gp = a.gate_plane;
    const int lo = rs * a.RS, hi = min(a.R, lo + a.RS);

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float sum_i = 0.f, sum_h = 0.f;

    for (int p = lo; p < hi; p += 2) {
        const int lpos_raw = p + half;
        const bool valid = lpos_raw < hi;
        const int lpos = valid ? lpos_raw : hi - 1;
        const int orow = a.rows[lpos];
        const float dh = a.d_hout[(size_t)orow * a.ld_dhout + f];
        const float* gq = a.gates + (size_t)orow * H + f;
        const float r = gq[0], z = gq[gp], n = gq[2 * gp], hn = gq[3 * gp];
        const float hp = a.h[(size_t)orow * a.ld_h + f];
        const float dn = dh * (1.0f - z) * (1.0f - n * n);
        float ai, ah;
        if (g == 0) { ai = dn * hn * r * (1.0f - r); ah = ai; }
        else if (g == 1) { ai = dh * (hp - n) * z * (1.0f - z); ah = ai; }
        else { ai = dn; ah = dn * r; }
        if (!valid) { ai = 0.f; ah = 0.f; }
        sum_i += ai;
        sum_h += ah;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ntw) {
                const int vcol = ch * 128 + t * 32 + c;
                if (ch * 128 + t * 32 < a.IN) {
                    acc[t] = mfma32(ai, load_x1<XMODE>(a, lpos, orow, vcol), acc[t]);
                } else {
                    acc[t] = mfma32(ah, a.h[(size_t)orow * a.ld_h + vcol - a.IN], acc[t]);
                }
            }
        }
    }
    float* sw = a.slab_w + (size_t)rs * (3 * H) * XH;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < ntw) {
            const int vcol = ch * 128 + t * 32 + c;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int j = q * 32 + acc_row(reg, half);
                sw[(size_t)j * XH + vcol] = acc[t][reg];
            }
        }
    }
    if (ch == 0) {
        sum_i += __shfl_xor(sum_i, 32);
        sum_h += __shfl_xor(sum_h, 32);
        if (half == 0) {
            float* sb = a.slab_b + (size_t)rs * 2 * (3 * H);
            sb[q * 32 + c] = sum_i;
            sb[3 * H + q * 32 + c] = sum_h;
        }
    }
}

// dW_ih[j][k] += sum_rs slab[rs][j][k], k < IN ; dW_hh[j][k-IN] += ... ; biases likewise
__global__ void k_gru_reduce_w(const float* __restrict__ slab_w, const float* __restrict__ slab_b, int n_rs,
                               int IN, int H, float* __restrict__ dW_ih, float* __restrict__ dW_hh,
                               float* __restrict__ db_ih, float* __restrict__ db_hh) {
    const int XH = IN + H;
    const size_t nW = (size_t)3 * H * XH;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nW) {
        float s = 0.f;
        for (int k = 0; k < n_rs; ++k) s += slab_w[(size_t)k * nW + i];
        const int j = (int)(i / XH), col = (int)(i % XH);
        if (col < IN) dW_ih[(size_t)j * IN + col] += s;
        else dW_hh[(size_t)j * H + (col - IN)] += s;
    } else if (i < nW + (size_t)6 * H) {
        const int b = (int)(i - nW);
        float s = 0.f;
        for (int k = 0; k < n_rs; ++k) s += slab_b[(size_t)k * 6 * H + b];
        if (b < 3 * H) db_ih[b] += s;
        else db_hh[b - 3 * H] += s;
    }
}

static void plan_weights(int R, int IN, int H, int* n_rs, int* RS, int* NQ, int* NCH) {
    *NQ = 3 * H / 32;
    *NCH = (IN + H + 127) / 128;
    long per = (long)(*NQ) * (*NCH);
    long want = 4096 / per;                  // ~16 waves per CU in flight
    if (want < 1) want = 1;
    long by_rows = (R + 63) / 64;            // at least 64 rows per slab
    long n = by_rows < want ? by_rows : want;
    if (n < 1) n = 1;
    long rs = (R + n - 1) / n;
    rs = (rs + 1) & ~1L;                     // even: one MFMA step eats two rows
    if (rs < 2) rs = 2;
    n = (R + rs - 1) / rs;
    if (n < 1) n = 1;
    *n_rs = (int)n;
    *RS = (int)rs;
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_gru_fwd(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst, const float* msg,
                  int ld_msg, int msg_compact, int IN, const float* h, int ld_h, int H, const float* wih_t, const float* whh_t,
                  const float* b_ih, const float* b_hh, float* h_out, int ld_out, float* gates, size_t gate_plane,
                  tmpnn_stream stream) {
    TM_REQUIRE(supported_H(H), "gru_fwd: unsupported H=%d", H);
    TM_REQUIRE(R >= 0, "gru_fwd: R=%d", R);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(xmode >= 0 && xmode <= 2, "gru_fwd: xmode=%d", xmode);
    TM_REQUIRE(IN == (xmode == 2 ? 2 * H : (xmode == 1 ? H : IN)) && IN % 32 == 0 && IN > 0,
               "gru_fwd: IN=%d does not match xmode=%d H=%d", IN, xmode, H);
    TM_REQUIRE(rows && h && wih_t && whh_t && b_ih && b_hh && h_out, "gru_fwd: null pointer");
    TM_REQUIRE(xmode == 0 ? (msg != nullptr && ld_msg >= IN && (ld_msg & 3) == 0 && aligned16(msg))
                          : (src != nullptr && dst != nullptr),
               "gru_fwd: message source missing/misaligned for xmode=%d", xmode);
    TM_REQUIRE(ld_h >= H && ld_out >= H && (ld_h & 3) == 0 && aligned16(h), "gru_fwd: bad state layout");
    TM_REQUIRE(gates == nullptr || gate_plane >= (size_t)H, "gru_fwd: gate_plane too small");
    GruFwdArgs a{rows, R, src, dst, msg, ld_msg, IN, msg_compact, h, ld_h, H, wih_t, whh_t, b_ih, b_hh, h_out, ld_out, gates,
                 gate_plane};
    const int CT = (H % 64 == 0) ? 2 : 1;
    dim3 grid(ceil_div(R, 128), H / (32 * CT)), block(256);
    hipStream_t st = as_stream(stream);
#define L(C, X) hipLaunchKernelGGL((k_gru_fwd<C, X>), grid, block, 0, st, a)
    if (CT == 2) { if (xmode == 0) L(2, 0); else if (xmode == 1) L(2, 1); else L(2, 2); }
    else         { if (xmode == 0) L(1, 0); else if (xmode == 1) L(1, 1); else L(1, 2); }
#undef L
    return check_launch("gru_fwd");
}

int tmpnn_gru_bwd_data(const int32_t* rows, int R, int IN, const float* h, int ld_h, int H, const float* w_ih,
                       const float* w_hh, const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout,
                       float* d_msg, int ld_dmsg, float* d_h, int ld_dh, tmpnn_stream stream) {
    TM_REQUIRE(supported_H(H), "gru_bwd_data: unsupported H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(R > 0 && IN > 0 && IN % 32 == 0 && (H % 64 != 0 || IN % 64 == 0), "gru_bwd_data: R=%d IN=%d", R, IN);
    TM_REQUIRE(rows && h && w_ih && w_hh && gates && d_hout && d_msg && d_h, "gru_bwd_data: null pointer");
    TM_REQUIRE((ld_h & 3) == 0 && (ld_dhout & 3) == 0 && aligned16(h) && aligned16(d_hout) && aligned16(gates) &&
                   (gate_plane & 3) == 0,
               "gru_bwd_data: rows must be 16-byte aligned");
    TM_REQUIRE(ld_dmsg >= IN && ld_dh >= H && ld_h >= H && ld_dhout >= H, "gru_bwd_data: leading dimension too small");
    GruBwdDataArgs a{rows, R, IN, h, ld_h, H, w_ih, w_hh, gates, gate_plane, d_hout, ld_dhout, d_msg, ld_dmsg, d_h,
                     ld_dh};
    const int NT = (H % 64 == 0) ? 2 : 1;
    dim3 grid(ceil_div(R, 128), (IN + H) / (32 * NT)), block(256);
    hipStream_t st = as_stream(stream);
    if (NT == 2) hipLaunchKernelGGL((k_gru_bwd_data<2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_gru_bwd_data<1>), grid, block, 0, st, a);
    return check_launch("gru_bwd_data");
}

size_t tmpnn_gru_bwd_weights_ws(int R, int IN, int H) {
    if (R <= 0) return 0;
    int n_rs, RS, NQ, NCH;
    plan_weights(R, IN, H, &n_rs, &RS, &NQ, &NCH);
    return ((size_t)n_rs * 3 * H * (IN + H) + (size_t)n_rs * 6 * H) * sizeof(float);
}

int tmpnn_gru_bwd_weights(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                          const float* msg, int ld_msg, int msg_compact, int IN, const float* h, int ld_h, int H,
                          const float* gates,
                          size_t gate_plane, const float* d_hout, int ld_dhout, float* dW_ih, float* dW_hh,
                          float* db_ih, float* db_hh, void* ws, size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(supported_H(H), "gru_bwd_weights: unsupported H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(R > 0 && xmode >= 0 && xmode <= 2 && IN % 32 == 0 && IN > 0, "gru_bwd_weights: R=%d xmode=%d IN=%d", R,
               xmode, IN);
    TM_REQUIRE(IN == (xmode == 2 ? 2 * H : (xmode == 1 ? H : IN)), "gru_bwd_weights: IN=%d vs xmode=%d", IN, xmode);
    TM_REQUIRE(rows && h && gates && d_hout && dW_ih && dW_hh && db_ih && db_hh, "gru_bwd_weights: null pointer");
    TM_REQUIRE(xmode == 0 ? msg != nullptr : (src != nullptr && dst != nullptr), "gru_bwd_weights: message source");
    const size_t need = tmpnn_gru_bwd_weights_ws(R, IN, H);
    if (ws == nullptr || ws_bytes < need)
        return set_error(TMPNN_EWORKSPACE, "gru_bwd_weights: workspace %zu < %zu bytes", ws_bytes, need);
    int n_rs, RS, NQ, NCH;
    plan_weights(R, IN, H, &n_rs, &RS, &NQ, &NCH);
    float* slab_w = reinterpret_cast<float*>(ws);
    float* slab_b = slab_w + (size_t)n_rs * 3 * H * (IN + H);
    GruBwdWArgs a{rows, R, src, dst, msg, ld_msg, IN, msg_compact, h, ld_h, H, gates, gate_plane, d_hout, ld_dhout, slab_w, slab_b,
                  n_rs, RS, NQ, NCH};
    const long nworkers = (long)n_rs * NQ * NCH;
    dim3 grid(ceil_div(nworkers, 4)), block(256);
    hipStream_t st = as_stream(stream);
    if (xmode == 0) hipLaunchKernelGGL((k_gru_bwd_weights<0>), grid, block, 0, st, a);
    else if (xmode == 1) hipLaunchKernelGGL((k_gru_bwd_weights<1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_gru_bwd_weights<2>), grid, block, 0, st, a);
    int rc = check_launch("gru_bwd_weights");
    if (rc) return rc;
    const size_t n = (size_t)3 * H * (IN + H) + (size_t)6 * H;
    hipLaunchKernelGGL(k_gru_reduce_w, dim3(ceil_div((long)n, 256)), dim3(256), 0, st, slab_w, slab_b, n_rs, IN, H,
                       dW_ih, dW_hh, db_ih, db_hh);
    return check_launch("gru_reduce_w");
}

}  // extern "C"
