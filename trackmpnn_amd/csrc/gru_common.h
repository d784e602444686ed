// Shared pieces of the GRU-cell kernels (csrc/gru_fwd.hip, csrc/gru_bwd.hip): the bf16x6 split-product helpers, the argument
// structs of the generic kernels and the transposing 32 x 32 store through a per-wave LDS tile.
#pragma once
#include "common.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>

namespace tmpnn {
typedef float f32x4 __attribute__((ext_vector_type(4)));

// sigmoid / tanh on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each).  Both stay within
// ~3e-7 of the libm results, far inside the 1e-4 parity budget, at a fraction of the VALU cost
// (the gate epilogue competes with the MFMAs for issue slots).
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // 1 - 2/(1+e^{2x}); saturates cleanly: e^{2x} -> inf gives 1, -> 0 gives -1
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe ("bf16x6").  The f32-input MFMA runs at 1/16 of the bf16 rate, so a
// GEMM whose operands are split into three bf16 pieces each, a = a1 + a2 + a3 (round-to-nearest residuals,
// |a - a1 - a2 - a3| <= 2^-27 |a|), and evaluated as
//     a.b ~= a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1        (six v_mfma_f32_32x32x16_bf16)
// costs 6/16 of the f32 instruction time.  Every partial product of two 8-bit significands is exact in the
// f32 accumulator; the dropped terms (a2 b3, a3 b2, a3 b3) are below 2^-25 |a.b|, i.e. under the rounding of
// one f32 fma, so the result is as accurate as the f32 MFMA chain (checked per stage against fp64).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {       // v_cvt_pk_bf16_f32: lo -> bits 0..15
    bf16x2 v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = pk_bf16(r0, r1);
    r0 -= __uint_as_float(p2 << 16);
    r1 -= __uint_as_float(p2 & 0xFFFF0000u);
    p3 = pk_bf16(r0, r1);
}
struct Split8 { uint4 p1, p2, p3; };            // eight consecutive k values as three packed-bf16 pieces
__device__ __forceinline__ Split8 split8(const float4& u, const float4& v) {
    Split8 s;
    split_pair(u.x, u.y, s.p1.x, s.p2.x, s.p3.x);
    split_pair(u.z, u.w, s.p1.y, s.p2.y, s.p3.y);
    split_pair(v.x, v.y, s.p1.z, s.p2.z, s.p3.z);
    split_pair(v.z, v.w, s.p1.w, s.p2.w, s.p3.w);
    return s;
}
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// acc += A B over 16 k values, smallest terms first
__device__ __forceinline__ f32x16 mfma_x6(const uint4& a1, const uint4& a2, const uint4& a3, const Split8& b, f32x16 c) {
    c = mfma_bf16(a3, b.p1, c);
    c = mfma_bf16(a1, b.p3, c);
    c = mfma_bf16(a2, b.p2, c);
    c = mfma_bf16(a2, b.p1, c);
    c = mfma_bf16(a1, b.p2, c);
    c = mfma_bf16(a1, b.p1, c);
    return c;
}
// one fp32 value -> its three bf16 pieces (bit patterns)
__device__ __forceinline__ void split1(float x, uint16_t& q1, uint16_t& q2, uint16_t& q3) {
    uint32_t p1, p2, p3;
    split_pair(x, 0.f, p1, p2, p3);
    q1 = (uint16_t)p1; q2 = (uint16_t)p2; q3 = (uint16_t)p3;
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

// upstream gradient of a cell's output: dh[row][f] = d_hout[row][f] (if given) + dy[row] * w_head[f] (if given).
// The second term is the output head's contribution (track_mpnn.py:73), folded in here so that the
// summed gradient never has to be materialised in HBM.
struct DhSrc {
    const float* d_hout; int ld_dhout;
    const float* dy; const float* w_head;
};
__device__ __forceinline__ float dh_at(const DhSrc& s, int row, int f) {
    float v = s.d_hout ? s.d_hout[(size_t)row * s.ld_dhout + f] : 0.f;
    if (s.dy) v += s.dy[row] * s.w_head[f];
    return v;
}
__device__ __forceinline__ void dh_load16(const DhSrc& s, int row, int f0, float* v) {
    if (s.d_hout) load16(s.d_hout + (size_t)row * s.ld_dhout + f0, v);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
    }
    if (s.dy) {
        const float d = s.dy[row];
        float w[16];
        load16(s.w_head + f0, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += d * w[i];
    }
}

// compile-time variants (UP bit 0: d_hout present, bit 1: head term present) for the hot LDS kernels
template <int UP>
__device__ __forceinline__ float dh_at_t(const DhSrc& s, int row, int f) {
    float v = (UP & 1) ? s.d_hout[(size_t)row * s.ld_dhout + f] : 0.f;
    if (UP & 2) v += s.dy[row] * s.w_head[f];
    return v;
}
template <int UP>
__device__ __forceinline__ void dh_load16_t(const DhSrc& s, int row, int f0, float* v) {
    if (UP & 1) load16(s.d_hout + (size_t)row * s.ld_dhout + f0, v);
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
    }
    if (UP & 2) {
        const float d = s.dy[row];
        float w[16];
        load16(s.w_head + f0, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += d * w[i];
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
    // optional fused output head (track_mpnn.py:73): logit_part[cw][row] = w_head[cols of column wave cw] . h_out[row]
    const float* w_head; float* logit_part; size_t part_stride;
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


struct GruBwdDataArgs {
    const int32_t* rows; int R; int IN;
    const float* h; int ld_h; int H;
    const float* w_ih; const float* w_hh;
    const float* gates; size_t gate_plane;
    DhSrc up;
    float* d_msg; int ld_dmsg;
    float* d_h; int ld_dh;
    // optional fused adjoint of the edge -> node sum (row F): d_h[row] += add_msg[add_src[r]] - add_msg[add_dst[r]]
    const int32_t* add_src; const int32_t* add_dst; const float* add_msg; int ld_add;
};

struct GruBwdWArgs {
    const int32_t* rows; int R; const int32_t* src; const int32_t* dst;
    const float* msg; int ld_msg; int IN; int msg_compact;
    const float* h; int ld_h; int H;
    const float* gates; size_t gate_plane;
    DhSrc up;
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

// ==========================================================================================
// LDS-resident-weight variants for the headline width (H <= 64).
//
// At H = 64 both weight matrices of a cell (2 x 48 KiB fp32) fit the 160 KiB LDS of a CU, so a
// block of 8 waves (2 per SIMD) loads them once and then streams row tiles through the fp32
// MFMAs: the B operand becomes a conflict-free ds_read_b32 (a half-wave reads 32 consecutive
// floats), nothing but state rows and gates crosses L2/HBM, and the grid is persistent
// (<= 1 block per CU) so the weight load is amortised over the whole launch.
// ==========================================================================================
// Transposing store of one 32 x 32 fp32 tile held one ROW per lane pair (lane (c, half) owns columns
// 8q + 4*half .. +3, q = 0..3, of row c) through a private LDS tile: written as ds_write_b128, read back
// so that lane l of pass k holds 16 bytes of row (64k + l)/8 -- eight lanes cover one 128-byte row segment.
// The staging row stride of 36 floats keeps both the writes and the reads bank-conflict free.
constexpr int STG_LD = 36;
template <bool NT = false, bool ACC = false>
__device__ __forceinline__ void stage_store32(float* stg, int c, int half, int lane, const f32x16& v,
                                              float* dst, int ld, int col0, int row, int r0, int R) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(stg + c * STG_LD + 8 * q + 4 * half) =
            make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same-wave LDS ops are ordered; keep the compiler honest
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = k * 64 + lane;
        const int rr = idx >> 3, ch = idx & 7;
        const float4 x = *reinterpret_cast<const float4*>(stg + rr * STG_LD + ch * 4);
        const int orow = __shfl(row, rr, 64);               // lane rr (< 32) owns row r0 + rr
        if (r0 + rr < R) {
            float* p = dst + (size_t)orow * ld + col0 + ch * 4;
            if (NT) {      // streamed once, read back only by the backward pass: keep it out of the way of L2
                __builtin_nontemporal_store(x.x, p); __builtin_nontemporal_store(x.y, p + 1);
                __builtin_nontemporal_store(x.z, p + 2); __builtin_nontemporal_store(x.w, p + 3);
            } else if (ACC) {
                const float4 o = *reinterpret_cast<const float4*>(p);
                *reinterpret_cast<float4*>(p) = make_float4(o.x + x.x, o.y + x.y, o.z + x.z, o.w + x.w);
            } else {
                *reinterpret_cast<float4*>(p) = x;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// output row of staging pass k for this lane = row owned by lane (64k + lane)/8 (lanes 0..31 own the tile's rows)
__device__ __forceinline__ void stage_rows(int row, int lane, int* orow4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) orow4[k] = __shfl(row, (k * 64 + lane) >> 3, 64);
}

// raw A-operand slice of one 32-wide k tile (16 floats per lane; the diff message needs two rows)
struct ATile { float u[16]; float w[16]; };

// row ids of one work item for this lane: list position, state row, and the two endpoint ids
// (XMODE 1/2: state rows of src/dst; XMODE 3: det indices into the projected buffer)
struct TileIdx { int li, row, s, d; };

template <int XMODE>
__device__ __forceinline__ TileIdx tile_idx(const GruFwdArgs& a, int r0, int c) {
    TileIdx t;
    t.li = min(r0 + c, a.R - 1);
    t.row = a.rows[t.li];
    t.s = (XMODE != 0) ? a.src[t.li] : 0;
    t.d = (XMODE != 0) ? a.dst[t.li] : 0;
    return t;
}


// TMPNN_SPLIT=0 keeps every GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32)
inline bool split_enabled() {
    static const int on = [] { const char* e = getenv("TMPNN_SPLIT"); return (e && e[0] == '0') ? 0 : 1; }();
    return on != 0;
}

}  // namespace tmpnn
