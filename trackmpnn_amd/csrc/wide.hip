// Wide GRU cells (H = 128 / 256; BASELINE.json C5: H = 256, 4.4 M edge rows per iteration) on the bf16 matrix pipe.
//
// At H <= 64 the cell's weight matrices fit the LDS and the kernels of gru_fwd.hip / gru_bwd.hip keep them resident.  At H = 256 they
// are 0.8 MB each, so round 1 streamed them from L2 into f32-input MFMAs (64 / 45 TFLOP/s forward / backward-data).
// Here the products are LDS-tiled GEMMs on bf16x6 split products (three bf16 pieces per fp32 operand, six
// v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulate: as accurate as the f32 MFMA chain, see gru_common.h), 2.7x
// the f32-MFMA ceiling:
//
//   block tile 128 rows x 128 (or 3 x 64) columns, K step 32; 8 waves, each a 32 x 64 (or 32 x 3 x 32) sub-tile;
//   A rows (fp32, optionally gathered through a row list) are split into pieces on their way into the LDS;
//   B = the weights, split ONCE per forward call into piece images laid out [K/32][piece][N][32] so that a tile is a
//   linear 16-byte copy (tmpnn_wide_prepare); the next K-step's operands are loaded into registers behind the
//   current step's MFMAs and written to the second LDS buffer afterwards (one barrier per step).
//
//   forward   P = h[dets] W_ih^T (Dn rows) ; per edge row gi = P[src] - P[dst] (linearity of the diff message) and
//             gh = h W_hh^T by the tiled GEMM with the GRU gates, the merge and the saved gate planes in its epilogue
//   backward  d_gi / d_gh are formed once by an elementwise pass (the gates are read exactly once), then
//             d_x = d_gi W_ih and d_h = dh z + d_gh W_hh are two tiled GEMMs with plain / accumulating stores.
//             dW = d_g^T [x | h] from the same materialised gate gradients: k_wide_dw (K = rows, transposing LDS reads).
#include "common.h"

namespace tmpnn {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float w_sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float w_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ uint32_t w_pk(float lo, float hi) {
    wbf16x2 v;
    v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
// two fp32 -> three packed-bf16 pieces (round-to-nearest residuals)
__device__ __forceinline__ void w_split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = w_pk(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = w_pk(r0, r1);
    r0 -= __uint_as_float(p2 << 16);
    r1 -= __uint_as_float(p2 & 0xFFFF0000u);
    p3 = w_pk(r0, r1);
}
__device__ __forceinline__ f32x16 w_mfma(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, a), __builtin_bit_cast(wbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int w_acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------------------------
// weight images: img[((kt * 3 + piece) * N + n) * 32 + sw(kk, n)] (bf16) = piece of B[32 kt + kk][n]
// ------------------------------------------------------------------------------------------------------------
// B[k][n] = W[n * ldw + k] (TRANS: forward, B = W^T) or W[k * ldw + n] (backward-data, B = W)
__global__ __launch_bounds__(256) void k_wide_prep(const float* __restrict__ W, int ldw, int K, int N, int trans,
                                                   uint16_t* __restrict__ img) {
    const long total = (long)K * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i / N), n = (int)(i % N);
        const float v = trans ? W[(size_t)n * ldw + k] : W[(size_t)k * ldw + n];
        uint32_t p1, p2, p3;
        w_split2(v, 0.f, p1, p2, p3);
        const int kt = k >> 5, kk = k & 31;
        // the 16-byte chunk (kk >> 3) of column n is stored at chunk position (kk >> 3) ^ ((n >> 2) & 3): the LDS
        // swizzle of w_sw, applied here once so that every consumer copies a tile LINEARLY (registers or LDS-DMA)
        const size_t base = ((size_t)kt * 3 * N + n) * 32 + ((((kk >> 3) ^ ((n >> 2) & 3)) << 3) | (kk & 7));
        img[base] = (uint16_t)p1;
        img[base + (size_t)N * 32] = (uint16_t)p2;
        img[base + (size_t)2 * N * 32] = (uint16_t)p3;
    }
}

// Half-step image of the forward W_hh operand for the ring kernel: img16[((j * 3 + piece) * N + n) * 16 + pos] = piece of
// B[16 j + kk][n], the two 16-byte chunks of a column swapped where (n >> 4) & 1 (pos = ((kk >> 3) ^ ((n >> 4) & 1)) * 8 +
// (kk & 7)): a [piece][192 columns][16] tile is a LINEAR copy and ds_read_b128 of chunk c ^ ((col >> 4) & 1) with
// lane = column is conflict-free (32-byte rows: the 16 lanes served together sit 2 r + c slots apart mod 16).
__global__ __launch_bounds__(256) void k_wide_prep16(const float* __restrict__ W, int ldw, int K, int N, int trans,
                                                     uint16_t* __restrict__ img) {
    const long total = (long)K * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i / N), n = (int)(i % N);
        const float v = trans ? W[(size_t)n * ldw + k] : W[(size_t)k * ldw + n];      // B = W^T (forward) or W (backward-data)
        uint32_t p1, p2, p3;
        w_split2(v, 0.f, p1, p2, p3);
        const int j = k >> 4, kk = k & 15;
        const size_t base = ((size_t)j * 3 * N + n) * 16 + ((((kk >> 3) ^ ((n >> 4) & 1)) << 3) | (kk & 7));
        img[base] = (uint16_t)p1;
        img[base + (size_t)N * 16] = (uint16_t)p2;
        img[base + (size_t)2 * N * 16] = (uint16_t)p3;
    }
}

// Half-step image of the forward W_hh operand for k_wide_gru_fwd_pp: the 18-KB block one HALF of the block reads in one half
// step of one item -- [piece 3][gate 3][64 hidden columns][16 k] -- is CONTIGUOUS, so its 18 one-KB request pieces are a
// linear copy: imgpp[((((j * (H / 128) + c) * 2 + hx) * 9 + piece * 3 + gate) * 64 + col) * 16 + pos] = piece of
// B[16 j + kk][gate H + 128 c + 64 hx + col], pos = ((kk >> 3) ^ ((col >> 4) & 1)) * 8 + (kk & 7) (the chunk swap of k_wide_prep16).
__global__ __launch_bounds__(256) void k_wide_prep_pp(const float* __restrict__ W, int ldw, int H, uint16_t* __restrict__ img) {
    const long total = (long)H * 3 * H;
    const int nchunk = H >> 7;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i / (3 * H)), n = (int)(i % (3 * H));
        const float v = W[(size_t)n * ldw + k];                                          // B = W_hh^T
        uint32_t p1, p2, p3;
        w_split2(v, 0.f, p1, p2, p3);
        const int j = k >> 4, kk = k & 15, gate = n / H, hcol = n % H, c = hcol >> 7, hx = (hcol >> 6) & 1, col = hcol & 63;
        const size_t blk = ((size_t)(j * nchunk + c) * 2 + hx) * 9;
        const size_t pos = (size_t)col * 16 + ((((kk >> 3) ^ ((col >> 4) & 1)) << 3) | (kk & 7));
        img[(blk + gate) * 1024 + pos] = (uint16_t)p1;
        img[(blk + 3 + gate) * 1024 + pos] = (uint16_t)p2;
        img[(blk + 6 + gate) * 1024 + pos] = (uint16_t)p3;
    }
}

// ------------------------------------------------------------------------------------------------------------
// the tiled product
// ------------------------------------------------------------------------------------------------------------
static constexpr int W_BM = 128, W_KT = 32, W_LD = 32;     // LDS rows of 32 bf16 = 64 B = four 16-byte chunks, no padding
// Chunk c of image row r sits at chunk c ^ ((r >> 2) & 3): the 16 lanes that ds_read_b128 serves together (lanes
// {0-3, 12-15, 20-27}, ... of a wave whose lane = row) then cover all sixteen 16-byte slots of the 256-byte bank row
// (rows equal mod 4 share a 64-byte quarter and differ in (r >> 2) & 3), and the 8-lane groups of ds_write_b128 (two
// rows x four chunks) still write 128 contiguous bytes.  Round 2 padded the rows to 80 B instead: 25 % more LDS.
__device__ __forceinline__ int w_sw(int row, int chunk) { return row * W_LD + ((chunk ^ ((row >> 2) & 3)) << 3); }

struct WideArgs {
    // A: R rows of K fp32, row r at A + (a_rows ? a_rows[r] : r) * lda
    const float* A; int lda; const int32_t* a_rows; int R; int K;
    int kskip_at, kskip;                        // A column of product index k: k < kskip_at ? k : k + kskip (kskip = 0: plain)
    const uint16_t* img; int N;                 // weight image [K/32][3][N][32]
    // STORE epilogue: C[(c_rows ? c_rows[r] : r) * ldc + n] (=|+=) acc
    float* C; int ldc; const int32_t* c_rows; int accumulate;
    // ... plus, in the ring form, the adjoint of row F: C[row r] += add_msg[add_src[r]] - add_msg[add_dst[r]] (columns n)
    const float* add_msg; int ld_add; const int32_t* add_src; const int32_t* add_dst;
    // GRU epilogue (forward): P [Dn][3H] projected det rows, src/dst det INDEX per row, state, biases, outputs
    const float* P; int ldp; const int32_t* src_pos; const int32_t* dst_pos;
    const float* h; int ld_h; int H; const float* b_ih; const float* b_hh;
    float* h_out; int ld_out; float* gates; size_t gate_plane; const int32_t* rows;
};

// One K-step of operands in registers: the global loads are issued BEFORE the MFMAs of the previous step and
// consumed (split, written to the other LDS buffer) after them.  512 threads: A 128 rows x 32 k -> 8 floats per
// thread; B NSEG * 64 columns x 32 k x 3 pieces -> NSEG * 768 chunks of 16 bytes.
template <int NSEG>
__device__ __forceinline__ void wide_load(const WideArgs& a, int r0, int kt, int n_base, int seg_stride, float4& qa0, float4& qa1,
                                          uint4 (&qb)[(NSEG * 768 + 511) / 512]) {
    const int tid = threadIdx.x;
    const int row = tid >> 2, kq = tid & 3;
    const int r = r0 + row;
    if (r < a.R) {
        int kc = kt * W_KT + 8 * kq;
        if (kc >= a.kskip_at) kc += a.kskip;
        const float4* p = reinterpret_cast<const float4*>(a.A + (size_t)(a.a_rows ? a.a_rows[r] : r) * a.lda + kc);
        qa0 = p[0]; qa1 = p[1];
    } else {
        qa0 = make_float4(0.f, 0.f, 0.f, 0.f); qa1 = qa0;
    }
    constexpr int CH = NSEG * 64 * 4;               // 16-byte chunks per piece
    constexpr int NB = (3 * CH + 511) / 512;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = min(tid + i * 512, 3 * CH - 1);      // (the surplus threads of the last pass re-load a valid chunk)
        const int p = idx / CH, c = idx % CH;
        const int col = c >> 2, qq = c & 3;
        const int n = min(n_base + (col >> 6) * seg_stride + (col & 63), a.N - 1);     // segment s starts at n_base + s * seg_stride
        const uint4 v = *reinterpret_cast<const uint4*>(a.img + (((size_t)kt * 3 + p) * a.N + n) * 32 + 8 * qq);
        qb[i] = v;
    }
}

template <int NSEG>
__device__ __forceinline__ void wide_store(const float4& qa0, const float4& qa1, const uint4 (&qb)[(NSEG * 768 + 511) / 512],
                                           uint16_t* sA, uint16_t* sB) {
    const int tid = threadIdx.x;
    {
        const int row = tid >> 2, kq = tid & 3;
        uint32_t q1[4], q2[4], q3[4];
        w_split2(qa0.x, qa0.y, q1[0], q2[0], q3[0]);
        w_split2(qa0.z, qa0.w, q1[1], q2[1], q3[1]);
        w_split2(qa1.x, qa1.y, q1[2], q2[2], q3[2]);
        w_split2(qa1.z, qa1.w, q1[3], q2[3], q3[3]);
        uint16_t* d = sA + w_sw(row, kq);
        constexpr int PL = W_BM * W_LD;
        *reinterpret_cast<uint4*>(d) = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        *reinterpret_cast<uint4*>(d + PL) = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        *reinterpret_cast<uint4*>(d + 2 * PL) = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    }
    constexpr int CH = NSEG * 64 * 4;
    constexpr int PLB = NSEG * 64 * W_LD;
    constexpr int NB = (3 * CH + 511) / 512;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + i * 512;
        if (idx < 3 * CH) {
            const int p = idx / CH, c = idx % CH;
            const int col = c >> 2, qq = c & 3;
            *reinterpret_cast<uint4*>(sB + p * PLB + col * W_LD + 8 * qq) = qb[i];      // (the image is pre-swizzled)
        }
    }
}

// acc[ct] += A(32 rows of this wave) x B(column tiles of this wave) over the staged K-step
// (column tile ct of the wave sits at LDS image row bcol_first + ct * BSTEP: 32 apart for the plain product, 64 apart --
//  one gate segment -- for the GRU product)
template <int NCT, int NSEG>
__device__ __forceinline__ void wide_mma(const uint16_t* sA, const uint16_t* sB, int arow0, int bcol_first, int lane,
                                         f32x16 (&acc)[NCT]) {
    constexpr int BSTEP = NSEG == 3 ? 64 : 32;
    constexpr int PL = W_BM * W_LD, PLB = NSEG * 64 * W_LD;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint4 af[3], bf[NCT][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
            af[p] = *reinterpret_cast<const uint4*>(sA + p * PL + w_sw(arow0 + r, 2 * s + hh));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bf[ct][p] = *reinterpret_cast<const uint4*>(sB + p * PLB + w_sw(bcol_first + ct * BSTEP + r, 2 * s + hh));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x16 c = acc[ct];
            c = w_mfma(af[2], bf[ct][0], c);       // smallest terms first
            c = w_mfma(af[0], bf[ct][2], c);
            c = w_mfma(af[1], bf[ct][1], c);
            c = w_mfma(af[1], bf[ct][0], c);
            c = w_mfma(af[0], bf[ct][1], c);
            c = w_mfma(af[0], bf[ct][0], c);
            acc[ct] = c;
        }
    }
}

// the K loop: register prefetch of step kt + 1 behind the MFMAs of step kt, two LDS buffers, one barrier per step
template <int NCT, int NSEG>
__device__ __forceinline__ void wide_kloop(const WideArgs& a, int r0, int n_base, int seg_stride, uint16_t* lds, int arow0,
                                           int bcol_first, int lane, f32x16 (&acc)[NCT]) {
    constexpr int SA = 3 * W_BM * W_LD, SB = 3 * NSEG * 64 * W_LD, BUF = SA + SB;
    // two register sets: the loads of step kt + 2 are issued while step kt computes, so every load has two MFMA
    // phases to land (one phase is shorter than an HBM round trip under load: measured, the K-step waited for it)
    constexpr int NB = (NSEG * 768 + 511) / 512;
    float4 xa0, xa1, ya0, ya1;
    uint4 xb[NB], yb[NB];
    const int nk = a.K / W_KT;
    uint16_t* buf0 = lds;
    uint16_t* buf1 = lds + BUF;
    wide_load<NSEG>(a, r0, 0, n_base, seg_stride, xa0, xa1, xb);
    if (nk > 1) wide_load<NSEG>(a, r0, 1, n_base, seg_stride, ya0, ya1, yb);
    wide_store<NSEG>(xa0, xa1, xb, buf0, buf0 + SA);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) wide_load<NSEG>(a, r0, kt + 2, n_base, seg_stride, xa0, xa1, xb);
        wide_mma<NCT, NSEG>(buf0, buf0 + SA, arow0, bcol_first, lane, acc);
        if (kt + 1 < nk) wide_store<NSEG>(ya0, ya1, yb, buf1, buf1 + SA);
        __syncthreads();
        if (kt + 1 >= nk) break;
        if (kt + 3 < nk) wide_load<NSEG>(a, r0, kt + 3, n_base, seg_stride, ya0, ya1, yb);
        wide_mma<NCT, NSEG>(buf1, buf1 + SA, arow0, bcol_first, lane, acc);
        if (kt + 2 < nk) wide_store<NSEG>(xa0, xa1, xb, buf0, buf0 + SA);
        __syncthreads();
    }
}

// XCD-aware tile order.  Blocks are dealt round-robin over the 8 XCDs (each with its own L2), so the gx column blocks of
// one row tile -- which read the SAME A rows -- are given ids that differ by 8: same XCD, dispatched back to back, and
// the rows are fetched from HBM once instead of gx times.  1-D grid of gx * 8 * ceil(gy / 8) blocks.
__device__ __forceinline__ bool wide_tile(int gx, int gy, int& bx, int& by) {
    const int b = blockIdx.x;
    const int xcd = b & 7, s = b >> 3;
    bx = s % gx;
    by = (s / gx) * 8 + xcd;
    return by < gy;
}
static int wide_grid(int gx, int gy) { return gx * 8 * ((gy + 7) / 8); }

static constexpr size_t W_STORE_SHM = sizeof(uint16_t) * 2 * (3 * W_BM * W_LD + 3 * 128 * W_LD);
static constexpr size_t W_GRU_SHM = sizeof(uint16_t) * 2 * (3 * W_BM * W_LD + 3 * 192 * W_LD);

// C = A B with a plain (or accumulating) store.  Tiles (ceil(N / 128) x ceil(R / 128)) in wide_tile order, 8 waves:
// wave = (32-row group, 64-column half)
__global__ __launch_bounds__(512) void k_wide_gemm_store(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    int bx, by;
    if (!wide_tile((a.N + 127) / 128, (a.R + W_BM - 1) / W_BM, bx, by)) return;
    const int r0 = by * W_BM, n0 = bx * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave & 3, wc = wave >> 2;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    wide_kloop<2, 2>(a, r0, n0, 64, w_dyn, 32 * wr, 64 * wc, lane, acc);
    // epilogue through the (now free) LDS: the accumulators hold one column per lane; rows leave as 16-byte accesses
    constexpr int LDC = 128 + 4;
    float* sC = reinterpret_cast<float*>(w_dyn);
    {
        const int c = lane & 31, half = lane >> 5;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                sC[(32 * wr + w_acc_row(reg, half)) * LDC + 64 * wc + 32 * ct + c] = acc[ct][reg];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < W_BM * 32; it += 512) {
        const int row = it >> 5, q = it & 31;
        const int r = r0 + row, n = n0 + 4 * q;
        if (r >= a.R || n >= a.N) continue;
        float4 v = *reinterpret_cast<const float4*>(sC + row * LDC + 4 * q);
        float* o = a.C + (size_t)(a.c_rows ? a.c_rows[r] : r) * a.ldc + n;
        if (a.accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(o);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *reinterpret_cast<float4*>(o) = v;
    }
}

// gh = h W_hh^T for 64 hidden units (3 x 64 gate columns) of 128 rows, GRU gates in the epilogue.
// Tiles (H / 64 hidden chunks x ceil(R / 128)) in wide_tile order: the hidden chunks of a row tile run on one XCD, so
// the A rows they share come out of its L2.  8 waves: wave = (32-row group, 32-hidden-unit half).
__global__ __launch_bounds__(512) void k_wide_gru_fwd(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    __shared__ int sRow[W_BM], sS[W_BM], sD[W_BM];
    const int H = a.H;
    int bx, by;
    if (!wide_tile(H / 64, (a.R + W_BM - 1) / W_BM, bx, by)) return;
    const int r0 = by * W_BM, hc0 = bx * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave & 3, wc = wave >> 2;
    if (threadIdx.x < W_BM) {
        const int r = r0 + threadIdx.x;
        const int rc = r < a.R ? r : a.R - 1;
        sRow[threadIdx.x] = a.rows[rc];
        sS[threadIdx.x] = a.src_pos[rc];
        sD[threadIdx.x] = a.dst_pos[rc];
    }
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    wide_kloop<3, 3>(a, r0, hc0, H, w_dyn, 32 * wr, 32 * wc, lane, acc);
    // epilogue through the (now free) LDS: gh tile [128 rows][3 gates x 64 hidden] fp32, then one thread per
    // (row, 4 hidden units): every global access of the gate math is a 16-byte one
    constexpr int LDC = 192 + 4;
    float* sC = reinterpret_cast<float*>(w_dyn);
    {
        const int c = lane & 31, half = lane >> 5;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                sC[(32 * wr + w_acc_row(reg, half)) * LDC + 64 * gate + 32 * wc + c] = acc[gate][reg];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < W_BM * 16; it += 512) {
        const int lr = it >> 4, q = it & 15;
        if (r0 + lr >= a.R) continue;
        const int grow = sRow[lr];
        const int col = hc0 + 4 * q;
        const float* ps = a.P + (size_t)sS[lr] * a.ldp + col;
        const float* pd = a.P + (size_t)sD[lr] * a.ldp + col;
        const float4 sr = *reinterpret_cast<const float4*>(ps), dr_ = *reinterpret_cast<const float4*>(pd);
        const float4 sz = *reinterpret_cast<const float4*>(ps + H), dz_ = *reinterpret_cast<const float4*>(pd + H);
        const float4 sn = *reinterpret_cast<const float4*>(ps + 2 * H), dn_ = *reinterpret_cast<const float4*>(pd + 2 * H);
        const float4 hp4 = *reinterpret_cast<const float4*>(a.h + (size_t)grow * a.ld_h + col);
        const float4 bir = *reinterpret_cast<const float4*>(a.b_ih + col), biz = *reinterpret_cast<const float4*>(a.b_ih + H + col);
        const float4 bin_ = *reinterpret_cast<const float4*>(a.b_ih + 2 * H + col);
        const float4 bhr = *reinterpret_cast<const float4*>(a.b_hh + col), bhz = *reinterpret_cast<const float4*>(a.b_hh + H + col);
        const float4 bhn = *reinterpret_cast<const float4*>(a.b_hh + 2 * H + col);
        const float4 ghr = *reinterpret_cast<const float4*>(sC + lr * LDC + 4 * q);
        const float4 ghz = *reinterpret_cast<const float4*>(sC + lr * LDC + 64 + 4 * q);
        const float4 ghn = *reinterpret_cast<const float4*>(sC + lr * LDC + 128 + 4 * q);
        const float gir[4] = {sr.x - dr_.x, sr.y - dr_.y, sr.z - dr_.z, sr.w - dr_.w};
        const float giz[4] = {sz.x - dz_.x, sz.y - dz_.y, sz.z - dz_.z, sz.w - dz_.w};
        const float gin[4] = {sn.x - dn_.x, sn.y - dn_.y, sn.z - dn_.z, sn.w - dn_.w};
        const float vr[4] = {ghr.x + bhr.x + bir.x, ghr.y + bhr.y + bir.y, ghr.z + bhr.z + bir.z, ghr.w + bhr.w + bir.w};
        const float vz[4] = {ghz.x + bhz.x + biz.x, ghz.y + bhz.y + biz.y, ghz.z + bhz.z + biz.z, ghz.w + bhz.w + biz.w};
        const float vhn[4] = {ghn.x + bhn.x, ghn.y + bhn.y, ghn.z + bhn.z, ghn.w + bhn.w};
        const float vbn[4] = {bin_.x, bin_.y, bin_.z, bin_.w}, hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        float orr[4], ozz[4], onn[4], oh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            orr[j] = w_sigm(gir[j] + vr[j]);
            ozz[j] = w_sigm(giz[j] + vz[j]);
            onn[j] = w_tanh(gin[j] + vbn[j] + orr[j] * vhn[j]);
            oh[j] = (1.0f - ozz[j]) * onn[j] + ozz[j] * hpv[j];
        }
        *reinterpret_cast<float4*>(a.h_out + (size_t)grow * a.ld_out + col) = make_float4(oh[0], oh[1], oh[2], oh[3]);
        if (a.gates) {
            float* gp = a.gates + (size_t)grow * H + col;
            *reinterpret_cast<float4*>(gp) = make_float4(orr[0], orr[1], orr[2], orr[3]);
            *reinterpret_cast<float4*>(gp + a.gate_plane) = make_float4(ozz[0], ozz[1], ozz[2], ozz[3]);
            *reinterpret_cast<float4*>(gp + 2 * a.gate_plane) = make_float4(onn[0], onn[1], onn[2], onn[3]);
            *reinterpret_cast<float4*>(gp + 3 * a.gate_plane) = make_float4(vhn[0], vhn[1], vhn[2], vhn[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// the same cell over EDGE TILES (struct tmpnn_edge_tiles): block-structured, LDS-staged projected det rows
// ------------------------------------------------------------------------------------------------------------
// A frame block of the rolling graph is a dense [A srcs x D_t dsts] set of edge rows (utils/graph.py:285-301), so 128
// edge rows chosen as (8 srcs x 16 dsts) touch 24 rows of the projected table P instead of 256 gathered ones.  The tile
// list (built once per graph: trackmpnn_amd.graph.build_edge_tiles) gives, per tile, its 128 graph rows, the list of the
// DISTINCT det indices it touches and each row's two positions in that list.  A persistent block walks tiles; for each
// of the H / 64 hidden chunks of a tile it runs a K loop and an epilogue that takes P[src] - P[dst] from an LDS copy of
// the tile's distinct P rows (fetched once per item by LDS-DMA while the K loop runs), the gate biases from LDS and the
// previous state from registers requested before the loop ends -- no global load is left on the epilogue's critical
// path.  A tile whose det list exceeds WT_DMAX rows (ragged graphs, the seams between frame blocks) reads P through the
// list from global memory instead.
struct WideTiles {
    const int32_t* t_row; const int32_t* t_loc; const int32_t* t_dptr; const int32_t* t_dets; int T;
};
static constexpr int WT_DMAX = 40;
// build-time ablations of k_wide_gru_fwd_ring (tools/build_variant.sh wide -DWT_NOEPI ...; timing only, wrong results):
// WT_NOEPI no gate math / stores, WT_NOMMA no MFMAs, WT_NODMA no operand DMA in the K loop, WT_NOREAD / WT_NOSPLIT no
// fragment reads / A split, WT_NOBAR no K-loop barriers.  The numbers they produced: DESIGN.md section 11.2

// (opaque(): the index arithmetic below is invariant over the kernel's persistent loop; hoisted out of it, the offsets
//  of all its call sites pile up in registers the K loop needs and the A operand in flight gets spilled)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// LDS-DMA: 16 bytes per lane from a per-lane global address to (wave-uniform LDS byte address) + 16 * lane; no register
// holds the data and no ds_write is issued.  Written as inline asm ON PURPOSE: hipcc waits vmcnt(0) before every LDS read
// that follows a __builtin_amdgcn_global_load_lds it cannot disambiguate (here: before the first ds_read of every K-step,
// i.e. the DMA of step k + 1 never ran under the MFMAs of step k).  An asm DMA is not in hipcc's bookkeeping, so its
// completion is waited for by hand: wait_dma() in every wave, then the barrier, then the ds_reads.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(const char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_wave_base) {
    unsigned keep;
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

typedef float wf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gru_gate4(const float4& ps_r, const float4& pd_r, const float4& ps_z, const float4& pd_z,
                                          const float4& ps_n, const float4& pd_n, const float4& ghr, const float4& ghz,
                                          const float4& ghn, const float* sBias, int q, const float4& hp4, float4& o_h,
                                          float4& o_r, float4& o_z, float4& o_n, float4& o_hn) {
    const float4 bir = *reinterpret_cast<const float4*>(sBias + 4 * q), biz = *reinterpret_cast<const float4*>(sBias + 64 + 4 * q);
    const float4 bin_ = *reinterpret_cast<const float4*>(sBias + 128 + 4 * q);
    const float4 bhr = *reinterpret_cast<const float4*>(sBias + 192 + 4 * q), bhz = *reinterpret_cast<const float4*>(sBias + 256 + 4 * q);
    const float4 bhn = *reinterpret_cast<const float4*>(sBias + 320 + 4 * q);
    const float gir[4] = {ps_r.x - pd_r.x, ps_r.y - pd_r.y, ps_r.z - pd_r.z, ps_r.w - pd_r.w};
    const float giz[4] = {ps_z.x - pd_z.x, ps_z.y - pd_z.y, ps_z.z - pd_z.z, ps_z.w - pd_z.w};
    const float gin[4] = {ps_n.x - pd_n.x, ps_n.y - pd_n.y, ps_n.z - pd_n.z, ps_n.w - pd_n.w};
    const float vr[4] = {ghr.x + bhr.x + bir.x, ghr.y + bhr.y + bir.y, ghr.z + bhr.z + bir.z, ghr.w + bhr.w + bir.w};
    const float vz[4] = {ghz.x + bhz.x + biz.x, ghz.y + bhz.y + biz.y, ghz.z + bhz.z + biz.z, ghz.w + bhz.w + biz.w};
    const float vhn[4] = {ghn.x + bhn.x, ghn.y + bhn.y, ghn.z + bhn.z, ghn.w + bhn.w};
    const float vbn[4] = {bin_.x, bin_.y, bin_.z, bin_.w}, hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    float orr[4], ozz[4], onn[4], oh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        orr[j] = w_sigm(gir[j] + vr[j]);
        ozz[j] = w_sigm(giz[j] + vz[j]);
        onn[j] = w_tanh(gin[j] + vbn[j] + orr[j] * vhn[j]);
        oh[j] = (1.0f - ozz[j]) * onn[j] + ozz[j] * hpv[j];
    }
    o_h = make_float4(oh[0], oh[1], oh[2], oh[3]);
    o_r = make_float4(orr[0], orr[1], orr[2], orr[3]);
    o_z = make_float4(ozz[0], ozz[1], ozz[2], ozz[3]);
    o_n = make_float4(onn[0], onn[1], onn[2], onn[3]);
    o_hn = make_float4(vhn[0], vhn[1], vhn[2], vhn[3]);
}

// ------------------------------------------------------------------------------------------------------------
// the tiled cell, second form: everything by LDS-DMA into a ring of four HALF K-steps
// ------------------------------------------------------------------------------------------------------------
// Measured on the first tiled kernel (C5, per 4.41 M rows): 13.2 ms; without the gate epilogue 10.2; without the MFMAs
// 9.7; without the operand traffic 9.4 -- the operand pipeline and the matrix phase each take ~6.5 ms and mostly ADD UP:
// a step's tiles are requested at the start of the step before and waited for at its end (one ~1 us phase of cover for a
// ~1.5 us round trip), and every phase has a vector-ALU head (DMA addresses) and tail (wait, A split, ds_write) that both
// waves of a SIMD run at the same time, between the barriers.  This form removes both:
//   * the A tile comes in by LDS-DMA too, as RAW fp32 rows (no register staging, no split + ds_write pass: each wave
//     splits its own 32 x 16 fragment after the ds_read -- twice the split work, but in front of the wave's own MFMAs
//     where the other wave of the SIMD can run under it -- and reads 2 instead of 3 A operands per step);
//   * K advances in half steps of 16 through a ring of FOUR slots (8 KB of A + 18 KB of weights each): the DMA of half
//     step j + 3 is issued in half step j, so every tile has three matrix phases to land, and a wave waits with a COUNTED
//     vmcnt (the two youngest half steps stay in flight) before a raw s_barrier.
// Products, their order and the split are those of k_wide_gru_fwd: bit-identical results.
static constexpr int RG_A = 128 * 64, RG_B = 3 * 192 * 32, RG_SLOT = RG_A + RG_B;      // bytes
static constexpr size_t W_RING_SHM = 4 * RG_SLOT + sizeof(float) * (WT_DMAX * 192 + 6 * 64) + sizeof(int) * (128 + 128 + 256);

// 16 bytes per lane from (uniform 64-bit base) + (per-lane 32-bit byte offset) to LDS (uniform address) + 16 * lane
__device__ __forceinline__ const void* uniform_ptr(const void* p) {          // (values hipcc cannot prove wave-uniform
    const uint64_t v = reinterpret_cast<uint64_t>(p);                        //  would reach an "s" operand in a VGPR)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void glds16_so(const void* sbase, uint32_t voff, uint32_t lds_wave_base) {
    unsigned keep;
    sbase = uniform_ptr(sbase);
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_wave_base) {      // 4 bytes per lane
    unsigned keep;
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait_barrier() {       // own DMAs down to the N youngest, own LDS writes, then the barrier
    if constexpr (N >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct RingCtx {
    const char* pa;              // this thread's A source: row a_row, floats 4 * (slot chunk) of half step 0
    uint32_t ob0, ob1, ob2;      // this thread's three weight-tile chunks (byte offsets within a half-step image)
    uint32_t lds_a, lds_b0, lds_b1, lds_b2;   // wave-uniform LDS addresses within slot 0
    bool b2_on;
};

// DMA of half step j of (tile rows pa, hidden chunk hc0) into ring slot `slot`: 1 A + 3 weight instructions per wave
__device__ __forceinline__ void ring_dma(const WideArgs& a, const RingCtx& c, int j, int hc0, int slot) {
    const uint32_t so = (uint32_t)slot * RG_SLOT;
    glds16(c.pa + (size_t)j * 64, c.lds_a + so);
    const char* wb = reinterpret_cast<const char*>(a.img) + ((size_t)j * 3 * a.N + hc0) * 32;
    glds16_so(wb, c.ob0, c.lds_b0 + so);
    glds16_so(wb, c.ob1, c.lds_b1 + so);
    if (c.b2_on) glds16_so(wb, c.ob2, c.lds_b2 + so);      // (lanes 0-15: the tile's last 128 chunks, 16 per wave)
    else asm volatile("s_nop 0" ::: "memory");
}

// One half step in two parts, so that a wave can fetch step p + 1's operands while its MFMAs of step p run:
//   ring_read   : the A fragment (raw fp32, two 16-byte reads) and the nine weight fragments of a slot into registers
//   ring_split  : raw A -> three bf16 pieces (vector ALU; independent of the MFMAs it is scheduled between)
//   ring_compute: 18 MFMAs on operands that are all in registers
struct RingOps { float4 lo, hi; uint4 af[3]; uint4 bf[9]; };
__device__ __forceinline__ void ring_read(const char* slot_base, int a_off0, int a_off1, int b_off, RingOps& o) {
    o.lo = *reinterpret_cast<const float4*>(slot_base + a_off0);
    o.hi = *reinterpret_cast<const float4*>(slot_base + a_off1);
    const char* sb = slot_base + RG_A + b_off;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int p = 0; p < 3; ++p) o.bf[ct * 3 + p] = *reinterpret_cast<const uint4*>(sb + ct * 64 * 32 + p * 192 * 32);
}
__device__ __forceinline__ void ring_split(RingOps& o) {
    w_split2(o.lo.x, o.lo.y, o.af[0].x, o.af[1].x, o.af[2].x);
    w_split2(o.lo.z, o.lo.w, o.af[0].y, o.af[1].y, o.af[2].y);
    w_split2(o.hi.x, o.hi.y, o.af[0].z, o.af[1].z, o.af[2].z);
    w_split2(o.hi.z, o.hi.w, o.af[0].w, o.af[1].w, o.af[2].w);
}
// (the pieces are pinned where they are formed: left alone, hipcc sinks the whole split into the next step, in front of
//  the MFMAs that consume it, and the wave runs split and MFMAs back to back again)
__device__ __forceinline__ void ring_pin(RingOps& o) {
    asm volatile("" : "+v"(o.af[0].x), "+v"(o.af[0].y), "+v"(o.af[0].z), "+v"(o.af[0].w), "+v"(o.af[1].x), "+v"(o.af[1].y),
                      "+v"(o.af[1].z), "+v"(o.af[1].w), "+v"(o.af[2].x), "+v"(o.af[2].y), "+v"(o.af[2].z), "+v"(o.af[2].w));
}
__device__ __forceinline__ void ring_compute(const RingOps& o, f32x16 (&acc)[3]) {
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        f32x16 c = acc[ct];
        c = w_mfma(o.af[2], o.bf[ct * 3], c);       // smallest terms first (as wide_mma)
        c = w_mfma(o.af[0], o.bf[ct * 3 + 2], c);
        c = w_mfma(o.af[1], o.bf[ct * 3 + 1], c);
        c = w_mfma(o.af[1], o.bf[ct * 3], c);
        c = w_mfma(o.af[0], o.bf[ct * 3 + 1], c);
        c = w_mfma(o.af[0], o.bf[ct * 3], c);
        acc[ct] = c;
    }
}

#define RING_MMA(...) ring_compute(__VA_ARGS__)
#define RING_DMA(...) ring_dma(__VA_ARGS__)
// the next step's A split (about 50 vector-ALU instructions) woven between this step's 18 MFMAs: three MFMAs cover the
// LDS round trip of the raw fragment, then three vector instructions per MFMA
#define RING_WEAVE()                                                                                   \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                             \
        _Pragma("unroll") for (int w_ = 0; w_ < 15; ++w_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        }                                                                                              \
    } while (0)
#ifdef TMPNN_KEEP_VARIANTS      // superseded by the opposite-phase form (round 5): comparison builds only
__global__ __launch_bounds__(512) void k_wide_gru_fwd_ring(WideArgs a, WideTiles tl) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    char* const ring = reinterpret_cast<char*>(w_dyn);
    float* const sC = reinterpret_cast<float*>(w_dyn);                    // [128][192], aliases the ring
    float* const sP = reinterpret_cast<float*>(ring + 4 * RG_SLOT);       // [WT_DMAX][3 gates x 64]
    float* const sBias = sP + WT_DMAX * 192;
    int* const sRow = reinterpret_cast<int*>(sBias + 6 * 64);
    int* const sLoc = sRow + 128;
    int* const sDet = sLoc + 128;
    const int H = a.H, nchunk = H >> 6, nsub = H >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave & 3, wc = wave >> 2;
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= tl.T) return;
    // ---- per-thread constants of the DMA and of the fragment reads
    RingCtx c;
    {
        const int N = a.N;
        auto boff = [&](int idx) {                         // chunk idx of a [3][192][2 x 16 B] weight tile -> image byte offset
            const int p = idx / 384, rem = idx % 384, col = rem >> 1, q = rem & 1;
            return (uint32_t)((p * N + (col >> 6) * H + (col & 63)) * 32 + q * 16);
        };
        c.ob0 = boff(tid); c.ob1 = boff(512 + tid);
        c.b2_on = lane < 16;
        c.ob2 = boff(1024 + 16 * wave + (lane & 15));
        const uint32_t base = lds_addr(ring);
        c.lds_a = base + 1024u * wave;
        c.lds_b0 = base + RG_A + 1024u * wave;
        c.lds_b1 = base + RG_A + 8192u + 1024u * wave;
        c.lds_b2 = base + RG_A + 16384u + 256u * wave;
    }
    const int r = lane & 31, hh = lane >> 5;
    const int arow = 32 * wr + r, fa = (arow >> 2) & 3;
    const int a_off0 = arow * 64 + (((2 * hh) ^ fa) << 4), a_off1 = arow * 64 + (((2 * hh + 1) ^ fa) << 4);
    const int b_off = (32 * wc + r) * 32 + ((hh ^ ((r >> 4) & 1)) << 4);
    const int drow = tid >> 2, dchunk = (tid & 3) ^ ((drow >> 2) & 3);        // A DMA: LDS chunk tid <- source chunk dchunk
    // ---- descriptor of the first tile
    int d_row = tid < 128 ? tl.t_row[(size_t)t * 128 + tid] : 0;
    int d_loc = tid < 128 ? tl.t_loc[(size_t)t * 128 + tid] : 0;
    int dp0 = tl.t_dptr[t], nd_next = tl.t_dptr[t + 1] - dp0;
    int d_det = (tid < 256 && tid < nd_next) ? tl.t_dets[dp0 + tid] : 0;
    int a_row_next = tl.t_row[(size_t)t * 128 + drow];
    // (the det-list bounds of the tile AFTER the next one: loaded a tile early, so that the next tile's descriptor
    //  request is not a chain of two dependent loads whose first wait would also drain every DMA in flight)
    const int tq = min(t + G, tl.T - 1);
    int dq0 = tl.t_dptr[tq], dq1 = tl.t_dptr[tq + 1];
    int nd = 0;
    const char* pa_next = nullptr;
    c.pa = reinterpret_cast<const char*>(a.A + (size_t)max(a_row_next, 0) * a.lda + 4 * dchunk);
    // the gate biases of a hidden chunk (b_ih r, z, n | b_hh r, z, n; 64 each) by 4-byte DMA: waves 0-5, one gate per wave
    auto bias_dma = [&](int hc0) {
        if (wave < 6) glds4((wave < 3 ? a.b_ih : a.b_hh) + (wave % 3) * H + hc0 + lane, lds_addr(sBias) + 256u * wave);
    };
    bias_dma(0);
    ring_dma(a, c, 0, 0, 0);
    ring_dma(a, c, 1, 0, 1);
    ring_dma(a, c, 2, 0, 2);
    ring_dma(a, c, 3, 0, 3);
    for (int bx = 0;;) {
        const int hc0 = bx << 6;
        if (bx == 0) {
            if (tid < 128) { sRow[tid] = d_row; sLoc[tid] = d_loc; }
            if (tid < 256) sDet[tid] = d_det;
            nd = __builtin_amdgcn_readfirstlane(nd_next);
        }
        ring_wait_barrier<12>();                              // half step 0 has landed; descriptor + biases visible
        const bool last_chunk = bx + 1 == nchunk;
        const bool more_tiles = t + G < tl.T;
        if (last_chunk && more_tiles) {                       // the next tile's descriptor, a whole item ahead of its use
            const int tn = t + G;
            d_row = tid < 128 ? tl.t_row[(size_t)tn * 128 + tid] : 0;
            d_loc = tid < 128 ? tl.t_loc[(size_t)tn * 128 + tid] : 0;
            dp0 = dq0; nd_next = dq1 - dq0;
            d_det = (tid < 256 && tid < nd_next) ? tl.t_dets[dp0 + tid] : 0;
            a_row_next = tl.t_row[(size_t)tn * 128 + drow];
            const int tq2 = min(tn + G, tl.T - 1);
            dq0 = tl.t_dptr[tq2]; dq1 = tl.t_dptr[tq2 + 1];
        }
        const bool staged = nd <= WT_DMAX;
        if (staged) {                                         // the tile's distinct P rows, this chunk's 3 x 64 columns
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx0 = 512 * i + 64 * wave;
                const int idx = idx0 + opaque(lane);
                if (idx < nd * 48) {
                    const int rw = idx / 48, cc = idx % 48;
                    glds16(a.P + (size_t)sDet[rw] * a.ldp + (cc >> 4) * H + hc0 + 4 * (cc & 15), lds_addr(sP) + 16u * idx0);
                }
            }
        }
        f32x16 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        wf32x4 hp[4];
        RingOps R[2];
        // Step p: DMA of step p + 4 into the slot step p's operands have just left (they sit in R[p & 1]); fetch step
        // p + 1's operands into R[(p + 1) & 1] and split its A fragment UNDER the MFMAs of step p; then wait until step
        // p + 2 has landed (steps p + 3, p + 4 stay in flight) and until this wave's LDS reads have returned; barrier.
        ring_read(ring, a_off0, a_off1, b_off, R[0]);
        ring_split(R[0]);
        ring_wait_barrier<8>();                               // step 1 has landed; every wave has step 0 in registers
        int p0 = 0;
        for (; p0 + 8 <= nsub; p0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                RING_DMA(a, c, p0 + u + 4, hc0, u);
                ring_read(ring + ((u + 1) & 3) * RG_SLOT, a_off0, a_off1, b_off, R[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);            // requests first; then the MFMAs with the split woven in
                RING_MMA(R[u & 1], acc);
                ring_split(R[(u + 1) & 1]);
                RING_WEAVE();
                __builtin_amdgcn_sched_barrier(0);
                ring_pin(R[(u + 1) & 1]);
                ring_wait_barrier<8>();
            }
        }
        // ---- the last four steps (p0 = nsub - 4): nothing left to request
        ring_read(ring + RG_SLOT, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        RING_MMA(R[0], acc);
        ring_split(R[1]);
        RING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        ring_pin(R[1]);
        ring_wait_barrier<4>();                               // step nsub - 2 has landed
        ring_read(ring + 2 * RG_SLOT, a_off0, a_off1, b_off, R[0]);
        __builtin_amdgcn_sched_barrier(0);
        RING_MMA(R[1], acc);
        ring_split(R[0]);
        RING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        ring_pin(R[0]);
        ring_wait_barrier<0>();                               // step nsub - 1 has landed
        {                                                     // previous state of the epilogue's rows (L2 hits: the A tile)
            const int th = opaque(tid);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int grow = sRow[(th >> 4) + 32 * i];
                hp[i] = *reinterpret_cast<const wf32x4*>(a.h + (size_t)max(grow, 0) * a.ld_h + hc0 + 4 * (th & 15));
            }
        }
        ring_read(ring + 3 * RG_SLOT, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        RING_MMA(R[0], acc);
        ring_split(R[1]);
        RING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        ring_pin(R[1]);
        // (the next tile's A source is formed HERE, where the queue is drained anyway: hipcc waits vmcnt(0) at the first
        //  use of the row id it loaded an item ago, and anywhere else that wait would drain the DMAs in flight)
        if (last_chunk && more_tiles) pa_next = reinterpret_cast<const char*>(a.A + (size_t)max(a_row_next, 0) * a.lda + 4 * dchunk);
        ring_wait_barrier<0>();                               // every wave has its last operands: sC may overwrite the ring
        // (hipcc must see the state rows as landed HERE: with them still pending in its books -- and a branch-dependent
        //  number of stores behind them -- it waits vmcnt(0) before each epilogue row, i.e. for the previous row's stores:
        //  12.21 -> 11.95 ms per C5 iteration)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(hp[i]));
        RING_MMA(R[1], acc);
        const int nbx = last_chunk ? 0 : bx + 1;
        const bool more = !last_chunk || more_tiles;
        {
            const int lw = opaque(lane), cl = lw & 31, half = lw >> 5;
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    sC[(32 * wr + w_acc_row(reg, half)) * 192 + 64 * gate + 32 * wc + cl] = acc[gate][reg];
        }
        __syncthreads();
        {
            const int te = opaque(tid), q = te & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lr = (te >> 4) + 32 * i;
                const int grow = sRow[lr];
                if (grow < 0) continue;
                const int loc = sLoc[lr];
                const int ls = loc & 0xFFFF, ldd = loc >> 16;
                float4 s_r, s_z, s_n, d_r, d_z, d_n;
                if (staged) {
                    const float* ps = sP + ls * 192 + 4 * q;
                    const float* pd = sP + ldd * 192 + 4 * q;
                    s_r = *reinterpret_cast<const float4*>(ps); s_z = *reinterpret_cast<const float4*>(ps + 64);
                    s_n = *reinterpret_cast<const float4*>(ps + 128);
                    d_r = *reinterpret_cast<const float4*>(pd); d_z = *reinterpret_cast<const float4*>(pd + 64);
                    d_n = *reinterpret_cast<const float4*>(pd + 128);
                } else {
                    const float* ps = a.P + (size_t)sDet[ls] * a.ldp + hc0 + 4 * q;
                    const float* pd = a.P + (size_t)sDet[ldd] * a.ldp + hc0 + 4 * q;
                    s_r = *reinterpret_cast<const float4*>(ps); s_z = *reinterpret_cast<const float4*>(ps + H);
                    s_n = *reinterpret_cast<const float4*>(ps + 2 * H);
                    d_r = *reinterpret_cast<const float4*>(pd); d_z = *reinterpret_cast<const float4*>(pd + H);
                    d_n = *reinterpret_cast<const float4*>(pd + 2 * H);
                }
                const float4 ghr = *reinterpret_cast<const float4*>(sC + lr * 192 + 4 * q);
                const float4 ghz = *reinterpret_cast<const float4*>(sC + lr * 192 + 64 + 4 * q);
                const float4 ghn = *reinterpret_cast<const float4*>(sC + lr * 192 + 128 + 4 * q);
                float4 o_h, o_r, o_z, o_n, o_hn;
                gru_gate4(s_r, d_r, s_z, d_z, s_n, d_n, ghr, ghz, ghn, sBias, q, make_float4(hp[i][0], hp[i][1], hp[i][2], hp[i][3]), o_h, o_r, o_z, o_n, o_hn);
                const int col = hc0 + 4 * q;
                *reinterpret_cast<float4*>(a.h_out + (size_t)grow * a.ld_out + col) = o_h;
                if (a.gates) {
                    float* gp = a.gates + (size_t)grow * H + col;
                    auto nt4 = [](float* p, const float4& v) {
                        wf32x4 x; x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
                        __builtin_nontemporal_store(x, reinterpret_cast<wf32x4*>(p));
                    };
                    nt4(gp, o_r); nt4(gp + a.gate_plane, o_z); nt4(gp + 2 * a.gate_plane, o_n); nt4(gp + 3 * a.gate_plane, o_hn);
                }
            }
        }
        __syncthreads();
        if (!more) break;
        if (last_chunk) {
            t += G;
            c.pa = pa_next;
        }
        bx = nbx;
        bias_dma(bx << 6);
        ring_dma(a, c, 0, bx << 6, 0);                        // the next item's first four half steps
        ring_dma(a, c, 1, bx << 6, 1);
        ring_dma(a, c, 2, bx << 6, 2);
        ring_dma(a, c, 3, bx << 6, 3);
    }
}
#endif  // TMPNN_KEEP_VARIANTS

// ------------------------------------------------------------------------------------------------------------
// the tiled cell, third form (round 5): 128 x 384 items, the two halves of the block in OPPOSITE phases
// ------------------------------------------------------------------------------------------------------------
// What the ring form above left on the table (C5, per 4.41 M rows: 11.7 ms against 5.6-6.3 ms of MFMAs; PMC traffic 1.51 x):
//   * its eight waves all run the same program between the same barriers -- request, read, split, 18 MFMAs -- so both waves
//     of a SIMD want the matrix pipe at the same time and wait at the same time; per half step a wave issues four LDS-DMA
//     pieces (~100 cycles of issue each) and ~50 vector instructions of A split for only 18 MFMAs (576 pipe cycles);
//   * an item is 128 rows x 192 gate columns, so the A rows of a tile are fetched once per 64 hidden columns (4 x at H = 256);
//   * the epilogue goes through a C tile that aliases the ring: two more barriers, and nothing can be in flight under it.
// Here:
//   * an item is 128 rows x (3 gates x 128 hidden columns): the A rows are fetched H / 128 times (2 x at H = 256);
//   * wave (wr, hx) owns rows 32 wr .. + 31 and the hidden columns 64 hx .. + 63 of all three gates: six 32 x 32 accumulators,
//     36 MFMAs per half step from ONE split of its A fragment (half the split work per MFMA) -- which only fits the register
//     file because nothing is double buffered: a wave's operands are read into registers in one barrier interval (its LOAD
//     segment) and consumed in the next (its MMA segment);
//   * the halves hx = 0 (waves 0-3, "X") and hx = 1 (waves 4-7, "Y": one wave of each per SIMD) run in opposite phases: while
//     X issues the 36 MFMAs of half step p, Y reads and splits its operands of step p and issues the requests of the
//     interval; then they swap.  The matrix pipe of a SIMD always has exactly one wave feeding it, and everything else (DMA
//     issue, LDS reads, the split, the waits) sits beside another wave's MFMAs by construction;
//   * A travels through a ring of four 8-KB slots (raw fp32 rows, as above), the weights through TWO sub-slots per half
//     (X reads only its 64 hidden columns of each gate, Y only its own: [piece][gate][64 columns][32 B] = 18 KB each), so a
//     sub-slot is free one interval after its reader's LOAD segment and the next-but-one step's pieces go out a whole half
//     step ahead.  The request stream is PERIODIC across items: the last steps of an item request the first steps of the
//     next one (same slots), so every LOAD segment issues the same number of pieces and every wait is a constant counted vmcnt;
//   * the tile's projected det rows (<= PP_NDMAX rows x 384 floats) are staged by one more piece per LOAD segment;
//   * the epilogue runs straight from the accumulators: with the row operand first, a lane holds ONE column of all three
//     gates for 16 rows per accumulator, so the gate arithmetic needs no C tile, no barrier and no LDS but the staged P
//     rows; outputs leave as 128-byte row segments (4 B per lane), the gate planes nontemporal.
// Same products in the same order as the forms above: bit-identical results.
static constexpr int PP_NDMAX = 32;                                   // det rows of a tile staged in LDS
static constexpr int PP_A = 128 * 64, PP_NA = 4;                      // A ring: four slots of 128 rows x 16 fp32
static constexpr int PP_BH = 9 * 2048;                                // weight sub-slot of one half: [piece 3][gate 3][64 columns][32 B]
static constexpr int PP_OFF_B = PP_NA * PP_A;                         // weight ring: [step parity 2][half 2]
static constexpr int PP_OFF_P = PP_OFF_B + 4 * PP_BH;                 // staged P rows [PP_NDMAX][384]
static constexpr int PP_OFF_D = PP_OFF_P + PP_NDMAX * 384 * 4;        // tile descriptors, double buffered: [row 128][loc 128][det 256]
static constexpr int PP_DESC = (128 + 128 + 256) * 4;
static constexpr int PP_OFF_X = PP_OFF_D + 2 * PP_DESC;               // 1 KB nobody reads: target of the filler piece
static constexpr size_t W_PP_SHM = PP_OFF_X + 1024;
#define PP_STAMP(k) do { } while (0)
#define PP_LGKM0() do { } while (0)

struct PpOps { uint4 af[3]; uint4 bf[18]; };      // bf[(ct * 3 + gate) * 3 + piece], ct = 32-column group of the wave's 64 hidden columns

// own requests down to the N youngest (the ones this segment issued), own LDS reads, then the barrier
template <int N>
__device__ __forceinline__ void pp_wait_barrier() {
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// what a wave needs to issue its share of an interval's requests
struct PpDma {
    const char* img;             // contiguous-block weight image (k_wide_prep_pp)
    int nchunk;                  // hidden chunks of 128 per half step of the image
    int nb;                      // weight pieces of this wave per interval: 5 (waves 0, 1 of a half) or 4
    uint32_t lane16;             // 16 * lane
    uint32_t lds;                // LDS byte address of the block's dynamic LDS
    uint32_t lds_a;              // + 1024 * wave: this wave's piece of an A slot
    int wi;                      // wave index within its half
};

// the weight pieces of half step j of hidden chunk c, for half hx_dst, into sub-slot (par, hx_dst): pieces wi, wi + 4, ...
// of the 18 (a linear copy of a contiguous 18-KB block of the image)
// A wave's weight pieces are CONSECUTIVE 1-KB pieces of the sub-slot (waves 0 / 1 of a half: pieces 0-4 / 5-9, waves 2 / 3:
// 10-13 / 14-17) and the image block is a linear copy of the sub-slot.  What a request costs is its INSTRUCTION, not its
// bytes or its M0 write (measured with s_memtime stamps, seven requests per LOAD segment beside the partner's MFMAs: ~205
// ticks each; the same with 4-byte pieces, with cache-hot sources, with one M0 write per group, and as plain 16-byte loads into
// registers) -- so PP_NM of a wave's weight pieces are issued from its own MMA segment, one behind each of the first
// accumulators' six MFMAs, where an issue costs a fraction of that.
#ifndef PP_LOAD_PRIO
#define PP_LOAD_PRIO 2     // s_setprio of a wave inside its LOAD segment (0 inside its MMA segment): its few requests / reads / split
#endif                     // instructions then win the SIMD's issue arbitration against the partner's MFMA stream, which needs a slot every 32 cycles only
__device__ __forceinline__ void pp_dma_b1(const PpDma& d, int j, int c, int hx_dst, int par, int k) {       // piece k of the wave's group
    const int q = (d.wi < 2 ? 5 * d.wi : 2 + 4 * d.wi) + k;
    const void* src = uniform_ptr(d.img + (size_t)((j * d.nchunk + c) * 2 + hx_dst) * PP_BH + 1024u * q);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(d.lds + PP_OFF_B + (uint32_t)(par * 2 + hx_dst) * PP_BH + 1024u * q);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(d.lane16), "s"(src), "s"(dst) : "memory", "m0");
}
// pieces [K0, 4) and the fifth where the wave has one (the part a LOAD segment issues); K0 = 0: the whole group
template <int K0>
__device__ __forceinline__ void pp_dma_b(const PpDma& d, int j, int c, int hx_dst, int par) {
#pragma unroll
    for (int k = K0; k < 4; ++k) pp_dma_b1(d, j, c, hx_dst, par, k);
    if (d.nb == 5) pp_dma_b1(d, j, c, hx_dst, par, 4);
}

__device__ __forceinline__ void pp_read(const char* lds, int aslot, int par, int hx, int a_off0, int a_off1, int b_off, float4& lo,
                                        float4& hi, PpOps& o) {
    const char* sa = lds + aslot * PP_A;
    lo = *reinterpret_cast<const float4*>(sa + a_off0);
    hi = *reinterpret_cast<const float4*>(sa + a_off1);
    const char* sb = lds + PP_OFF_B + (par * 2 + hx) * PP_BH + b_off;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                o.bf[(ct * 3 + g) * 3 + pc] = *reinterpret_cast<const uint4*>(sb + (pc * 3 + g) * 2048 + ct * 1024);
}
// w_split2 with its four subtractions as v_sub_f32 (inline asm): hipcc pairs them into v_pk_add_f32, and a PACKED fp32 add of the
// LOAD wave does not run beside the partner's MFMAs (s_memtime: ~50 vector instructions of split took 700-940 ticks in the LOAD
// segment; plain adds, converts and shifts do overlap -- profiles/r03_ubench_mfma_valu_overlap.md).  Same values bit for bit.
__device__ __forceinline__ float pp_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pp_split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = w_pk(x0, x1);
    float r0 = pp_sub(x0, __uint_as_float(p1 << 16)), r1 = pp_sub(x1, __uint_as_float(p1 & 0xFFFF0000u));
    p2 = w_pk(r0, r1);
    r0 = pp_sub(r0, __uint_as_float(p2 << 16));
    r1 = pp_sub(r1, __uint_as_float(p2 & 0xFFFF0000u));
    p3 = w_pk(r0, r1);
}
__device__ __forceinline__ void pp_split(const float4& lo, const float4& hi, PpOps& o) {
    pp_split2(lo.x, lo.y, o.af[0].x, o.af[1].x, o.af[2].x);
    pp_split2(lo.z, lo.w, o.af[0].y, o.af[1].y, o.af[2].y);
    pp_split2(hi.x, hi.y, o.af[0].z, o.af[1].z, o.af[2].z);
    pp_split2(hi.z, hi.w, o.af[0].w, o.af[1].w, o.af[2].w);
}
__device__ __forceinline__ void pp_pin(PpOps& o) {      // (as ring_pin: the split stays in the LOAD segment that formed it)
    asm volatile("" : "+v"(o.af[0].x), "+v"(o.af[0].y), "+v"(o.af[0].z), "+v"(o.af[0].w), "+v"(o.af[1].x), "+v"(o.af[1].y),
                      "+v"(o.af[1].z), "+v"(o.af[1].w), "+v"(o.af[2].x), "+v"(o.af[2].y), "+v"(o.af[2].z), "+v"(o.af[2].w));
}
// The 36 MFMAs of a half step with the wave's requests of the interval woven in, one behind the six MFMAs of each
// accumulator: its weight pieces of (half step jm, chunk cm) for half hxm (four, and the fifth where the wave has one), then
// -- youngest, so that the counted wait of the next LOAD segment may leave it in flight -- its A piece (asrc -> adst).
__device__ __forceinline__ void pp_mma(const PpOps& o, f32x16 (&acc)[6], const PpDma& d, int jm, int cm, int hxm, int parm, const char* asrc,
                                       uint32_t adst) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        f32x16 c = acc[t];
        c = w_mfma(o.af[2], o.bf[t * 3], c);        // smallest terms first (as wide_mma / ring_compute)
        c = w_mfma(o.af[0], o.bf[t * 3 + 2], c);
        c = w_mfma(o.af[1], o.bf[t * 3 + 1], c);
        c = w_mfma(o.af[1], o.bf[t * 3], c);
        c = w_mfma(o.af[0], o.bf[t * 3 + 1], c);
        c = w_mfma(o.af[0], o.bf[t * 3], c);
        acc[t] = c;
        __builtin_amdgcn_sched_barrier(0);
        if (t < 4) pp_dma_b1(d, jm, cm, hxm, parm, t);
        else if (t == 4) { if (d.nb == 5) pp_dma_b1(d, jm, cm, hxm, parm, 4); }
        else glds16(asrc, adst);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one element of the cell (the expressions of gru_gate4, so that both forms round alike)
__device__ __forceinline__ void gru_gate1(float ps_r, float pd_r, float ps_z, float pd_z, float ps_n, float pd_n, float ghr, float ghz,
                                          float ghn, float bir, float biz, float bin_, float bhr, float bhz, float bhn, float hp,
                                          float& o_h, float& o_r, float& o_z, float& o_n, float& o_hn) {
    const float gir = ps_r - pd_r, giz = ps_z - pd_z, gin = ps_n - pd_n;
    const float vr = ghr + bhr + bir, vz = ghz + bhz + biz, vhn = ghn + bhn;
    const float orr = w_sigm(gir + vr);
    const float ozz = w_sigm(giz + vz);
    const float onn = w_tanh(gin + bin_ + orr * vhn);
    o_h = (1.0f - ozz) * onn + ozz * hp;
    o_r = orr; o_z = ozz; o_n = onn; o_hn = vhn;
}

// The epilogue of one wave, straight from its six accumulators: lane = hidden column (cl of the 32-column group ct), registers =
// rows (w_acc_row), all three gates of an element in the same lane.  Outputs leave as 128-byte row segments (4 B per lane).
template <int HX, bool STAGED, bool GATES>
__device__ __forceinline__ void pp_epilogue(const WideArgs& a, const f32x16 (&acc)[6], const float* sP, const int* dsc, int hc0, int wr,
                                            int hh, int lane) {
    const int H = a.H;
    const int cl = opaque(lane) & 31;
    const int colw = hc0 + 64 * HX + cl;                           // + 32 ct
    float bir[2], biz[2], bin_[2], bhr[2], bhz[2], bhn[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = colw + 32 * ct;
        bir[ct] = a.b_ih[c]; biz[ct] = a.b_ih[H + c]; bin_[ct] = a.b_ih[2 * H + c];
        bhr[ct] = a.b_hh[c]; bhz[ct] = a.b_hh[H + c]; bhn[ct] = a.b_hh[2 * H + c];
    }
    // Eight outputs (four rows x two column groups) per pass, STAGE BY STAGE: the element's chain (three differences, two
    // sigmoids, a tanh, the merge: ~16 dependent vector / transcendental instructions) is latency-bound one element at a time
    // with two waves per SIMD; eight independent chains side by side keep the issue port busy.
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int lr0 = 32 * wr + 8 * rg + 4 * hh;
        const int4 rows4 = *reinterpret_cast<const int4*>(dsc + lr0);
        const int4 locs4 = *reinterpret_cast<const int4*>(dsc + 128 + lr0);
        const int rws[4] = {rows4.x, rows4.y, rows4.z, rows4.w}, lcs[4] = {locs4.x, locs4.y, locs4.z, locs4.w};
        float hp[8], xr[8], xz[8], xn[8], vhn[8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) hp[2 * j + ct] = a.h[(size_t)rws[j] * a.ld_h + colw + 32 * ct];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int reg = 4 * rg + j;
            const int ls = lcs[j] & 0xFFFF, ldd = lcs[j] >> 16;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float s_r, s_z, s_n, d_r, d_z, d_n;
                if constexpr (STAGED) {
                    const float* ps = sP + ls * 384 + 64 * HX + 32 * ct + cl;
                    const float* pd = sP + ldd * 384 + 64 * HX + 32 * ct + cl;
                    s_r = ps[0]; s_z = ps[128]; s_n = ps[256];
                    d_r = pd[0]; d_z = pd[128]; d_n = pd[256];
                } else {
                    const float* ps = a.P + (size_t)dsc[256 + ls] * a.ldp + colw + 32 * ct;
                    const float* pd = a.P + (size_t)dsc[256 + ldd] * a.ldp + colw + 32 * ct;
                    s_r = ps[0]; s_z = ps[H]; s_n = ps[2 * H];
                    d_r = pd[0]; d_z = pd[H]; d_n = pd[2 * H];
                }
                // (the expressions of gru_gate4, so that both forms round alike)
                const int e = 2 * j + ct;
                const float gir = s_r - d_r, giz = s_z - d_z, gin = s_n - d_n;
                const float vr = acc[ct * 3][reg] + bhr[ct] + bir[ct], vz = acc[ct * 3 + 1][reg] + bhz[ct] + biz[ct];
                vhn[e] = acc[ct * 3 + 2][reg] + bhn[ct];
                xr[e] = gir + vr;
                xz[e] = giz + vz;
                xn[e] = gin + bin_[ct];
            }
        }
        float orr[8], ozz[8], onn[8], oh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) orr[e] = w_sigm(xr[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) ozz[e] = w_sigm(xz[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) onn[e] = w_tanh(xn[e] + orr[e] * vhn[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) oh[e] = (1.0f - ozz[e]) * onn[e] + ozz[e] * hp[e];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int e = 2 * j + ct;
                {
                    a.h_out[(size_t)rws[j] * a.ld_out + colw + 32 * ct] = oh[e];
                    if constexpr (GATES) {
                        float* gp = a.gates + (size_t)rws[j] * H + colw + 32 * ct;
                        __builtin_nontemporal_store(orr[e], gp);
                        __builtin_nontemporal_store(ozz[e], gp + a.gate_plane);
                        __builtin_nontemporal_store(onn[e], gp + 2 * a.gate_plane);
                        __builtin_nontemporal_store(vhn[e], gp + 3 * a.gate_plane);
                    }
                }
            }
    }
}

// The whole persistent loop of one half (HX = 0: X, waves 0-3; HX = 1: Y, waves 4-7).  Two instantiations instead of branches
// on the half inside one loop: the operand registers of X live across an item boundary (loaded after X's epilogue, consumed
// by the next item's first MMA segment), those of Y do not, and with both in one control-flow graph hipcc kept two copies
// of the 84 operand registers alive (94 spilled registers, scratch reloads in front of the requests).
template <int HX>
__device__ __forceinline__ void pp_half(const WideArgs& a, const WideTiles& tl, char* const lds) {
    const float* const sP = reinterpret_cast<const float*>(lds + PP_OFF_P);
    const int H = a.H, nchunk = H >> 7, nsub = H >> 4;                     // hidden chunks of 128 per tile; half steps per item
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3;                                               // row group of the wave
    const int G = gridDim.x;
    int t = blockIdx.x;
    // ---- per-wave constants of the requests
    PpDma d;
    d.img = reinterpret_cast<const char*>(a.img);
    d.nchunk = nchunk;
    d.wi = wr;
    d.nb = wr < 2 ? 5 : 4;
    d.lds = lds_addr(lds);
    d.lds_a = d.lds + 1024u * wave;
    const uint32_t lane16 = 16u * lane;
    d.lane16 = lane16;
    // ---- fragment addresses
    const int r = lane & 31, hh = lane >> 5;
    const int arow = 32 * wr + r, fa = (arow >> 2) & 3;
    const int a_off0 = arow * 64 + (((2 * hh) ^ fa) << 4), a_off1 = arow * 64 + (((2 * hh + 1) ^ fa) << 4);
    const int b_off = r * 32 + ((hh ^ ((r >> 4) & 1)) << 4);
    const int drow = tid >> 2, dchunk = (tid & 3) ^ ((drow >> 2) & 3);     // A piece: LDS chunk tid <- (row drow, chunk dchunk)
    // ---- tile descriptors (padding slots -- the last tile only -- take slot 0's row and dets: they then compute and
    //      store what slot 0 computes, byte for byte, and the epilogue needs no row guard)
    auto desc_row = [&](int tile, int slot) {
        const int rw = tl.t_row[(size_t)tile * 128 + slot];
        return max(rw < 0 ? tl.t_row[(size_t)tile * 128] : rw, 0);
    };
    auto desc_loc = [&](int tile, int slot) {
        const int rw = tl.t_row[(size_t)tile * 128 + slot];
        return tl.t_loc[(size_t)tile * 128 + (rw < 0 ? 0 : slot)];
    };
    auto a_src = [&](int row) { return reinterpret_cast<const char*>(a.A + (size_t)row * a.lda + 4 * dchunk); };
    int par_d = 0;                                                         // descriptor buffer of the current tile
    int nd;
    {
        int* ds = reinterpret_cast<int*>(lds + PP_OFF_D);
        const int dp0 = tl.t_dptr[t];
        nd = tl.t_dptr[t + 1] - dp0;
        if (tid < 128) { ds[tid] = desc_row(t, tid); ds[128 + tid] = desc_loc(t, tid); }
        if (tid < 256) ds[256 + tid] = tid < nd ? tl.t_dets[dp0 + tid] : 0;
    }
    const char* pa = a_src(desc_row(t, drow));
    // ---- the first item's head, as the tail of a previous item would have requested it: A steps 0-2, the weights of step 0
    //      for both halves and of step 1 for X
    if (HX == 0) {
        pp_dma_b<0>(d, 0, 0, 1, 0);
    } else {
        pp_dma_b<0>(d, 0, 0, 0, 0);
        pp_dma_b<0>(d, 1, 0, 0, 1);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) glds16(pa + (size_t)j * 64, d.lds_a + (uint32_t)j * PP_A);
    pp_wait_barrier<0>();
    PpOps o;
    float4 lo, hi;
    if (HX == 0) {
        pp_read(lds, 0, 0, 0, a_off0, a_off1, b_off, lo, hi, o);
        pp_split(lo, hi, o);
        pp_pin(o);
    }
    pp_barrier();
    int bx = 0;
    for (;;) {
        const int hc0 = bx << 7;
        const bool last_chunk = bx + 1 == nchunk;
        const bool more_tiles = t + G < tl.T;
        const bool more = !last_chunk || more_tiles;
        // the item after this one (its first steps are requested by this item's last ones); none: this item again, unread
        const int bx_n = more ? (last_chunk ? 0 : bx + 1) : bx;
        const int* const dsc = reinterpret_cast<const int*>(lds + PP_OFF_D + par_d * PP_DESC);
        const bool staged = nd <= PP_NDMAX;
        const int np_need = staged ? (nd * 96 + 63) >> 6 : 0;             // 1-KB pieces of the tile's P rows (this chunk's columns)
        int a_row_next = 0;
        if (last_chunk && more_tiles) a_row_next = desc_row(t + G, drow);  // (needed by the last four steps' A requests)
        const char* pa_n = pa;
        f32x16 acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        // the staged-P piece of LOAD segment `seg` (0 .. 2 nsub - 1 in time order): piece 4 seg + wi, or the filler
        auto dma_p = [&](int seg) {
            const int piece = 4 * seg + d.wi;
            if (piece < np_need && seg <= 2 * nsub - 3) {
                const int idx = piece * 64 + (int)opaque(lane);
                const int rw = min(idx / 96, nd - 1), cc = idx % 96;
                glds16(a.P + (size_t)dsc[256 + rw] * a.ldp + (cc >> 5) * H + hc0 + 4 * (cc & 31), d.lds + PP_OFF_P + 1024u * piece);
            } else {
                glds16(d.img + lane16, d.lds + PP_OFF_X);
            }
        };
        // One half step p.  X: MMA(p) | LOAD(p + 1): requests weights(p + 2) of half Y, A rows 0-63 of step p + 4, reads slot
        // p + 1.  Y: LOAD(p): requests weights(p + 2) of half X, A rows 64-127 of step p + 3, reads slot p | MMA(p).
        // U = p & 3 is a compile-time constant (slots).  Steps past the item's end are the next item's first ones.
        // READ = false: X's last step of an item -- the requests of the interval, but the reads of the next item's step 0 wait
        // until X's epilogue is through (its operand registers would otherwise be alive across the epilogue).
#define PP_REQ(x) x
        // A step's requests: the weight pieces and the A piece ride in the wave's MMA segment (pp_mma), the staged-P piece in
        // its LOAD segment.  X, MMA(p): weights of step p + 1 for half Y (JB_, CH_), A rows 0-63 of step p + 3 (JA_);
        // Y, MMA(p): weights of step p + 2 for half X, A rows 64-127 of step p + 3.  LOAD segments: the reads are requested
        // FIRST (their latency runs under the request's issue), then the piece, then the split; the counted wait leaves the
        // segment's piece and the preceding MMA segment's A piece in flight.
#define PP_STEP(U, p, PA_, CH_, JB_, JA_, READ)                                                                    \
        do {                                                                                                       \
            const int sk_ = ((p) - 4) * 6;                                                                         \
            if (HX == 0) {                                                                                         \
                PP_STAMP(sk_);                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                pp_mma(o, acc, d, (JB_), (CH_), 1, ((U) + 1) & 1, (PA_) + (size_t)(JA_) * 64,                      \
                       d.lds_a + (uint32_t)(((U) + 3) & 3) * PP_A);                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                PP_STAMP(sk_ + 1);                                                                                 \
                pp_barrier();                                                                                      \
                PP_STAMP(sk_ + 2);                                                                                 \
                if (PP_LOAD_PRIO) __builtin_amdgcn_s_setprio(PP_LOAD_PRIO);                                        \
                if (READ) pp_read(lds, ((U) + 1) & 3, ((U) + 1) & 1, 0, a_off0, a_off1, b_off, lo, hi, o);         \
                PP_REQ(dma_p(2 * (p) + 1));                                                                        \
                PP_STAMP(sk_ + 3);                                                                                 \
                PP_LGKM0();                                                                                        \
                PP_STAMP(sk_ + 4);                                                                                 \
                if (READ) {                                                                                        \
                    pp_split(lo, hi, o);                                                                           \
                    pp_pin(o);                                                                                     \
                }                                                                                                  \
                PP_STAMP(sk_ + 5);                                                                                 \
                if (PP_LOAD_PRIO) __builtin_amdgcn_s_setprio(0);                                                   \
                pp_wait_barrier<2>();                                                                              \
            } else {                                                                                               \
                PP_STAMP(sk_);                                                                                     \
                if (PP_LOAD_PRIO) __builtin_amdgcn_s_setprio(PP_LOAD_PRIO);                                        \
                pp_read(lds, (U) & 3, (U) & 1, 1, a_off0, a_off1, b_off, lo, hi, o);                               \
                PP_REQ(dma_p(2 * (p)));                                                                            \
                PP_STAMP(sk_ + 1);                                                                                 \
                pp_split(lo, hi, o);                                                                               \
                pp_pin(o);                                                                                         \
                PP_STAMP(sk_ + 2);                                                                                 \
                if (PP_LOAD_PRIO) __builtin_amdgcn_s_setprio(0);                                                   \
                pp_wait_barrier<2>();                                                                              \
                PP_STAMP(sk_ + 3);                                                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                pp_mma(o, acc, d, (JB_), (CH_), 0, (U) & 1, (PA_) + (size_t)(JA_) * 64,                            \
                       d.lds_a + (uint32_t)(((U) + 3) & 3) * PP_A);                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                PP_STAMP(sk_ + 4);                                                                                 \
                pp_barrier();                                                                                      \
                PP_STAMP(sk_ + 5);                                                                                 \
            }                                                                                                      \
        } while (0)
        constexpr int DB = HX == 0 ? 1 : 2;                               // MMA(p) carries the weights of step p + DB (X: for Y's LOAD(p + 1); Y: for X's LOAD(p + 2))
        int p = 0;
        for (; p + 4 < nsub; p += 4) {
            PP_STEP(0, p, pa, bx, p + DB, p + 3, true);
            PP_STEP(1, p + 1, pa, bx, p + 1 + DB, p + 4, true);
            PP_STEP(2, p + 2, pa, bx, p + 2 + DB, p + 5, true);
            PP_STEP(3, p + 3, pa, bx, p + 3 + DB, p + 6, true);
        }
        // p = nsub - 4: the last four steps; requests past the item's end are the next item's
        if (last_chunk && more_tiles) pa_n = a_src(a_row_next);
        if (HX == 0) {
            PP_STEP(0, p, pa, bx, p + 1, p + 3, true);
            PP_STEP(1, p + 1, pa_n, bx, p + 2, 0, true);
            PP_STEP(2, p + 2, pa_n, bx, p + 3, 1, true);
            PP_STEP(3, p + 3, pa_n, bx_n, 0, 2, false);
        } else {
            PP_STEP(0, p, pa, bx, p + 2, p + 3, true);
            PP_STEP(1, p + 1, pa_n, bx, p + 3, 0, true);
            PP_STEP(2, p + 2, pa_n, bx_n, 0, 1, true);
            PP_STEP(3, p + 3, pa_n, bx_n, 1, 2, true);
        }
#undef PP_STEP
#undef PP_REQ
        // ---- the next tile's descriptor: requested here, written into the OTHER buffer behind the epilogue (nothing reads
        //      that buffer before the barrier below)
        int d_row = 0, d_loc = 0, d_det = 0, nd_next = nd;
        if (last_chunk && more_tiles) {
            const int tn = t + G;
            const int dp0 = tl.t_dptr[tn];
            nd_next = tl.t_dptr[tn + 1] - dp0;
            if (tid < 128) { d_row = desc_row(tn, tid); d_loc = desc_loc(tn, tid); }
            if (tid < 256 && tid < nd_next) d_det = tl.t_dets[dp0 + tid];
        }
        // ---- epilogue, from the accumulators (staged P rows or through the tile's det list; with / without the gate planes:
        //      compile-time forms -- a pointer that may be LDS or global becomes a flat load, which waits for everything)
        PP_STAMP(31);
        if (staged) {
            if (a.gates) pp_epilogue<HX, true, true>(a, acc, sP, dsc, hc0, wr, hh, lane);
            else pp_epilogue<HX, true, false>(a, acc, sP, dsc, hc0, wr, hh, lane);
        } else {
            if (a.gates) pp_epilogue<HX, false, true>(a, acc, sP, dsc, hc0, wr, hh, lane);
            else pp_epilogue<HX, false, false>(a, acc, sP, dsc, hc0, wr, hh, lane);
        }
        if (!more) break;
        if (last_chunk) {
            int* ds = reinterpret_cast<int*>(lds + PP_OFF_D + (par_d ^ 1) * PP_DESC);
            if (tid < 128) { ds[tid] = d_row; ds[128 + tid] = d_loc; }
            if (tid < 256) ds[256 + tid] = d_det;
        }
        if (HX == 0) {                                                     // X: the next item's step 0 (slot 0, parity 0), beside Y's epilogue
            pp_read(lds, 0, 0, 0, a_off0, a_off1, b_off, lo, hi, o);
            pp_split(lo, hi, o);
            pp_pin(o);
        }
        pp_barrier();                                                      // every wave is done with sP and the descriptor
        if (last_chunk) {
            t += G;
            pa = pa_n;
            par_d ^= 1;
            nd = __builtin_amdgcn_readfirstlane(nd_next);
            bx = 0;
        } else {
            ++bx;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the unread re-requests of the last item must land before the LDS is handed on)
}

__global__ __launch_bounds__(512) void k_wide_gru_fwd_pp(WideArgs a, WideTiles tl) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    if ((int)blockIdx.x >= tl.T) return;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) pp_half<0>(a, tl, reinterpret_cast<char*>(w_dyn));
    else pp_half<1>(a, tl, reinterpret_cast<char*>(w_dyn));
}

// ------------------------------------------------------------------------------------------------------------
// C (+)= A B with the ring structure of k_wide_gru_fwd_ring (the backward product d_h += d_gh W_hh: 4.41 M x 768 x 256)
// ------------------------------------------------------------------------------------------------------------
// k_wide_gemm_store stages both operands through registers with a barrier per 32-deep K-step (12.7 ms per C5 iteration for
// 5.6 ms of matrix-pipe time).  Here: 128 x 128 output tiles, K in half steps of 16 through a ring of four LDS slots
// (8 KB of raw fp32 A rows -- gathered through a row list, with the column skip of the [dr | dz | dn | dn r] image -- and
// 12 KB of pre-swizzled weight pieces), everything by LDS-DMA three half steps ahead, operands of step p + 1 fetched into
// registers and the A split woven under the MFMAs of step p; a persistent block takes the N / 128 column blocks of a row
// tile back to back (the second read of its A rows comes from L2).  Same products in the same order as
// k_wide_gemm_store: bit-identical.
static constexpr int GR_A = 128 * 64, GR_B = 3 * 128 * 32, GR_SLOT = GR_A + GR_B;       // bytes
static constexpr size_t W_GEMM_RING_SHM = 4 * GR_SLOT;                                   // sC [128][132] fp32 aliases it

template <int NCT> struct GRingOpsT { float4 lo, hi; uint4 af[3]; uint4 bf[3 * NCT]; };   // NCT 32-column tiles per wave
using GRingOps = GRingOpsT<2>;
template <int NCT>
__device__ __forceinline__ void gring_read(const char* slot_base, int a_off0, int a_off1, int b_off, GRingOpsT<NCT>& o) {
    o.lo = *reinterpret_cast<const float4*>(slot_base + a_off0);
    o.hi = *reinterpret_cast<const float4*>(slot_base + a_off1);
    const char* sb = slot_base + GR_A + b_off;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int p = 0; p < 3; ++p) o.bf[ct * 3 + p] = *reinterpret_cast<const uint4*>(sb + ct * 32 * 32 + p * (64 * NCT) * 32);
}
template <int NCT>
__device__ __forceinline__ void gring_split(GRingOpsT<NCT>& o) {
    w_split2(o.lo.x, o.lo.y, o.af[0].x, o.af[1].x, o.af[2].x);
    w_split2(o.lo.z, o.lo.w, o.af[0].y, o.af[1].y, o.af[2].y);
    w_split2(o.hi.x, o.hi.y, o.af[0].z, o.af[1].z, o.af[2].z);
    w_split2(o.hi.z, o.hi.w, o.af[0].w, o.af[1].w, o.af[2].w);
}
template <int NCT>
__device__ __forceinline__ void gring_pin(GRingOpsT<NCT>& o) {
    asm volatile("" : "+v"(o.af[0].x), "+v"(o.af[0].y), "+v"(o.af[0].z), "+v"(o.af[0].w), "+v"(o.af[1].x), "+v"(o.af[1].y),
                      "+v"(o.af[1].z), "+v"(o.af[1].w), "+v"(o.af[2].x), "+v"(o.af[2].y), "+v"(o.af[2].z), "+v"(o.af[2].w));
}
template <int NCT>
__device__ __forceinline__ void gring_compute(const GRingOpsT<NCT>& o, f32x16 (&acc)[NCT]) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        f32x16 c = acc[ct];
        c = w_mfma(o.af[2], o.bf[ct * 3], c);
        c = w_mfma(o.af[0], o.bf[ct * 3 + 2], c);
        c = w_mfma(o.af[1], o.bf[ct * 3 + 1], c);
        c = w_mfma(o.af[1], o.bf[ct * 3], c);
        c = w_mfma(o.af[0], o.bf[ct * 3 + 1], c);
        c = w_mfma(o.af[0], o.bf[ct * 3], c);
        acc[ct] = c;
    }
}
template <int N>
__device__ __forceinline__ void gring_wait_barrier() {      // three DMA instructions per wave and half step
    if constexpr (N >= 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define GRING_WEAVE()                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                             \
        _Pragma("unroll") for (int w_ = 0; w_ < 10; ++w_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        }                                                                                              \
    } while (0)

__global__ __launch_bounds__(512) void k_wide_gemm_ring(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    char* const ring = reinterpret_cast<char*>(w_dyn);
    float* const sC = reinterpret_cast<float*>(w_dyn);                    // [128][132], aliases the ring
    constexpr int LDC = 128 + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave & 3, wc = wave >> 2;
    const int nchunk = a.N >> 7, nsub = a.K >> 4;
    const int ntile = (a.R + 127) >> 7;
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntile) return;
    const uint32_t base = lds_addr(ring);
    const uint32_t lds_a = base + 1024u * wave, lds_b0 = base + GR_A + 1024u * wave, lds_b1 = base + GR_A + 8192u + 512u * wave;
    // weight-tile chunks of this thread: idx = (p * 128 + col) * 2 + q -> image byte offset
    auto boff = [&](int idx) { const int p = idx >> 8, rem = idx & 255; return (uint32_t)((p * a.N + (rem >> 1)) * 32 + (rem & 1) * 16); };
    const uint32_t ob0 = boff(tid), ob1 = boff(512 + 32 * wave + (lane & 31));
    const bool b1_on = lane < 32;
    const int r = lane & 31, hh = lane >> 5;
    const int arow = 32 * wr + r, fa = (arow >> 2) & 3;
    const int a_off0 = arow * 64 + (((2 * hh) ^ fa) << 4), a_off1 = arow * 64 + (((2 * hh + 1) ^ fa) << 4);
    const int b_off = (64 * wc + r) * 32 + ((hh ^ ((r >> 4) & 1)) << 4);
    const int drow = tid >> 2, dchunk = (tid & 3) ^ ((drow >> 2) & 3);
    auto a_row = [&](int tile) {
        const int rr = min(tile * 128 + drow, a.R - 1);                   // (rows past R repeat the last one: never stored)
        return a.a_rows ? a.a_rows[rr] : rr;
    };
    auto a_src = [&](int row) { return reinterpret_cast<const char*>(a.A + (size_t)row * a.lda + 4 * dchunk); };
    const char* pa = a_src(a_row(t));
    int row_next = 0;
    auto dma = [&](int j, int n0, int slot) {
        const uint32_t so = (uint32_t)slot * GR_SLOT;
        int kc = 16 * j;
        if (kc >= a.kskip_at) kc += a.kskip;
        glds16(pa + (size_t)kc * 4, lds_a + so);
        const char* wb = reinterpret_cast<const char*>(a.img) + ((size_t)j * 3 * a.N + n0) * 32;
        glds16_so(wb, ob0, lds_b0 + so);
        if (b1_on) glds16_so(wb, ob1, lds_b1 + so);
        else asm volatile("s_nop 0" ::: "memory");
    };
    dma(0, 0, 0); dma(1, 0, 1); dma(2, 0, 2); dma(3, 0, 3);
    for (int bx = 0;;) {
        const int n0 = bx << 7;
        gring_wait_barrier<9>();                                 // half step 0 has landed
        const bool last_chunk = bx + 1 == nchunk;
        const bool more_tiles = t + G < ntile;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        GRingOps R[2];
        gring_read(ring, a_off0, a_off1, b_off, R[0]);
        gring_split(R[0]);
        gring_wait_barrier<6>();                                 // step 1 has landed; every wave has step 0 in registers
        int p0 = 0;
        for (; p0 + 8 <= nsub; p0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dma(p0 + u + 4, n0, u);
                gring_read(ring + ((u + 1) & 3) * GR_SLOT, a_off0, a_off1, b_off, R[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                gring_compute(R[u & 1], acc);
                gring_split(R[(u + 1) & 1]);
                GRING_WEAVE();
                __builtin_amdgcn_sched_barrier(0);
                gring_pin(R[(u + 1) & 1]);
                gring_wait_barrier<6>();
            }
        }
        gring_read(ring + GR_SLOT, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[0], acc); gring_split(R[1]); GRING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[1]);
        gring_wait_barrier<3>();
        gring_read(ring + 2 * GR_SLOT, a_off0, a_off1, b_off, R[0]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[1], acc); gring_split(R[0]); GRING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[0]);
        gring_wait_barrier<0>();
        gring_read(ring + 3 * GR_SLOT, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[0], acc); gring_split(R[1]); GRING_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[1]);
        gring_wait_barrier<0>();                                 // every wave has its last operands: sC may overwrite the ring
        if (last_chunk && more_tiles) row_next = a_row(t + G);   // (requested with nothing in flight, used after the epilogue)
        gring_compute(R[1], acc);
        const int r0 = t * 128;
        // the accumulate operand of the epilogue's rows, requested before the tile is staged
        float4 cprev[8];
        int crow[8], asrc[8], adst[8];
        {
            const int te = opaque(tid);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rr = min(r0 + (te >> 5) + 16 * i, a.R - 1);
                crow[i] = a.c_rows ? a.c_rows[rr] : rr;
                if (a.add_msg) { asrc[i] = a.add_src[rr]; adst[i] = a.add_dst[rr]; }
            }
            if (a.accumulate) {
#pragma unroll
                for (int i = 0; i < 8; ++i) cprev[i] = *reinterpret_cast<const float4*>(a.C + (size_t)crow[i] * a.ldc + n0 + 4 * (te & 31));
            }
        }
        {
            const int lw = opaque(lane), cl = lw & 31, half = lw >> 5;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    sC[(32 * wr + w_acc_row(reg, half)) * LDC + 64 * wc + 32 * ct + cl] = acc[ct][reg];
        }
        __syncthreads();
        {
            const int te = opaque(tid), q = te & 31;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lr = (te >> 5) + 16 * i;
                if (r0 + lr >= a.R) continue;
                float4 v = *reinterpret_cast<const float4*>(sC + lr * LDC + 4 * q);
                if (a.accumulate) { v.x += cprev[i].x; v.y += cprev[i].y; v.z += cprev[i].z; v.w += cprev[i].w; }
                if (a.add_msg) {       // (det rows of add_msg: L2-resident; the sum in the order of tmpnn_gather_diff_fwd(accumulate))
                    const float4 ms = *reinterpret_cast<const float4*>(a.add_msg + (size_t)asrc[i] * a.ld_add + n0 + 4 * q);
                    const float4 md = *reinterpret_cast<const float4*>(a.add_msg + (size_t)adst[i] * a.ld_add + n0 + 4 * q);
                    v.x = (ms.x - md.x) + v.x; v.y = (ms.y - md.y) + v.y; v.z = (ms.z - md.z) + v.z; v.w = (ms.w - md.w) + v.w;
                }
                *reinterpret_cast<float4*>(a.C + (size_t)crow[i] * a.ldc + n0 + 4 * q) = v;
            }
        }
        __syncthreads();
        if (last_chunk && !more_tiles) break;
        if (last_chunk) { t += G; pa = a_src(row_next); bx = 0; } else ++bx;
        const int nn = bx << 7;
        dma(0, nn, 0); dma(1, nn, 1); dma(2, nn, 2); dma(3, nn, 3);
    }
}

// The same ring on 128 x 256 output tiles, for N in whole 256-column blocks (H = 256, 512, ...): a wave owns 32 rows x 128
// columns, so one split of its A rows feeds 24 MFMAs instead of 12 and the A rows of a tile are fetched once per 256
// columns.  Slots of 8 KB + 24 KB (four: 128 KB), four DMA instructions per wave and half step; sC [128][256] aliases
// the ring exactly, with the column-bit-5 flip on rows 4..7 (mod 8) that keeps the two halves of a wave on different
// banks without the pad.  Same products in the same order: bit-identical to the 128-column form.
static constexpr int GR_B4 = 3 * 256 * 32, GR_SLOT4 = GR_A + GR_B4;
static constexpr size_t W_GEMM_RING4_SHM = 4 * GR_SLOT4;
template <int N>
__device__ __forceinline__ void gring4_wait_barrier() {
    if constexpr (N >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define GRING4_WEAVE()                                                                                 \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                             \
        _Pragma("unroll") for (int w_ = 0; w_ < 22; ++w_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        }                                                                                              \
    } while (0)

__global__ __launch_bounds__(512) void k_wide_gemm_ring256(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    char* const ring = reinterpret_cast<char*>(w_dyn);
    float* const sC = reinterpret_cast<float*>(w_dyn);                    // [128][256], aliases the ring
    constexpr int LDC = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave & 3, wc = wave >> 2;
    const int nchunk = a.N >> 8, nsub = a.K >> 4;
    const int ntile = (a.R + 127) >> 7;
    const int G = gridDim.x;
    int t = blockIdx.x;
    if (t >= ntile) return;
    const uint32_t base = lds_addr(ring);
    const uint32_t lds_a = base + 1024u * wave, lds_b = base + GR_A + 1024u * wave;      // piece p at lds_b + 8192 p
    // weight-tile chunks of this thread: idx = (p * 256 + col) * 2 + q, p = 0..2 -> image byte offset of piece 0
    const uint32_t ob = (uint32_t)((tid >> 1) * 32 + (tid & 1) * 16);
    const uint32_t piece = (uint32_t)a.N * 32u;
    const int r = lane & 31, hh = lane >> 5;
    const int arow = 32 * wr + r, fa = (arow >> 2) & 3;
    const int a_off0 = arow * 64 + (((2 * hh) ^ fa) << 4), a_off1 = arow * 64 + (((2 * hh + 1) ^ fa) << 4);
    const int b_off = (128 * wc + r) * 32 + ((hh ^ ((r >> 4) & 1)) << 4);
    const int drow = tid >> 2, dchunk = (tid & 3) ^ ((drow >> 2) & 3);
    auto a_row = [&](int tile) {
        const int rr = min(tile * 128 + drow, a.R - 1);                   // (rows past R repeat the last one: never stored)
        return a.a_rows ? a.a_rows[rr] : rr;
    };
    auto a_src = [&](int row) { return reinterpret_cast<const char*>(a.A + (size_t)row * a.lda + 4 * dchunk); };
    const char* pa = a_src(a_row(t));
    int row_next = 0;
    auto dma = [&](int j, int n0, int slot) {
        const uint32_t so = (uint32_t)slot * GR_SLOT4;
        int kc = 16 * j;
        if (kc >= a.kskip_at) kc += a.kskip;
        glds16(pa + (size_t)kc * 4, lds_a + so);
        const char* wb = reinterpret_cast<const char*>(a.img) + ((size_t)j * 3 * a.N + n0) * 32;
        glds16_so(wb, ob, lds_b + so);
        glds16_so(wb, ob + piece, lds_b + so + 8192u);
        glds16_so(wb, ob + 2u * piece, lds_b + so + 16384u);
    };
    dma(0, 0, 0); dma(1, 0, 1); dma(2, 0, 2); dma(3, 0, 3);
    for (int bx = 0;;) {
        const int n0 = bx << 8;
        gring4_wait_barrier<12>();                               // half step 0 has landed
        const bool last_chunk = bx + 1 == nchunk;
        const bool more_tiles = t + G < ntile;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        GRingOpsT<4> R[2];
        gring_read(ring, a_off0, a_off1, b_off, R[0]);
        gring_split(R[0]);
        gring4_wait_barrier<8>();                                // step 1 has landed; every wave has step 0 in registers
        int p0 = 0;
        for (; p0 + 8 <= nsub; p0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dma(p0 + u + 4, n0, u);
                gring_read(ring + ((u + 1) & 3) * GR_SLOT4, a_off0, a_off1, b_off, R[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                gring_compute(R[u & 1], acc);
                gring_split(R[(u + 1) & 1]);
                GRING4_WEAVE();
                __builtin_amdgcn_sched_barrier(0);
                gring_pin(R[(u + 1) & 1]);
                gring4_wait_barrier<8>();
            }
        }
        gring_read(ring + GR_SLOT4, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[0], acc); gring_split(R[1]); GRING4_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[1]);
        gring4_wait_barrier<4>();
        gring_read(ring + 2 * GR_SLOT4, a_off0, a_off1, b_off, R[0]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[1], acc); gring_split(R[0]); GRING4_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[0]);
        gring4_wait_barrier<0>();
        gring_read(ring + 3 * GR_SLOT4, a_off0, a_off1, b_off, R[1]);
        __builtin_amdgcn_sched_barrier(0);
        gring_compute(R[0], acc); gring_split(R[1]); GRING4_WEAVE();
        __builtin_amdgcn_sched_barrier(0);
        gring_pin(R[1]);
        gring4_wait_barrier<0>();                                // every wave has its last operands: sC may overwrite the ring
        if (last_chunk && more_tiles) row_next = a_row(t + G);   // (requested with nothing in flight, used after the epilogue)
        gring_compute(R[1], acc);
        const int r0 = t * 128;
        // epilogue rows of this wave: wave + 8 i (wave-uniform: the row lists come through the scalar unit)
        const int q = opaque(lane);                              // float4 column 0..63
        int crow[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = min(r0 + wave + 8 * i, a.R - 1);
            crow[i] = a.c_rows ? a.c_rows[rr] : rr;
        }
        {
            const int cl = q & 31, half = q >> 5;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = 32 * wr + w_acc_row(reg, half);
                    sC[row * LDC + ((128 * wc + 32 * ct + cl) ^ (((row >> 2) & 1) << 5))] = acc[ct][reg];
                }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float4 cprev[8];
            if (a.accumulate) {
#pragma unroll
                for (int i = 0; i < 8; ++i) cprev[i] = *reinterpret_cast<const float4*>(a.C + (size_t)crow[8 * e + i] * a.ldc + n0 + 4 * q);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int lr = wave + 8 * (8 * e + i);
                if (r0 + lr >= a.R) continue;
                float4 v = *reinterpret_cast<const float4*>(sC + lr * LDC + ((4 * q) ^ (((lr >> 2) & 1) << 5)));
                if (a.accumulate) { v.x += cprev[i].x; v.y += cprev[i].y; v.z += cprev[i].z; v.w += cprev[i].w; }
                if (a.add_msg) {       // (det rows of add_msg: L2-resident; the sum in the order of tmpnn_gather_diff_fwd(accumulate))
                    const int rr = r0 + lr;
                    const int as = a.add_src[rr], ad = a.add_dst[rr];
                    const float4 ms = *reinterpret_cast<const float4*>(a.add_msg + (size_t)as * a.ld_add + n0 + 4 * q);
                    const float4 md = *reinterpret_cast<const float4*>(a.add_msg + (size_t)ad * a.ld_add + n0 + 4 * q);
                    v.x = (ms.x - md.x) + v.x; v.y = (ms.y - md.y) + v.y; v.z = (ms.z - md.z) + v.z; v.w = (ms.w - md.w) + v.w;
                }
                *reinterpret_cast<float4*>(a.C + (size_t)crow[8 * e + i] * a.ldc + n0 + 4 * q) = v;
            }
        }
        __syncthreads();
        if (last_chunk && !more_tiles) break;
        if (last_chunk) { t += G; pa = a_src(row_next); bx = 0; } else ++bx;
        const int nn = bx << 8;
        dma(0, nn, 0); dma(1, nn, 1); dma(2, nn, 2); dma(3, nn, 3);
    }
}

// ------------------------------------------------------------------------------------------------------------
// C (+)= A B on 128 x 256 tiles with the block's halves in opposite phases (round 5; the structure of k_wide_gru_fwd_pp)
// ------------------------------------------------------------------------------------------------------------
// The backward product d_h += d_gh W_hh (C5: 4.41 M x 768 x 256) in the ring form above runs all eight waves through the same
// request / read / split / MFMA phases between the same barriers (9.9 ms for 5.6 ms of matrix-pipe time).  Here, as in
// k_wide_gru_fwd_pp: half X (waves 0-3) owns output columns 0-127, half Y columns 128-255; a wave's tile is 32 rows x 128
// columns (four accumulators, 24 MFMAs per half step) with its operands read in one barrier interval (LOAD) and consumed in the
// next (MMA); X and Y alternate, so each SIMD's matrix pipe always has exactly one wave feeding it; the requests of an
// interval ride between the MMA wave's MFMAs (three weight pieces and the wave's A piece), the LOAD wave runs at raised
// priority, the A split uses scalar subtractions.  A tile's K loop (3H / 16 half steps) runs into the next tile's without
// draining (the request stream is periodic); the epilogue comes straight from the accumulators (lane = column): read-modify-
// write of C and the fused row-F adjoint in 128-byte row segments.  Same products in the same order as the forms above.
static constexpr int GP_A = 128 * 64, GP_NA = 4;                       // A ring: four slots of 128 rows x 16 fp32
static constexpr int GP_BH = 3 * 128 * 32;                             // weight sub-slot of one half: [piece 3][128 columns][32 B]
static constexpr int GP_OFF_B = GP_NA * GP_A;
static constexpr int GP_OFF_S = GP_OFF_B + 4 * GP_BH;                 // the split A pieces X publishes for Y: [step parity 2][piece 3][128 rows][32 B]
static constexpr int GP_S = 3 * 128 * 32;
static constexpr size_t W_GEMM_PP_SHM = GP_OFF_S + 2 * GP_S;          // 104 KB
#define GP_STAMP(k) do { } while (0)
struct GpOps { uint4 af[3]; uint4 bf[12]; };                           // bf[ct * 3 + piece]

// 4 x 4 transpose across the four lanes of a quad (DPP quad_perm): in, lane k holds column k of rows 0..3 in a0..a3; out,
// lane k holds row k, columns 0..3.  Two exchange stages (lane bit 0 / register bit 0, then bit 1 / bit 1): 16 vector instructions.
__device__ __forceinline__ float pp_dpp_x1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float pp_dpp_x2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void pp_transpose4(float& a0, float& a1, float& a2, float& a3, bool b0, bool b1) {
    float r;
    r = pp_dpp_x1(b0 ? a0 : a1); if (b0) a0 = r; else a1 = r;
    r = pp_dpp_x1(b0 ? a2 : a3); if (b0) a2 = r; else a3 = r;
    r = pp_dpp_x2(b1 ? a0 : a2); if (b1) a0 = r; else a2 = r;
    r = pp_dpp_x2(b1 ? a1 : a3); if (b1) a1 = r; else a3 = r;
}

// Half-step image of the backward W_hh operand for k_wide_gemm_pp256: the 12-KB block one half reads in one half step --
// [piece 3][128 columns][16 k] -- is contiguous: imgpp[(((j * 2 + hx) * 3 + piece) * 128 + col) * 16 + pos] = piece of
// B[16 j + kk][128 hx + col], pos as k_wide_prep16.  B[k][n] = W[k * ldw + n] (K = 3H gate rows, N = 256 hidden columns).
__global__ __launch_bounds__(256) void k_wide_prep_gpp(const float* __restrict__ W, int ldw, int K, uint16_t* __restrict__ img) {
    const long total = (long)K * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i >> 8), n = (int)(i & 255);
        const float v = W[(size_t)k * ldw + n];
        uint32_t p1, p2, p3;
        w_split2(v, 0.f, p1, p2, p3);
        const int j = k >> 4, kk = k & 15, hx = n >> 7, col = n & 127;
        const size_t blk = ((size_t)j * 2 + hx) * 3;
        const size_t pos = (size_t)col * 16 + ((((kk >> 3) ^ ((col >> 4) & 1)) << 3) | (kk & 7));
        img[(blk + 0) * 2048 + pos] = (uint16_t)p1;
        img[(blk + 1) * 2048 + pos] = (uint16_t)p2;
        img[(blk + 2) * 2048 + pos] = (uint16_t)p3;
    }
}

__device__ __forceinline__ void gp_read(const char* lds, int aslot, int par, int hx, int a_off0, int a_off1, int b_off, float4& lo,
                                        float4& hi, GpOps& o) {
    const char* sa = lds + aslot * GP_A;
    lo = *reinterpret_cast<const float4*>(sa + a_off0);
    hi = *reinterpret_cast<const float4*>(sa + a_off1);
    const char* sb = lds + GP_OFF_B + (par * 2 + hx) * GP_BH + b_off;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) o.bf[ct * 3 + pc] = *reinterpret_cast<const uint4*>(sb + pc * 4096 + ct * 1024);
}
__device__ __forceinline__ void gp_split(const float4& lo, const float4& hi, GpOps& o) {
    pp_split2(lo.x, lo.y, o.af[0].x, o.af[1].x, o.af[2].x);
    pp_split2(lo.z, lo.w, o.af[0].y, o.af[1].y, o.af[2].y);
    pp_split2(hi.x, hi.y, o.af[0].z, o.af[1].z, o.af[2].z);
    pp_split2(hi.z, hi.w, o.af[0].w, o.af[1].w, o.af[2].w);
    asm volatile("" : "+v"(o.af[0].x), "+v"(o.af[0].y), "+v"(o.af[0].z), "+v"(o.af[0].w), "+v"(o.af[1].x), "+v"(o.af[1].y),
                      "+v"(o.af[1].z), "+v"(o.af[1].w), "+v"(o.af[2].x), "+v"(o.af[2].y), "+v"(o.af[2].z), "+v"(o.af[2].w));
}
// The A fragment of (row group, half step) is needed by the X wave and by the Y wave of that row group; X reads it first (its
// LOAD segment comes an interval earlier), so X alone splits it and publishes the three bf16 pieces in LDS, and Y reads 48
// bytes instead of 32 raw ones + ~50 vector instructions of split: with 24 MFMAs per segment the LOAD segments were the longer
// ones (9.7 ms per C5 launch, no better than the ring form), and the split is the larger half of a LOAD segment.
__device__ __forceinline__ void gp_publish(char* lds, int par, int s_off, const GpOps& o) {
    char* sp = lds + GP_OFF_S + par * GP_S + s_off;
    *reinterpret_cast<uint4*>(sp) = o.af[0];
    *reinterpret_cast<uint4*>(sp + 4096) = o.af[1];
    *reinterpret_cast<uint4*>(sp + 8192) = o.af[2];
}
__device__ __forceinline__ void gp_read_y(const char* lds, int par, int s_off, int b_off, GpOps& o) {
    const char* sp = lds + GP_OFF_S + par * GP_S + s_off;
    o.af[0] = *reinterpret_cast<const uint4*>(sp);
    o.af[1] = *reinterpret_cast<const uint4*>(sp + 4096);
    o.af[2] = *reinterpret_cast<const uint4*>(sp + 8192);
    const char* sb = lds + GP_OFF_B + (par * 2 + 1) * GP_BH + b_off;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) o.bf[ct * 3 + pc] = *reinterpret_cast<const uint4*>(sb + pc * 4096 + ct * 1024);
}
// the wave's requests of an interval, woven behind the accumulators' MFMAs: its three weight pieces of (half step jb, half hxb)
// into sub-slot (parb, hxb), then -- youngest -- its A piece
__device__ __forceinline__ void gp_mma(const GpOps& o, f32x16 (&acc)[4], const char* img, int jb, int hxb, int parb, int wi, uint32_t lane16,
                                       uint32_t lds0, const char* asrc, uint32_t adst) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 c = acc[t];
        c = w_mfma(o.af[2], o.bf[t * 3], c);        // smallest terms first (as gring_compute)
        c = w_mfma(o.af[0], o.bf[t * 3 + 2], c);
        c = w_mfma(o.af[1], o.bf[t * 3 + 1], c);
        c = w_mfma(o.af[1], o.bf[t * 3], c);
        c = w_mfma(o.af[0], o.bf[t * 3 + 1], c);
        c = w_mfma(o.af[0], o.bf[t * 3], c);
        acc[t] = c;
        __builtin_amdgcn_sched_barrier(0);
        if (t < 3) {                                 // pieces wi, wi + 4, wi + 8 of the twelve
            const int q = wi + 4 * t;
            glds16_so(img + (size_t)(jb * 2 + hxb) * GP_BH + 1024u * q, lane16, lds0 + GP_OFF_B + (uint32_t)(parb * 2 + hxb) * GP_BH + 1024u * q);
        } else {
            glds16(asrc, adst);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// S = 32-row sub-tiles per wave: S = 1 -> 128-row tiles (above), S = 2 -> 256 x 256 tiles (below).
// Why 256 rows: what bounds these kernels is the CU's memory pipe, ~12-14 B per clock whatever the instruction mix (s_memtime:
// an epilogue of 512 KB took 45 k ticks as 4-byte and as 16-byte accesses alike; a forward item moves 1.1 MB in ~80 k ticks; a
// 128-row tile of this product 2.0 MB -- 0.38 of A, 1.15 of WEIGHTS re-streamed from L2, 0.5 of epilogue -- in ~150 k against
// 79 k of MFMAs).  The weights are the same for every row tile: with 256 rows per tile they are streamed half as often (1.47 MB
// per 128 rows).  A wave then owns 64 rows x 128 columns: eight accumulators (128 registers), 48 MFMAs per half step from one set
// of twelve weight fragments; LDS: A ring 4 x 16 KB, weights 48 KB, published pieces 2 x 24 KB = 160 KB.
template <int S> struct GqOps { uint4 af[S][3]; uint4 bf[12]; };
template <int S> static constexpr int gq_a() { return S * GP_A; }                      // bytes of an A slot
template <int S> static constexpr int gq_off_b() { return GP_NA * S * GP_A; }
template <int S> static constexpr int gq_off_s() { return gq_off_b<S>() + 4 * GP_BH; }
template <int S> static constexpr int gq_s() { return S * GP_S; }                     // bytes of a published-piece slot
template <int S> static constexpr size_t gq_shm() { return (size_t)gq_off_s<S>() + 2 * gq_s<S>(); }

template <int S>
__device__ __forceinline__ void gq_mma(const GqOps<S>& o, f32x16 (&acc)[S][4], const char* img, int jb, int hxb, int parb, int wi,
                                       uint32_t lane16, uint32_t lds0, const char* (&asrc)[S], size_t aoff, uint32_t adst) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            f32x16 c = acc[sb][t];
            c = w_mfma(o.af[sb][2], o.bf[t * 3], c);        // smallest terms first (as gring_compute)
            c = w_mfma(o.af[sb][0], o.bf[t * 3 + 2], c);
            c = w_mfma(o.af[sb][1], o.bf[t * 3 + 1], c);
            c = w_mfma(o.af[sb][1], o.bf[t * 3], c);
            c = w_mfma(o.af[sb][0], o.bf[t * 3 + 1], c);
            c = w_mfma(o.af[sb][0], o.bf[t * 3], c);
            acc[sb][t] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t < 3) {                                 // weight pieces wi, wi + 4, wi + 8 of the twelve
            const int q = wi + 4 * t;
            glds16_so(img + (size_t)(jb * 2 + hxb) * GP_BH + 1024u * q, lane16, lds0 + gq_off_b<S>() + (uint32_t)(parb * 2 + hxb) * GP_BH + 1024u * q);
        } else {                                     // youngest: the wave's A pieces (rows 16 w .. of each 128-row half of the tile)
#pragma unroll
            for (int sb = 0; sb < S; ++sb) glds16(asrc[sb] + aoff, adst + (uint32_t)sb * GP_A);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int HX, int S>
__device__ __forceinline__ void gq_half(const WideArgs& a, char* const lds) {
    constexpr int TR = 128 * S;                                            // rows of a tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave & 3;
    const int nsub = a.K >> 4;                                             // half steps per tile (multiple of 8)
    const int ntile = (a.R + TR - 1) / TR, G = gridDim.x;
    int t = blockIdx.x;
    const char* const img = reinterpret_cast<const char*>(a.img);
    const uint32_t lds0 = lds_addr(lds), lds_a = lds0 + 1024u * wave, lane16 = 16u * lane;
    const int r = lane & 31, hh = lane >> 5;
    // the wave's rows: 32 S wr + 32 sb + r.  An A slot is [S][128 rows][64 B] (sub-slot sb = rows 128 sb ..): row R of the tile
    // sits in sub-slot R >> 7 at row R & 127
    int a_off0[S], a_off1[S], s_off[S];
#pragma unroll
    for (int sb = 0; sb < S; ++sb) {
        const int trow = 32 * S * wr + 32 * sb + r, sl = trow >> 7, ar = trow & 127, fa = (ar >> 2) & 3;
        a_off0[sb] = sl * GP_A + ar * 64 + (((2 * hh) ^ fa) << 4);
        a_off1[sb] = sl * GP_A + ar * 64 + (((2 * hh + 1) ^ fa) << 4);
        s_off[sb] = sl * GP_S + ar * 32 + ((hh ^ ((ar >> 4) & 1)) << 4);
    }
    const int b_off = r * 32 + ((hh ^ ((r >> 4) & 1)) << 4);
    const int drow = tid >> 2, dchunk = (tid & 3) ^ ((drow >> 2) & 3);
    auto a_src = [&](int tile, int sb) {
        const int rr = min(tile * TR + 128 * sb + drow, a.R - 1);         // (rows past R repeat the last one: never stored)
        const int row = a.a_rows ? a.a_rows[rr] : rr;
        return reinterpret_cast<const char*>(a.A + (size_t)row * a.lda + 4 * dchunk);
    };
    auto kcol = [&](int j) { int kc = 16 * j; if (kc >= a.kskip_at) kc += a.kskip; return (size_t)kc * 4; };   // byte offset of half step j in an A row
    auto dma_b = [&](int j, int hxb, int par) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int q = wr + 4 * k;
            glds16_so(img + (size_t)(j * 2 + hxb) * GP_BH + 1024u * q, lane16, lds0 + gq_off_b<S>() + (uint32_t)(par * 2 + hxb) * GP_BH + 1024u * q);
        }
    };
    auto rd_b = [&](int par, int hx, GqOps<S>& o) {
        const char* sbp = lds + gq_off_b<S>() + (par * 2 + hx) * GP_BH + b_off;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) o.bf[ct * 3 + pc] = *reinterpret_cast<const uint4*>(sbp + pc * 4096 + ct * 1024);
    };
    // X: raw A rows of (slot), split, publish for Y at (par), and this half's weights
    auto load_x = [&](int aslot, int par, GqOps<S>& o) {
        float4 lo[S], hi[S];
        const char* sa = lds + aslot * gq_a<S>();
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            lo[sb] = *reinterpret_cast<const float4*>(sa + a_off0[sb]);
            hi[sb] = *reinterpret_cast<const float4*>(sa + a_off1[sb]);
        }
        rd_b(par, 0, o);
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            pp_split2(lo[sb].x, lo[sb].y, o.af[sb][0].x, o.af[sb][1].x, o.af[sb][2].x);
            pp_split2(lo[sb].z, lo[sb].w, o.af[sb][0].y, o.af[sb][1].y, o.af[sb][2].y);
            pp_split2(hi[sb].x, hi[sb].y, o.af[sb][0].z, o.af[sb][1].z, o.af[sb][2].z);
            pp_split2(hi[sb].z, hi[sb].w, o.af[sb][0].w, o.af[sb][1].w, o.af[sb][2].w);
            char* sp = lds + gq_off_s<S>() + par * gq_s<S>() + s_off[sb];
            *reinterpret_cast<uint4*>(sp) = o.af[sb][0];
            *reinterpret_cast<uint4*>(sp + 4096) = o.af[sb][1];
            *reinterpret_cast<uint4*>(sp + 8192) = o.af[sb][2];
        }
    };
    auto load_y = [&](int par, GqOps<S>& o) {
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            const char* sp = lds + gq_off_s<S>() + par * gq_s<S>() + s_off[sb];
            o.af[sb][0] = *reinterpret_cast<const uint4*>(sp);
            o.af[sb][1] = *reinterpret_cast<const uint4*>(sp + 4096);
            o.af[sb][2] = *reinterpret_cast<const uint4*>(sp + 8192);
        }
        rd_b(par, 1, o);
    };
    const char* pa[S];
#pragma unroll
    for (int sb = 0; sb < S; ++sb) pa[sb] = a_src(t, sb);
    // ---- the first tile's head, as the tail of a previous tile would have requested it: A steps 0-2, the weights of step 0
    //      for both halves and of step 1 for X
    if (HX == 0) { dma_b(0, 1, 0); }
    else { dma_b(0, 0, 0); dma_b(1, 0, 1); }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int sb = 0; sb < S; ++sb) glds16(pa[sb] + kcol(j), lds_a + (uint32_t)(j * gq_a<S>() + sb * GP_A));
    pp_wait_barrier<0>();
    GqOps<S> o;
    if (HX == 0) load_x(0, 0, o);
    pp_barrier();
    for (;;) {
        const bool more = t + G < ntile;
        f32x16 acc[S][4];
#pragma unroll
        for (int sb = 0; sb < S; ++sb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[sb][j][q] = 0.f;
        GqOps<S> oy;
        // X, MMA(p): weights of step p + 1 for half Y, A rows of step p + 3 (this wave's pieces); Y, MMA(p): weights of step p + 2 for
        // half X, its A pieces of step p + 3 (steps past the tile's end: the next tile's first ones).  LOAD: reads (X: split, publish).
#define GQ_STEP(U, JB_, JA_, READ)                                                                                 \
        do {                                                                                                       \
            if (HX == 0) {                                                                                         \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                gq_mma<S>(o, acc, img, (JB_), 1, ((U) + 1) & 1, wr, lane16, lds0, pa, kcol(JA_), lds_a + (uint32_t)(((U) + 3) & 3) * gq_a<S>()); \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                pp_barrier();                                                                                      \
                __builtin_amdgcn_s_setprio(2);                                                                     \
                if (READ) load_x(((U) + 1) & 3, ((U) + 1) & 1, o);                                                 \
                __builtin_amdgcn_s_setprio(0);                                                                     \
                pp_wait_barrier<S>();                                                                              \
            } else {                                                                                               \
                __builtin_amdgcn_s_setprio(2);                                                                     \
                load_y((U) & 1, oy);                                                                               \
                __builtin_amdgcn_s_setprio(0);                                                                     \
                pp_wait_barrier<S>();                                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                gq_mma<S>(oy, acc, img, (JB_), 0, (U) & 1, wr, lane16, lds0, pa, kcol(JA_), lds_a + (uint32_t)(((U) + 3) & 3) * gq_a<S>()); \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                pp_barrier();                                                                                      \
            }                                                                                                      \
        } while (0)
        constexpr int DB = HX == 0 ? 1 : 2;
        int p = 0;
        for (; p + 4 < nsub; p += 4) {
            GQ_STEP(0, p + DB, p + 3, true);
            GQ_STEP(1, p + 1 + DB, p + 4, true);
            GQ_STEP(2, p + 2 + DB, p + 5, true);
            GQ_STEP(3, p + 3 + DB, p + 6, true);
        }
        // the last four steps: after the first, every A request is the next tile's (none: this tile again, unread)
        GQ_STEP(0, p + DB, p + 3, true);
        if (more) {
#pragma unroll
            for (int sb = 0; sb < S; ++sb) pa[sb] = a_src(t + G, sb);
        }
        if (HX == 0) {
            GQ_STEP(1, p + 2, 0, true);
            GQ_STEP(2, p + 3, 1, true);
            GQ_STEP(3, 0, 2, false);
        } else {
            GQ_STEP(1, p + 3, 0, true);
            GQ_STEP(2, 0, 1, true);
            GQ_STEP(3, 1, 2, true);
        }
#undef GQ_STEP
        // ---- epilogue from the accumulators: lane = column (cl of the 32-column group ct), registers = rows; 16 outputs per pass
        {
            const int cl = opaque(lane) & 31;
            const int col0 = 128 * HX + cl;
            const int r0 = t * TR;
#pragma unroll
            for (int sb = 0; sb < S; ++sb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    __builtin_amdgcn_sched_barrier(0);
                    int crow[4], asr[4], adr[4];
                    bool ok[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = r0 + 32 * S * wr + 32 * sb + 8 * rg + 4 * hh + j;
                        ok[j] = rr < a.R;
                        const int rc = min(rr, a.R - 1);
                        crow[j] = a.c_rows ? a.c_rows[rc] : rc;
                        asr[j] = a.add_msg ? a.add_src[rc] : 0;
                        adr[j] = a.add_msg ? a.add_dst[rc] : 0;
                    }
                    float cp[4][4], ms[4][4], md[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            cp[j][ct] = a.accumulate ? a.C[(size_t)crow[j] * a.ldc + col0 + 32 * ct] : 0.f;
                            ms[j][ct] = a.add_msg ? a.add_msg[(size_t)asr[j] * a.ld_add + col0 + 32 * ct] : 0.f;
                            md[j][ct] = a.add_msg ? a.add_msg[(size_t)adr[j] * a.ld_add + col0 + 32 * ct] : 0.f;
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            float v = acc[sb][ct][4 * rg + j];
                            if (a.accumulate) v += cp[j][ct];
                            if (a.add_msg) v = (ms[j][ct] - md[j][ct]) + v;       // (the sum in the order of tmpnn_gather_diff_fwd(accumulate))
                            if (ok[j]) a.C[(size_t)crow[j] * a.ldc + col0 + 32 * ct] = v;
                        }
                }
        }
        if (!more) break;
        if (HX == 0) load_x(0, 0, o);                                      // X: the next tile's step 0 (slot 0, parity 0)
        pp_barrier();
        t += G;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the unread re-requests of the last tile must land before the LDS is handed on)
}

template <int S>
__global__ __launch_bounds__(512) void k_wide_gemm_pp256(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w_dyn[];
    if ((int)blockIdx.x >= (a.R + 128 * S - 1) / (128 * S)) return;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) gq_half<0, S>(a, reinterpret_cast<char*>(w_dyn));
    else gq_half<1, S>(a, reinterpret_cast<char*>(w_dyn));
}

// ------------------------------------------------------------------------------------------------------------
// backward, elementwise pass: d_gi = [dr | dz | dn], d_gh = [dr | dz | dn r] (compact rows), d_h[row] = dh z
// ------------------------------------------------------------------------------------------------------------
struct WideBwdArgs {
    const int32_t* rows; int R; int H;
    const float* h; int ld_h;
    const float* gates; size_t gate_plane;
    const float* d_hout; int ld_dhout; const float* dy; const float* w_head;
    float* dgi; float* dgh;          // [R][3H]
    float* d_h; int ld_dh;
};

__global__ __launch_bounds__(256) void k_wide_gates_bwd(WideBwdArgs a) {
    const int H = a.H, lpr = H >> 2;
    const long total = (long)a.R * lpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / lpr), c4 = (int)(i % lpr) * 4;
        const int row = a.rows[r];
        float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.d_hout) dh = *reinterpret_cast<const float4*>(a.d_hout + (size_t)row * a.ld_dhout + c4);
        if (a.dy) {
            const float d = a.dy[row];
            const float4 w = *reinterpret_cast<const float4*>(a.w_head + c4);
            dh.x += d * w.x; dh.y += d * w.y; dh.z += d * w.z; dh.w += d * w.w;
        }
        const float* gp = a.gates + (size_t)row * H + c4;
        const float4 gr = *reinterpret_cast<const float4*>(gp);
        const float4 gz = *reinterpret_cast<const float4*>(gp + a.gate_plane);
        const float4 gn = *reinterpret_cast<const float4*>(gp + 2 * a.gate_plane);
        const float4 gh = *reinterpret_cast<const float4*>(gp + 3 * a.gate_plane);
        const float4 hp = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + c4);
        const float dv[4] = {dh.x, dh.y, dh.z, dh.w}, rv[4] = {gr.x, gr.y, gr.z, gr.w}, zv[4] = {gz.x, gz.y, gz.z, gz.w};
        const float nv[4] = {gn.x, gn.y, gn.z, gn.w}, hv[4] = {gh.x, gh.y, gh.z, gh.w}, pv[4] = {hp.x, hp.y, hp.z, hp.w};
        float dr[4], dz[4], dn[4], dnr[4], dhz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = dv[j] * (1.0f - zv[j]) * (1.0f - nv[j] * nv[j]);
            dn[j] = t;
            dnr[j] = t * rv[j];
            dr[j] = t * hv[j] * rv[j] * (1.0f - rv[j]);
            dz[j] = dv[j] * (pv[j] - nv[j]) * zv[j] * (1.0f - zv[j]);
            dhz[j] = dv[j] * zv[j];
        }
        float* gi = a.dgi + (size_t)r * 3 * H + c4;
        float* gg = a.dgh + (size_t)r * 3 * H + c4;
        *reinterpret_cast<float4*>(gi) = make_float4(dr[0], dr[1], dr[2], dr[3]);
        *reinterpret_cast<float4*>(gi + H) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        *reinterpret_cast<float4*>(gi + 2 * H) = make_float4(dn[0], dn[1], dn[2], dn[3]);
        *reinterpret_cast<float4*>(gg) = make_float4(dr[0], dr[1], dr[2], dr[3]);
        *reinterpret_cast<float4*>(gg + H) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        *reinterpret_cast<float4*>(gg + 2 * H) = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
        *reinterpret_cast<float4*>(a.d_h + (size_t)row * a.ld_dh + c4) = make_float4(dhz[0], dhz[1], dhz[2], dhz[3]);
    }
}

// The same pass for the det-side form of the W_ih products (tmpnn_wide_gru_bwd_diff): ONE image per row, indexed by the
// graph row, dg4[row] = [dr | dz | dn | dn r] (4H floats: d_gi = columns 0..3H, d_gh = columns 0..2H and 3H..4H), and the
// column sums of dn (the third of db_ih that the W_hh-side weight-gradient launch does not see) as one [H] slab per block.
__global__ __launch_bounds__(256) void k_wide_gates_bwd4(WideBwdArgs a, float* __restrict__ dn_slabs) {
    __shared__ float red[256 * 4];
    // thread = (row slot, column group): 256 / lpr rows per pass (the surplus threads of widths whose H / 4 does not divide
    // 256 -- 384, 640, ... -- idle), so that a thread keeps its four columns over the whole grid stride
    const int H = a.H, lpr = H >> 2, rpb = 256 / lpr;
    const int slot = threadIdx.x / lpr, c4 = (threadIdx.x % lpr) * 4;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long rl = (long)blockIdx.x * rpb + slot; slot < rpb && rl < a.R; rl += (long)gridDim.x * rpb) {
        const int r = (int)rl;
        const int row = a.rows[r];
        float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.d_hout) dh = *reinterpret_cast<const float4*>(a.d_hout + (size_t)row * a.ld_dhout + c4);
        if (a.dy) {
            const float d = a.dy[row];
            const float4 w = *reinterpret_cast<const float4*>(a.w_head + c4);
            dh.x += d * w.x; dh.y += d * w.y; dh.z += d * w.z; dh.w += d * w.w;
        }
        const float* gp = a.gates + (size_t)row * H + c4;
        const float4 gr = *reinterpret_cast<const float4*>(gp);
        const float4 gz = *reinterpret_cast<const float4*>(gp + a.gate_plane);
        const float4 gn = *reinterpret_cast<const float4*>(gp + 2 * a.gate_plane);
        const float4 gh = *reinterpret_cast<const float4*>(gp + 3 * a.gate_plane);
        const float4 hp = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + c4);
        const float dv[4] = {dh.x, dh.y, dh.z, dh.w}, rv[4] = {gr.x, gr.y, gr.z, gr.w}, zv[4] = {gz.x, gz.y, gz.z, gz.w};
        const float nv[4] = {gn.x, gn.y, gn.z, gn.w}, hv[4] = {gh.x, gh.y, gh.z, gh.w}, pv[4] = {hp.x, hp.y, hp.z, hp.w};
        float dr[4], dz[4], dn[4], dnr[4], dhz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = dv[j] * (1.0f - zv[j]) * (1.0f - nv[j] * nv[j]);
            dn[j] = t;
            dnr[j] = t * rv[j];
            dr[j] = t * hv[j] * rv[j] * (1.0f - rv[j]);
            dz[j] = dv[j] * (pv[j] - nv[j]) * zv[j] * (1.0f - zv[j]);
            dhz[j] = dv[j] * zv[j];
            cs[j] += t;
        }
        float* gi = a.dgi + (size_t)row * 4 * H + c4;
        *reinterpret_cast<float4*>(gi) = make_float4(dr[0], dr[1], dr[2], dr[3]);
        *reinterpret_cast<float4*>(gi + H) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        *reinterpret_cast<float4*>(gi + 2 * H) = make_float4(dn[0], dn[1], dn[2], dn[3]);
        *reinterpret_cast<float4*>(gi + 3 * H) = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
        *reinterpret_cast<float4*>(a.d_h + (size_t)row * a.ld_dh + c4) = make_float4(dhz[0], dhz[1], dhz[2], dhz[3]);
    }
    // fixed-order sum over the 256 / lpr threads that share a column group
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x * 4 + j] = cs[j];
    __syncthreads();
    if ((int)threadIdx.x < lpr) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = threadIdx.x; t < rpb * lpr; t += lpr)
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] += red[t * 4 + j];
        *reinterpret_cast<float4*>(dn_slabs + (size_t)blockIdx.x * H + 4 * threadIdx.x) = make_float4(s4[0], s4[1], s4[2], s4[3]);
    }
}


// ------------------------------------------------------------------------------------------------------------
// weight gradient from the materialised gate gradients: dW[j][k] = sum_r dg[r][j] * X[r][k]   (bf16x6, K = rows)
// ------------------------------------------------------------------------------------------------------------
// Block (mt, nt, slab): 192 gate columns x 128 operand columns, streamed over the slab's rows in 32-row chunks.  The
// product contracts over ROWS, so both MFMA operands need 8 consecutive rows of one column per lane -- the transpose of
// the HBM layout: a chunk is staged row-major as bf16 pieces (16 threads per row; dg [32][256 (192 used)], X [32][128];
// 64-byte chunks XOR-swizzled by row & 3 as in k_gru_bwd_weights_split) and fetched with ds_read_b64_tr_b16.  8 waves,
// each 3 (A tiles) x 1 (B tile) accumulator tiles over the whole slab; the next chunk's rows are in registers behind the
// current chunk's MFMAs and go to the second LDS buffer afterwards (one barrier per chunk).  One [3H][H] slab per
// row slab, bias = column sums of dg taken by the staging threads (blocks with nt = 0).
typedef short ws16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 w_read_tr(const uint16_t* p) {
    const ws16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}
template <int COLS>
__device__ __forceinline__ int w_swz(int row, int col) { return row * COLS + ((((col >> 5) ^ (row & 3)) << 5) | (col & 31)); }

struct WideDwArgs {
    const float* dg; int ldg;                       // gate gradients: row r at dg + (grows ? grows[r] : r) * ldg, product column
    const int32_t* grows; int gskip_at, gskip;      // j at image column j < gskip_at ? j : j + gskip (the 4H image's d_gh)
    const float* X; int ldx;                        // operand rows: X[xa[r]] (- X[xb[r]] when xb != NULL)
    const int32_t* xa; const int32_t* xb;
    int R; int H; int rows_per_slab;
    float* slabs; float* bslabs;                    // [n_slab][3H][H], [n_slab][3H]
};

static constexpr int DW_A = 3 * 32 * 256, DW_B = 3 * 32 * 128;          // elements of one A / B image set
static constexpr int DW_SHM = 2 * (DW_A + DW_B) * 2;                    // two buffers, bf16

struct DwRaw { float4 a[3]; float4 b0[2], b1[2]; bool valid; };

__device__ __forceinline__ void dw_issue(const WideDwArgs& q, int r, int r_end, int ca, int cb, DwRaw& w) {
    w.valid = r < r_end;
    const int rr = w.valid ? r : r_end - 1;
    const float* pa = q.dg + (size_t)(q.grows ? q.grows[rr] : rr) * q.ldg;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        int cg = ca + 4 * g;
        if (cg >= q.gskip_at) cg += q.gskip;
        w.a[g] = *reinterpret_cast<const float4*>(pa + cg);
    }
    const float4* pb = reinterpret_cast<const float4*>(q.X + (size_t)q.xa[rr] * q.ldx + cb);
    w.b0[0] = pb[0]; w.b0[1] = pb[1];
    if (q.xb) {
        const float4* pc = reinterpret_cast<const float4*>(q.X + (size_t)q.xb[rr] * q.ldx + cb);
        w.b1[0] = pc[0]; w.b1[1] = pc[1];
    }
}

__global__ __launch_bounds__(512) void k_wide_dw(WideDwArgs q, int mt_count, int nt_count) {
    extern __shared__ float lds[];
    uint16_t* const base = reinterpret_cast<uint16_t*>(lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, half = lane >> 5;
    const int tile = blockIdx.x, mt = tile / nt_count, nt = tile % nt_count, slab = blockIdx.y;
    const int H = q.H;
    const int srow = tid >> 4, c16 = tid & 15;
    const int ca = mt * 192 + c16 * 12, cb = nt * 128 + c16 * 8;      // this thread's global columns of dg / X
    const int r_lo = slab * q.rows_per_slab, r_end = min(q.R, r_lo + q.rows_per_slab);
    // roles: A tiles 3 * (wave & 1) + {0, 1, 2} (32 gate columns each), B tile wave >> 1 (32 operand columns)
    const int jt0 = (wave & 1) * 3, bt = wave >> 1;
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, tg = (lane >> 4) & 1;
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    float cs[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) cs[i] = 0.f;

    auto stage = [&](const DwRaw& w, uint16_t* sA, uint16_t* sB) {
        float av[12], bv[8];
        const float4* a4 = w.a;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            av[4 * g] = a4[g].x; av[4 * g + 1] = a4[g].y; av[4 * g + 2] = a4[g].z; av[4 * g + 3] = a4[g].w;
        }
        bv[0] = w.b0[0].x; bv[1] = w.b0[0].y; bv[2] = w.b0[0].z; bv[3] = w.b0[0].w;
        bv[4] = w.b0[1].x; bv[5] = w.b0[1].y; bv[6] = w.b0[1].z; bv[7] = w.b0[1].w;
        if (q.xb) {
            bv[0] -= w.b1[0].x; bv[1] -= w.b1[0].y; bv[2] -= w.b1[0].z; bv[3] -= w.b1[0].w;
            bv[4] -= w.b1[1].x; bv[5] -= w.b1[1].y; bv[6] -= w.b1[1].z; bv[7] -= w.b1[1].w;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) { av[i] = w.valid ? av[i] : 0.f; cs[i] += av[i]; }
#pragma unroll
        for (int g = 0; g < 3; ++g) {                    // four columns = 8 bytes per piece
            uint2 p1, p2, p3;
            w_split2(av[4 * g], av[4 * g + 1], p1.x, p2.x, p3.x);
            w_split2(av[4 * g + 2], av[4 * g + 3], p1.y, p2.y, p3.y);
            const int off = w_swz<256>(srow, c16 * 12 + 4 * g);
            *reinterpret_cast<uint2*>(sA + off) = p1;
            *reinterpret_cast<uint2*>(sA + 32 * 256 + off) = p2;
            *reinterpret_cast<uint2*>(sA + 2 * 32 * 256 + off) = p3;
        }
        {
            uint4 p1, p2, p3;
            w_split2(bv[0], bv[1], p1.x, p2.x, p3.x); w_split2(bv[2], bv[3], p1.y, p2.y, p3.y);
            w_split2(bv[4], bv[5], p1.z, p2.z, p3.z); w_split2(bv[6], bv[7], p1.w, p2.w, p3.w);
            const int off = w_swz<128>(srow, c16 * 8);
            *reinterpret_cast<uint4*>(sB + off) = p1;
            *reinterpret_cast<uint4*>(sB + 32 * 128 + off) = p2;
            *reinterpret_cast<uint4*>(sB + 2 * 32 * 128 + off) = p3;
        }
    };

    DwRaw raw;
    const int nchunk = r_end > r_lo ? (r_end - r_lo + 31) / 32 : 0;
    if (nchunk > 0) {
        dw_issue(q, r_lo + srow, r_end, ca, cb, raw);
        stage(raw, base, base + DW_A);
        if (nchunk > 1) dw_issue(q, r_lo + 32 + srow, r_end, ca, cb, raw);
    }
    __syncthreads();
    // (measured and not kept: the fragments of step s + 1 requested before the MFMAs of step s through two rotating operand
    //  sets, the next chunk's staging moved into the middle of the matrix phase -- bit-identical and 1 ms per C5 iteration
    //  SLOWER than the compiler's own placement below, same box)
    for (int ch = 0; ch < nchunk; ++ch) {
        uint16_t* sA = base + (ch & 1) * (DW_A + DW_B);
        uint16_t* sB = sA + DW_A;
        uint16_t* nA = base + ((ch & 1) ^ 1) * (DW_A + DW_B);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row0 = kb * 16 + 8 * half + tq;
            uint4 b1, b2, b3;
            {
                const int col = bt * 32 + 16 * tg + 4 * tp;
                const uint16_t* p0 = sB + w_swz<128>(row0, col);
                const uint16_t* p1 = sB + w_swz<128>(row0 + 4, col);
                const uint2 u0 = w_read_tr(p0), u1 = w_read_tr(p1);
                const uint2 v0 = w_read_tr(p0 + 32 * 128), v1 = w_read_tr(p1 + 32 * 128);
                const uint2 w0 = w_read_tr(p0 + 2 * 32 * 128), w1 = w_read_tr(p1 + 2 * 32 * 128);
                b1 = make_uint4(u0.x, u0.y, u1.x, u1.y); b2 = make_uint4(v0.x, v0.y, v1.x, v1.y);
                b3 = make_uint4(w0.x, w0.y, w1.x, w1.y);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = (jt0 + j) * 32 + 16 * tg + 4 * tp;
                const uint16_t* p0 = sA + w_swz<256>(row0, col);
                const uint16_t* p1 = sA + w_swz<256>(row0 + 4, col);
                const uint2 u0 = w_read_tr(p0), u1 = w_read_tr(p1);
                const uint2 v0 = w_read_tr(p0 + 32 * 256), v1 = w_read_tr(p1 + 32 * 256);
                const uint2 w0 = w_read_tr(p0 + 2 * 32 * 256), w1 = w_read_tr(p1 + 2 * 32 * 256);
                const uint4 a1 = make_uint4(u0.x, u0.y, u1.x, u1.y), a2 = make_uint4(v0.x, v0.y, v1.x, v1.y);
                const uint4 a3 = make_uint4(w0.x, w0.y, w1.x, w1.y);
                acc[j] = w_mfma(a3, b1, acc[j]);
                acc[j] = w_mfma(a1, b3, acc[j]);
                acc[j] = w_mfma(a2, b2, acc[j]);
                acc[j] = w_mfma(a2, b1, acc[j]);
                acc[j] = w_mfma(a1, b2, acc[j]);
                acc[j] = w_mfma(a1, b1, acc[j]);
            }
        }
        if (ch + 1 < nchunk) {
            stage(raw, nA, nA + DW_A);
            if (ch + 2 < nchunk) dw_issue(q, r_lo + (ch + 2) * 32 + srow, r_end, ca, cb, raw);
        }
        __syncthreads();
    }
    // ---- the block's 192 x 128 tile of this slab
    float* sw = q.slabs + (size_t)slab * 3 * H * H;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int jj = mt * 192 + (jt0 + j) * 32 + w_acc_row(reg, half);
            sw[(size_t)jj * H + nt * 128 + bt * 32 + c] = acc[j][reg];
        }
    if (nt == 0) {                                           // bias gradient: column sums of dg over the slab's rows
        float* red = lds;                                    // [32][192] floats (the images are dead)
#pragma unroll
        for (int i = 0; i < 12; ++i) red[srow * 192 + c16 * 12 + i] = cs[i];
        __syncthreads();
        if (tid < 192) {
            float sum = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) sum += red[rr * 192 + tid];
            q.bslabs[(size_t)slab * 3 * H + mt * 192 + tid] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same weight gradient on a 384 x 128 block tile in 16-row chunks (k_wide_dw2, default; -DWT_DW_OLD keeps k_wide_dw)
// ------------------------------------------------------------------------------------------------------------
// k_wide_dw's 192 x 128 tile makes every wave read 4 operand fragments sets per 3 output tiles and the launch read dg
// twice and X four times (45 GB at C5).  Here a wave owns 3 x 2 output tiles (96 accumulators): 5 fragment sets per 6 tiles
// (30 transposing LDS reads per 36 MFMAs instead of 48), dg is read twice and X twice (36 GB), a thread stages 16 values per
// chunk instead of 20; chunks are ONE 16-row k block, so the MFMAs between two barriers stay 36 per wave and the two image
// sets fit in 96 KB.  Per accumulator the 16-row products come in the same order as in k_wide_dw; the slab partition (and with
// it the order of the final ordered reduction) differs.
static constexpr int DW2_A = 3 * 16 * 384, DW2_B = 3 * 16 * 128;        // elements of one A / B image set
static constexpr int DW2_SHM = 2 * (DW2_A + DW2_B) * 2;                 // two buffers, bf16

struct Dw2Raw { wf32x4 a[3]; wf32x4 b0, b1; bool valid; int ig, ia, ib, rr; bool vnext; };

// The rows of a chunk are addressed through row lists (edge_row, det_row): index load -> row load is a dependent chain,
// and the wait for the index in front of the row requests is a wait for EVERYTHING older in the in-order queue.  So the
// indices are requested a chunk before the rows that use them (they have landed when the rows are requested), and both
// are unconditional (rows past the slab's end repeat its last row and are staged as zeros).
template <bool XB>
__device__ __forceinline__ void dw2_index(const WideDwArgs& q, int r, int r_end, Dw2Raw& w) {
    w.vnext = r < r_end;
    const int rr = min(r, r_end - 1);
    w.rr = rr;
    w.ig = (q.grows ? q.grows : q.xa)[rr];               // (always a load, the select where it is used: a branch here would
    w.ia = q.xa[rr];                                     //  make the index a phi whose copy -- and wait -- sits behind the load)
    if constexpr (XB) w.ib = q.xb[rr]; else w.ib = 0;
}
template <bool XB>
__device__ __forceinline__ void dw2_rows(const WideDwArgs& q, int ca, int cb, Dw2Raw& w) {
    w.valid = w.vnext;
    // (the indices are first USED here: left alone, hipcc forms the row addresses earlier and waits for the index there --
    //  and with it for the row requests in front of it in the queue)
    asm volatile("" : "+v"(w.ig), "+v"(w.ia), "+v"(w.ib) : : "memory");
    const float* pa = q.dg + (size_t)(q.grows ? w.ig : w.rr) * q.ldg;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        int cg = ca + 4 * g;
        if (cg >= q.gskip_at) cg += q.gskip;
        w.a[g] = *reinterpret_cast<const wf32x4*>(pa + cg);
    }
    w.b0 = *reinterpret_cast<const wf32x4*>(q.X + (size_t)w.ia * q.ldx + cb);
    if constexpr (XB) w.b1 = *reinterpret_cast<const wf32x4*>(q.X + (size_t)w.ib * q.ldx + cb);
}

template <bool XB>
__global__ __launch_bounds__(512) void k_wide_dw2(WideDwArgs q, int mt_count, int nt_count) {
    extern __shared__ float lds[];
    uint16_t* const base = reinterpret_cast<uint16_t*>(lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 31, half = lane >> 5;
    // The tiles of a slab read the same dg / X rows (dg once per column tile, X once per row tile).  Workgroups go to the
    // eight XCDs round-robin in launch order: with the slab count a multiple of eight, a slab's tiles take consecutive
    // slots of ONE XCD, so the second reader of a row finds it in that XCD's L2 (same partition and sums: only the
    // placement of the blocks changes; -DWT_DW_PLAIN_ORDER keeps tile = blockIdx.x, slab = blockIdx.y).
    int tile = blockIdx.x, slab = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int id = blockIdx.x + gridDim.x * blockIdx.y, slot = id >> 3;
        tile = slot % (int)gridDim.x;
        slab = (id & 7) + 8 * (slot / (int)gridDim.x);
    }
    const int mt = tile / nt_count, nt = tile % nt_count;
    const int H = q.H;
    const int srow = tid >> 5, c32 = tid & 31;                       // staging: 16 rows x 32 threads
    const int ca = mt * 384 + c32 * 12, cb = nt * 128 + c32 * 4;     // this thread's global columns of dg / X
    const int r_lo = slab * q.rows_per_slab, r_end = min(q.R, r_lo + q.rows_per_slab);
    // roles: A tiles 3 * (wave & 3) + {0, 1, 2} of the block's twelve, B tiles 2 * (wave >> 2) + {0, 1} of its four
    const int jt0 = (wave & 3) * 3, bt0 = (wave >> 2) * 2;
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, tg = (lane >> 4) & 1;
    f32x16 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][b][i] = 0.f;
    float cs[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) cs[i] = 0.f;

    auto stage = [&](const Dw2Raw& w, uint16_t* sA, uint16_t* sB) {
        float av[12];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            av[4 * g] = w.a[g].x; av[4 * g + 1] = w.a[g].y; av[4 * g + 2] = w.a[g].z; av[4 * g + 3] = w.a[g].w;
        }
        float bv[4] = {w.b0.x, w.b0.y, w.b0.z, w.b0.w};
        if constexpr (XB) { bv[0] -= w.b1.x; bv[1] -= w.b1.y; bv[2] -= w.b1.z; bv[3] -= w.b1.w; }
#pragma unroll
        for (int i = 0; i < 12; ++i) { av[i] = w.valid ? av[i] : 0.f; cs[i] += av[i]; }
#pragma unroll
        for (int g = 0; g < 3; ++g) {                    // four columns = 8 bytes per piece
            uint2 p1, p2, p3;
            w_split2(av[4 * g], av[4 * g + 1], p1.x, p2.x, p3.x);
            w_split2(av[4 * g + 2], av[4 * g + 3], p1.y, p2.y, p3.y);
            const int off = w_swz<384>(srow, c32 * 12 + 4 * g);
            *reinterpret_cast<uint2*>(sA + off) = p1;
            *reinterpret_cast<uint2*>(sA + 16 * 384 + off) = p2;
            *reinterpret_cast<uint2*>(sA + 2 * 16 * 384 + off) = p3;
        }
        {
            uint2 p1, p2, p3;
            w_split2(bv[0], bv[1], p1.x, p2.x, p3.x); w_split2(bv[2], bv[3], p1.y, p2.y, p3.y);
            const int off = w_swz<128>(srow, c32 * 4);
            *reinterpret_cast<uint2*>(sB + off) = p1;
            *reinterpret_cast<uint2*>(sB + 16 * 128 + off) = p2;
            *reinterpret_cast<uint2*>(sB + 2 * 16 * 128 + off) = p3;
        }
    };

    // The rows of a chunk are requested TWO chunks before they are staged (two register sets A / B, the loop unrolled by
    // two; their indices two chunks before that): a chunk's matrix phase (~1.5 us) is shorter than a loaded memory round
    // trip, and with one chunk of cover a launch took matrix time + memory time (11.3 ms at C5, 6.3 for the MFMAs alone).
    // Everything is unconditional: rows past the slab's end repeat its last row and are staged as zeros, a chunk past
    // its end is zeros into a buffer nobody reads.
    Dw2Raw rawA, rawB;
    rawA.b1 = rawB.b1 = wf32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunk = r_end > r_lo ? (r_end - r_lo + 15) / 16 : 0;
    if (nchunk > 0) {
        dw2_index<XB>(q, r_lo + srow, r_end, rawA);
        dw2_index<XB>(q, r_lo + 16 + srow, r_end, rawB);
        dw2_rows<XB>(q, ca, cb, rawA);                   // chunk 0
        dw2_index<XB>(q, r_lo + 32 + srow, r_end, rawA);
        stage(rawA, base, base + DW2_A);
        dw2_rows<XB>(q, ca, cb, rawB);                   // chunk 1 (staged by the iteration of chunk 0)
        dw2_index<XB>(q, r_lo + 48 + srow, r_end, rawB);
        dw2_rows<XB>(q, ca, cb, rawA);                   // chunk 2
        dw2_index<XB>(q, r_lo + 64 + srow, r_end, rawA);
    }
    __syncthreads();
    const int row0 = 8 * half + tq;
    auto compute = [&](const uint16_t* sA) {
        const uint16_t* sB = sA + DW2_A;
        uint4 bq[2][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = (bt0 + b) * 32 + 16 * tg + 4 * tp;
            const uint16_t* p0 = sB + w_swz<128>(row0, col);
            const uint16_t* p1 = sB + w_swz<128>(row0 + 4, col);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const uint2 u0 = w_read_tr(p0 + pc * 16 * 128), u1 = w_read_tr(p1 + pc * 16 * 128);
                bq[b][pc] = make_uint4(u0.x, u0.y, u1.x, u1.y);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = (jt0 + j) * 32 + 16 * tg + 4 * tp;
            const uint16_t* p0 = sA + w_swz<384>(row0, col);
            const uint16_t* p1 = sA + w_swz<384>(row0 + 4, col);
            uint4 aq[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const uint2 u0 = w_read_tr(p0 + pc * 16 * 384), u1 = w_read_tr(p1 + pc * 16 * 384);
                aq[pc] = make_uint4(u0.x, u0.y, u1.x, u1.y);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x16 v = acc[j][b];
                v = w_mfma(aq[2], bq[b][0], v);
                v = w_mfma(aq[0], bq[b][2], v);
                v = w_mfma(aq[1], bq[b][1], v);
                v = w_mfma(aq[1], bq[b][0], v);
                v = w_mfma(aq[0], bq[b][1], v);
                v = w_mfma(aq[0], bq[b][0], v);
                acc[j][b] = v;
            }
        }
    };
    // iteration of chunk ch with the register set that holds chunk ch + 1 (B for even ch, A for odd ch)
    // (Also measured: the two waves of a SIMD taking the matrix phase and the staging in opposite order, so that one's ~150
    //  vector instructions run under the other's MFMAs -- 10.8 ms against 10.1: the extra branch costs more than it hides.)
    auto body = [&](int ch, Dw2Raw& raw) {
        compute(base + (ch & 1) * (DW2_A + DW2_B));
        // (the rows are first USED here: the split is pure arithmetic, and hipcc otherwise puts it -- with the wait for the
        //  rows -- in front of the MFMAs; nor may a copy of a loaded register wait at the loop's back edge)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(raw.a[0]), "+v"(raw.a[1]), "+v"(raw.a[2]), "+v"(raw.b0), "+v"(raw.b1) : : "memory");
        uint16_t* nA = base + ((ch & 1) ^ 1) * (DW2_A + DW2_B);
        stage(raw, nA, nA + DW2_A);                      // chunk ch + 1
        dw2_rows<XB>(q, ca, cb, raw);                    // chunk ch + 3 (its indices: requested two chunks ago)
        dw2_index<XB>(q, r_lo + (ch + 5) * 16 + srow, r_end, raw);
        __syncthreads();
    };
    for (int ch = 0; ch < nchunk; ch += 2) {
        body(ch, rawB);
        if (ch + 1 < nchunk) body(ch + 1, rawA);
    }
    // ---- the block's 384 x 128 tile of this slab
    float* sw = q.slabs + (size_t)slab * 3 * H * H;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int jj = mt * 384 + (jt0 + j) * 32 + w_acc_row(reg, half);
                sw[(size_t)jj * H + nt * 128 + (bt0 + b) * 32 + c] = acc[j][b][reg];
            }
    if (nt == 0) {                                           // bias gradient: column sums of dg over the slab's rows
        float* red = lds;                                    // [16][384] floats (the images are dead)
#pragma unroll
        for (int i = 0; i < 12; ++i) red[srow * 384 + c32 * 12 + i] = cs[i];
        __syncthreads();
        if (tid < 384) {
            float sum = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 16; ++rr) sum += red[rr * 384 + tid];
            q.bslabs[(size_t)slab * 3 * H + mt * 384 + tid] = sum;
        }
    }
}

static constexpr int DW_TILE_M = 384, DW_CHUNK = 16;
// one launch of the weight-gradient kernel over q.R rows in nslab slabs of q.rows_per_slab rows
static int launch_dw(const WideDwArgs& q, int nslab, hipStream_t st) {
    const int mt = 3 * q.H / DW_TILE_M, nt = q.H / 128;
    if (q.xb) {
        TM_SHM_ONCE(k_wide_dw2<true>, DW2_SHM);
        hipLaunchKernelGGL(k_wide_dw2<true>, dim3(mt * nt, nslab), dim3(512), DW2_SHM, st, q, mt, nt);
    } else {
        TM_SHM_ONCE(k_wide_dw2<false>, DW2_SHM);
        hipLaunchKernelGGL(k_wide_dw2<false>, dim3(mt * nt, nslab), dim3(512), DW2_SHM, st, q, mt, nt);
    }
    return check_launch("wide_dw");
}

// the ring form needs whole 128-column blocks, K in whole groups of four half steps (>= 8) and 16-byte aligned rows
static int launch_gemm_ring(const WideArgs& a, hipStream_t st, const uint16_t* img_gpp = nullptr) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int ntile = ceil_div(a.R, 128);
    // default since round 5 where N = 256 (C5): the opposite-phase form; TMPNN_WIDE_GEMM_RING=1 keeps the ring form for A/B runs
#ifdef TMPNN_KEEP_VARIANTS
    static const bool ring_form = [] { const char* e = getenv("TMPNN_WIDE_GEMM_RING"); return e && e[0] == '1'; }();
    // 256-row tiles (the weights streamed half as often) unless TMPNN_WIDE_GEMM_ROWS=128
    static const bool rows128 = [] { const char* e = getenv("TMPNN_WIDE_GEMM_ROWS"); return e && e[0] == '1' && e[1] == '2'; }();
#else
    constexpr bool ring_form = false, rows128 = false;
#endif
    if (!ring_form && a.N == 256 && a.K % 128 == 0 && img_gpp != nullptr) {
        WideArgs b = a;
        b.img = img_gpp;
        if (rows128) {
            TM_SHM_ONCE(k_wide_gemm_pp256<1>, gq_shm<1>());
            hipLaunchKernelGGL(k_wide_gemm_pp256<1>, dim3(ntile < cus ? ntile : cus), dim3(512), gq_shm<1>(), st, b);
        } else {
            const int nt2 = ceil_div(a.R, 256);
            TM_SHM_ONCE(k_wide_gemm_pp256<2>, gq_shm<2>());
            hipLaunchKernelGGL(k_wide_gemm_pp256<2>, dim3(nt2 < cus ? nt2 : cus), dim3(512), gq_shm<2>(), st, b);
        }
        return check_launch("wide_gemm_pp256");
    }
    if (a.N % 256 == 0) {
        TM_SHM_ONCE(k_wide_gemm_ring256, W_GEMM_RING4_SHM);
        hipLaunchKernelGGL(k_wide_gemm_ring256, dim3(ntile < cus ? ntile : cus), dim3(512), W_GEMM_RING4_SHM, st, a);
        return check_launch("wide_gemm_ring256");
    }
    TM_SHM_ONCE(k_wide_gemm_ring, W_GEMM_RING_SHM);
    hipLaunchKernelGGL(k_wide_gemm_ring, dim3(ntile < cus ? ntile : cus), dim3(512), W_GEMM_RING_SHM, st, a);
    return check_launch("wide_gemm_ring");
}

static int launch_store(const WideArgs& a, hipStream_t st) {
    dim3 grid(wide_grid(ceil_div(a.N, 128), ceil_div(a.R, W_BM)));
    TM_SHM_ONCE(k_wide_gemm_store, W_STORE_SHM);
    hipLaunchKernelGGL(k_wide_gemm_store, grid, dim3(512), W_STORE_SHM, st, a);
    return check_launch("wide_gemm_store");
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_wide_supported(int H, int IN) { return (H >= 128 && H <= 1024 && H % 128 == 0 && IN == H) ? 1 : 0; }

// bytes of the four weight images of one cell: forward hh (K = H, N = 3H), forward ih (K = IN, N = 3H),
// backward ih (K = 3H, N = IN), backward hh (K = 3H, N = H)
size_t tmpnn_wide_prep_bytes(int H, int IN) {
    if (H <= 0 || IN <= 0) return 0;
    // + the half-step images of the forward and backward-data W_hh operands (k_wide_gru_fwd_ring, k_wide_gemm_ring)
    // + the contiguous-block image of the forward W_hh operand (k_wide_gru_fwd_pp)
    return sizeof(uint16_t) * 3 * ((size_t)H * 3 * H + (size_t)IN * 3 * H + (size_t)3 * H * IN + (size_t)3 * H * H + 4 * (size_t)H * 3 * H);      // ... and of the backward W_hh operand (k_wide_gemm_pp256)
}

int tmpnn_wide_prepare(const float* w_ih, const float* w_hh, int IN, int H, void* prep, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_wide_supported(H, IN), "wide_prepare: H=%d IN=%d (need H in {128, 256}, IN = H)", H, IN);
    TM_REQUIRE(w_ih && w_hh && prep && aligned16(prep), "wide_prepare: null / misaligned pointer");
    hipStream_t st = as_stream(stream);
    uint16_t* f_hh = reinterpret_cast<uint16_t*>(prep);
    uint16_t* f_ih = f_hh + (size_t)3 * H * 3 * H;
    uint16_t* b_ih = f_ih + (size_t)3 * IN * 3 * H;
    uint16_t* b_hh = b_ih + (size_t)3 * 3 * H * IN;
    const int g1 = ceil_div((long)H * 3 * H, 256), g2 = ceil_div((long)IN * 3 * H, 256);
    hipLaunchKernelGGL(k_wide_prep, dim3(g1), dim3(256), 0, st, w_hh, H, H, 3 * H, 1, f_hh);
    hipLaunchKernelGGL(k_wide_prep, dim3(g2), dim3(256), 0, st, w_ih, IN, IN, 3 * H, 1, f_ih);
    hipLaunchKernelGGL(k_wide_prep, dim3(g2), dim3(256), 0, st, w_ih, IN, 3 * H, IN, 0, b_ih);
    hipLaunchKernelGGL(k_wide_prep, dim3(g1), dim3(256), 0, st, w_hh, H, 3 * H, H, 0, b_hh);
    uint16_t* f_hh16 = b_hh + (size_t)3 * 3 * H * H;
    hipLaunchKernelGGL(k_wide_prep16, dim3(g1), dim3(256), 0, st, w_hh, H, H, 3 * H, 1, f_hh16);
    hipLaunchKernelGGL(k_wide_prep16, dim3(g1), dim3(256), 0, st, w_hh, H, 3 * H, H, 0, f_hh16 + (size_t)3 * H * 3 * H);
    hipLaunchKernelGGL(k_wide_prep_pp, dim3(g1), dim3(256), 0, st, w_hh, H, H, f_hh16 + (size_t)2 * 3 * H * 3 * H);
    if (H == 256) hipLaunchKernelGGL(k_wide_prep_gpp, dim3(g1), dim3(256), 0, st, w_hh, H, 3 * H, f_hh16 + (size_t)3 * 3 * H * 3 * H);
    return check_launch("wide_prepare");
}

/* Edge (or any) cell forward, diff message through the projected det rows:
 *   P [Dn][3H] = h[det_rows] W_ih^T (written here) ; for r < R: gi = P[src_pos[r]] - P[dst_pos[r]], gh = h[rows[r]] W_hh^T,
 *   h_out[rows[r]] = GRU gates ; gates: NULL or 4 planes.  */
int tmpnn_wide_gru_fwd(const void* prep, const int32_t* det_rows, int Dn, const int32_t* rows, int R,
                       const int32_t* src_pos, const int32_t* dst_pos, const float* h, int ld_h, int H,
                       const float* b_ih, const float* b_hh, float* P, float* h_out, int ld_out, float* gates,
                       size_t gate_plane, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_wide_supported(H, H), "wide_gru_fwd: H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(prep && det_rows && rows && src_pos && dst_pos && h && b_ih && b_hh && P && h_out && Dn > 0 && R > 0,
               "wide_gru_fwd: null pointer / empty det table");
    TM_REQUIRE(aligned16(prep) && aligned16(h) && (ld_h & 3) == 0 && ld_h >= H && ld_out >= H && aligned16(P) &&
                   aligned16(h_out) && (ld_out & 3) == 0 && aligned16(b_ih) && aligned16(b_hh) &&
                   (gates == nullptr || (aligned16(gates) && (gate_plane & 3) == 0)), "wide_gru_fwd: layout (16-byte alignment)");
    TM_REQUIRE(gates == nullptr || gate_plane >= (size_t)H, "wide_gru_fwd: gate_plane too small");
    hipStream_t st = as_stream(stream);
    const uint16_t* f_hh = reinterpret_cast<const uint16_t*>(prep);
    const uint16_t* f_ih = f_hh + (size_t)3 * H * 3 * H;
    WideArgs p{};
    p.A = h; p.lda = ld_h; p.a_rows = det_rows; p.R = Dn; p.K = H; p.img = f_ih; p.N = 3 * H;
    p.C = P; p.ldc = 3 * H; p.c_rows = nullptr; p.accumulate = 0;
    int rc = launch_store(p, st);
    if (rc) return rc;
    WideArgs a{};
    a.A = h; a.lda = ld_h; a.a_rows = rows; a.R = R; a.K = H; a.img = f_hh; a.N = 3 * H;
    a.P = P; a.ldp = 3 * H; a.src_pos = src_pos; a.dst_pos = dst_pos; a.h = h; a.ld_h = ld_h; a.H = H;
    a.b_ih = b_ih; a.b_hh = b_hh; a.h_out = h_out; a.ld_out = ld_out; a.gates = gates; a.gate_plane = gate_plane; a.rows = rows;
    TM_SHM_ONCE(k_wide_gru_fwd, W_GRU_SHM);
    hipLaunchKernelGGL(k_wide_gru_fwd, dim3(wide_grid(H / 64, ceil_div(R, W_BM))), dim3(512), W_GRU_SHM, st, a);
    return check_launch("wide_gru_fwd");
}

/* The same cell over edge tiles (struct tmpnn_edge_tiles, rows_per_tile = 128): results bit-identical to
 * tmpnn_wide_gru_fwd; the projected det rows of a tile are staged in LDS once instead of gathered per edge row. */
int tmpnn_wide_gru_fwd_tiled(const void* prep, const int32_t* det_rows, int Dn, const tmpnn_edge_tiles* tiles, int R,
                             const float* h, int ld_h, int H, const float* b_ih, const float* b_hh, float* P,
                             float* h_out, int ld_out, float* gates, size_t gate_plane, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_wide_supported(H, H), "wide_gru_fwd_tiled: H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(tiles != nullptr, "wide_gru_fwd_tiled: tiles is null");
    TM_REQUIRE(tiles->rows_per_tile == 128 && tiles->T > 0 && (long)tiles->T * 128 >= R && (long)(tiles->T - 1) * 128 < R &&
                   tiles->t_row && tiles->t_loc && tiles->t_dptr && tiles->t_dets,
               "wide_gru_fwd_tiled: tile list (T=%d, rows_per_tile=%d) does not cover R=%d rows in 128-row tiles",
               tiles->T, tiles->rows_per_tile, R);
    TM_REQUIRE(prep && det_rows && h && b_ih && b_hh && P && h_out && Dn > 0 && R > 0,
               "wide_gru_fwd_tiled: null pointer / empty det table");
    TM_REQUIRE(aligned16(prep) && aligned16(h) && (ld_h & 3) == 0 && ld_h >= H && ld_out >= H && aligned16(P) &&
                   aligned16(h_out) && (ld_out & 3) == 0 && aligned16(b_ih) && aligned16(b_hh) &&
                   (gates == nullptr || (aligned16(gates) && (gate_plane & 3) == 0)), "wide_gru_fwd_tiled: layout (16-byte alignment)");
    TM_REQUIRE(gates == nullptr || gate_plane >= (size_t)H, "wide_gru_fwd_tiled: gate_plane too small");
    hipStream_t st = as_stream(stream);
    const uint16_t* f_hh = reinterpret_cast<const uint16_t*>(prep);
    const uint16_t* f_ih = f_hh + (size_t)3 * H * 3 * H;
    WideArgs p{};
    p.A = h; p.lda = ld_h; p.a_rows = det_rows; p.R = Dn; p.K = H; p.img = f_ih; p.N = 3 * H;
    p.C = P; p.ldc = 3 * H; p.c_rows = nullptr; p.accumulate = 0;
    int rc = launch_store(p, st);
    if (rc) return rc;
    WideArgs a{};
    a.A = h; a.lda = ld_h; a.R = R; a.K = H; a.img = f_hh; a.N = 3 * H;
    a.P = P; a.ldp = 3 * H; a.h = h; a.ld_h = ld_h; a.H = H;
    a.b_ih = b_ih; a.b_hh = b_hh; a.h_out = h_out; a.ld_out = ld_out; a.gates = gates; a.gate_plane = gate_plane;
    WideTiles tl{tiles->t_row, tiles->t_loc, tiles->t_dptr, tiles->t_dets, tiles->T};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = tiles->T < cus ? tiles->T : cus;          // one persistent block per CU (LDS)
    a.img = f_hh + (size_t)3 * (3 * H * H + 3 * H * H + 3 * H * H + 3 * H * H);      // the half-step image (fifth of prep)
    const uint16_t* img_pp = a.img + (size_t)2 * 3 * H * 3 * H;                       // the contiguous-block image (seventh)
    // default since round 5: the opposite-phase form on 128 x 384 items (k_wide_gru_fwd_pp); TMPNN_WIDE_FWD_RING=1 keeps the
    // ring form of round 3 (same results bit for bit) for A/B runs
#ifdef TMPNN_KEEP_VARIANTS
    static const bool ring_form = [] { const char* e = getenv("TMPNN_WIDE_FWD_RING"); return e && e[0] == '1'; }();
    if (ring_form) {
        TM_SHM_ONCE(k_wide_gru_fwd_ring, W_RING_SHM);
        hipLaunchKernelGGL(k_wide_gru_fwd_ring, dim3(grid), dim3(512), W_RING_SHM, st, a, tl);
        return check_launch("wide_gru_fwd_tiled (ring form)");
    }
#endif
    a.img = img_pp;
    TM_SHM_ONCE(k_wide_gru_fwd_pp, W_PP_SHM);
    hipLaunchKernelGGL(k_wide_gru_fwd_pp, dim3(grid), dim3(512), W_PP_SHM, st, a, tl);
    return check_launch("wide_gru_fwd_tiled");
}

#ifdef TMPNN_KEEP_VARIANTS
size_t tmpnn_wide_gru_bwd_data_ws(int R, int H) { return R > 0 ? sizeof(float) * 2 * (size_t)R * 3 * H : 0; }
#endif  // TMPNN_KEEP_VARIANTS

/* Data gradient of the cell (arguments as tmpnn_gru_bwd_data, IN = H): d_msg[rows[r]][0:H] = d_gi W_ih,
 * d_h[rows[r]] = dh z + d_gh W_hh.  ws: tmpnn_wide_gru_bwd_data_ws bytes. */
#ifdef TMPNN_KEEP_VARIANTS
int tmpnn_wide_gru_bwd_data(const void* prep, const int32_t* rows, int R, const float* h, int ld_h, int H,
                            const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy,
                            const float* w_head, float* d_msg, int ld_dmsg, float* d_h, int ld_dh, void* ws,
                            size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_wide_supported(H, H), "wide_gru_bwd_data: H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(prep && rows && h && gates && d_msg && d_h && ws, "wide_gru_bwd_data: null pointer");
    TM_REQUIRE(d_hout != nullptr || dy != nullptr, "wide_gru_bwd_data: no upstream gradient");
    TM_REQUIRE(dy == nullptr || (w_head != nullptr && aligned16(w_head)), "wide_gru_bwd_data: dy needs a 16-byte aligned w_head");
    TM_REQUIRE(aligned16(h) && aligned16(gates) && aligned16(d_h) && aligned16(ws) && (ld_h & 3) == 0 && (ld_dh & 3) == 0 &&
                   aligned16(d_msg) && (ld_dmsg & 3) == 0 &&
                   (gate_plane & 3) == 0 && (d_hout == nullptr || (aligned16(d_hout) && (ld_dhout & 3) == 0)),
               "wide_gru_bwd_data: 16-byte alignment");
    if (ws_bytes < tmpnn_wide_gru_bwd_data_ws(R, H))
        return set_error(TMPNN_EWORKSPACE, "wide_gru_bwd_data: workspace %zu < %zu bytes", ws_bytes,
                         tmpnn_wide_gru_bwd_data_ws(R, H));
    hipStream_t st = as_stream(stream);
    float* dgi = reinterpret_cast<float*>(ws);
    float* dgh = dgi + (size_t)R * 3 * H;
    WideBwdArgs b{rows, R, H, h, ld_h, gates, gate_plane, d_hout, ld_dhout, dy, w_head, dgi, dgh, d_h, ld_dh};
    long blocks = ((long)R * (H / 4) + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    hipLaunchKernelGGL(k_wide_gates_bwd, dim3((int)blocks), dim3(256), 0, st, b);
    int rc = check_launch("wide_gates_bwd");
    if (rc) return rc;
    const uint16_t* f_hh = reinterpret_cast<const uint16_t*>(prep);
    const uint16_t* b_ih = f_hh + (size_t)3 * H * 3 * H + (size_t)3 * H * 3 * H;
    const uint16_t* b_hh = b_ih + (size_t)3 * 3 * H * H;
    WideArgs x{};
    x.A = dgi; x.lda = 3 * H; x.a_rows = nullptr; x.R = R; x.K = 3 * H; x.img = b_ih; x.N = H;
    x.C = d_msg; x.ldc = ld_dmsg; x.c_rows = rows; x.accumulate = 0;
    if ((rc = launch_store(x, st))) return rc;
    WideArgs y = x;
    y.A = dgh; y.img = b_hh; y.C = d_h; y.ldc = ld_dh; y.accumulate = 1;
    return launch_store(y, st);
}
#endif  // TMPNN_KEEP_VARIANTS

static int dw_slabs(int R, int H) {
    const int tiles = (3 * H / DW_TILE_M) * (H / 128);
    int s = 256 / tiles;
    const int by_rows = (R + 2047) / 2048;                 // at least 2048 rows per slab
    if (s > by_rows) s = by_rows;
    return s < 1 ? 1 : s;
}

#ifdef TMPNN_KEEP_VARIANTS
size_t tmpnn_wide_gru_bwd_weights_ws(int R, int H) {
    if (R <= 0) return 0;
    const int n = dw_slabs(R, H);
    const size_t per = (size_t)3 * H * H + (size_t)3 * H;
    return sizeof(float) * ((size_t)n * per + reduce_slabs_ws_floats(n, (size_t)3 * H * H));
}
#endif  // TMPNN_KEEP_VARIANTS

/* Weight gradient of the cell from the gate gradients tmpnn_wide_gru_bwd_data left in ITS workspace (`dg_ws`: d_gi then
 * d_gh, [R][3H] each): dW_ih += d_gi^T (h[src] - h[dst]), dW_hh += d_gh^T h[rows], db_* += column sums.  */
#ifdef TMPNN_KEEP_VARIANTS
int tmpnn_wide_gru_bwd_weights(const void* dg_ws, const int32_t* rows, int R, const int32_t* src, const int32_t* dst,
                               const float* h, int ld_h, int H, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                               void* ws, size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(tmpnn_wide_supported(H, H), "wide_gru_bwd_weights: H=%d", H);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(dg_ws && rows && src && dst && h && dW_ih && dW_hh && db_ih && db_hh && ws && R > 0,
               "wide_gru_bwd_weights: null pointer");
    TM_REQUIRE(aligned16(dg_ws) && aligned16(h) && (ld_h & 3) == 0 && ld_h >= H && aligned16(ws),
               "wide_gru_bwd_weights: 16-byte alignment");
    if (ws_bytes < tmpnn_wide_gru_bwd_weights_ws(R, H))
        return set_error(TMPNN_EWORKSPACE, "wide_gru_bwd_weights: workspace %zu < %zu bytes", ws_bytes,
                         tmpnn_wide_gru_bwd_weights_ws(R, H));
    hipStream_t st = as_stream(stream);
    const int n = dw_slabs(R, H);
    const int rps = ((R + n - 1) / n + 31) / 32 * 32;
    const float* dgi = reinterpret_cast<const float*>(dg_ws);
    const float* dgh = dgi + (size_t)R * 3 * H;
    float* slabs = reinterpret_cast<float*>(ws);
    float* bslabs = slabs + (size_t)n * 3 * H * H;
    float* fold = bslabs + (size_t)n * 3 * H;
    const int nslab = (R + rps - 1) / rps;
    for (int which = 0; which < 2; ++which) {
        WideDwArgs q{which == 0 ? dgi : dgh, 3 * H, nullptr, 3 * H, 0, h, ld_h, which == 0 ? src : rows,
                     which == 0 ? dst : nullptr, R, H, rps, slabs, bslabs};
        int rc = launch_dw(q, nslab, st);
        if (rc) return rc;
        if ((rc = launch_reduce_slabs(slabs, (size_t)3 * H * H, nslab, which == 0 ? dW_ih : dW_hh, (size_t)3 * H * H, 1, st, fold)))
            return rc;
        if ((rc = launch_reduce_slabs(bslabs, (size_t)3 * H, nslab, which == 0 ? db_ih : db_hh, (size_t)3 * H, 1, st, fold)))
            return rc;
    }
    return TMPNN_OK;
}
#endif  // TMPNN_KEEP_VARIANTS

// ---- det-side form of the W_ih products -------------------------------------------------------------------
static int gates4_blocks(int R, int H) {
    long blocks = ((long)R * (H / 4) + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    return (int)(blocks < 1 ? 1 : blocks);
}
// floats: dg4 [N][4H] | S [Dn][3H] | dn slabs [nb][H] | dW slabs + bias slabs | fold
static void diff_ws_layout(int N, int R, int Dn, int H, size_t* o_S, size_t* o_dn, size_t* o_slabs, size_t* o_bslabs,
                           size_t* o_fold, size_t* total) {
    const int nb = gates4_blocks(R, H);
    const int n_e = dw_slabs(R, H), n_d = dw_slabs(Dn, H);
    const int n = n_e > n_d ? n_e : n_d;
    size_t fold = reduce_slabs_ws_floats(n, (size_t)3 * H * H);
    const size_t f2 = reduce_slabs_ws_floats(nb, (size_t)H);
    if (f2 > fold) fold = f2;
    auto up = [](size_t v) { return (v + 63) / 64 * 64; };          // 256-byte aligned sections
    size_t o = 0;
    o += up((size_t)N * 4 * H);          *o_S = o;
    o += up((size_t)Dn * 3 * H);         *o_dn = o;
    o += up((size_t)nb * H);             *o_slabs = o;
    o += up((size_t)n * 3 * H * H);      *o_bslabs = o;
    o += up((size_t)n * 3 * H);          *o_fold = o;
    o += up(fold);
    *total = o;
}

size_t tmpnn_wide_gru_bwd_diff_ws(int N, int R, int Dn, int H) {
    if (N <= 0 || R <= 0 || Dn <= 0 || !tmpnn_wide_supported(H, H)) return 0;
    size_t a, b, c, d, e, total;
    diff_ws_layout(N, R, Dn, H, &a, &b, &c, &d, &e, &total);
    return sizeof(float) * total;
}

/* Whole backward of a wide EDGE cell whose input is the diff message x[e] = h[src e] - h[dst e], with both W_ih products
 * taken on the DET side by linearity (the backward twin of the forward's projected det rows):
 *     S[d] = sum_{e: src = d} d_gi[e] - sum_{e: dst = d} d_gi[e]            (signed segment sum over the det's CSR run)
 *     message adjoint   d_h[det_row[d]] += S[d] W_ih          instead of  d_x = d_gi W_ih per edge + its segment sum
 *     dW_ih += S^T h[det rows]                                 instead of  d_gi^T (h[src] - h[dst]) over the edges
 * i.e. two of the four (R x 3H x H) products run over Dn rows instead of E (C5: 15 000 instead of 4.4 M).
 * d_h[edge_row[e]] = dh z + d_gh W_hh (plain store), dW_hh += d_gh^T h[edge rows], db_ih / db_hh += column sums. */
static int wide_gru_bwd_diff_impl(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                                  size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                                  float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                                  size_t ws_bytes, tmpnn_stream stream, tmpnn_stream aux_stream, tmpnn_event ev_fork_, tmpnn_event ev_join_, const float* add_msg,
                                  int ld_add) {
    TM_REQUIRE(tmpnn_wide_supported(H, H), "wide_gru_bwd_diff: H=%d", H);
    TM_REQUIRE(add_msg == nullptr || (aligned16(add_msg) && (ld_add & 3) == 0 && ld_add >= H && g && g->src && g->dst),
               "wide_gru_bwd_diff: fused row-F adjoint needs a 16-byte aligned table of >= H columns and the graph's src / dst");
    TM_REQUIRE(g != nullptr, "wide_gru_bwd_diff: graph is null");
    const int N = g->N, R = g->E, Dn = g->Dn;
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(N > 0 && Dn > 0 && (long)R + Dn == N && g->edge_row && g->det_row && g->rowptr && g->inc,
               "wide_gru_bwd_diff: graph arrays (N=%d E=%d Dn=%d)", N, R, Dn);
    TM_REQUIRE(prep && h && gates && d_h && dW_ih && dW_hh && db_ih && db_hh && ws, "wide_gru_bwd_diff: null pointer");
    TM_REQUIRE(d_hout != nullptr || dy != nullptr, "wide_gru_bwd_diff: no upstream gradient");
    TM_REQUIRE(dy == nullptr || (w_head != nullptr && aligned16(w_head)), "wide_gru_bwd_diff: dy needs a 16-byte aligned w_head");
    TM_REQUIRE(aligned16(prep) && aligned16(h) && aligned16(gates) && aligned16(d_h) && aligned16(ws) && (ld_h & 3) == 0 &&
                   ld_h >= H && (ld_dh & 3) == 0 && ld_dh >= H && (gate_plane & 3) == 0 && gate_plane >= (size_t)H &&
                   (d_hout == nullptr || (aligned16(d_hout) && (ld_dhout & 3) == 0)),
               "wide_gru_bwd_diff: layout (16-byte alignment)");
    size_t oS, odn, oslabs, obslabs, ofold, total;
    diff_ws_layout(N, R, Dn, H, &oS, &odn, &oslabs, &obslabs, &ofold, &total);
    if (ws_bytes < sizeof(float) * total)
        return set_error(TMPNN_EWORKSPACE, "wide_gru_bwd_diff: workspace %zu < %zu bytes", ws_bytes, sizeof(float) * total);
    hipStream_t st = as_stream(stream);
    float* base = reinterpret_cast<float*>(ws);
    float* dg4 = base;
    float* S = base + oS;
    float* dn_slabs = base + odn;
    float* slabs = base + oslabs;
    float* bslabs = base + obslabs;
    float* fold = base + ofold;
    int rc;
    // 1. gate gradients, once: dg4[row] = [dr | dz | dn | dn r], d_h[row] = dh z, column sums of dn
    const int nb = gates4_blocks(R, H);
    WideBwdArgs b{g->edge_row, R, H, h, ld_h, gates, gate_plane, d_hout, ld_dhout, dy, w_head, dg4, nullptr, d_h, ld_dh};
    hipLaunchKernelGGL(k_wide_gates_bwd4, dim3(nb), dim3(256), 0, st, b, dn_slabs);
    if ((rc = check_launch("wide_gates_bwd4"))) return rc;
    if ((rc = launch_reduce_slabs(dn_slabs, (size_t)H, nb, db_ih + 2 * H, (size_t)H, 1, st, fold))) return rc;
    // 2. d_h[edge rows] += d_gh W_hh   (d_gh = image columns 0..2H and 3H..4H)
    const uint16_t* f_hh = reinterpret_cast<const uint16_t*>(prep);
    const uint16_t* b_ih = f_hh + (size_t)3 * H * 3 * H + (size_t)3 * H * 3 * H;
    const uint16_t* b_hh = b_ih + (size_t)3 * 3 * H * H;
    // With an auxiliary stream the det-side branch (3, 4: the signed segment sums of d_gi and the message adjoint -- row movers
    // and a Dn-row product, 4.9 ms per C5 iteration) runs NEXT TO the two E-row matrix kernels (2, 5a: 22.6 ms, matrix-pipe
    // bound): both only read dg4 and write disjoint rows of d_h.  Fork after 1, join before 5b (which reads S and reuses the
    // slab buffer).  Same kernels on the same data in the same order per buffer: bit-identical to the one-stream form.
    hipStream_t sx = aux_stream ? as_stream(aux_stream) : st;
    hipEvent_t ev_fork = reinterpret_cast<hipEvent_t>(ev_fork_), ev_join = reinterpret_cast<hipEvent_t>(ev_join_);
    if (aux_stream) {                              // the caller's events: nothing is created or destroyed here
        TM_REQUIRE(ev_fork && ev_join, "wide_gru_bwd_diff: the two-stream form needs the caller's fork / join events");
        if (hipEventRecord(ev_fork, st) != hipSuccess || hipStreamWaitEvent(sx, ev_fork, 0) != hipSuccess)
            return set_error(TMPNN_ELAUNCH, "wide_gru_bwd_diff: fork onto the auxiliary stream failed");
    }
    bool joined = false;
    auto join = [&]() -> int {                     // `stream` waits for everything enqueued on the auxiliary stream
        if (!aux_stream || joined) return TMPNN_OK;
        joined = true;
        if (hipEventRecord(ev_join, sx) != hipSuccess || hipStreamWaitEvent(st, ev_join, 0) != hipSuccess)
            return set_error(TMPNN_ELAUNCH, "wide_gru_bwd_diff: join of the auxiliary stream failed");
        return TMPNN_OK;
    };
    auto done = [&](int code) {                    // on an error path too: never return with the auxiliary stream un-joined
        const int j = join();
        return code ? code : j;
    };
    WideArgs y{};
    y.A = dg4; y.lda = 4 * H; y.a_rows = g->edge_row; y.R = R; y.K = 3 * H; y.kskip_at = 2 * H; y.kskip = H;
    y.img = b_hh; y.N = H; y.C = d_h; y.ldc = ld_dh; y.c_rows = g->edge_row; y.accumulate = 1;
    // (the E-row product of the backward: the ring form; its weight image is the sixth of prep)
    y.img = f_hh + (size_t)3 * (4 * 3 * H * H) + (size_t)3 * H * 3 * H;
    y.add_msg = add_msg; y.ld_add = ld_add; y.add_src = g->src; y.add_dst = g->dst;
    if ((rc = launch_gemm_ring(y, st, H == 256 ? y.img + (size_t)2 * 3 * H * 3 * H : nullptr))) return done(rc);
    // 3. S[d] = signed segment sum of d_gi (image columns 0..3H) over the det's incident edges, compact rows
    for (int k = 0; k < 3; ++k)
        if ((rc = tmpnn_segsum_fwd(g, dg4 + (size_t)k * H, 4 * H, S + (size_t)k * H, 3 * H, H, 0, 1,
                                   aux_stream ? aux_stream : stream))) return done(rc);
    // 4. message adjoint on the det rows: d_h[det_row[d]] += S[d] W_ih
    WideArgs x{};
    x.A = S; x.lda = 3 * H; x.a_rows = nullptr; x.R = Dn; x.K = 3 * H; x.kskip_at = 3 * H; x.kskip = 0;
    x.img = b_ih; x.N = H; x.C = d_h; x.ldc = ld_dh; x.c_rows = g->det_row; x.accumulate = 1;
    if ((rc = launch_store(x, sx))) return done(rc);
    // 5. weight gradients: dW_hh over the edge rows (bias sums: db_hh, and db_ih's r / z thirds), dW_ih over the det rows
    {
        const int n = dw_slabs(R, H);
        const int rps = ((R + n - 1) / n + 31) / 32 * 32;
        const int nslab = (R + rps - 1) / rps;
        WideDwArgs q{dg4, 4 * H, g->edge_row, 2 * H, H, h, ld_h, g->edge_row, nullptr, R, H, rps, slabs, bslabs};
        if ((rc = launch_dw(q, nslab, st))) return done(rc);
        if ((rc = launch_reduce_slabs(slabs, (size_t)3 * H * H, nslab, dW_hh, (size_t)3 * H * H, 1, st, fold))) return done(rc);
        if ((rc = launch_reduce_slabs(bslabs, (size_t)3 * H, nslab, db_hh, (size_t)3 * H, 1, st, fold))) return done(rc);
        if ((rc = launch_reduce_slabs(bslabs, (size_t)3 * H, nslab, db_ih, (size_t)2 * H, 1, st, fold))) return done(rc);
    }
    if ((rc = join())) return rc;                                       // S is complete, d_h's det rows are final
    {
        const int n = dw_slabs(Dn, H);
        const int rps = ((Dn + n - 1) / n + 31) / 32 * 32;
        const int nslab = (Dn + rps - 1) / rps;
        WideDwArgs q{S, 3 * H, nullptr, 3 * H, 0, h, ld_h, g->det_row, nullptr, Dn, H, rps, slabs, bslabs};
        if ((rc = launch_dw(q, nslab, st))) return done(rc);
        if ((rc = launch_reduce_slabs(slabs, (size_t)3 * H * H, nslab, dW_ih, (size_t)3 * H * H, 1, st, fold))) return done(rc);
    }
    return done(TMPNN_OK);
}

int tmpnn_wide_gru_bwd_diff(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                            size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                            float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                            size_t ws_bytes, tmpnn_stream stream) {
    return wide_gru_bwd_diff_impl(prep, g, h, ld_h, H, gates, gate_plane, d_hout, ld_dhout, dy, w_head, d_h, ld_dh, dW_ih, dW_hh,
                                  db_ih, db_hh, ws, ws_bytes, stream, nullptr, nullptr, nullptr, nullptr, 0);
}

#ifdef TMPNN_KEEP_VARIANTS
int tmpnn_wide_gru_bwd_diff_aux(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                                size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                                float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                                size_t ws_bytes, tmpnn_stream stream, tmpnn_stream aux_stream, tmpnn_event ev_fork, tmpnn_event ev_join) {
    TM_REQUIRE(aux_stream != nullptr && aux_stream != stream, "wide_gru_bwd_diff_aux: needs a second stream");
    return wide_gru_bwd_diff_impl(prep, g, h, ld_h, H, gates, gate_plane, d_hout, ld_dhout, dy, w_head, d_h, ld_dh, dW_ih, dW_hh,
                                  db_ih, db_hh, ws, ws_bytes, stream, aux_stream, ev_fork, ev_join, nullptr, 0);
}
#endif  // TMPNN_KEEP_VARIANTS

int tmpnn_wide_gru_bwd_diff_fused(const void* prep, const tmpnn_graph* g, const float* h, int ld_h, int H, const float* gates,
                                  size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy, const float* w_head,
                                  float* d_h, int ld_dh, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                                  size_t ws_bytes, const float* add_msg, int ld_add, tmpnn_stream stream,
                                  tmpnn_stream aux_stream, tmpnn_event ev_fork, tmpnn_event ev_join) {
    TM_REQUIRE(add_msg != nullptr, "wide_gru_bwd_diff_fused: add_msg is null");
    TM_REQUIRE(aux_stream != stream || aux_stream == nullptr, "wide_gru_bwd_diff_fused: aux_stream must differ from stream (or be null)");
    return wide_gru_bwd_diff_impl(prep, g, h, ld_h, H, gates, gate_plane, d_hout, ld_dhout, dy, w_head, d_h, ld_dh, dW_ih, dW_hh,
                                  db_ih, db_hh, ws, ws_bytes, stream, aux_stream, ev_fork, ev_join, add_msg, ld_add);
}

}  // extern "C"
