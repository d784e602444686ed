// Small dense stages of the hot path: the input transform with (segmented) BatchNorm
// (models/track_mpnn.py:45-52,59-61) and the masked output heads
// (row J; models/track_mpnn.py:72-75), forward + backward.
//
// These touch only the NEW det rows of a call (tens of rows per tracking window) and one scalar
// per state row, i.e. < 1 % of the bytes and flops of the GRU/aggregation kernels, so they are
// plain vector-ALU kernels: an LDS-tiled strided GEMM, deterministic two-level column sums and
// a handful of elementwise passes.  No float atomics: every reduction has a fixed order.
#include "common.h"

namespace tmpnn {

static constexpr float BN_EPS = 1e-5f;
static constexpr float BN_MOMENTUM = 0.1f;

// ------------------------------------------------------------------------------------------
// generic strided GEMM, 64x64 tile, BK=16, 4x4 per thread
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g, int kper, float* slab) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * kper;
    const int kend = min(g.K, kbeg + kper);
    float acc[4][4] = {};
    // 16-byte alignment of the contiguous dimension of A / B (kbeg is a multiple of 16)
    const bool va = (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && ((g.sak == 1 ? g.sam : g.sak) & 3) == 0;
    const bool vb = (reinterpret_cast<uintptr_t>(g.B) & 15) == 0 && ((g.sbn == 1 ? g.sbk : g.sbn) & 3) == 0;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        // tile loads: 16-byte accesses where a whole 64 x 16 tile is in range and the contiguous dimension is aligned
        // (row lists, ragged edges and odd strides take the element-wise path); the sums below do not depend on which
        const bool fullk = k0 + 16 <= kend;
        if (fullk && m0 + 64 <= g.M && !g.a_krows && g.sak == 1 && va) {          // A row-major: k contiguous
            const int m = tid >> 2, kq = (tid & 3) * 4;
            const long ar = g.a_rows ? g.a_rows[m0 + m] : (m0 + m);
            const float4 v = *reinterpret_cast<const float4*>(g.A + ar * g.sam + k0 + kq);
            As[kq][m] = v.x; As[kq + 1][m] = v.y; As[kq + 2][m] = v.z; As[kq + 3][m] = v.w;
        } else if (fullk && m0 + 64 <= g.M && !g.a_rows && g.sam == 1 && va) {    // A column-major: m contiguous
            const int k = tid >> 4, mq = (tid & 15) * 4;
            const long ak = g.a_krows ? g.a_krows[k0 + k] : (k0 + k);
            *reinterpret_cast<float4*>(&As[k][mq]) = *reinterpret_cast<const float4*>(g.A + ak * g.sak + m0 + mq);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * 256;
                int m, k;
                if (g.sak == 1) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
                float v = 0.f;
                if (m0 + m < g.M && k0 + k < kend) {
                    const long ar = g.a_rows ? g.a_rows[m0 + m] : (m0 + m);
                    const long ak = g.a_krows ? g.a_krows[k0 + k] : (k0 + k);
                    v = g.A[ar * g.sam + ak * g.sak];
                }
                As[k][m] = v;
            }
        }
        if (fullk && n0 + 64 <= g.N && g.sbn == 1 && vb) {                         // B: n contiguous
            const int k = tid >> 4, nq = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&Bs[k][nq]) = *reinterpret_cast<const float4*>(g.B + (long)(k0 + k) * g.sbk + n0 + nq);
        } else if (fullk && n0 + 64 <= g.N && g.sbk == 1 && vb) {                  // B: k contiguous
            const int n = tid >> 2, kq = (tid & 3) * 4;
            const float4 v = *reinterpret_cast<const float4*>(g.B + (long)(n0 + n) * g.sbn + k0 + kq);
            Bs[kq][n] = v.x; Bs[kq + 1][n] = v.y; Bs[kq + 2][n] = v.z; Bs[kq + 3][n] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * 256;
                int n, k;
                if (g.sbn == 1) { n = idx & 63; k = idx >> 6; } else { k = idx & 15; n = idx >> 4; }
                float v = 0.f;
                if (n0 + n < g.N && k0 + k < kend) v = g.B[(long)(k0 + k) * g.sbk + (long)(n0 + n) * g.sbn];
                Bs[k][n] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
    const bool vc = n0 + tx * 4 + 3 < g.N &&
                    (slab ? (g.N & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0
                          : (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                                (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= g.M) continue;
        if (vc) {                                   // four columns in one 16-byte access (same values as below)
            const int n = n0 + tx * 4;
            float4 v = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            if (slab) {
                *reinterpret_cast<float4*>(slab + (size_t)blockIdx.z * g.M * g.N + (size_t)m * g.N + n) = v;
            } else {
                if (g.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                const long cr = g.c_rows ? g.c_rows[m] : m;
                float4* p = reinterpret_cast<float4*>(g.C + cr * g.ldc + n);
                if (g.accumulate) { const float4 o = *p; v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w; }
                *p = v;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= g.N) continue;
            if (slab) {
                slab[(size_t)blockIdx.z * g.M * g.N + (size_t)m * g.N + n] = acc[i][j];
            } else {
                const long cr = g.c_rows ? g.c_rows[m] : m;
                float v = acc[i][j] + (g.bias ? g.bias[n] : 0.f);
                float* p = g.C + cr * g.ldc + n;
                *p = g.accumulate ? *p + v : v;
            }
        }
    }
}

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return TMPNN_OK;
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), 1), block(256);
    hipLaunchKernelGGL(k_gemm, grid, block, 0, st, g, g.K > 0 ? g.K : 1, (float*)nullptr);
    return check_launch("gemm");
}

static int splitk_plan(int K, int* kper) {
    int nsplit = (K + 127) / 128;
    if (nsplit > 1024) nsplit = 1024;
    if (nsplit < 1) nsplit = 1;
    int kp = (K + nsplit - 1) / nsplit;
    kp = (kp + 15) & ~15;
    if (kp < 16) kp = 16;
    nsplit = (K + kp - 1) / kp;
    if (nsplit < 1) nsplit = 1;
    *kper = kp;
    return nsplit;
}

size_t gemm_splitk_ws_floats(int M, int N, int K) {
    int kper;
    const int ns = splitk_plan(K, &kper);
    return (size_t)ns * M * N + reduce_slabs_ws_floats(ns, (size_t)M * N);
}

int launch_gemm_splitk(const GemmArgs& g, float* ws, size_t ws_floats, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return TMPNN_OK;
    if (g.K <= 0) return TMPNN_OK;
    int kper;
    const int ns = splitk_plan(g.K, &kper);
    if (ws_floats < gemm_splitk_ws_floats(g.M, g.N, g.K))
        return set_error(TMPNN_EWORKSPACE, "gemm_splitk: workspace too small");
    if (ns == 1) return launch_gemm(g, st);      // one slab: the same sum straight into C (a launch less on batch-1 graphs)
    float* ws2 = ws + (size_t)ns * g.M * g.N;
    dim3 grid(ceil_div(g.N, 64), ceil_div(g.M, 64), ns), block(256);
    hipLaunchKernelGGL(k_gemm, grid, block, 0, st, g, kper, ws);
    int rc = check_launch("gemm_splitk");
    if (rc) return rc;
    // C is dense [M][ldc] here (weight gradients); reduce row by row when ldc != N
    if (g.ldc == g.N)
        return launch_reduce_slabs(ws, (size_t)g.M * g.N, ns, g.C, (size_t)g.M * g.N, g.accumulate, st, ws2);
    for (int m = 0; m < g.M; ++m) {
        rc = launch_reduce_slabs(ws + (size_t)m * g.N, (size_t)g.M * g.N, ns, g.C + (size_t)m * g.ldc, g.N,
                                 g.accumulate, st, ws2);
        if (rc) return rc;
    }
    return TMPNN_OK;
}

// ------------------------------------------------------------------------------------------
// rows-contracting product on the matrix pipe:  C[i][n] (+)= sum_r X[xrow(r)][i] * Y[r][n]      (weight gradients of the
// attention projections: X = h[det rows], Y = d_ha).  v_mfma_f32_32x32x2_f32 takes A[m][k] from lane (m = l % 32,
// k = l / 32) and B[k][n] from lane (n = l % 32, k = l / 32): with k = two consecutive ROWS r both operands are 32
// consecutive floats of one row per half wave -- coalesced straight from global memory, no transpose, no LDS.  A wave owns
// a (32 MT) x (32 NT) output tile over a slab of rows (exact fp32 products and accumulation); slabs are combined in a fixed
// order by launch_reduce_slabs.
// ------------------------------------------------------------------------------------------
typedef float f32x16_d __attribute__((ext_vector_type(16)));
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_rows_outer(const float* __restrict__ X, int ldx, const int32_t* __restrict__ x_rows,
                                                    const float* __restrict__ Y, int ldy, int R, int rows_per_slab,
                                                    int M, int N, float* __restrict__ slabs) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int slab = blockIdx.x * 4 + wv;
    const int i0 = blockIdx.y * (32 * MT), n0 = blockIdx.z * (32 * NT);
    const int r_lo = slab * rows_per_slab, r_hi = min(R, r_lo + rows_per_slab);
    f32x16_d acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    constexpr int UN = 8;                                  // row pairs in flight
    for (int r = r_lo; r < r_hi; r += 2 * UN) {
        float av[UN][MT], bv[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int rr = r + 2 * u + half;
            const bool ok = rr < r_hi;
            const int rc = ok ? rr : r_lo;
            const long xr = x_rows ? x_rows[rc] : rc;
#pragma unroll
            for (int a = 0; a < MT; ++a) { const float t = X[xr * ldx + i0 + 32 * a + c]; av[u][a] = ok ? t : 0.f; }
#pragma unroll
            for (int b = 0; b < NT; ++b) { const float t = Y[(size_t)rc * ldy + n0 + 32 * b + c]; bv[u][b] = ok ? t : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
    }
    // accumulator register i of lane (c, half): row 8 (i / 4) + 4 half + (i % 4), column c of the 32 x 32 tile
    float* o = slabs + (size_t)slab * M * N;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = i0 + 32 * a + 8 * (i >> 2) + 4 * half + (i & 3);
                o[(size_t)row * N + n0 + 32 * b + c] = acc[a][b][i];
            }
}

// The same product for M = 64 through LDS: a block owns one slab and a 64 x 64 block of C; 32-row chunks of X (gathered rows) and Y
// arrive as coalesced 16-byte loads, one chunk ahead in registers, and are read back as MFMA operands from rows of 96 floats (the
// two halves of a wave land in different bank halves: conflict-free).  The per-lane 4-byte global loads of k_rows_outer kept
// 24 requests in flight per lane and still ran at 2.0 TB/s (0.28 ms per 734 k rows at N = 128); same MFMAs on the same rows in
// the same order: bit-identical slabs.
static constexpr int RO_LD = 96;
__global__ __launch_bounds__(256) void k_rows_outer_lds(const float* __restrict__ X, int ldx, const int32_t* __restrict__ x_rows,
                                                        const float* __restrict__ Y, int ldy, int R, int rows_per_slab, int N,
                                                        float* __restrict__ slabs) {
    __shared__ float sX[2][32 * RO_LD], sY[2][32 * RO_LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int slab = blockIdx.x, n0 = blockIdx.y * 64;
    const int r_lo = slab * rows_per_slab, r_hi = min(R, r_lo + rows_per_slab);
    const int ti = (wv >> 1) * 32, tj = (wv & 1) * 32;          // this wave's 32 x 32 tile of the block's 64 x 64
    // staging: thread -> (row = tid >> 3 of the chunk, 16-byte pieces (tid & 7) and (tid & 7) + 8 of the 64 columns)
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    float4 xa, xb, ya, yb;
    auto fetch = [&](int r0) {
        const int rr = r0 + srow;
        const bool ok = rr < r_hi;
        const int rc = ok ? rr : r_lo;
        const long xr = x_rows ? x_rows[rc] : rc;
        xa = *reinterpret_cast<const float4*>(X + xr * ldx + sc4);
        xb = *reinterpret_cast<const float4*>(X + xr * ldx + 32 + sc4);
        ya = *reinterpret_cast<const float4*>(Y + (size_t)rc * ldy + n0 + sc4);
        yb = *reinterpret_cast<const float4*>(Y + (size_t)rc * ldy + n0 + 32 + sc4);
        if (!ok) xa = xb = ya = yb = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&sX[buf][srow * RO_LD + sc4]) = xa;
        *reinterpret_cast<float4*>(&sX[buf][srow * RO_LD + 32 + sc4]) = xb;
        *reinterpret_cast<float4*>(&sY[buf][srow * RO_LD + sc4]) = ya;
        *reinterpret_cast<float4*>(&sY[buf][srow * RO_LD + 32 + sc4]) = yb;
    };
    f32x16_d acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (r_lo < r_hi) {
        fetch(r_lo);
        stage(0);
        __syncthreads();
        int buf = 0;
        for (int r = r_lo; r < r_hi; r += 32) {
            const bool more = r + 32 < r_hi;
            if (more) fetch(r + 32);
            const float* px = &sX[buf][half * RO_LD + ti + c];
            const float* py = &sY[buf][half * RO_LD + tj + c];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(px[2 * u * RO_LD], py[2 * u * RO_LD], acc, 0, 0, 0);
            if (more) stage(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    float* o = slabs + (size_t)slab * 64 * N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = ti + 8 * (i >> 2) + 4 * half + (i & 3);
        o[(size_t)row * N + n0 + tj + c] = acc[i];
    }
}

static int rows_outer_plan(int R, int* per) {
    int ns = (R + 255) / 256;                 // >= 256 rows per slab
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    ns = (ns + 3) & ~3;                        // four slabs (waves) per block
    int p = (R + ns - 1) / ns;
    p = (p + 1) & ~1;                          // row pairs
    if (p < 2) p = 2;
    *per = p;
    return ns;
}

size_t rows_outer_ws_floats(int M, int N, int R) {
    int per;
    const int ns = rows_outer_plan(R > 0 ? R : 1, &per);
    return (size_t)ns * M * N + reduce_slabs_ws_floats(ns, (size_t)M * N);
}

int launch_rows_outer(const float* X, int ldx, const int32_t* x_rows, const float* Y, int ldy, int R, int M, int N, float* C,
                      int ldc, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
    if (M <= 0 || N <= 0) return TMPNN_OK;
    TM_REQUIRE(M % 32 == 0 && N % 32 == 0 && ldc == N, "rows_outer: M=%d N=%d must be multiples of 32, C dense", M, N);
    if (ws_floats < rows_outer_ws_floats(M, N, R)) return set_error(TMPNN_EWORKSPACE, "rows_outer: workspace too small");
    int per;
    const int ns = rows_outer_plan(R > 0 ? R : 1, &per);
    // small tiles: the slab memory (ns x M x N) does not depend on the tiling, and (M / 32) x (N / 64) times as many waves keep
    // more row requests in flight (a 64 x 128 tile per wave left one wave per SIMD: 0.32 ms per 734 k rows, latency-bound)
    if (M == 64 && N % 64 == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && aligned16(X) && aligned16(Y)) {
        hipLaunchKernelGGL(k_rows_outer_lds, dim3(ns, N / 64), dim3(256), 0, st, X, ldx, x_rows, Y, ldy, R, per, N, ws);
        int rc = check_launch("rows_outer (LDS form)");
        if (rc) return rc;
        return launch_reduce_slabs(ws, (size_t)M * N, ns, C, (size_t)M * N, accumulate, st, ws + (size_t)ns * M * N);
    }
    const int mt = 1;
    const int nt = (N % 64 == 0) ? 2 : 1;
    dim3 grid(ns / 4, M / (32 * mt), N / (32 * nt)), block(256);
#define RO(A_, B_) hipLaunchKernelGGL((k_rows_outer<A_, B_>), grid, block, 0, st, X, ldx, x_rows, Y, ldy, R, per, M, N, ws)
    if (mt == 2) { if (nt == 4) RO(2, 4); else if (nt == 2) RO(2, 2); else RO(2, 1); }
    else         { if (nt == 4) RO(1, 4); else if (nt == 2) RO(1, 2); else RO(1, 1); }
#undef RO
    int rc = check_launch("rows_outer");
    if (rc) return rc;
    return launch_reduce_slabs(ws, (size_t)M * N, ns, C, (size_t)M * N, accumulate, st, ws + (size_t)ns * M * N);
}

// ------------------------------------------------------------------------------------------
// column sums
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ src, long ld,
                                                        const float* __restrict__ mul, long ldm, int rows, int cols,
                                                        float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int slot = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 128;
    const int r1 = min(rows, r0 + 128);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + slot; r < r1; r += 4) {
            float v = src[(size_t)r * ld + c];
            if (mul) v *= mul[(size_t)r * ldm + c];
            s += v;
        }
    red[slot][threadIdx.x & 63] = s;
    __syncthreads();
    if (slot == 0 && c < cols)
        part[(size_t)blockIdx.x * cols + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

size_t colsum_ws_floats(int rows, int cols) {
    const size_t nch = ceil_div(rows > 0 ? rows : 1, 128);
    return nch * cols + reduce_slabs_ws_floats((int)nch, cols);
}

int launch_colsum(const float* src, long ld, const float* mul, long ldm, int rows, int cols, float* dst,
                  int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
    if (cols <= 0) return TMPNN_OK;
    if (rows <= 0) {
        if (!accumulate) (void)hipMemsetAsync(dst, 0, sizeof(float) * cols, st);
        return TMPNN_OK;
    }
    const int nch = ceil_div(rows, 128);
    if (ws_floats < colsum_ws_floats(rows, cols)) return set_error(TMPNN_EWORKSPACE, "colsum: workspace too small");
    hipLaunchKernelGGL(k_colsum_partial, dim3(nch, ceil_div(cols, 64)), dim3(256), 0, st, src, ld, mul, ldm, rows,
                       cols, ws);
    int rc = check_launch("colsum_partial");
    if (rc) return rc;
    return launch_reduce_slabs(ws, cols, nch, dst, cols, accumulate, st, ws + (size_t)nch * cols);
}

// ------------------------------------------------------------------------------------------
// BatchNorm pieces (segmented).  y [nd][H] = Lin1 output on det rows; all-zero rows give b1.
// ------------------------------------------------------------------------------------------
// one block per segment, H threads
// blockDim = (H, G): G row groups share a segment (G = 1 for the usual short windows: plain in-order sums; long segments --
// one giant window such as BASELINE C5's 15 000 dets -- take 1024 / H groups whose partial sums are combined in a fixed
// order: 7.4 -> ~0.3 ms there, still no atomics)
__global__ void k_bn_stats(const float* __restrict__ y, const int32_t* __restrict__ seg_ptr,
                           const int32_t* __restrict__ seg_cnt, int H, const float* __restrict__ b1,
                           float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float red[1024];
    const int s = blockIdx.x, j = threadIdx.x, g = threadIdx.y, G = blockDim.y;
    const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    const float cnt = (float)seg_cnt[s];
    const float nz = cnt - (float)(p1 - p0);      // all-zero rows of the segment
    const float b = b1[j];
    float sum = G == 1 ? nz * b : 0.f;
#pragma unroll 4
    for (int i = p0 + g; i < p1; i += G) sum += y[(size_t)i * H + j];
    if (G > 1) {
        red[g * H + j] = sum;
        __syncthreads();
        sum = nz * b;
        for (int k = 0; k < G; ++k) sum += red[k * H + j];
        __syncthreads();
    }
    const float m = sum / cnt;
    float sq = G == 1 ? nz * (b - m) * (b - m) : 0.f;
#pragma unroll 4
    for (int i = p0 + g; i < p1; i += G) {
        const float d = y[(size_t)i * H + j] - m;
        sq += d * d;
    }
    if (G > 1) {
        red[g * H + j] = sq;
        __syncthreads();
        sq = nz * (b - m) * (b - m);
        for (int k = 0; k < G; ++k) sq += red[k * H + j];
    }
    const float var = sq / cnt;
    if (g == 0) {
        mean[(size_t)s * H + j] = m;
        rstd[(size_t)s * H + j] = rsqrtf(var + BN_EPS);
    }
}
// row groups per segment for the two per-segment reductions (a property of the batch's shape)
static int bn_row_groups(int nd, int S, int H) { return (S > 0 && nd / S >= 256 && H <= 256) ? 1024 / H : 1; }

// running statistics: windows are seen one after another in the reference, so the momentum update is
// the EMA over segments in order: rm <- 0.9^S rm + sum_s 0.1 * 0.9^(S-1-s) mean_s (unbiased variance
// likewise).  A segment L places from the end carries 0.1 * 0.9^L, so only the last 320 can reach an fp32
// result (0.9^320 = 2e-15).  One 64-lane block per feature, lanes striding over those segments.
__global__ __launch_bounds__(64) void k_bn_running(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const int32_t* __restrict__ seg_cnt, int S, int H,
                                                   float* __restrict__ rm, float* __restrict__ rv) {
    const int j = blockIdx.x;
    const int lane = threadIdx.x;
    const int s0 = S > 320 ? S - 320 : 0;
    float m = 0.f, v = 0.f;
    for (int s = s0 + lane; s < S; s += 64) {
        const float w = BN_MOMENTUM * powf(1.0f - BN_MOMENTUM, (float)(S - 1 - s));
        const float cnt = (float)seg_cnt[s];
        const float r = rstd[(size_t)s * H + j];
        const float var = 1.0f / (r * r) - BN_EPS;
        m += w * mean[(size_t)s * H + j];
        v += w * var * (cnt / (cnt - 1.0f));
    }
    for (int off = 32; off >= 1; off >>= 1) { m += __shfl_xor(m, off); v += __shfl_xor(v, off); }
    if (lane == 0) {
        const float keep = s0 > 0 ? 0.f : powf(1.0f - BN_MOMENTUM, (float)S);
        rm[j] = keep * rm[j] + m;
        rv[j] = keep * rv[j] + v;
    }
}

__device__ __forceinline__ int seg_of(const int32_t* __restrict__ seg_ptr, int S, int i) {
    int lo = 0, hi = S - 1;            // last s with seg_ptr[s] <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_ptr[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// yhat = (y - mean) * rstd ; a = relu(yhat * gamma + beta).  mean/rstd per segment (training) or
// running (eval: seg_ptr == nullptr, mean/rstd hold one row).
// (four columns of one row per thread: H is a multiple of 4 and every buffer is 16-byte aligned; `nd * H / 4` threads)
__global__ void k_bn_apply(const float* __restrict__ y, int nd, int H, const int32_t* __restrict__ seg_ptr, int S,
                           const int32_t* __restrict__ seg_of_det,
                           const float* __restrict__ mean, const float* __restrict__ rstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta,
                           float* __restrict__ yhat_out, float* __restrict__ a_out) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= (size_t)nd * H) return;
    const int i = (int)(idx / H), j = (int)(idx % H);
    const int s = seg_ptr ? (seg_of_det ? seg_of_det[i] : seg_of(seg_ptr, S, i)) : 0;
    const float4 yv = *reinterpret_cast<const float4*>(y + idx);
    const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)s * H + j);
    const float4 r = *reinterpret_cast<const float4*>(rstd + (size_t)s * H + j);
    const float4 yh = make_float4((yv.x - m.x) * r.x, (yv.y - m.y) * r.y, (yv.z - m.z) * r.z, (yv.w - m.w) * r.w);
    if (yhat_out) *reinterpret_cast<float4*>(yhat_out + idx) = yh;
    if (a_out) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + j), b = *reinterpret_cast<const float4*>(beta + j);
        *reinterpret_cast<float4*>(a_out + idx) = make_float4(fmaxf(yh.x * g.x + b.x, 0.f), fmaxf(yh.y * g.y + b.y, 0.f),
                                                              fmaxf(yh.z * g.z + b.z, 0.f), fmaxf(yh.w * g.w + b.w, 0.f));
    }
}

__global__ void k_running_to_stats(const float* __restrict__ rm, const float* __restrict__ rv, int H,
                                   float* __restrict__ mean, float* __restrict__ rstd) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    mean[j] = rm[j];
    rstd[j] = rsqrtf(rv[j] + BN_EPS);
}

// dyhat = da * (yhat*gamma+beta > 0) * gamma   (in place over da); dz (= da*mask) is written to dz_out
__global__ void k_bn_bwd_act(float* __restrict__ da, const float* __restrict__ yhat, int nd, int H,
                             const float* __restrict__ gamma, const float* __restrict__ beta,
                             float* __restrict__ dz_out) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // four columns of one row per thread
    if (idx >= (size_t)nd * H) return;
    const int j = (int)(idx % H);
    const float4 yh = *reinterpret_cast<const float4*>(yhat + idx), d = *reinterpret_cast<const float4*>(da + idx);
    const float4 g = *reinterpret_cast<const float4*>(gamma + j), b = *reinterpret_cast<const float4*>(beta + j);
    const float4 dz = make_float4(yh.x * g.x + b.x > 0.f ? d.x : 0.f, yh.y * g.y + b.y > 0.f ? d.y : 0.f,
                                  yh.z * g.z + b.z > 0.f ? d.z : 0.f, yh.w * g.w + b.w > 0.f ? d.w : 0.f);
    *reinterpret_cast<float4*>(dz_out + idx) = dz;
    *reinterpret_cast<float4*>(da + idx) = make_float4(dz.x * g.x, dz.y * g.y, dz.z * g.z, dz.w * g.w);
}

// per segment: s1 = sum dyhat, s2 = sum dyhat*yhat (zero rows contribute nothing: their output is
// masked, track_mpnn.py:61).  Then s1 <- dy0, the gradient wrt the Lin1 output of a zero row.
__global__ void k_bn_bwd_seg(const float* __restrict__ dyhat, const float* __restrict__ yhat,
                             const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_cnt, int H,
                             const float* __restrict__ b1, const float* __restrict__ mean,
                             const float* __restrict__ rstd, float* __restrict__ s1, float* __restrict__ s2) {
    __shared__ float red[2048];
    const int s = blockIdx.x, j = threadIdx.x, g = threadIdx.y, G = blockDim.y;      // row groups as in k_bn_stats
    const int p0 = seg_ptr[s], p1 = seg_ptr[s + 1];
    float a = 0.f, b = 0.f;
#pragma unroll 4
    for (int i = p0 + g; i < p1; i += G) {
        const float d = dyhat[(size_t)i * H + j];
        a += d;
        b += d * yhat[(size_t)i * H + j];
    }
    if (G > 1) {
        red[g * H + j] = a;
        red[1024 + g * H + j] = b;
        __syncthreads();
        a = 0.f; b = 0.f;
        for (int k = 0; k < G; ++k) { a += red[k * H + j]; b += red[1024 + k * H + j]; }
    }
    if (g == 0) {
        s1[(size_t)s * H + j] = a;
        s2[(size_t)s * H + j] = b;
    }
}

// dy_i = rstd/cnt * (cnt*dyhat_i - s1 - yhat_i*s2)  (in place over dyhat)
__global__ void k_bn_bwd_dy(float* __restrict__ dyhat, const float* __restrict__ yhat, int nd, int H,
                            const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_cnt, int S,
                            const int32_t* __restrict__ seg_of_det,
                            const float* __restrict__ rstd, const float* __restrict__ s1,
                            const float* __restrict__ s2, int training) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // four columns of one row per thread
    if (idx >= (size_t)nd * H) return;
    const int i = (int)(idx / H), j = (int)(idx % H);
    float4 d = *reinterpret_cast<const float4*>(dyhat + idx);
    if (!training) {
        const float4 r = *reinterpret_cast<const float4*>(rstd + j);
        *reinterpret_cast<float4*>(dyhat + idx) = make_float4(d.x * r.x, d.y * r.y, d.z * r.z, d.w * r.w);
        return;
    }
    const int s = seg_of_det ? seg_of_det[i] : seg_of(seg_ptr, S, i);
    const float cnt = (float)seg_cnt[s];
    const size_t sj = (size_t)s * H + j;
    const float4 r = *reinterpret_cast<const float4*>(rstd + sj), a1 = *reinterpret_cast<const float4*>(s1 + sj);
    const float4 a2 = *reinterpret_cast<const float4*>(s2 + sj), yh = *reinterpret_cast<const float4*>(yhat + idx);
    d.x = r.x / cnt * (cnt * d.x - a1.x - yh.x * a2.x);
    d.y = r.y / cnt * (cnt * d.y - a1.y - yh.y * a2.y);
    d.z = r.z / cnt * (cnt * d.z - a1.z - yh.z * a2.z);
    d.w = r.w / cnt * (cnt * d.w - a1.w - yh.w * a2.w);
    *reinterpret_cast<float4*>(dyhat + idx) = d;
}

// dy0[s][j] = rstd/cnt * (-s1 - yhat0*s2), yhat0 = (b1 - mean)*rstd ; written over s1.
// zsum[j] = sum_s (cnt_s - nd_s) * dy0[s][j]  is added to db1 by the caller via colsum of `s2` <- nz*dy0.
__global__ void k_bn_bwd_dy0(float* __restrict__ s1, float* __restrict__ s2, const int32_t* __restrict__ seg_ptr,
                             const int32_t* __restrict__ seg_cnt, int S, int H, const float* __restrict__ b1,
                             const float* __restrict__ mean, const float* __restrict__ rstd) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)S * H) return;
    const int s = (int)(idx / H), j = (int)(idx % H);
    const float cnt = (float)seg_cnt[s];
    const float nz = cnt - (float)(seg_ptr[s + 1] - seg_ptr[s]);
    const float yh0 = (b1[j] - mean[idx]) * rstd[idx];
    const float dy0 = rstd[idx] / cnt * (-s1[idx] - yh0 * s2[idx]);
    s1[idx] = dy0;
    s2[idx] = nz * dy0;
}

__global__ void k_gather_rows(const float* __restrict__ src, long ld, const int32_t* __restrict__ rows, int n, int H,
                              float* __restrict__ dst) {
    const size_t idx = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // four columns of one row per thread
    if (idx >= (size_t)n * H) return;
    const int i = (int)(idx / H), j = (int)(idx % H);
    *reinterpret_cast<float4*>(dst + idx) = *reinterpret_cast<const float4*>(src + (size_t)rows[i] * ld + j);
}

// ------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------
// 16 lanes per row, float4 chunks
__global__ __launch_bounds__(256) void k_heads_fwd(const float* __restrict__ h, int ld_h, int C, int N,
                                                   const uint8_t* __restrict__ is_edge,
                                                   const float* __restrict__ w_node, const float* __restrict__ b_node,
                                                   const float* __restrict__ w_edge, const float* __restrict__ b_edge,
                                                   float* __restrict__ logits, float* __restrict__ scores) {
    const int sub = threadIdx.x & 15;
    const long rows_per_pass = (long)gridDim.x * 16;
    for (long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4); i < N; i += rows_per_pass) {
        const bool e = is_edge[i] != 0;
        const float* w = e ? w_edge : w_node;
        const float* hr = h + (size_t)i * ld_h;
        float s = 0.f;
        for (int c4 = sub * 4; c4 < C; c4 += 64) {
            const float4 x = *reinterpret_cast<const float4*>(hr + c4);
            const float4 ww = *reinterpret_cast<const float4*>(w + c4);
            s += x.x * ww.x + x.y * ww.y + x.z * ww.z + x.w * ww.w;
        }
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (sub == 0) {
            const float y = s + (e ? b_edge[0] : b_node[0]);
            logits[i] = y;
            scores[i] = 1.0f / (1.0f + expf(-y));
        }
    }
}

// logits[i] = sum_p parts[p][i] + b_type(i) ; scores = sigmoid   (finishes the head fused into tmpnn_gru_fwd)
__global__ void k_heads_finish(const float* __restrict__ parts, size_t stride, int nparts, int N,
                               const uint8_t* __restrict__ is_edge, const float* __restrict__ b_node,
                               const float* __restrict__ b_edge, float* __restrict__ logits,
                               float* __restrict__ scores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float y = is_edge[i] ? b_edge[0] : b_node[0];
    for (int p = 0; p < nparts; ++p) y += parts[(size_t)p * stride + i];
    logits[i] = y;
    scores[i] = 1.0f / (1.0f + expf(-y));
}

// d_h[i] (+)= dy_i * w_type ; per-block partial dw_node/dw_edge [C], db_node, db_edge
// block = 256 threads: cpt = C/4 threads per row (float4), slots = 256/cpt rows per pass.
// partial layout per block: [2][C] then [2]
__global__ __launch_bounds__(256) void k_heads_bwd(const float* __restrict__ h, int ld_h, int C, int N,
                                                   const uint8_t* __restrict__ is_edge,
                                                   const float* __restrict__ w_node, const float* __restrict__ w_edge,
                                                   const float* __restrict__ scores, const float* __restrict__ d_logits,
                                                   const float* __restrict__ d_scores, float* __restrict__ dy_out,
                                                   float* __restrict__ d_h,
                                                   int ld_dh, int accumulate, int rows_per_block,
                                                   float* __restrict__ part) {
    extern __shared__ float sm[];          // [slots][2][C] + [slots][2]
    const int cpt = C >> 2;
    const int slots = 256 / cpt;
    const int slot = threadIdx.x / cpt;
    const int c4 = (threadIdx.x % cpt) * 4;
    const bool active = slot < slots;
    float4 an = make_float4(0.f, 0.f, 0.f, 0.f), ae = an;
    float bn = 0.f, be = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min((long)N, r0 + rows_per_block);
    if (active) {
        // four rows per trip with every load issued up front: one row per trip leaves the loop latency-bound
        // (a dependent scalar + 16-byte load chain per row and ~200 trips per thread)
        constexpr int UR = 4;
        for (long i0 = r0 + slot; i0 < r1; i0 += (long)slots * UR) {
            float dyv[UR];
            bool ev[UR], live[UR];
            float4 xv[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long i = i0 + (long)u * slots;
                live[u] = i < r1;
                const long ic = live[u] ? i : r1 - 1;
                float dy = d_logits ? d_logits[ic] : 0.f;
                if (d_scores) {
                    const float sc = scores[ic];
                    dy += d_scores[ic] * sc * (1.0f - sc);
                }
                dyv[u] = live[u] ? dy : 0.f;
                ev[u] = is_edge[ic] != 0;
                xv[u] = *reinterpret_cast<const float4*>(h + (size_t)ic * ld_h + c4);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if (!live[u]) continue;
                const long i = i0 + (long)u * slots;
                const float dy = dyv[u];
                const bool e = ev[u];
                const float4 x = xv[u];
                if (dy_out && c4 == 0) dy_out[i] = dy;
                if (d_h) {
                    const float4 w = *reinterpret_cast<const float4*>((e ? w_edge : w_node) + c4);
                    float* dp = d_h + (size_t)i * ld_dh + c4;
                    float4 o = make_float4(dy * w.x, dy * w.y, dy * w.z, dy * w.w);
                    if (accumulate) {
                        const float4 p = *reinterpret_cast<const float4*>(dp);
                        o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
                    }
                    *reinterpret_cast<float4*>(dp) = o;
                }
                if (e) { ae.x += dy * x.x; ae.y += dy * x.y; ae.z += dy * x.z; ae.w += dy * x.w; if (c4 == 0) be += dy; }
                else   { an.x += dy * x.x; an.y += dy * x.y; an.z += dy * x.z; an.w += dy * x.w; if (c4 == 0) bn += dy; }
            }
        }
        float* my = sm + (size_t)slot * 2 * C;
        *reinterpret_cast<float4*>(my + c4) = an;
        *reinterpret_cast<float4*>(my + C + c4) = ae;
        if (c4 == 0) {
            sm[(size_t)slots * 2 * C + slot * 2 + 0] = bn;
            sm[(size_t)slots * 2 * C + slot * 2 + 1] = be;
        }
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * (2 * C + 2);
    for (int j = threadIdx.x; j < 2 * C + 2; j += 256) {
        float s = 0.f;
        if (j < 2 * C) for (int k = 0; k < slots; ++k) s += sm[(size_t)k * 2 * C + j];
        else for (int k = 0; k < slots; ++k) s += sm[(size_t)slots * 2 * C + k * 2 + (j - 2 * C)];
        out[j] = s;
    }
}

// scatter the reduced [2C+2] vector into the four gradient buffers (+=)
__global__ void k_heads_bwd_final(const float* __restrict__ red, int C, float* __restrict__ dw_node,
                                  float* __restrict__ db_node, float* __restrict__ dw_edge,
                                  float* __restrict__ db_edge) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * C + 2) return;
    const float s = red[j];
    if (j < C) dw_node[j] += s;
    else if (j < 2 * C) dw_edge[j - C] += s;
    else if (j == 2 * C) db_node[0] += s;
    else db_edge[0] += s;
}

static int heads_rows_per_block(int N) {
    long rpb = ((long)N + 2047) / 2048;     // at most 2048 partial blocks
    if (rpb < 64) rpb = 64;
    return (int)rpb;
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_input_bn_fwd(const float* xdet, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int H, int training, const float* w1, const float* b1, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, const float* w2, const float* b2,
                       float* y_save, float* mean, float* rstd, float* ws_a, const int32_t* out_row, float* h_new,
                       int ld_h, tmpnn_stream stream) {
    TM_REQUIRE(supported_H_cell(H), "input_bn_fwd: unsupported H=%d", H);
    TM_REQUIRE(F > 0 && nd >= 0 && S >= 0, "input_bn_fwd: F=%d nd=%d S=%d", F, nd, S);
    TM_REQUIRE(w1 && b1 && gamma && beta && running_mean && running_var && w2 && b2 && mean && rstd,
               "input_bn_fwd: null parameter pointer");
    TM_REQUIRE(!training || (seg_ptr && seg_cnt && S > 0), "input_bn_fwd: training needs segments");
    hipStream_t st = as_stream(stream);
    int rc;
    if (nd > 0) {
        TM_REQUIRE(xdet && y_save && ws_a && out_row && h_new && ld_x >= F && ld_h >= H, "input_bn_fwd: null/short buffers");
        TM_REQUIRE(aligned16(y_save) && aligned16(ws_a) && aligned16(mean) && aligned16(rstd) && aligned16(gamma) && aligned16(beta),
                   "input_bn_fwd: y_save / ws_a / mean / rstd / gamma / beta must be 16-byte aligned");
        GemmArgs g{xdet, ld_x, 1, nullptr, nullptr, w1, 1, F, b1, y_save, H, nullptr, nd, H, F, 0};   // y = x W1^T + b1
        if ((rc = launch_gemm(g, st))) return rc;
    }
    if (training) {
        hipLaunchKernelGGL(k_bn_stats, dim3(S), dim3(H, bn_row_groups(nd, S, H)), 0, st, y_save, seg_ptr, seg_cnt, H, b1, mean, rstd);
        if ((rc = check_launch("bn_stats"))) return rc;
        hipLaunchKernelGGL(k_bn_running, dim3(H), dim3(64), 0, st, mean, rstd, seg_cnt, S, H, running_mean,
                           running_var);
        if ((rc = check_launch("bn_running"))) return rc;
    } else {
        hipLaunchKernelGGL(k_running_to_stats, dim3(ceil_div(H, 64)), dim3(64), 0, st, running_mean, running_var, H,
                           mean, rstd);
        if ((rc = check_launch("running_to_stats"))) return rc;
    }
    if (nd == 0) return TMPNN_OK;
    hipLaunchKernelGGL(k_bn_apply, dim3(ceil_div((long)nd * H / 4, 256)), dim3(256), 0, st, y_save, nd, H,
                       training ? seg_ptr : nullptr, S, seg_of_det, mean, rstd, gamma, beta, (float*)nullptr, ws_a);
    if ((rc = check_launch("bn_apply"))) return rc;
    GemmArgs g2{ws_a, H, 1, nullptr, nullptr, w2, 1, H, b2, h_new, ld_h, out_row, nd, H, H, 0};        // out = a W2^T + b2
    return launch_gemm(g2, st);
}

int tmpnn_input_bn_bwd(const float* xdet, int ld_x, int F, int nd, const int32_t* seg_ptr, const int32_t* seg_cnt,
                       const int32_t* seg_of_det, int S, int H, int training, const float* w1, const float* b1, const float* gamma,
                       const float* beta, const float* w2, const float* y_save, const float* mean, const float* rstd,
                       const int32_t* out_row, const float* d_h, int ld_dh, float* d_xdet, int ld_dx, float* d_xzero,
                       float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, float* ws,
                       size_t ws_floats, tmpnn_stream stream) {
    TM_REQUIRE(supported_H_cell(H), "input_bn_bwd: unsupported H=%d", H);
    TM_REQUIRE(F > 0 && nd >= 0 && S >= 0, "input_bn_bwd: F=%d nd=%d S=%d", F, nd, S);
    TM_REQUIRE(w1 && b1 && gamma && beta && w2 && mean && rstd && dw1 && db1 && dgamma && dbeta && dw2 && db2,
               "input_bn_bwd: null parameter pointer");
    hipStream_t st = as_stream(stream);
    if (nd == 0) {
        if (d_xzero && S > 0) (void)hipMemsetAsync(d_xzero, 0, sizeof(float) * (size_t)S * F, st);
        return TMPNN_OK;   // no det rows: nothing reaches the loss through this transform
    }
    TM_REQUIRE(xdet && y_save && out_row && d_h && ws, "input_bn_bwd: null buffers");
    const size_t ndH = (size_t)nd * H, SH = (size_t)(S > 0 ? S : 1) * H;
    const size_t need = tmpnn_input_bn_bwd_ws(nd, S, H, F);
    if (ws_floats < need) return set_error(TMPNN_EWORKSPACE, "input_bn_bwd: workspace %zu < %zu floats", ws_floats, need);
    TM_REQUIRE(aligned16(y_save) && aligned16(mean) && aligned16(rstd) && aligned16(gamma) && aligned16(beta) && aligned16(ws) &&
                   aligned16(d_h) && (ld_dh & 3) == 0,
               "input_bn_bwd: y_save / mean / rstd / gamma / beta / ws / d_h must be 16-byte aligned (ld_dh a multiple of 4)");
    float* B0 = ws;              // d_out -> dy
    float* B1 = B0 + ndH;        // yhat
    float* B2 = B1 + ndH;        // a -> da -> dyhat
    float* s1 = B2 + ndH;        // [S][H]
    float* s2 = s1 + SH;
    float* scratch = s2 + SH;
    const size_t scratch_n = ws_floats - (3 * ndH + 2 * SH);
    const int gridE = ceil_div((long)ndH / 4, 256);          // four columns per thread
    int rc;
    hipLaunchKernelGGL(k_gather_rows, dim3(gridE), dim3(256), 0, st, d_h, (long)ld_dh, out_row, nd, H, B0);
    if ((rc = check_launch("gather_rows"))) return rc;
    hipLaunchKernelGGL(k_bn_apply, dim3(gridE), dim3(256), 0, st, y_save, nd, H, training ? seg_ptr : nullptr, S,
                       seg_of_det, mean, rstd, gamma, beta, B1, B2);
    if ((rc = check_launch("bn_apply"))) return rc;
    // dW2 += d_out^T a ; db2 += colsum(d_out)
    {
        GemmArgs g{B0, 1, H, nullptr, nullptr, B2, H, 1, nullptr, dw2, H, nullptr, H, H, nd, 1};
        if ((rc = launch_gemm_splitk(g, scratch, scratch_n, st))) return rc;
        if ((rc = launch_colsum(B0, H, nullptr, 0, nd, H, db2, 1, scratch, scratch_n, st))) return rc;
    }
    // da = d_out W2  -> B2
    {
        GemmArgs g{B0, H, 1, nullptr, nullptr, w2, H, 1, nullptr, B2, H, nullptr, nd, H, H, 0};
        if ((rc = launch_gemm(g, st))) return rc;
    }
    // dz = da*mask -> B0 ; dyhat = dz*gamma -> B2 ; dgamma += sum dz*yhat ; dbeta += sum dz
    hipLaunchKernelGGL(k_bn_bwd_act, dim3(gridE), dim3(256), 0, st, B2, B1, nd, H, gamma, beta, B0);
    if ((rc = check_launch("bn_bwd_act"))) return rc;
    if ((rc = launch_colsum(B0, H, B1, H, nd, H, dgamma, 1, scratch, scratch_n, st))) return rc;
    if ((rc = launch_colsum(B0, H, nullptr, 0, nd, H, dbeta, 1, scratch, scratch_n, st))) return rc;
    if (training) {
        hipLaunchKernelGGL(k_bn_bwd_seg, dim3(S), dim3(H, bn_row_groups(nd, S, H)), 0, st, B2, B1, seg_ptr, seg_cnt, H, b1, mean, rstd, s1, s2);
        if ((rc = check_launch("bn_bwd_seg"))) return rc;
    }
    hipLaunchKernelGGL(k_bn_bwd_dy, dim3(gridE), dim3(256), 0, st, B2, B1, nd, H, seg_ptr, seg_cnt, S, seg_of_det, rstd,
                       s1, s2, training);
    if ((rc = check_launch("bn_bwd_dy"))) return rc;
    // now B2 = dy (det rows)
    // dW1 += dy^T x ; db1 += colsum(dy) (+ zero-row part below) ; d_xdet = dy W1
    {
        GemmArgs g{B2, 1, H, nullptr, nullptr, xdet, ld_x, 1, nullptr, dw1, F, nullptr, H, F, nd, 1};
        if ((rc = launch_gemm_splitk(g, scratch, scratch_n, st))) return rc;
        if ((rc = launch_colsum(B2, H, nullptr, 0, nd, H, db1, 1, scratch, scratch_n, st))) return rc;
        if (d_xdet) {
            GemmArgs gx{B2, H, 1, nullptr, nullptr, w1, F, 1, nullptr, d_xdet, ld_dx, nullptr, nd, F, H, 0};
            if ((rc = launch_gemm(gx, st))) return rc;
        }
    }
    if (training) {
        hipLaunchKernelGGL(k_bn_bwd_dy0, dim3(ceil_div((long)S * H, 256)), dim3(256), 0, st, s1, s2, seg_ptr, seg_cnt,
                           S, H, b1, mean, rstd);
        if ((rc = check_launch("bn_bwd_dy0"))) return rc;
        // db1 += sum_s nz_s * dy0_s ; d_xzero = dy0 W1
        const size_t cs2 = colsum_ws_floats(S, H);
        if (scratch_n < cs2) return set_error(TMPNN_EWORKSPACE, "input_bn_bwd: workspace too small for segments");
        if ((rc = launch_colsum(s2, H, nullptr, 0, S, H, db1, 1, scratch, scratch_n, st))) return rc;
        if (d_xzero) {
            GemmArgs gz{s1, H, 1, nullptr, nullptr, w1, F, 1, nullptr, d_xzero, F, nullptr, S, F, H, 0};
            if ((rc = launch_gemm(gz, st))) return rc;
        }
    } else if (d_xzero && S > 0) {
        (void)hipMemsetAsync(d_xzero, 0, sizeof(float) * (size_t)S * F, st);
    }
    return TMPNN_OK;
}

size_t tmpnn_input_bn_bwd_ws(int nd, int S, int H, int F) {
    if (nd <= 0) return 0;
    const size_t ndH = (size_t)nd * H, SH = (size_t)(S > 0 ? S : 1) * H;
    size_t x = colsum_ws_floats(nd, H);
    const size_t a = gemm_splitk_ws_floats(H, H, nd), b = gemm_splitk_ws_floats(H, F, nd),
                 c = colsum_ws_floats(S > 0 ? S : 1, H);
    if (a > x) x = a;
    if (b > x) x = b;
    if (c > x) x = c;
    return 3 * ndH + 2 * SH + x;
}

int tmpnn_heads_fwd(const float* h, int ld_h, int C, int N, const uint8_t* is_edge, const float* w_node,
                    const float* b_node, const float* w_edge, const float* b_edge, float* logits, float* scores,
                    tmpnn_stream stream) {
    TM_REQUIRE(C > 0 && (C & 3) == 0 && N >= 0, "heads_fwd: C=%d N=%d", C, N);
    if (N == 0) return TMPNN_OK;
    TM_REQUIRE(h && is_edge && w_node && b_node && w_edge && b_edge && logits && scores, "heads_fwd: null pointer");
    TM_REQUIRE(ld_h >= C && (ld_h & 3) == 0 && aligned16(h) && aligned16(w_node) && aligned16(w_edge),
               "heads_fwd: rows must be 16-byte aligned");
    long b = ((long)N + 15) / 16;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(k_heads_fwd, dim3((int)b), dim3(256), 0, as_stream(stream), h, ld_h, C, N, is_edge, w_node,
                       b_node, w_edge, b_edge, logits, scores);
    return check_launch("heads_fwd");
}

int tmpnn_heads_finish(const float* parts, size_t part_stride, int nparts, int N, const uint8_t* is_edge,
                       const float* b_node, const float* b_edge, float* logits, float* scores, tmpnn_stream stream) {
    TM_REQUIRE(nparts > 0 && N >= 0, "heads_finish: nparts=%d N=%d", nparts, N);
    if (N == 0) return TMPNN_OK;
    TM_REQUIRE(parts && is_edge && b_node && b_edge && logits && scores && part_stride >= (size_t)N, "heads_finish: null pointer");
    hipLaunchKernelGGL(k_heads_finish, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), parts, part_stride, nparts, N,
                       is_edge, b_node, b_edge, logits, scores);
    return check_launch("heads_finish");
}

size_t tmpnn_heads_bwd_ws(int N, int C) {
    if (N <= 0) return 0;
    const int rpb = heads_rows_per_block(N);
    const int nblk = ceil_div(N, rpb);
    const size_t n = 2 * (size_t)C + 2;
    return ((size_t)nblk * n + n + reduce_slabs_ws_floats(nblk, n)) * sizeof(float);
}

int tmpnn_heads_bwd(const float* h, int ld_h, int C, int N, const uint8_t* is_edge, const float* w_node,
                    const float* w_edge, const float* scores, const float* d_logits, const float* d_scores,
                    float* dy_out, float* d_h, int ld_dh, int accumulate, float* dw_node, float* db_node,
                    float* dw_edge, float* db_edge, void* ws, size_t ws_bytes, tmpnn_stream stream) {
    TM_REQUIRE(C > 0 && (C & 3) == 0 && C <= 1024 && N >= 0, "heads_bwd: C=%d N=%d", C, N);
    if (N == 0) return TMPNN_OK;
    TM_REQUIRE(h && is_edge && w_node && w_edge && dw_node && db_node && dw_edge && db_edge, "heads_bwd: null pointer");
    TM_REQUIRE(d_scores == nullptr || scores != nullptr, "heads_bwd: d_scores needs scores");
    TM_REQUIRE((ld_h & 3) == 0 && aligned16(h) && aligned16(w_node) && aligned16(w_edge) &&
                   (d_h == nullptr || ((ld_dh & 3) == 0 && aligned16(d_h))),
               "heads_bwd: rows must be 16-byte aligned");
    const size_t need = tmpnn_heads_bwd_ws(N, C);
    if (!ws || ws_bytes < need) return set_error(TMPNN_EWORKSPACE, "heads_bwd: workspace %zu < %zu", ws_bytes, need);
    const int rpb = heads_rows_per_block(N);
    const int nblk = ceil_div(N, rpb);
    const int cpt = C >> 2;
    TM_REQUIRE(cpt <= 256, "heads_bwd: C too wide");
    const int slots = 256 / cpt;
    const size_t shm = ((size_t)slots * 2 * C + (size_t)slots * 2) * sizeof(float);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_heads_bwd, dim3(nblk), dim3(256), shm, st, h, ld_h, C, N, is_edge, w_node, w_edge, scores,
                       d_logits, d_scores, dy_out, d_h, ld_dh, accumulate, rpb, reinterpret_cast<float*>(ws));
    int rc = check_launch("heads_bwd");
    if (rc) return rc;
    const size_t nred = 2 * (size_t)C + 2;
    float* part = reinterpret_cast<float*>(ws);
    float* red = part + (size_t)nblk * nred;
    if ((rc = launch_reduce_slabs(part, nred, nblk, red, nred, 0, st, red + nred))) return rc;
    hipLaunchKernelGGL(k_heads_bwd_final, dim3(ceil_div(2 * C + 2, 256)), dim3(256), 0, st, red, C, dw_node, db_node,
                       dw_edge, db_edge);
    return check_launch("heads_bwd_final");
}

}  // extern "C"
