// Fused message-passing iteration for ONE small graph (batch-1 path; include/tmpnn.h: tmpnn_mp_iter_*).
//
// The reference's real workload is one tracking window per call (train.py:92-107, infer.py:60-87; batch size 1,
// utils/graph.py:117): a few hundred to a few thousand rows, H = 64.  At that size the staged entry points are
// launch-bound (a dozen launches per group forward, two dozen backward) and their LDS-resident persistent kernels
// pay a 100+ KB weight load per block for a handful of rows.  Here a forward call is TWO launches and a backward
// TWO, each sized to the graph:
//
//   k_small_bn_fwd      rows C, D     input transform of the new det rows (Lin-BN-ReLU-Lin, batch statistics over ALL
//                                     new rows with the analytic zero-row terms), zeros on new edge rows
//   k_small_iter_fwd    rows E-J      16-row tiles: edge tiles form h[src]-h[dst] (or the concat) on the fly, det
//                                     tiles reduce their incident edge rows (CSR, fixed order); both cells' GEMMs on
//                                     v_mfma_f32_16x16x4_f32 with weight operands pre-arranged for coalesced 16-byte
//                                     loads (tmpnn_mp_iter_prepare); GRU gates, merge (row indirection) and the
//                                     output head in the epilogue
//   k_small_iter_bwd    row K         per tile: gate gradients, d_x = d_gi W_ih, d_h = dh z + d_gh W_hh, and the
//                                     weight / bias / head gradients accumulated in REGISTERS over a persistent
//                                     block's tiles (one slab per block)
//   k_small_bwd_finish  row K         adjoints of rows E and F (CSR segment sums / gathers of d_x), slab reduction and --
//                                     in G further blocks of the same launch -- the input-transform backward
//
// Sizes (E, Dn) are read from the graph's device-side meta (tmpnn_dgraph): grids are sized from N alone and
// surplus blocks exit, so the host never synchronises.  fp32 throughout; every reduction has a fixed order.
#include "common.h"
#include "small_bn_dev.h"

namespace tmpnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TR = 16;            // rows per tile
static constexpr int SMALL_BWD_BLOCKS = 96;

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS image of a [TR][W] operand tile whose k index is PERMUTED so that the four k values a lane needs for four
// consecutive MFMA steps (k = 4*kq + (lane >> 4), kq = 4i .. 4i+3) are one aligned float4:
//   element k of a row lives at position (k & 3) * (W / 4) + (k >> 2).
__device__ __forceinline__ int perm_pos(int k, int W) { return (k & 3) * (W >> 2) + (k >> 2); }

// ------------------------------------------------------------------------------------------------------------
// weight operand images
// ------------------------------------------------------------------------------------------------------------
// W [3H][IN] row-major (reference layout).  Forward image (B operand of  g = x W^T : B[k][n] = W[n][k]):
//   Wf[(((gate * (H/16) + cs) * (IN/16) + i) * 64 + lane) * 4 + jj] = W[gate*H + 16 cs + (lane & 15)][4 (4i + jj) + (lane >> 4)]
// Backward-data image (B operand of  d_x = d_g W : B[k][n] = W[k][n]):
//   Wb[((ct * (3H/16) + i) * 64 + lane) * 4 + jj] = W[4 (4i + jj) + (lane >> 4)][16 ct + (lane & 15)]
struct PrepJob { const float* W; int IN; size_t off_f, off_b; };
struct PrepArgs { PrepJob job[12]; int njobs; int H; };

__global__ __launch_bounds__(256) void k_small_prepare(PrepArgs a, float* __restrict__ prep) {
    const PrepJob jb = a.job[blockIdx.y];
    const int H = a.H, IN = jb.IN;
    const int total = 3 * H * IN;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        {   // forward image
            const int jj = idx & 3, lane = (idx >> 2) & 63;
            int rest = idx >> 8;
            const int i = rest % (IN / 16); rest /= (IN / 16);
            const int cs = rest % (H / 16);
            const int gate = rest / (H / 16);
            prep[jb.off_f + idx] = jb.W[(size_t)(gate * H + 16 * cs + (lane & 15)) * IN + 4 * (4 * i + jj) + (lane >> 4)];
        }
        {   // backward-data image
            const int jj = idx & 3, lane = (idx >> 2) & 63;
            int rest = idx >> 8;
            const int i = rest % (3 * H / 16);
            const int ct = rest / (3 * H / 16);
            prep[jb.off_b + idx] = jb.W[(size_t)(4 * (4 * i + jj) + (lane >> 4)) * IN + 16 * ct + (lane & 15)];
        }
    }
}

// offsets (floats) of the images inside `prep`: per group [edge ih f|b][edge hh f|b][node ih f|b][node hh f|b]
struct PrepLayout {
    size_t per_group, e_ih_f, e_ih_b, e_hh_f, e_hh_b, n_ih_f, n_ih_b, n_hh_f, n_hh_b;
};
__host__ __device__ inline PrepLayout prep_layout(int H, int IN_e) {
    PrepLayout L;
    const size_t a = (size_t)3 * H * IN_e, b = (size_t)3 * H * H;
    L.e_ih_f = 0; L.e_ih_b = a; L.e_hh_f = 2 * a; L.e_hh_b = 2 * a + b;
    L.n_ih_f = 2 * a + 2 * b; L.n_ih_b = L.n_ih_f + b; L.n_hh_f = L.n_ih_b + b; L.n_hh_b = L.n_hh_f + b;
    L.per_group = 2 * a + 6 * b;
    return L;
}

// (saved-for-backward layout: small_bn_dev.h)

// ------------------------------------------------------------------------------------------------------------
// input transform, forward: one block per feature group (the transform itself: small_bn_dev.h)
// ------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void k_small_bn_fwd(BnFwdArgs a) {
    d_small_bn_fwd<H>(a, (int)blockIdx.x, BnSrcGraph{a.g.is_edge, a.g.N - a.n_new, a.x, a.ld_x});
}

// ------------------------------------------------------------------------------------------------------------
// the iteration, forward
// ------------------------------------------------------------------------------------------------------------
struct IterFwdArgs {
    tmpnn_mp_params P;
    tmpnn_dgraph g;
    const float* prep;
    const float* h;            // [N][G*H]  (h_cat)
    float* h_out;              // [N][G*H]
    float* logits; float* scores;
    float* gates;              // [G][4][N][H] or NULL
    float* es;                 // [G][N][H]   or NULL (nothing saved)
    int es_given;              // != 0: es already holds the edge -> node aggregate of every det (the attention stage wrote it:
                               // tmpnn_att_fwd between the two launches of tmpnn_mp_iter_fwd_parts); det tiles read it
    int det_score_one;         // != 0: scores[det rows] = 1 (a model without TP classifier at inference, infer.py:53-56, 77-80)
    int det_tile;              // dets per det tile (<= TR): see k_small_iter_fwd
};

// gi/gh MFMA loop of one column slice: acc[gate] += A(tile rows, K = W) x image.  The weight operands (3 gates x W/16
// float4 per lane, L2-resident) are requested up front in one burst: issued inside the loop, every 16-k step waited
// for its own L2 round trip (measured: 8 such waits were a third of the kernel at N = 375).
template <int W, int H>
struct CellB { float4 v[3][W / 16]; };

template <int W, int H>
__device__ __forceinline__ void cell_load_b(const float* __restrict__ img, int cs, int lane, CellB<W, H>& b) {
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int i = 0; i < W / 16; ++i)
            b.v[gate][i] = *reinterpret_cast<const float4*>(img + ((size_t)((gate * (H / 16) + cs) * (W / 16) + i) * 64 + lane) * 4);
}

template <int W, int H>
__device__ __forceinline__ void cell_gemm(const float* __restrict__ sA, int ldA, const CellB<W, H>& b, int lane, f32x4 (&acc)[3]) {
    const int row = lane & 15, kh = lane >> 4;
    const float* ap = sA + row * ldA + kh * (W / 4);
#pragma unroll
    for (int i = 0; i < W / 16; ++i) {
        const float4 av = *reinterpret_cast<const float4*>(ap + 4 * i);
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
            acc[gate] = mfma16(av.x, b.v[gate][i].x, acc[gate]);
            acc[gate] = mfma16(av.y, b.v[gate][i].y, acc[gate]);
            acc[gate] = mfma16(av.z, b.v[gate][i].z, acc[gate]);
            acc[gate] = mfma16(av.w, b.v[gate][i].w, acc[gate]);
        }
    }
}

template <int H, int IN_E>
__global__ __launch_bounds__(256) void k_small_iter_fwd(IterFwdArgs a) {
    if (a.g.meta[2] != 0) {
        // the adjacency failed validation (tmpnn_graph_from_coo presents it as empty): nothing may look like a result
        // until the host reads the status -- every output row becomes NaN
        const float qnan = __int_as_float(0x7fc00000);
        const long total = (long)a.g.N * a.P.G * H;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) a.h_out[i] = qnan;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.g.N; i += gridDim.x * 256) { a.logits[i] = qnan; a.scores[i] = qnan; }
        return;
    }
    // A det tile's staging sums the det's incident edge rows (~50 per det in a KITTI-sized window): with TR dets per tile that is
    // ~800 row loads from ONE CU, bound by its L1's outstanding misses (10 us of the launch's 16 at the C2 shape, 18 of 23 at the
    // BDD shape, while the edge tiles are done after 2) -- and a window has only 2-5 such tiles.  Small graphs therefore take
    // det_tile = 4 dets per tile (four times the CUs on the aggregation; the matrix phase computes TR rows either way, rows
    // beyond the tile's dets are never stored): same arithmetic per row, bit-identical results (round 6).
    const int E = a.g.meta[0], Dn = a.g.meta[1];
    const int TD = a.det_tile;
    const int nEt = (E + TR - 1) / TR, nDt = (Dn + TD - 1) / TD;
    const int b = blockIdx.x;
    if (b >= nEt + nDt) return;
    const bool is_e = b < nEt;
    const int r0 = (is_e ? b : b - nEt) * (is_e ? TR : TD);
    const int R = is_e ? E : min(Dn, r0 + TD);       // (rows of the tile at or beyond R are not part of it)
    const int G = a.P.G, GH = G * H, N = a.g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int LDX = IN_E + 4, LDH = H + 4;
    __shared__ __attribute__((aligned(16))) float sX[TR * LDX];
    __shared__ __attribute__((aligned(16))) float sH[TR * LDH];
    __shared__ float sHp[TR * (H + 1)];        // h_prev of the tile, natural order (merge term of the epilogue)
    __shared__ float sLog[4][TR];
    __shared__ int sRow[TR], sS[TR], sD[TR];
    // det tiles: the CSR runs of the tile's dets are one contiguous stretch of inc[] -- staged in LDS in one coalesced pass, so that
    // the aggregation below is straight-line row loads, sixteen in flight per thread, instead of a dependent (incidence -> row)
    // pair of round trips per four rows: ~26 of them in a row for a det of a KITTI-sized window
    constexpr int SINC = 1024;
    __shared__ int sPtr[TR + 1], sInc[SINC];
    if (tid < TR) {
        const int q = r0 + tid;
        const bool ok = q < R;
        const int qc = ok ? q : R - 1;
        sRow[tid] = is_e ? a.g.edge_row[qc] : a.g.det_row[qc];
        sS[tid] = is_e ? a.g.src[qc] : 0;
        sD[tid] = is_e ? a.g.dst[qc] : 0;
    } else if (tid < 2 * TR + 1 && !is_e) {
        sPtr[tid - TR] = a.g.rowptr[min(r0 + tid - TR, R)];
    }
    float lsum[4] = {0.f, 0.f, 0.f, 0.f};
    const PrepLayout PL = prep_layout(H, IN_E);
    const int IN = is_e ? IN_E : H;
    const float* w_head = is_e ? a.P.w_edge : a.P.w_node;
    __syncthreads();
    bool inc_lds = false;
    if (!is_e && !a.es_given) {                                            // (block-uniform)
        const int pb = sPtr[0], cnt = sPtr[TR] - pb;
        inc_lds = cnt <= SINC;
        if (inc_lds) {
            for (int i = tid; i < cnt; i += 256) sInc[i] = a.g.inc[pb + i];
            __syncthreads();
        }
    }
    for (int gi = 0; gi < G; ++gi) {
        const float* hg = a.h + gi * H;
        const float* img = a.prep + (size_t)gi * PL.per_group;
        const float* img_ih = img + (is_e ? PL.e_ih_f : PL.n_ih_f);
        const float* img_hh = img + (is_e ? PL.e_hh_f : PL.n_hh_f);
        const float* b_ih = is_e ? a.P.e_bih[gi] : a.P.n_bih[gi];
        const float* b_hh = is_e ? a.P.e_bhh[gi] : a.P.n_bhh[gi];
        // weight operands of this wave's column slice (one slice per wave at H = 64): requested now, consumed after
        // the tile has been staged
        CellB<IN_E, H> bI_e;
        CellB<H, H> bI_n, bH;
        float bir = 0.f, biz = 0.f, bin_ = 0.f, bhr = 0.f, bhz = 0.f, bhn = 0.f, wh = 0.f;
        if (wave < H / 16) {
            if (is_e) cell_load_b<IN_E, H>(img_ih, wave, lane, bI_e);
            else cell_load_b<H, H>(img_ih, wave, lane, bI_n);
            cell_load_b<H, H>(img_hh, wave, lane, bH);
            const int col0 = 16 * wave + (lane & 15);
            bir = b_ih[col0]; biz = b_ih[H + col0]; bin_ = b_ih[2 * H + col0];
            bhr = b_hh[col0]; bhz = b_hh[H + col0]; bhn = b_hh[2 * H + col0];
            wh = w_head[gi * H + col0];
        }
        // ---- stage the operand tiles
        if (tid < TR * (H / 4)) {
            const int row = tid / (H / 4), c4 = tid % (H / 4);
            const bool ok = r0 + row < R;
            const int grow = sRow[row];
            float4 hp = make_float4(0.f, 0.f, 0.f, 0.f), x0 = hp, x1 = hp;
            if (ok) {
                hp = *reinterpret_cast<const float4*>(hg + (size_t)grow * GH + 4 * c4);
                if (is_e) {
                    x0 = *reinterpret_cast<const float4*>(hg + (size_t)sS[row] * GH + 4 * c4);
                    x1 = *reinterpret_cast<const float4*>(hg + (size_t)sD[row] * GH + 4 * c4);
                    if (IN_E == H) { x0.x -= x1.x; x0.y -= x1.y; x0.z -= x1.z; x0.w -= x1.w; }
                } else if (a.es_given) {
                    // attention-weighted aggregate (models/layers.py:105-112), computed by tmpnn_att_fwd into the save area
                    x0 = *reinterpret_cast<const float4*>(a.es + ((size_t)gi * N + (r0 + row)) * H + 4 * c4);
                } else {
                    // edge -> node aggregation (models/layers.py:103): signed sum over the det's incident edge rows,
                    // CSR order (ascending edge row), four rows in flight
                    const int d = r0 + row;
                    const int p0 = sPtr[row], p1 = sPtr[row + 1];
                    if (inc_lds) {
                        constexpr int UA = 16;
                        const int* keys = sInc - sPtr[0];
                        for (int p = p0; p < p1; p += UA) {
                            int key[UA];
                            float4 v[UA];
#pragma unroll
                            for (int u = 0; u < UA; ++u) key[u] = p + u < p1 ? keys[p + u] : 0;
#pragma unroll
                            for (int u = 0; u < UA; ++u)
                                v[u] = *reinterpret_cast<const float4*>(hg + (size_t)(key[u] & 0x7fffffff) * GH + 4 * c4);
#pragma unroll
                            for (int u = 0; u < UA; ++u)
                                if (p + u < p1) {
                                    const float sgn = key[u] < 0 ? -1.0f : 1.0f;
                                    x0.x += sgn * v[u].x; x0.y += sgn * v[u].y; x0.z += sgn * v[u].z; x0.w += sgn * v[u].w;
                                }
                        }
                    } else
                    for (int p = p0; p < p1; p += 4) {
                        float4 v[4];
                        float sg[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool live = p + u < p1;
                            const int key = live ? a.g.inc[p + u] : 0;
                            sg[u] = live ? (key < 0 ? -1.0f : 1.0f) : 0.f;
                            v[u] = *reinterpret_cast<const float4*>(hg + (size_t)(key & 0x7fffffff) * GH + 4 * c4);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (sg[u] != 0.f) { x0.x += sg[u] * v[u].x; x0.y += sg[u] * v[u].y; x0.z += sg[u] * v[u].z; x0.w += sg[u] * v[u].w; }
                    }
                    if (a.es) *reinterpret_cast<float4*>(a.es + ((size_t)gi * N + d) * H + 4 * c4) = x0;
                }
            }
            const float xv[4] = {x0.x, x0.y, x0.z, x0.w}, yv[4] = {x1.x, x1.y, x1.z, x1.w}, hv[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * c4 + j;
                sX[row * LDX + perm_pos(k, IN)] = xv[j];
                if (is_e && IN_E == 2 * H) sX[row * LDX + perm_pos(H + k, IN)] = yv[j];
                sH[row * LDH + perm_pos(k, H)] = hv[j];
                sHp[row * (H + 1) + k] = hv[j];
            }
        }
        __syncthreads();
        for (int cs = wave; cs < H / 16; cs += 4) {
            f32x4 gi_[3], gh_[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { gi_[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; gh_[q] = gi_[q]; }
            if (is_e) cell_gemm<IN_E, H>(sX, LDX, bI_e, lane, gi_);
            else cell_gemm<H, H>(sX, LDX, bI_n, lane, gi_);
            cell_gemm<H, H>(sH, LDH, bH, lane, gh_);
            const int col = 16 * cs + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                if (r0 + row >= R) continue;
                const float rr = sigm(gi_[0][r] + bir + gh_[0][r] + bhr);
                const float zz = sigm(gi_[1][r] + biz + gh_[1][r] + bhz);
                const float hn = gh_[2][r] + bhn;
                const float nn = tanh_(gi_[2][r] + bin_ + rr * hn);
                const float hp = sHp[row * (H + 1) + col];
                const float hv = (1.0f - zz) * nn + zz * hp;
                const int grow = sRow[row];
                a.h_out[(size_t)grow * GH + gi * H + col] = hv;
                if (a.gates) {
                    float* gp = a.gates + (size_t)gi * 4 * N * H + (size_t)grow * H + col;
                    const size_t plane = (size_t)N * H;
                    gp[0] = rr; gp[plane] = zz; gp[2 * plane] = nn; gp[3 * plane] = hn;
                }
                lsum[r] += wh * hv;
            }
        }
        __syncthreads();
    }
    // ---- output head (models/track_mpnn.py:72-75): sum over the 16 columns a lane group holds, then over the waves
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = lsum[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if ((lane & 15) == 0) sLog[wave][4 * (lane >> 4) + r] = v;
    }
    __syncthreads();
    if (tid < TR && r0 + tid < R) {
        const float bias = is_e ? a.P.b_edge[0] : a.P.b_node[0];
        const float y = ((sLog[0][tid] + sLog[1][tid]) + (sLog[2][tid] + sLog[3][tid])) + bias;
        const int grow = sRow[tid];
        a.logits[grow] = y;
        a.scores[grow] = (!is_e && a.det_score_one) ? 1.0f : sigm(y);
    }
}

// ------------------------------------------------------------------------------------------------------------
// the iteration, backward: row tiles
// ------------------------------------------------------------------------------------------------------------
// slab of one block and group (floats): [dW_ih 3H x IN][dW_hh 3H x H][db_ih 3H][db_hh 3H][dw_head H][db_head 1]
__host__ __device__ inline size_t slab_floats(int H, int IN_e) { return (size_t)3 * H * (IN_e + H) + 6 * H + H + 4; }

// how the persistent blocks of k_small_iter_bwd split between edge tiles and det tiles (also used by the finish kernel)
__device__ __forceinline__ int bwd_edge_blocks(int nEt, int nDt, int nb) {
    if (nEt == 0) return 0;
    if (nDt == 0) return nb;
    int d = (int)(((long)nb * nDt + (nEt + nDt) / 2) / (nEt + nDt));
    if (d < 1) d = 1;
    if (d > nb - 1) d = nb - 1;
    return nb - d;
}

struct IterBwdArgs {
    tmpnn_mp_params P;
    tmpnn_dgraph g;
    const float* prep;
    const float* h;            // h_cat
    const float* scores;
    const float* gates; const float* es;
    const float* d_scores; const float* d_logits; const float* d_hout;
    int st_ds, st_dl;          // element strides of d_scores / d_logits (0 = one broadcast value, e.g. the gradient of a sum)
    float* d_h;                // [N][G*H] written (dh z + d_gh W_hh)
    float* d_msg;              // [N][G*IN_e] written: d_x of every row (edge rows: IN_e columns, det rows: H)
    float* slabs;              // [gridDim.x][G][slab_floats]
};

template <int H, int IN_E>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_small_iter_bwd(IterBwdArgs a) {
    const int E = a.g.meta[0], Dn = a.g.meta[1];
    const int nEt = (E + TR - 1) / TR, nDt = (Dn + TR - 1) / TR;
    const int nb = gridDim.x;
    const int nbE = bwd_edge_blocks(nEt, nDt, nb);
    const bool is_e = (int)blockIdx.x < nbE;
    const int my0 = is_e ? blockIdx.x : blockIdx.x - nbE;
    const int mystride = is_e ? nbE : nb - nbE;
    const int ntiles = is_e ? nEt : nDt;
    const int R = is_e ? E : Dn;
    const int G = a.P.G, GH = G * H, N = a.g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int LDX = IN_E + 4, LDH = H + 4, LDG = 3 * H + 4;
    constexpr int NJT = (3 * H / 16 + 3) / 4;          // j-tiles (16 rows of dW) per wave
    constexpr int NCT_I = IN_E / 16, NCT_H = H / 16;   // column tiles of dW_ih (edge cell) / dW_hh
    constexpr int NDT = (NCT_I + NCT_H + 3) / 4;       // data-gradient column tiles per wave
    constexpr int O_X = 0, O_H = O_X + TR * LDX, O_GI = O_H + TR * LDH, O_GH = O_GI + TR * LDG, O_DHZ = O_GH + TR * LDG,
                  O_END = O_DHZ + TR * (H + 1);
    constexpr int RED = TR * (5 * H + 1);              // the end-of-group column sums reuse the tile images
    __shared__ __attribute__((aligned(16))) float smem[O_END > RED ? O_END : RED];
    float* sX = smem + O_X;
    float* sH = smem + O_H;
    float* sGi = smem + O_GI;        // [dr | dz | dn ]  permuted over 3H
    float* sGh = smem + O_GH;        // [dr | dz | dnr]
    float* sDhz = smem + O_DHZ;
    __shared__ int sRow[TR], sS[TR], sD[TR];
    __shared__ float sDy[TR];
    const PrepLayout PL = prep_layout(H, IN_E);
    const int IN = is_e ? IN_E : H;
    const int nct_i = is_e ? NCT_I : NCT_H;
    const float* w_head = is_e ? a.P.w_edge : a.P.w_node;
    const size_t SLF = slab_floats(H, IN_E);
    const size_t plane = (size_t)N * H;
    for (int gi = 0; gi < G; ++gi) {
        f32x4 wI[NJT][NCT_I], wH[NJT][NCT_H];
#pragma unroll
        for (int j = 0; j < NJT; ++j) {
#pragma unroll
            for (int t = 0; t < NCT_I; ++t) wI[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NCT_H; ++t) wH[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float4 cb[4], chw;                               // column sums of dr, dz, dn, dnr ; of dy * h'
        float cdy = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) cb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        chw = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* hg = a.h + gi * H;
        const float* img = a.prep + (size_t)gi * PL.per_group;
        const float* img_ih = img + (is_e ? PL.e_ih_b : PL.n_ih_b);
        const float* img_hh = img + (is_e ? PL.e_hh_b : PL.n_hh_b);
        // the weight operands of this wave's data-gradient column tiles stay in registers for the whole tile loop (the
        // block has one wave per SIMD and 512 registers each: loading them per tile cost an L2 round trip per 16-k step)
        float4 wB[NDT][3 * H / 16];
#pragma unroll
        for (int cq = 0; cq < NDT; ++cq) {
            const int ct = wave + 4 * cq;
            const bool ih = ct < nct_i;
            const int c = ih ? ct : ct - nct_i;
            const float* im = (ih ? img_ih : img_hh) + (size_t)c * (3 * H / 16) * 256;
#pragma unroll
            for (int i = 0; i < 3 * H / 16; ++i)
                wB[cq][i] = (ct < nct_i + NCT_H && my0 < ntiles) ? *reinterpret_cast<const float4*>(im + ((size_t)i * 64 + lane) * 4)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int tile = my0; tile < ntiles; tile += mystride) {
            const int r0 = tile * TR;
            __syncthreads();                               // the previous tile's LDS reads are done
            if (tid < TR) {
                const int q = r0 + tid;
                const bool ok = q < R;
                const int qc = ok ? q : R - 1;
                const int grow = is_e ? a.g.edge_row[qc] : a.g.det_row[qc];
                sRow[tid] = grow;
                sS[tid] = is_e ? a.g.src[qc] : 0;
                sD[tid] = is_e ? a.g.dst[qc] : 0;
                float dy = 0.f;
                if (ok) {
                    if (a.d_logits) dy += a.d_logits[(size_t)grow * a.st_dl];
                    if (a.d_scores) { const float s = a.scores[grow]; dy += a.d_scores[(size_t)grow * a.st_ds] * s * (1.0f - s); }
                }
                sDy[tid] = dy;
            }
            __syncthreads();
            if (tid < TR * (H / 4)) {
                const int row = tid / (H / 4), c4 = tid % (H / 4);
                const bool ok = r0 + row < R;
                const int grow = sRow[row];
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 dh = z4, gr = z4, gz = z4, gn = z4, ghn = z4, hp = z4, x0 = z4, x1 = z4;
                if (ok) {
                    const float* gp = a.gates + (size_t)gi * 4 * plane + (size_t)grow * H + 4 * c4;
                    gr = *reinterpret_cast<const float4*>(gp);
                    gz = *reinterpret_cast<const float4*>(gp + plane);
                    gn = *reinterpret_cast<const float4*>(gp + 2 * plane);
                    ghn = *reinterpret_cast<const float4*>(gp + 3 * plane);
                    hp = *reinterpret_cast<const float4*>(hg + (size_t)grow * GH + 4 * c4);
                    if (a.d_hout) dh = *reinterpret_cast<const float4*>(a.d_hout + (size_t)grow * GH + gi * H + 4 * c4);
                    const float dy = sDy[row];
                    const float4 wh = *reinterpret_cast<const float4*>(w_head + gi * H + 4 * c4);
                    dh.x += dy * wh.x; dh.y += dy * wh.y; dh.z += dy * wh.z; dh.w += dy * wh.w;
                    if (is_e) {
                        x0 = *reinterpret_cast<const float4*>(hg + (size_t)sS[row] * GH + 4 * c4);
                        x1 = *reinterpret_cast<const float4*>(hg + (size_t)sD[row] * GH + 4 * c4);
                        if (IN_E == H) { x0.x -= x1.x; x0.y -= x1.y; x0.z -= x1.z; x0.w -= x1.w; }
                    } else {
                        x0 = *reinterpret_cast<const float4*>(a.es + ((size_t)gi * N + (r0 + row)) * H + 4 * c4);
                    }
                    // head gradient needs h' = (1 - z) n + z h_prev (recomputed, not re-read)
                    chw.x += dy * ((1.0f - gz.x) * gn.x + gz.x * hp.x);
                    chw.y += dy * ((1.0f - gz.y) * gn.y + gz.y * hp.y);
                    chw.z += dy * ((1.0f - gz.z) * gn.z + gz.z * hp.z);
                    chw.w += dy * ((1.0f - gz.w) * gn.w + gz.w * hp.w);
                    if (c4 == 0) cdy += dy;
                }
                const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, rv[4] = {gr.x, gr.y, gr.z, gr.w}, zv[4] = {gz.x, gz.y, gz.z, gz.w};
                const float nv[4] = {gn.x, gn.y, gn.z, gn.w}, hnv[4] = {ghn.x, ghn.y, ghn.z, ghn.w}, hv[4] = {hp.x, hp.y, hp.z, hp.w};
                const float xv[4] = {x0.x, x0.y, x0.z, x0.w}, yv[4] = {x1.x, x1.y, x1.z, x1.w};
                float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = dhv[j] * (1.0f - zv[j]) * (1.0f - nv[j] * nv[j]);
                    dn[j] = t;
                    dnr[j] = t * rv[j];
                    dr[j] = t * hnv[j] * rv[j] * (1.0f - rv[j]);
                    dz[j] = dhv[j] * (hv[j] - nv[j]) * zv[j] * (1.0f - zv[j]);
                    const int k = 4 * c4 + j;
                    sGi[row * LDG + perm_pos(k, 3 * H)] = dr[j];
                    sGi[row * LDG + perm_pos(H + k, 3 * H)] = dz[j];
                    sGi[row * LDG + perm_pos(2 * H + k, 3 * H)] = dn[j];
                    sGh[row * LDG + perm_pos(k, 3 * H)] = dr[j];
                    sGh[row * LDG + perm_pos(H + k, 3 * H)] = dz[j];
                    sGh[row * LDG + perm_pos(2 * H + k, 3 * H)] = dnr[j];
                    sX[row * LDX + perm_pos(k, IN)] = xv[j];
                    if (is_e && IN_E == 2 * H) sX[row * LDX + perm_pos(H + k, IN)] = yv[j];
                    sH[row * LDH + perm_pos(k, H)] = hv[j];
                    sDhz[row * (H + 1) + k] = dhv[j] * zv[j];
                }
                cb[0].x += dr[0]; cb[0].y += dr[1]; cb[0].z += dr[2]; cb[0].w += dr[3];
                cb[1].x += dz[0]; cb[1].y += dz[1]; cb[1].z += dz[2]; cb[1].w += dz[3];
                cb[2].x += dn[0]; cb[2].y += dn[1]; cb[2].z += dn[2]; cb[2].w += dn[3];
                cb[3].x += dnr[0]; cb[3].y += dnr[1]; cb[3].z += dnr[2]; cb[3].w += dnr[3];
            }
            __syncthreads();
            // ---- data gradients: d_x = d_gi W_ih (IN columns), d_hprev = dh z + d_gh W_hh (H columns)
            {
                const int row = lane & 15, kh = lane >> 4;
#pragma unroll
                for (int cq = 0; cq < NDT; ++cq) {
                    const int ct = wave + 4 * cq;
                    if (ct >= nct_i + NCT_H) continue;
                    const bool ih = ct < nct_i;
                    const int c = ih ? ct : ct - nct_i;
                    const float* sA = (ih ? sGi : sGh) + row * LDG + kh * (3 * H / 4);
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 3 * H / 16; ++i) {
                        const float4 av = *reinterpret_cast<const float4*>(sA + 4 * i);
                        const float4 bv = wB[cq][i];
                        acc = mfma16(av.x, bv.x, acc);
                        acc = mfma16(av.y, bv.y, acc);
                        acc = mfma16(av.z, bv.z, acc);
                        acc = mfma16(av.w, bv.w, acc);
                    }
                    const int col = 16 * c + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rw = 4 * (lane >> 4) + r;
                        if (r0 + rw >= R) continue;
                        const int grow = sRow[rw];
                        if (ih) a.d_msg[(size_t)grow * (G * IN_E) + gi * IN_E + col] = acc[r];
                        else a.d_h[(size_t)grow * GH + gi * H + col] = acc[r] + sDhz[rw * (H + 1) + col];
                    }
                }
            }
            // ---- weight gradients: dW[j][c] += sum_rows d_g[row][j] * [x | h][row][c]   (K = the tile's 16 rows)
#pragma unroll
            for (int jq = 0; jq < NJT; ++jq) {
                const int jt = wave + 4 * jq;
                if (jt < 3 * H / 16) {
                    const int j = 16 * jt + (lane & 15);
                    float ai[4], ah[4];
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) {
                        const int rw = 4 * kq + (lane >> 4);
                        ai[kq] = sGi[rw * LDG + perm_pos(j, 3 * H)];
                        ah[kq] = sGh[rw * LDG + perm_pos(j, 3 * H)];
                    }
#pragma unroll
                    for (int t = 0; t < NCT_I; ++t) {
                        if (t < nct_i) {
                            const int c = 16 * t + (lane & 15);
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq)
                                wI[jq][t] = mfma16(ai[kq], sX[(4 * kq + (lane >> 4)) * LDX + perm_pos(c, IN)], wI[jq][t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NCT_H; ++t) {
                        const int c = 16 * t + (lane & 15);
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq)
                            wH[jq][t] = mfma16(ah[kq], sH[(4 * kq + (lane >> 4)) * LDH + perm_pos(c, H)], wH[jq][t]);
                    }
                }
            }
        }
        if (my0 >= ntiles) continue;                       // no tile, no slab (the reducer skips this block too)
        // ---- this block's slab for group gi
        float* sl = a.slabs + ((size_t)blockIdx.x * G + gi) * SLF;
        float* sl_hh = sl + (size_t)3 * H * IN;
#pragma unroll
        for (int jq = 0; jq < NJT; ++jq) {
            const int jt = wave + 4 * jq;
            if (jt < 3 * H / 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * (lane >> 4) + r;
#pragma unroll
                    for (int t = 0; t < NCT_I; ++t)
                        if (t < nct_i) sl[(size_t)j * IN + 16 * t + (lane & 15)] = wI[jq][t][r];
#pragma unroll
                    for (int t = 0; t < NCT_H; ++t) sl_hh[(size_t)j * H + 16 * t + (lane & 15)] = wH[jq][t][r];
                }
            }
        }
        // column sums: 16 threads (one per tile row) hold partial sums of the same 4 columns -> fixed-order sum in LDS
        __syncthreads();
        float* red = smem;                                 // [TR][5H + 1]
        if (tid < TR * (H / 4)) {
            const int row = tid / (H / 4), c4 = tid % (H / 4);
            float* rr = red + row * (5 * H + 1);
            rr[4 * c4 + 0] = cb[0].x; rr[4 * c4 + 1] = cb[0].y; rr[4 * c4 + 2] = cb[0].z; rr[4 * c4 + 3] = cb[0].w;
            rr[H + 4 * c4 + 0] = cb[1].x; rr[H + 4 * c4 + 1] = cb[1].y; rr[H + 4 * c4 + 2] = cb[1].z; rr[H + 4 * c4 + 3] = cb[1].w;
            rr[2 * H + 4 * c4 + 0] = cb[2].x; rr[2 * H + 4 * c4 + 1] = cb[2].y; rr[2 * H + 4 * c4 + 2] = cb[2].z; rr[2 * H + 4 * c4 + 3] = cb[2].w;
            rr[3 * H + 4 * c4 + 0] = cb[3].x; rr[3 * H + 4 * c4 + 1] = cb[3].y; rr[3 * H + 4 * c4 + 2] = cb[3].z; rr[3 * H + 4 * c4 + 3] = cb[3].w;
            rr[4 * H + 4 * c4 + 0] = chw.x; rr[4 * H + 4 * c4 + 1] = chw.y; rr[4 * H + 4 * c4 + 2] = chw.z; rr[4 * H + 4 * c4 + 3] = chw.w;
            if (c4 == 0) rr[5 * H] = cdy;
        }
        __syncthreads();
        float* sb = sl + (size_t)3 * H * (IN + H);
        for (int q = tid; q < 5 * H + 1; q += 256) {
            float s = 0.f;
#pragma unroll
            for (int row = 0; row < TR; ++row) s += red[row * (5 * H + 1) + q];
            // q: [0,H) dr | [H,2H) dz | [2H,3H) dn | [3H,4H) dnr | [4H,5H) dy*h' | 5H: dy
            if (q < 3 * H) sb[q] = s;                                   // db_ih = [dr | dz | dn]
            if (q < 2 * H) sb[3 * H + q] = s;                           // db_hh = [dr | dz | dnr]
            else if (q >= 3 * H && q < 4 * H) sb[3 * H + q - H] = s;
            else if (q >= 4 * H) sb[6 * H + (q - 4 * H)] = s;           // dw_head [H], then db_head
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, second launch: adjoints of the two aggregations + slab reduction into the gradient buffers
// ------------------------------------------------------------------------------------------------------------
struct FinishArgs {
    tmpnn_mp_params P;
    tmpnn_mp_params grads;
    tmpnn_dgraph g;
    const float* d_msg;
    float* d_h;
    const float* slabs;
    int nb_bwd;                // grid of k_small_iter_bwd
    int row_blocks;            // blocks [0, row_blocks): adjoints on the carried rows
    int red_blocks;            // then red_blocks blocks reduce the slabs; the remaining G blocks (if n_new > 0) run
                               // the input-transform backward of one feature group each
    int n_new, training;
    const float* x; int ld_x;
    const float* ysave; const float* mean; const float* rstd;
    const int* newdet;         // [G][n_new + 1]
    float* d_x;                // [n_new][F_total] or NULL
    float* scratch;            // [G][2][n_new][H]: used when the new det rows do not fit the LDS
    int skip_row_f;            // != 0: the adjoint of the edge -> node aggregation has been added to d_h already (tmpnn_att_bwd
                               // between the two launches of tmpnn_mp_iter_bwd_parts)
};

// gradient of the iteration's input state at row `row`, group gi, columns 4 c4 ..: what k_small_iter_bwd wrote
// (dh z + d_gh W_hh) plus the adjoint of the aggregation the row took part in
template <int H, int IN_E>
__device__ __forceinline__ float4 adjoint_at(const FinishArgs& a, int row, int gi, int c4) {
    const int G = a.P.G;
    const float* dm = a.d_msg + gi * IN_E + 4 * c4;
    const size_t ldm = (size_t)G * IN_E;
    const int p = a.g.pos[row];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.g.is_edge[row]) {
        // adjoint of row F: d h[e] += d_es[src(e)] - d_es[dst(e)]   (d_es = the node cell's d_x, stored at det rows)
        if (!a.skip_row_f) {
            const float4 u = *reinterpret_cast<const float4*>(dm + (size_t)a.g.src[p] * ldm);
            const float4 v = *reinterpret_cast<const float4*>(dm + (size_t)a.g.dst[p] * ldm);
            acc = make_float4(u.x - v.x, u.y - v.y, u.z - v.z, u.w - v.w);
        }
    } else {
        // adjoint of row E: d h[d] += sum_{e: src=d} d_x[e][0:H] -/+ sum_{e: dst=d} d_x[e][0:H | H:2H]; CSR order
        // (sixteen incidences, then their sixteen rows, each batch as straight-line loads: four at a time with the incidence
        //  behind a select was a dependent pair of round trips per four rows, ~26 in a row for a det of a KITTI-sized window)
        const int p0 = a.g.rowptr[p], p1 = a.g.rowptr[p + 1];
        constexpr int UA = 16;
        for (int q = p0; q < p1; q += UA) {
            int key[UA];
            float4 v[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) key[u] = a.g.inc[min(q + u, p1 - 1)];
#pragma unroll
            for (int u = 0; u < UA; ++u)
                v[u] = *reinterpret_cast<const float4*>(dm + (size_t)(key[u] & 0x7fffffff) * ldm + ((key[u] < 0 && IN_E == 2 * H) ? H : 0));
#pragma unroll
            for (int u = 0; u < UA; ++u)
                if (q + u < p1) {
                    const float w = (key[u] < 0 && IN_E == H) ? -1.0f : 1.0f;
                    acc.x += w * v[u].x; acc.y += w * v[u].y; acc.z += w * v[u].z; acc.w += w * v[u].w;
                }
        }
    }
    const float4 t = *reinterpret_cast<const float4*>(a.d_h + (size_t)row * (G * H) + gi * H + 4 * c4);
    return make_float4(t.x + acc.x, t.y + acc.y, t.z + acc.z, t.w + acc.w);
}

// input-transform backward of feature group gi by ONE block (models/track_mpnn.py:45-52,59-61 reversed).
// B0 / B1: [nd][ldb] work arrays (LDS when the new det rows fit, else global scratch).
static constexpr int BN_LDS_ROWS_C = 120;
template <int H, int IN_E>
__device__ void bn_bwd_block(const FinishArgs& a, int gi, float* B0, float* B1, int ldb, float* stage /* LDS, BN_STAGE floats */) {
    const int N = a.g.N, n = a.n_new, N_old = N - n;
    const int F = a.P.F[gi], Ft = a.P.F_total;
    int f0 = 0;
    for (int q = 0; q < gi; ++q) f0 += a.P.F[q];
    const int tid = threadIdx.x;
    const int* newdet = a.newdet + (size_t)gi * (n + 1);
    const int nd = newdet[n];
    const float* ysave = a.ysave + (size_t)gi * (n > 0 ? n : 1) * H;
    __shared__ float s_m1[H], s_m2[H], s_dyz[H], s_mean[H], s_rstd[H], s_gam[H];
    const float cntf = (float)n, nz = (float)(n - nd);
    const int c = tid % H, sub = tid / H;
    constexpr int NSUB = 256 / H;
    if (nd == 0) {
        if (a.d_x) for (int idx = tid; idx < n * F; idx += 256) a.d_x[(size_t)(idx / F) * Ft + f0 + idx % F] = 0.f;
        return;
    }
    if (tid < H) {
        s_mean[tid] = a.mean[(size_t)gi * H + tid];
        s_rstd[tid] = a.rstd[(size_t)gi * H + tid];
        s_gam[tid] = a.P.gamma[gi][tid];
    }
    // the small operands every later step re-reads -- W2 [H][H], W1 [H][F], the det rows' features [nd][F] -- are
    // copied into the LDS in ONE burst of independent loads, behind which the dependent index chain of step 1 hides;
    // each later step then costs LDS latencies instead of an L2 round trip (wide feature groups keep reading L2)
    const bool staged = F <= 16 && nd <= BN_LDS_ROWS_C;
    const float* W2 = a.P.w2[gi];
    const float* W1 = a.P.w1[gi];
    const float* xs = a.x + f0;
    int ldxs = a.ld_x;
    float* sW2 = stage;
    for (int idx = tid; idx < H * H; idx += 256) sW2[idx] = W2[idx];
    W2 = sW2;
    if (staged) {
        float* sW1 = stage + H * H;
        float* sx = sW1 + H * 16;
        for (int idx = tid; idx < H * F; idx += 256) sW1[idx] = W1[idx];
        for (int idx = tid; idx < nd * F; idx += 256) sx[idx] = a.x[(size_t)newdet[idx / F] * a.ld_x + f0 + idx % F];
        W1 = sW1;
        xs = sx;
        ldxs = F;
    }
    __syncthreads();
    // 1. d_out (complete gradient of the new det rows' state) and the recomputed activations a
    {
        constexpr int LPR = H / 4, RPP = 256 / LPR;
        const int c4 = tid % LPR, slot = tid / LPR;
        for (int i = slot; i < nd; i += RPP) {
            const float4 v = adjoint_at<H, IN_E>(a, N_old + newdet[i], gi, c4);
            float* o = B0 + (size_t)i * ldb + 4 * c4;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
        const float bet = a.P.beta[gi][c];
        for (int i = sub; i < nd; i += NSUB) {
            const float yh = (ysave[(size_t)i * H + c] - s_mean[c]) * s_rstd[c];
            B1[(size_t)i * ldb + c] = fmaxf(yh * s_gam[c] + bet, 0.f);
        }
    }
    __syncthreads();
    // 2. dW2[c][k] += sum_i d_out[i][c] a[i][k] ; db2[c] += sum_i d_out[i][c]
    {
        float* dW2 = a.grads.w2[gi];
        // (all partial sums first, then the read-modify-writes back to back: interleaved, every += waited for its own
        //  L2 round trip -- 16 of them in a row were half of this kernel)
        constexpr int NQ = H * H / 256;
        float sacc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + q * 256;
            const int cc = idx / H, k = idx % H;
            float s = 0.f;
            for (int i = 0; i < nd; ++i) s = fmaf(B0[(size_t)i * ldb + cc], B1[(size_t)i * ldb + k], s);
            sacc[q] = s;
        }
        float old[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) old[q] = dW2[tid + q * 256];
#pragma unroll
        for (int q = 0; q < NQ; ++q) dW2[tid + q * 256] = old[q] + sacc[q];
        if (tid < H) {
            float s = 0.f;
            for (int i = 0; i < nd; ++i) s += B0[(size_t)i * ldb + tid];
            a.grads.b2[gi][tid] += s;
        }
    }
    __syncthreads();
    // 3. d_a = d_out W2 ; d_pre = d_a [a > 0]  (overwrites a in B1)
    {
        for (int i = sub; i < nd; i += NSUB) {
            float s = 0.f;
#pragma unroll 8
            for (int cc = 0; cc < H; ++cc) s = fmaf(B0[(size_t)i * ldb + cc], W2[cc * H + c], s);
            const float av = B1[(size_t)i * ldb + c];
            B1[(size_t)i * ldb + c] = av > 0.f ? s : 0.f;
        }
    }
    __syncthreads();
    // 4. dgamma, dbeta, the two batch means of the BatchNorm backward
    if (tid < H) {
        float sg = 0.f, sb = 0.f;
        for (int i = 0; i < nd; ++i) {
            const float dp = B1[(size_t)i * ldb + tid];
            const float yh = (ysave[(size_t)i * H + tid] - s_mean[tid]) * s_rstd[tid];
            sg += dp * yh;
            sb += dp;
        }
        a.grads.gamma[gi][tid] += sg;
        a.grads.beta[gi][tid] += sb;
        s_m1[tid] = a.training ? s_gam[tid] * sb / cntf : 0.f;      // mean over ALL n rows of d_yhat (zero rows add 0)
        s_m2[tid] = a.training ? s_gam[tid] * sg / cntf : 0.f;      // mean of d_yhat * yhat
    }
    __syncthreads();
    // 5. d_y1 of the det rows into B0 ; the zero rows' d_y1 (one value per column)
    for (int i = sub; i < nd; i += NSUB) {
        const float yh = (ysave[(size_t)i * H + c] - s_mean[c]) * s_rstd[c];
        const float dyh = B1[(size_t)i * ldb + c] * s_gam[c];
        B0[(size_t)i * ldb + c] = s_rstd[c] * (dyh - s_m1[c] - yh * s_m2[c]);
    }
    if (tid < H) {
        const float yz = (a.P.b1[gi][tid] - s_mean[tid]) * s_rstd[tid];
        s_dyz[tid] = a.training ? s_rstd[tid] * (-s_m1[tid] - yz * s_m2[tid]) : 0.f;
    }
    __syncthreads();
    // 6. dW1[k][f] += sum_i d_y1[i][k] x[i][f] ; db1[k] += sum_i d_y1[i][k] + nz * d_y1_zero[k]
    {
        float* dW1 = a.grads.w1[gi];
        for (int base = 0; base < H * F; base += 256 * 8) {
            float sacc[8], old[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = base + tid + q * 256;
                float s = 0.f;
                if (idx < H * F) {
                    const int k = idx / F, f = idx % F;
                    for (int i = 0; i < nd; ++i)
                        s = fmaf(B0[(size_t)i * ldb + k], staged ? xs[i * ldxs + f] : a.x[(size_t)newdet[i] * a.ld_x + f0 + f], s);
                }
                sacc[q] = s;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int idx = base + tid + q * 256; old[q] = idx < H * F ? dW1[idx] : 0.f; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int idx = base + tid + q * 256; if (idx < H * F) dW1[idx] = old[q] + sacc[q]; }
        }
        if (tid < H) {
            float s = nz * s_dyz[tid];
            for (int i = 0; i < nd; ++i) s += B0[(size_t)i * ldb + tid];
            a.grads.b1[gi][tid] += s;
        }
    }
    // 7. d_x: det rows  d_y1 W1 ; zero rows  d_y1_zero W1 (what reaches them through the batch statistics)
    if (a.d_x) {
        for (int idx = tid; idx < n * F; idx += 256) {
            const int i = idx / F, f = idx % F;
            if (!a.g.is_edge[N_old + i]) continue;
            float s = 0.f;
            for (int k = 0; k < H; ++k) s = fmaf(s_dyz[k], W1[k * F + f], s);
            a.d_x[(size_t)i * Ft + f0 + f] = s;
        }
        for (int idx = tid; idx < nd * F; idx += 256) {
            const int i = idx / F, f = idx % F;
            float s = 0.f;
            for (int k = 0; k < H; ++k) s = fmaf(B0[(size_t)i * ldb + k], W1[k * F + f], s);
            a.d_x[(size_t)newdet[i] * Ft + f0 + f] = s;
        }
    }
}

static constexpr int BN_LDS_ROWS = 120;      // new det rows whose two work arrays fit the LDS (2 x 120 x 65 floats)
// LDS staging area of bn_bwd_block: W2 [H][H] + W1 [H][16] + features [120][16]
__host__ __device__ constexpr int bn_stage_floats(int H) { return H * H + H * 16 + BN_LDS_ROWS * 16; }

template <int H, int IN_E>
__global__ __launch_bounds__(256) void k_small_bwd_finish(FinishArgs a) {
    const int E = a.g.meta[0], Dn = a.g.meta[1];
    const int G = a.P.G, GH = G * H, N = a.g.N;
    const int tid = threadIdx.x;
    constexpr int LPR = H / 4, RPB = 256 / LPR;
    if ((int)blockIdx.x < a.row_blocks) {
        // ---- carried rows: d_h += adjoint (the new rows' gradient is only needed by the input transform below)
        if (E + Dn == 0) return;
        const int N_old = N - a.n_new;
        const int c4 = tid % LPR, slot = tid / LPR;
        for (int r = blockIdx.x * RPB + slot; r < N_old; r += a.row_blocks * RPB)
            for (int gi = 0; gi < G; ++gi) {
                const float4 v = adjoint_at<H, IN_E>(a, r, gi, c4);
                *reinterpret_cast<float4*>(a.d_h + (size_t)r * GH + gi * H + 4 * c4) = v;
            }
        return;
    }
    if ((int)blockIdx.x >= a.row_blocks + a.red_blocks) {
        // ---- input-transform backward, one block per feature group
        if (E + Dn == 0) return;
        const int gi = blockIdx.x - a.row_blocks - a.red_blocks;
        extern __shared__ float dyn[];
        const int nd = a.newdet[(size_t)gi * (a.n_new + 1) + a.n_new];
        float* stage = dyn + 2 * BN_LDS_ROWS * (H + 1);
        if (nd <= BN_LDS_ROWS) bn_bwd_block<H, IN_E>(a, gi, dyn, dyn + BN_LDS_ROWS * (H + 1), H + 1, stage);
        else {
            float* B0 = a.scratch + (size_t)gi * 2 * a.n_new * H;
            bn_bwd_block<H, IN_E>(a, gi, B0, B0 + (size_t)a.n_new * H, H, stage);
        }
        return;
    }
    // ---- slab reduction (fixed order over the blocks of each cell that had tiles)
    const int nEt = (E + TR - 1) / TR, nDt = (Dn + TR - 1) / TR;
    const int nbE = bwd_edge_blocks(nEt, nDt, a.nb_bwd);
    const int usedE = min(nbE, nEt), usedD = min(a.nb_bwd - nbE, nDt);
    const size_t SLF = slab_floats(H, IN_E);
    const size_t n_e = (size_t)3 * H * (IN_E + H) + 7 * H + 1, n_n = (size_t)3 * H * (2 * H) + 7 * H + 1;
    const size_t per_g = n_e + n_n;
    const size_t total = per_g * G;
    const int rb = a.red_blocks;
    const int bid = blockIdx.x - a.row_blocks;
    constexpr int UR = 4;
    for (size_t idx0 = (size_t)bid * 256 + tid; idx0 < total; idx0 += (size_t)rb * 256 * UR) {
        float* dst[UR];
        float old[UR], sum[UR];
        const float* sp[UR];
        int nb0[UR], nb1[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const size_t idx = idx0 + (size_t)u * rb * 256;
            dst[u] = nullptr; sp[u] = a.slabs; nb0[u] = nb1[u] = 0;
            if (idx >= total) continue;
            const int gi = (int)(idx / per_g);
            size_t q = idx % per_g;
            const bool edge = q < n_e;
            if (!edge) q -= n_e;
            const int IN = edge ? IN_E : H;
            nb0[u] = edge ? 0 : nbE; nb1[u] = edge ? usedE : nbE + usedD;
            sp[u] = a.slabs + (size_t)gi * SLF + q;
            const size_t nih = (size_t)3 * H * IN, nhh = (size_t)3 * H * H;
            if (q < nih) dst[u] = (edge ? a.grads.e_wih[gi] : a.grads.n_wih[gi]) + q;
            else if (q < nih + nhh) dst[u] = (edge ? a.grads.e_whh[gi] : a.grads.n_whh[gi]) + (q - nih);
            else {
                const size_t k = q - nih - nhh;
                if (k < (size_t)3 * H) dst[u] = (edge ? a.grads.e_bih[gi] : a.grads.n_bih[gi]) + k;
                else if (k < (size_t)6 * H) dst[u] = (edge ? a.grads.e_bhh[gi] : a.grads.n_bhh[gi]) + (k - 3 * H);
                else if (k < (size_t)7 * H) dst[u] = (edge ? a.grads.w_edge : a.grads.w_node) + gi * H + (k - 6 * H);
                // k == 7H: head bias, one scalar per cell type, below
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) old[u] = dst[u] ? *dst[u] : 0.f;            // the old values travel with the slab loads
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            float s = 0.f;
            for (int b = nb0[u]; b < nb1[u]; ++b) s += sp[u][(size_t)b * G * SLF];
            sum[u] = s;
        }
#pragma unroll
        for (int u = 0; u < UR; ++u)
            if (dst[u]) *dst[u] = old[u] + sum[u];
    }
    // head biases: the same dy sums appear in every group's slab; take group 0
    if (bid == 0 && tid < 2) {
        const bool edge = tid == 0;
        const int IN = edge ? IN_E : H;
        const int b0 = edge ? 0 : nbE, b1 = edge ? usedE : nbE + usedD;
        const size_t q = (size_t)3 * H * (IN + H) + 7 * H;
        float s = 0.f;
        for (int b = b0; b < b1; ++b) s += a.slabs[((size_t)b * G + 0) * SLF + q];
        float* dst = edge ? a.grads.b_edge : a.grads.b_node;
        *dst += s;
    }
}

static int check_params(const tmpnn_mp_params* P, const char* what, bool grads) {
    TM_REQUIRE(P != nullptr, "%s: null parameter struct", what);
    TM_REQUIRE(P->G >= 1 && P->G <= 3 && (P->H == 32 || P->H == 64) && (P->IN_e == P->H || P->IN_e == 2 * P->H),
               "%s: G=%d H=%d IN_e=%d (need G <= 3, H in {32, 64}, IN_e = H or 2H)", what, P->G, P->H, P->IN_e);
    int ft = 0;
    for (int g = 0; g < P->G; ++g) {
        TM_REQUIRE(P->F[g] > 0, "%s: F[%d]=%d", what, g, P->F[g]);
        ft += P->F[g];
        TM_REQUIRE(P->w1[g] && P->b1[g] && P->gamma[g] && P->beta[g] && P->w2[g] && P->b2[g] && P->e_wih[g] &&
                       P->e_whh[g] && P->e_bih[g] && P->e_bhh[g] && P->n_wih[g] && P->n_whh[g] && P->n_bih[g] &&
                       P->n_bhh[g], "%s: null pointer in group %d", what, g);
        if (!grads) TM_REQUIRE(P->run_mean[g] && P->run_var[g], "%s: null BatchNorm buffers in group %d", what, g);
    }
    TM_REQUIRE(ft == P->F_total, "%s: F_total=%d but the groups sum to %d", what, P->F_total, ft);
    TM_REQUIRE(P->w_node && P->b_node && P->w_edge && P->b_edge, "%s: null output head", what);
    return TMPNN_OK;
}

static int check_dgraph(const tmpnn_dgraph* g, const char* what) {
    TM_REQUIRE(g != nullptr && g->meta && g->is_edge && g->pos && g->src && g->dst && g->edge_row && g->det_row &&
                   g->rowptr && g->inc, "%s: unbound graph", what);
    TM_REQUIRE(g->N >= 0 && g->N <= TMPNN_DG_BIG_ROWS && g->N <= g->cap, "%s: graph N=%d cap=%d (limit %d)", what,
               g->N, g->cap, TMPNN_DG_BIG_ROWS);
    return TMPNN_OK;
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

size_t tmpnn_mp_iter_prep_floats(int G, int H, int IN_e) {
    if (G <= 0 || H <= 0) return 0;
    return (size_t)G * prep_layout(H, IN_e).per_group;
}

int tmpnn_mp_iter_prepare(const tmpnn_mp_params* P, float* prep, tmpnn_stream stream) {
    int rc = check_params(P, "mp_iter_prepare", true);
    if (rc) return rc;
    TM_REQUIRE(prep != nullptr && aligned16(prep), "mp_iter_prepare: prep must be a 16-byte aligned buffer");
    const int H = P->H, IN_e = P->IN_e;
    const PrepLayout L = prep_layout(H, IN_e);
    PrepArgs a;
    a.H = H;
    a.njobs = 0;
    for (int g = 0; g < P->G; ++g) {
        const size_t base = (size_t)g * L.per_group;
        a.job[a.njobs++] = PrepJob{P->e_wih[g], IN_e, base + L.e_ih_f, base + L.e_ih_b};
        a.job[a.njobs++] = PrepJob{P->e_whh[g], H, base + L.e_hh_f, base + L.e_hh_b};
        a.job[a.njobs++] = PrepJob{P->n_wih[g], H, base + L.n_ih_f, base + L.n_ih_b};
        a.job[a.njobs++] = PrepJob{P->n_whh[g], H, base + L.n_hh_f, base + L.n_hh_b};
    }
    hipLaunchKernelGGL(k_small_prepare, dim3(ceil_div(3L * H * IN_e, 256), a.njobs), dim3(256), 0, as_stream(stream), a, prep);
    return check_launch("mp_iter_prepare");
}

size_t tmpnn_mp_iter_save_floats(int N, int n_new, int G, int H) {
    if (N < 0 || n_new < 0 || G <= 0 || H <= 0) return 0;
    // + the per-group lists of new det rows (ints, stored behind the floats)
    return save_layout(N, n_new, G, H).total + (size_t)G * (n_new + 1) + 4;
}

size_t tmpnn_mp_iter_save_es_offset(int N, int n_new, int G, int H) {
    if (N < 0 || n_new < 0 || G <= 0 || H <= 0) return 0;
    return save_layout(N, n_new, G, H).es;
}

int tmpnn_mp_iter_fwd(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new, const float* x,
                      int ld_x, float* h, int training, float* h_out, float* logits, float* scores, float* save,
                      size_t save_floats, tmpnn_stream stream) {
    return tmpnn_mp_iter_fwd_parts(P, prep, g, n_new, x, ld_x, h, training, h_out, logits, scores, save, save_floats, 0, stream);
}

int tmpnn_mp_iter_fwd_parts(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new, const float* x,
                            int ld_x, float* h, int training, float* h_out, float* logits, float* scores, float* save,
                            size_t save_floats, int parts, tmpnn_stream stream) {
    int rc = check_params(P, "mp_iter_fwd", false);
    if (rc) return rc;
    if ((rc = check_dgraph(g, "mp_iter_fwd"))) return rc;
    const int N = g->N, G = P->G, H = P->H;
    TM_REQUIRE(n_new >= 0 && n_new <= N, "mp_iter_fwd: n_new=%d N=%d", n_new, N);
    if (N == 0) return TMPNN_OK;
    TM_REQUIRE(prep && h && h_out && logits && scores, "mp_iter_fwd: null buffer");
    TM_REQUIRE(aligned16(prep) && aligned16(h) && aligned16(h_out), "mp_iter_fwd: h / h_out / prep must be 16-byte aligned");
    TM_REQUIRE(n_new == 0 || (parts & 1) || (x != nullptr && ld_x >= P->F_total), "mp_iter_fwd: x [%d][ld %d] for F_total=%d", n_new, ld_x,
               P->F_total);
    TM_REQUIRE(!(training && n_new == 1), "Expected more than 1 value per channel when training, got input size [1, %d]", H);
    const SaveLayout SL = save_layout(N, n_new, G, H);
    float* sv = save;
    if (save != nullptr) {
        if (save_floats < tmpnn_mp_iter_save_floats(N, n_new, G, H))
            return set_error(TMPNN_EWORKSPACE, "mp_iter_fwd: save buffer %zu < %zu floats", save_floats,
                             tmpnn_mp_iter_save_floats(N, n_new, G, H));
        TM_REQUIRE(aligned16(save), "mp_iter_fwd: save must be 16-byte aligned");
    }
    hipStream_t st = as_stream(stream);
    TM_REQUIRE((parts & ~15) == 0 && (!(parts & 4) || sv != nullptr), "mp_iter_fwd_parts: parts=%d (an aggregate given by the "
               "caller lives in the save buffer)", parts);
    TM_REQUIRE(!(parts & 8) || !training, "mp_iter_fwd_parts: parts bit 3 (det scores = 1) is an inference rule");
    if (n_new > 0 && !(parts & 1)) {
        TM_REQUIRE(sv != nullptr, "mp_iter_fwd: a call with new rows needs the save buffer (the input transform keeps its "
                                  "Lin1 outputs and statistics there), also in inference");
        BnFwdArgs b{*P, *g, n_new, training, x, ld_x, h, sv + SL.ysave, sv + SL.mean, sv + SL.rstd,
                    reinterpret_cast<int*>(sv + SL.total)};
        const size_t shm = sizeof(float) * 64 * (H + 1);
        if (H == 64) hipLaunchKernelGGL((k_small_bn_fwd<64>), dim3(G), dim3(256), shm, st, b);
        else hipLaunchKernelGGL((k_small_bn_fwd<32>), dim3(G), dim3(256), shm, st, b);
        if ((rc = check_launch("small_bn_fwd"))) return rc;
    }
    if (parts & 2) return TMPNN_OK;
    IterFwdArgs a{*P, *g, prep, h, h_out, logits, scores, sv ? sv + SL.gates : nullptr, sv ? sv + SL.es : nullptr,
                  (parts & 4) ? 1 : 0, (parts & 8) ? 1 : 0, N <= TMPNN_DG_MAX_ROWS ? 4 : TR};
    // (sized from N alone: E / Dn live on the device.  ceil(E / TR) + ceil(Dn / det_tile) <= N / det_tile + 2; surplus blocks exit)
    const int grid = a.det_tile < TR ? (N + a.det_tile - 1) / a.det_tile + 2 : (N + TR - 1) / TR + 2;
    if (H == 64 && P->IN_e == 64) hipLaunchKernelGGL((k_small_iter_fwd<64, 64>), dim3(grid), dim3(256), 0, st, a);
    else if (H == 64) hipLaunchKernelGGL((k_small_iter_fwd<64, 128>), dim3(grid), dim3(256), 0, st, a);
    else if (P->IN_e == 32) hipLaunchKernelGGL((k_small_iter_fwd<32, 32>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_small_iter_fwd<32, 64>), dim3(grid), dim3(256), 0, st, a);
    return check_launch("small_iter_fwd");
}

static int bwd_blocks(int N) {
    int nb = (N + TR - 1) / TR + 2;
    if (nb > SMALL_BWD_BLOCKS) nb = SMALL_BWD_BLOCKS;
    if (nb < 2) nb = 2;
    return nb;
}

size_t tmpnn_mp_iter_bwd_ws(int N, int n_new, int G, int H, int IN_e) {
    if (N < 0 || n_new < 0 || G <= 0 || H <= 0) return 0;
    const size_t d_msg = (size_t)N * G * IN_e;
    const size_t slabs = (size_t)bwd_blocks(N) * G * slab_floats(H, IN_e);
    const size_t bn = (size_t)G * 2 * n_new * H;
    return sizeof(float) * (d_msg + slabs + bn + 16);
}

int tmpnn_mp_iter_bwd(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new, const float* x,
                      int ld_x, const float* h, const float* h_out, const float* scores, const float* save,
                      int training, const float* d_scores, int st_dscores, const float* d_logits, int st_dlogits,
                      const float* d_hout, float* d_h, float* d_x, const tmpnn_mp_params* grads, void* ws, size_t ws_bytes, tmpnn_stream stream) {
    return tmpnn_mp_iter_bwd_parts(P, prep, g, n_new, x, ld_x, h, h_out, scores, save, training, d_scores, st_dscores, d_logits,
                                   st_dlogits, d_hout, d_h, d_x, grads, ws, ws_bytes, 0, stream);
}

int tmpnn_mp_iter_bwd_parts(const tmpnn_mp_params* P, const float* prep, const tmpnn_dgraph* g, int n_new, const float* x,
                            int ld_x, const float* h, const float* h_out, const float* scores, const float* save,
                            int training, const float* d_scores, int st_dscores, const float* d_logits, int st_dlogits,
                            const float* d_hout, float* d_h, float* d_x, const tmpnn_mp_params* grads, void* ws, size_t ws_bytes,
                            int parts, tmpnn_stream stream) {
    int rc = check_params(P, "mp_iter_bwd", false);
    if (rc) return rc;
    if ((rc = check_params(grads, "mp_iter_bwd (grads)", true))) return rc;
    if ((rc = check_dgraph(g, "mp_iter_bwd"))) return rc;
    (void)h_out;
    const int N = g->N, G = P->G, H = P->H, IN_e = P->IN_e;
    TM_REQUIRE(grads->G == G && grads->H == H && grads->IN_e == IN_e, "mp_iter_bwd: grads struct does not match the parameters");
    TM_REQUIRE(n_new >= 0 && n_new <= N, "mp_iter_bwd: n_new=%d N=%d", n_new, N);
    if (N == 0) return TMPNN_OK;
    TM_REQUIRE(prep && h && save && d_h && ws, "mp_iter_bwd: null buffer");
    TM_REQUIRE(d_scores == nullptr || scores != nullptr, "mp_iter_bwd: d_scores needs the forward's scores");
    TM_REQUIRE(aligned16(h) && aligned16(save) && aligned16(d_h) && aligned16(ws) && (d_hout == nullptr || aligned16(d_hout)) &&
                   aligned16(P->w_node) && aligned16(P->w_edge), "mp_iter_bwd: 16-byte alignment");
    const size_t need = tmpnn_mp_iter_bwd_ws(N, n_new, G, H, IN_e);
    if (ws_bytes < need) return set_error(TMPNN_EWORKSPACE, "mp_iter_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    const SaveLayout SL = save_layout(N, n_new, G, H);
    float* d_msg = reinterpret_cast<float*>(ws);
    const int nb = bwd_blocks(N);
    float* slabs = d_msg + (size_t)N * G * IN_e;
    float* bn_scratch = slabs + (size_t)nb * G * slab_floats(H, IN_e);
    hipStream_t st = as_stream(stream);
    TM_REQUIRE(st_dscores >= 0 && st_dlogits >= 0, "mp_iter_bwd: negative gradient stride");
    TM_REQUIRE((parts & ~7) == 0, "mp_iter_bwd_parts: parts=%d", parts);
    if (!(parts & 1)) {
    IterBwdArgs a{*P, *g, prep, h, scores, save + SL.gates, save + SL.es, d_scores, d_logits, d_hout, st_dscores, st_dlogits,
                  d_h, d_msg, slabs};
    if (H == 64 && IN_e == 64) hipLaunchKernelGGL((k_small_iter_bwd<64, 64>), dim3(nb), dim3(256), 0, st, a);
    else if (H == 64) hipLaunchKernelGGL((k_small_iter_bwd<64, 128>), dim3(nb), dim3(256), 0, st, a);
    else if (IN_e == 32) hipLaunchKernelGGL((k_small_iter_bwd<32, 32>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_small_iter_bwd<32, 64>), dim3(nb), dim3(256), 0, st, a);
    if ((rc = check_launch("small_iter_bwd"))) return rc;
    }
    if (parts & 2) return TMPNN_OK;
    const int rpb = 256 / (H / 4);
    int row_blocks = (N - n_new + rpb - 1) / rpb;
    if (row_blocks > 256) row_blocks = 256;
    if (row_blocks < 1) row_blocks = 1;
    const size_t red_elems = (size_t)G * ((size_t)3 * H * (IN_e + 3 * H) + 14 * H + 2);
    int red_blocks = (int)((red_elems + 255) / 256);
    if (red_blocks > 64) red_blocks = 64;
    const int bn_blocks = n_new > 0 ? G : 0;
    if (n_new > 0) TM_REQUIRE(x != nullptr && ld_x >= P->F_total, "mp_iter_bwd: x");
    FinishArgs f{*P, *grads, *g, d_msg, d_h, slabs, nb, row_blocks, red_blocks, n_new, training, x, ld_x,
                 save + SL.ysave, save + SL.mean, save + SL.rstd, reinterpret_cast<const int*>(save + SL.total), d_x,
                 bn_scratch, (parts & 4) ? 1 : 0};
    const size_t shm = sizeof(float) * (2 * BN_LDS_ROWS * (H + 1) + bn_stage_floats(H));
    const dim3 fgrid(row_blocks + red_blocks + bn_blocks);
#define LF(HH, II)                                                                                           \
    do {                                                                                                     \
        TM_SHM_ONCE((k_small_bwd_finish<HH, II>), shm);                                                      \
        hipLaunchKernelGGL((k_small_bwd_finish<HH, II>), fgrid, dim3(256), shm, st, f);                      \
    } while (0)
    if (H == 64 && IN_e == 64) LF(64, 64);
    else if (H == 64) LF(64, 128);
    else if (IN_e == 32) LF(32, 32);
    else LF(32, 64);
#undef LF
    if ((rc = check_launch("small_bwd_finish"))) return rc;
    return TMPNN_OK;
}

}  // extern "C"
