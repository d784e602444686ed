// Aggregation kernels of the TrackMPNN hot path (SURVEY 8(a) rows E, F and their adjoints).
//
// Both are HBM-bound row movers over the bipartite det/edge graph:
//   gather : one output row per EDGE  = f(two det rows)          (models/layers.py:90-95)
//   segsum : one output row per DET   = signed sum of its edges   (models/layers.py:103)
// and each is the other's adjoint, so the same two kernels serve forward and backward.
//
// Layout choices for gfx950: a state row is H fp32 = 128 B .. 1 KiB; a row is always moved by
// H/4 adjacent lanes with one 16-byte access each (full 64/128-byte segments, dwordx4), several
// rows per 64-lane wave.  Index loads are wave-uniform per row group.  The per-det sum walks the
// CSR run in a fixed order and combines the row groups with xor-shuffles, so results are bitwise
// reproducible (no float atomics anywhere).
#include "common.h"
#include <stdlib.h>

namespace tmpnn {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------------------------------
// gather: out[edge_row[e]] = in[src[e]] - in[dst[e]]        (CONCAT: [in[src] | in[dst]])
// ------------------------------------------------------------------------------------------
#ifdef TMPNN_KEEP_VARIANTS      // round-1 row movers (not pipelined): comparison builds only (TMPNN_AGG=0)
template <bool CONCAT, bool ACC>
__global__ __launch_bounds__(256) void k_gather(int E, const int32_t* __restrict__ src,
                                                const int32_t* __restrict__ dst,
                                                const int32_t* __restrict__ edge_row,
                                                const float* __restrict__ in, int ld_in,
                                                float* __restrict__ out, int ld_out, int H) {
    const int lpr = H >> 2;                       // lanes per row (one float4 each)
    const int rpb = 256 / lpr;                    // rows per block pass
    const int c4 = (threadIdx.x % lpr) * 4;
    const int slot = threadIdx.x / lpr;
    constexpr int U = 4;                          // edges in flight per lane group (memory-level parallelism)
    const long stride = (long)gridDim.x * rpb;
    for (long e0 = (long)blockIdx.x * rpb + slot; e0 < E; e0 += stride * U) {
        int s[U], d[U], r[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long e = e0 + u * stride;
            ok[u] = e < E;
            const long ec = ok[u] ? e : e0;
            s[u] = src[ec]; d[u] = dst[ec]; r[u] = edge_row[ec];
        }
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = *reinterpret_cast<const float4*>(in + (size_t)s[u] * ld_in + c4);
            b[u] = *reinterpret_cast<const float4*>(in + (size_t)d[u] * ld_in + c4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float* o = out + (size_t)r[u] * ld_out + c4;
            if (!CONCAT) {
                float4 v = make_float4(a[u].x - b[u].x, a[u].y - b[u].y, a[u].z - b[u].z, a[u].w - b[u].w);
                if (ACC) {
                    const float4 p = *reinterpret_cast<const float4*>(o);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                *reinterpret_cast<float4*>(o) = v;
            } else {
                float4 va = a[u], vb = b[u];
                if (ACC) {
                    const float4 p = *reinterpret_cast<const float4*>(o);
                    const float4 q = *reinterpret_cast<const float4*>(o + H);
                    va.x += p.x; va.y += p.y; va.z += p.z; va.w += p.w;
                    vb.x += q.x; vb.y += q.y; vb.z += q.z; vb.w += q.w;
                }
                *reinterpret_cast<float4*>(o) = va;
                *reinterpret_cast<float4*>(o + H) = vb;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// segsum: out[det_row[d]] = sum_p  (neg_p ? wneg * in[row_p, cneg:cneg+H] : in[row_p, 0:H])
//   diff adjoint / row F : wneg = -1, cneg = 0        concat adjoint : wneg = +1, cneg = H
// One wave per det; the wave's 64/lpr row groups take CSR entries round-robin, four entries of a
// group in flight at a time (the CSR run of a det is short -- ~16 at KITTI sizes -- so the kernel is
// latency bound unless every lane group keeps several row loads outstanding).
// ------------------------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(256) void k_segsum(int Dn, const int32_t* __restrict__ det_row,
                                                const int32_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ inc,
                                                const int32_t* __restrict__ det_order,
                                                const float* __restrict__ in, int ld_in,
                                                float* __restrict__ out, int ld_out, int H,
                                                float wneg, int cneg, int compact_out) {
    const int lane = threadIdx.x & 63;
    const int lpr = H >> 2;
    const int ngrp = 64 / lpr;
    const int grp = lane / lpr;
    const int c4 = (lane % lpr) * 4;
    constexpr int U = 4;
    // a block walks SEG_CHUNK consecutive entries of the visiting order at a time (its four waves interleaved), so
    // dets that share edges -- neighbours in that order -- are reduced on the same CU at about the same time
    constexpr int SEG_CHUNK = 64;
    const int wv = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * SEG_CHUNK; base < Dn; base += (long)gridDim.x * SEG_CHUNK) {
        const long hi = base + SEG_CHUNK < Dn ? base + SEG_CHUNK : Dn;
        for (long i = base + wv; i < hi; i += 4) {
            const int d = det_order ? det_order[i] : (int)i;
            const int p0 = rowptr[d], p1 = rowptr[d + 1];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int pb = p0 + grp; pb < p1; pb += ngrp * U) {
                int v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = pb + u * ngrp;
                    v[u] = p < p1 ? inc[p] : 0x7fffffff;       // sentinel: nothing to add
                }
                float4 x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = v[u] != 0x7fffffff;
                    const int row = live ? (v[u] & 0x7fffffff) : 0;
                    x[u] = *reinterpret_cast<const float4*>(in + (size_t)row * ld_in + (v[u] < 0 ? cneg : 0) + c4);
                    const float w = v[u] < 0 ? wneg : 1.0f;
                    if (live) { x[u].x *= w; x[u].y *= w; x[u].z *= w; x[u].w *= w; }
                    else x[u] = make_float4(0.f, 0.f, 0.f, 0.f);     // (never multiply: row 0 may hold inf/nan)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
            }
            for (int off = lpr; off < 64; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off);
                acc.y += __shfl_xor(acc.y, off);
                acc.z += __shfl_xor(acc.z, off);
                acc.w += __shfl_xor(acc.w, off);
            }
            if (grp == 0) {
                float* o = out + (size_t)(compact_out ? d : det_row[d]) * ld_out + c4;
                if (ACC) {
                    const float4 p = *reinterpret_cast<const float4*>(o);
                    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
                }
                *reinterpret_cast<float4*>(o) = acc;
            }
        }
    }
}


#endif  // TMPNN_KEEP_VARIANTS

// ------------------------------------------------------------------------------------------
// Software-pipelined forms (round 2).  Both row movers are chains of DEPENDENT loads -- index -> row for the gather,
// visiting order -> rowptr -> incidence -> row for the segment sum -- and with one chain per lane group in flight
// the memory system idles while the indices travel (measured: the edge rows are fetched from HBM exactly once and the
// det table comes out of L2, yet both kernels sat at 3.0-3.2 TB/s of algorithmic traffic).  Here every lane group
// issues the index loads of its NEXT item(s) right behind the row loads of the current one, so the row loads of
// consecutive items follow each other without an index round trip in between.  Same arithmetic, same summation
// order, bit-identical results.
// ------------------------------------------------------------------------------------------
template <bool CONCAT, bool ACC, int U = 4>
__global__ __launch_bounds__(256) void k_gather_pipe(int E, const int32_t* __restrict__ src,
                                                     const int32_t* __restrict__ dst,
                                                     const int32_t* __restrict__ edge_row,
                                                     const float* __restrict__ in, int ld_in,
                                                     float* __restrict__ out, int ld_out, int H) {
    const int lpr = H >> 2;
    const int rpb = 256 / lpr;
    const int c4 = (threadIdx.x % lpr) * 4;
    const int slot = threadIdx.x / lpr;
    // a block owns CONSECUTIVE chunks of U * rpb edges: the dets an edge block refers to (one src per D_t consecutive
    // edges, the same D_t dst rows over and over) stay in this CU's L1 across the chunk
    const long chunk = (long)rpb * U;
    long e0 = (long)blockIdx.x * chunk + slot;
    const long step = (long)gridDim.x * chunk;
    int s[U], d[U], r[U];
    auto load_ids = [&](long base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long e = base + (long)u * rpb;
            const long ec = e < E ? e : (E - 1);
            s[u] = src[ec]; d[u] = dst[ec]; r[u] = edge_row[ec];
        }
    };
    if (e0 < E) load_ids(e0);
    for (; e0 < E; e0 += step) {
        float4 a[U], b[U];
        int rr[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = e0 + (long)u * rpb < E;
            rr[u] = r[u];
            a[u] = *reinterpret_cast<const float4*>(in + (size_t)s[u] * ld_in + c4);
            b[u] = *reinterpret_cast<const float4*>(in + (size_t)d[u] * ld_in + c4);
        }
        if (e0 + step < E) load_ids(e0 + step);            // next chunk's indices travel while the rows arrive
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float* o = out + (size_t)rr[u] * ld_out + c4;
            if (!CONCAT) {
                float4 v = make_float4(a[u].x - b[u].x, a[u].y - b[u].y, a[u].z - b[u].z, a[u].w - b[u].w);
                if (ACC) {
                    const float4 p = *reinterpret_cast<const float4*>(o);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                *reinterpret_cast<float4*>(o) = v;
            } else {
                float4 va = a[u], vb = b[u];
                if (ACC) {
                    const float4 p = *reinterpret_cast<const float4*>(o);
                    const float4 q = *reinterpret_cast<const float4*>(o + H);
                    va.x += p.x; va.y += p.y; va.z += p.z; va.w += p.w;
                    vb.x += q.x; vb.y += q.y; vb.z += q.z; vb.w += q.w;
                }
                *reinterpret_cast<float4*>(o) = va;
                *reinterpret_cast<float4*>(o + H) = vb;
            }
        }
    }
}

// DUAL (the second pass of the dense form below): incidence entries >= nsplit address rows of `part` ([.][H], dense) instead of
// rows of `in`
// LIM (tmpnn_segsum_fwd_live): rows >= row_limit are known to be all-zero -- the new edge rows of a call, which start at 0
// (models/track_mpnn.py:61) -- and are NOT read (a lane group's load is masked off: no request at all).  Adding their +-0
// changes no bit of a sum, so the result equals the full read's.
template <bool ACC, int U = 4, int SEG_CHUNK = 32, bool DUAL = false, bool LIM = false>
__global__ __launch_bounds__(256) void k_segsum_pipe(int Dn, const int32_t* __restrict__ det_row,
                                                     const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ inc,
                                                     const int32_t* __restrict__ det_order,
                                                     const float* __restrict__ in, int ld_in,
                                                     float* __restrict__ out, int ld_out, int H,
                                                     float wneg, int cneg, int compact_out,
                                                     const float* __restrict__ part = nullptr, int nsplit = 0,
                                                     int row_limit = 0x7fffffff) {
    const int lane = threadIdx.x & 63;
    const int lpr = H >> 2;
    const int ngrp = 64 / lpr;
    const int grp = lane / lpr;
    const int c4 = (lane % lpr) * 4;
    const int wv = threadIdx.x >> 6;
    // a wave's dets: positions base + wv, base + wv + 4, ... of consecutive 64-entry chunks of the visiting order,
    // flattened into one sequence so that the pipeline runs across chunk boundaries
    const long nchunk = (Dn + SEG_CHUNK - 1) / SEG_CHUNK;
    auto item_pos = [&](long k) -> long {             // k-th det position of this wave, or -1
        const long per = SEG_CHUNK / 4;                // 16 dets of a chunk per wave
        const long ch = (long)blockIdx.x + (k / per) * gridDim.x;
        if (ch >= nchunk) return -1;
        const long i = ch * SEG_CHUNK + wv + (k % per) * 4;
        return i < Dn ? i : -2;                        // -2: hole at the tail of the last chunk (skip, keep going)
    };
    // stage A: det id ; stage B: its CSR range ; stage C: its first U incidences per group
    long kA = 0;
    int dA = -1, dB = -1, p0B = 0, p1B = 0, dC = -1, p0C = 0, p1C = 0, vC[U];
    auto fetchA = [&]() {
        long pos;
        do { pos = item_pos(kA++); } while (pos == -2);
        dA = pos < 0 ? -1 : (det_order ? det_order[pos] : (int)pos);
    };
    auto advanceB = [&]() {
        dB = dA;
        if (dB >= 0) { p0B = rowptr[dB]; p1B = rowptr[dB + 1]; }
    };
    auto advanceC = [&]() {
        dC = dB; p0C = p0B; p1C = p1B;
        if (dC >= 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0C + grp + u * ngrp;
                vC[u] = p < p1C ? inc[p] : 0x7fffffff;
            }
        }
    };
    fetchA(); advanceB(); fetchA(); advanceC(); advanceB(); fetchA();
    int pbC = p0C + grp;                       // first CSR position of this lane group in the current pass
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    while (dC >= 0) {
        const int d = dC;
        float4 x[U];
        bool live[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = vC[u] != 0x7fffffff;
            const int row = live[u] ? (vC[u] & 0x7fffffff) : 0;
            w[u] = vC[u] < 0 ? wneg : 1.0f;
            if (LIM) {
                live[u] = live[u] && row < row_limit;
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live[u]) x[u] = *reinterpret_cast<const float4*>(in + (size_t)row * ld_in + (vC[u] < 0 ? cneg : 0) + c4);
            } else if (DUAL && row >= nsplit)
                x[u] = *reinterpret_cast<const float4*>(part + (size_t)(row - nsplit) * H + c4);
            else
                x[u] = *reinterpret_cast<const float4*>(in + (size_t)row * ld_in + (vC[u] < 0 ? cneg : 0) + c4);
        }
        // the next item's index loads are issued behind this pass's row loads: the det's next pass (a long CSR run
        // takes several) or, after its last pass, the next det of the pipeline
        const bool last = (pbC - grp) + ngrp * U >= p1C;          // wave-uniform
        if (!last) {
            pbC += ngrp * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = pbC + u * ngrp;
                vC[u] = p < p1C ? inc[p] : 0x7fffffff;
            }
        } else {
            advanceC(); advanceB(); fetchA();
            pbC = p0C + grp;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (live[u]) { x[u].x *= w[u]; x[u].y *= w[u]; x[u].z *= w[u]; x[u].w *= w[u]; }
            else x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
        if (!last) continue;
        for (int off = lpr; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            acc.w += __shfl_xor(acc.w, off);
        }
        if (grp == 0) {
            float* o = out + (size_t)(compact_out ? d : det_row[d]) * ld_out + c4;
            if (ACC) {
                const float4 p = *reinterpret_cast<const float4*>(o);
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            *reinterpret_cast<float4*>(o) = acc;
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// TMPNN_AGG=0 keeps the round-1 kernels (A/B measurements); default: the pipelined forms
#ifdef TMPNN_KEEP_VARIANTS
static int agg_variant() {
    static const int v = [] { const char* e = getenv("TMPNN_AGG"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
#else
static constexpr int agg_variant() { return 1; }
#endif

__global__ void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, size_t stride, int nslab,
                               float* __restrict__ dst, size_t n, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < nslab; ++k) s += slabs[(size_t)k * stride + i];
    dst[i] = accumulate ? dst[i] + s : s;
}

// fold groups of 32 slabs: out[g][i] = sum_{k in group g} slabs[k][i]
__global__ void k_fold_slabs(const float* __restrict__ slabs, size_t stride, int nslab, float* __restrict__ out,
                             size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k0 = blockIdx.y * 32, k1 = min(nslab, k0 + 32);
    float s = 0.f;
#pragma unroll 8
    for (int k = k0; k < k1; ++k) s += slabs[(size_t)k * stride + i];
    out[(size_t)blockIdx.y * n + i] = s;
}

size_t reduce_slabs_ws_floats(int nslab, size_t n) { return nslab > 64 ? (size_t)ceil_div(nslab, 32) * n : 0; }

int launch_reduce_slabs(const float* slabs, size_t stride, int nslab, float* dst, size_t n, int accumulate,
                        hipStream_t st, float* ws2) {
    if (n == 0) return TMPNN_OK;
    if (nslab > 64 && ws2 != nullptr) {
        const int ng = ceil_div(nslab, 32);
        hipLaunchKernelGGL(k_fold_slabs, dim3(ceil_div((long)n, 256), ng), dim3(256), 0, st, slabs, stride, nslab, ws2, n);
        int rc = check_launch("fold_slabs");
        if (rc) return rc;
        slabs = ws2;
        stride = n;
        nslab = ng;
    }
    hipLaunchKernelGGL(k_reduce_slabs, dim3(ceil_div((long)n, 256)), dim3(256), 0, st, slabs, stride, nslab, dst,
                       n, accumulate);
    return check_launch("reduce_slabs");
}

static int check_graph(const tmpnn_graph* g) {
    TM_REQUIRE(g != nullptr, "graph is null");
    TM_REQUIRE(g->N >= 0 && g->E >= 0 && g->Dn >= 0 && (long)g->E + g->Dn == g->N,
               "graph sizes inconsistent: N=%d E=%d Dn=%d", g->N, g->E, g->Dn);
    if (g->E > 0)
        TM_REQUIRE(g->src && g->dst && g->edge_row && g->inc, "graph edge arrays are null (E=%d)", g->E);
    if (g->Dn > 0) TM_REQUIRE(g->det_row && g->rowptr, "graph det arrays are null (Dn=%d)", g->Dn);
    return TMPNN_OK;
}

static int check_rows(const float* in, int ld_in, const float* out, int ld_out, int H, int w_in, int w_out) {
    TM_REQUIRE(supported_H(H), "unsupported hidden width H=%d (need 32/64/128/256)", H);
    TM_REQUIRE(in && out, "null feature pointer");
    TM_REQUIRE(ld_in >= w_in && ld_out >= w_out, "leading dimension too small (ld_in=%d ld_out=%d)", ld_in, ld_out);
    TM_REQUIRE((ld_in & 3) == 0 && (ld_out & 3) == 0 && aligned16(in) && aligned16(out),
               "feature rows must be 16-byte aligned (ld multiple of 4 floats)");
    return TMPNN_OK;
}

static int grid_for(long items, int per_block) {
    long b = (items + per_block - 1) / per_block;
    if (b > 256L * 64) b = 256L * 64;   // grid-stride beyond 64 blocks per CU (16: gather 0.49 -> 0.45 ms, segsum 0.47 -> 0.45
                                        // ms per 6 M edges; the tail of a coarser grid costs more than the extra dispatches)
    if (b < 1) b = 1;
    return (int)b;
}

static int gather(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H, int accumulate,
                  bool concat, tmpnn_stream stream) {
    int rc = check_graph(g);
    if (rc) return rc;
    rc = check_rows(in, ld_in, out, ld_out, H, H, concat ? 2 * H : H);
    if (rc) return rc;
    if (g->E == 0) return TMPNN_OK;
    const int rpb = 256 / (H >> 2);
    dim3 grid(grid_for(g->E, rpb)), block(256);
    hipStream_t st = as_stream(stream);
#define L(K, C, A) hipLaunchKernelGGL((K<C, A>), grid, block, 0, st, g->E, g->src, g->dst, g->edge_row, in, ld_in, out, ld_out, H)
    if (agg_variant() && !concat) {                 // eight edge rows in flight per thread (16 row loads): +4 % over four
        grid = dim3(grid_for(g->E, rpb * 8));
        if (accumulate) hipLaunchKernelGGL((k_gather_pipe<false, true, 8>), grid, block, 0, st, g->E, g->src, g->dst, g->edge_row, in, ld_in, out, ld_out, H);
        else hipLaunchKernelGGL((k_gather_pipe<false, false, 8>), grid, block, 0, st, g->E, g->src, g->dst, g->edge_row, in, ld_in, out, ld_out, H);
    } else
    if (agg_variant()) {
        grid = dim3(grid_for(g->E, rpb * 4));
        if (concat) { if (accumulate) L(k_gather_pipe, true, true); else L(k_gather_pipe, true, false); }
        else        { if (accumulate) L(k_gather_pipe, false, true); else L(k_gather_pipe, false, false); }
    }
#ifdef TMPNN_KEEP_VARIANTS
    else {
        if (concat) { if (accumulate) L(k_gather, true, true); else L(k_gather, true, false); }
        else        { if (accumulate) L(k_gather, false, true); else L(k_gather, false, false); }
    }
#endif
#undef L
    return check_launch("gather");
}


// ---- dense graphs (BASELINE C5: frame blocks of 300 x 300 edges): a segment sum that reads every edge row ONCE -----------------
// k_segsum_pipe reads a row once from either endpoint; in a batch of small windows the second read is an L2 / MALL hit, in a
// dense scene it is a second HBM read (PMC: 1.99 x the rows at C5, the kernel at ~5.9 TB/s of real traffic).  Here the edge set
// is cut into 8 src x 16 dst TILES over det indices (struct tmpnn_seg_plan, built once per graph by the host:
// trackmpnn_amd.graph.dense_seg_plan; a slot without an edge holds -1) and a tile is summed by the block that streams it: wave w keeps the running sums of dsts 4w .. 4w+3 in registers over a whole ITEM
// (<= a few dozen consecutive tiles that share their 16 dsts) and forms, per tile, its share of the 8 src sums, which the four
// waves combine through LDS in a fixed order.  Out go one partial row per (tile, src) and per (item, dst): 6 % + < 1 % of the
// bytes read.  A second pass (k_segsum_pipe<DUAL>) sums, per det, its partial rows -- a CSR over them (plan->rowptr2 / inc2;
// raw edge rows are allowed in it too).  fp32, no atomics, every sum in a fixed order.
__global__ __launch_bounds__(256) void k_segsum_tiles(int I, int T, const int32_t* __restrict__ items,
                                                      const int32_t* __restrict__ t_row, const float* __restrict__ in,
                                                      int ld_in, float* __restrict__ part) {
    __shared__ float4 S[4][8][64];                              // 32 KB: a wave's share of the tile's 8 src sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto add4 = [](float4 a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; };
    for (int k = blockIdx.x; k < I; k += gridDim.x) {
        const int t0 = items[2 * k], nt = items[2 * k + 1];
        float4 ad[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ad[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = t0; t < t0 + nt; ++t) {
            const int32_t* tr = t_row + (size_t)t * 128 + 4 * w;
            float4 ps[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 x[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = tr[(half * 4 + i) * 16 + j];        // wave-uniform; -1: no such edge
                        // (branch-free: a conditional load makes hipcc drain the request queue at the join)
                        x[i][j] = *reinterpret_cast<const float4*>(in + (size_t)(row < 0 ? 0 : row) * ld_in + 4 * lane);
                        if (row < 0) x[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ps[half * 4 + i] = add4(add4(add4(x[i][0], x[i][1]), x[i][2]), x[i][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ad[j] = add4(ad[j], x[i][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) S[w][i][lane] = ps[i];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = w + 4 * r;
                const float4 v = add4(add4(add4(S[0][i][lane], S[1][i][lane]), S[2][i][lane]), S[3][i][lane]);
                *reinterpret_cast<float4*>(part + ((size_t)t * 8 + i) * 256 + 4 * lane) = v;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(part + ((size_t)8 * T + (size_t)16 * k + 4 * w + j) * 256 + 4 * lane) = ad[j];
    }
}

static int segsum_dense(const tmpnn_graph* g, const tmpnn_seg_plan* pl, const float* in, int ld_in, float* out, int ld_out,
                        int accumulate, int compact_out, hipStream_t st) {
    TM_REQUIRE(pl->T >= 0 && pl->I >= 0 && pl->nsplit == g->N && pl->rowptr2 && pl->inc2 && (pl->T == 0 || (pl->t_row && pl->items)),
               "segsum (dense form): inconsistent plan (T=%d I=%d nsplit=%d N=%d)", pl->T, pl->I, pl->nsplit, g->N);
    const size_t need = ((size_t)8 * pl->T + (size_t)16 * pl->I) * 256;
    if (need > 0 && (pl->ws == nullptr || pl->ws_floats < need))
        return set_error(TMPNN_EWORKSPACE, "segsum (dense form): the plan's partial buffer holds %zu floats, %zu needed", pl->ws_floats, need);
    if (pl->I > 0) {
        const int grid = pl->I < 256 * 4 ? pl->I : 256 * 4;
        hipLaunchKernelGGL(k_segsum_tiles, dim3(grid), dim3(256), 0, st, pl->I, pl->T, pl->items, pl->t_row, in, ld_in, pl->ws);
        int rc = check_launch("segsum_tiles");
        if (rc) return rc;
    }
    dim3 grid(grid_for(g->Dn, 16)), block(256);
    if (accumulate)
        hipLaunchKernelGGL((k_segsum_pipe<true, 8, 16, true>), grid, block, 0, st, g->Dn, g->det_row, pl->rowptr2, pl->inc2,
                           (const int32_t*)nullptr, in, ld_in, out, ld_out, 256, -1.0f, 0, compact_out, pl->ws, pl->nsplit);
    else
        hipLaunchKernelGGL((k_segsum_pipe<false, 8, 16, true>), grid, block, 0, st, g->Dn, g->det_row, pl->rowptr2, pl->inc2,
                           (const int32_t*)nullptr, in, ld_in, out, ld_out, 256, -1.0f, 0, compact_out, pl->ws, pl->nsplit);
    return check_launch("segsum (dense form, second pass)");
}

#ifdef TMPNN_KEEP_VARIANTS      // measured 2 x slower than the CSR kernel (DESIGN_HISTORY 13.6): comparison builds only
// ------------------------------------------------------------------------------------------------------------
// segment sum of a BATCH OF SMALL WINDOWS with every edge row read once (round 5, struct tmpnn_win_plan)
// ------------------------------------------------------------------------------------------------------------
// k_segsum_pipe reads an edge row from either endpoint, 2 x 256 B per edge at H = 64, through a memory pipe that gives a CU ~12
// bytes per clock (its 0.43 ms per 6 M edges ARE that rate).  A window of the rolling graph is ~40-100 dets and 0.4-2 k edge rows
// that touch nothing outside it.  Here a workgroup owns (window, column half) JOBS; it walks the window's edge rows in ascending order
// in CHUNKS of 160 rows, staged in LDS once each (LDS-DMA, 128-byte row halves, three buffers: two chunks travel while one is
// summed) -- half the row requests.  The sums are those of k_segsum_pipe BIT FOR BIT: incidence i of a det's run belongs to the
// STREAM (det, i % 4), the lane group of k_segsum_pipe that adds it; a stream's partial sum lives in LDS for the length of the
// job, is always added to by the same eight lanes, in run order (the host deals the streams round the 128 lane groups and lists,
// per chunk and lane group, the incidences to add -- `recs`, trackmpnn_amd.graph.build_win_plan), and the four streams of a det
// are combined as k_segsum_pipe's xor tree combines its lane groups.  Windows beyond the capacities below are listed in the plan
// and take the CSR kernel.
//
// The request queue (vmcnt) retires in order and the compiler cannot see requests made by inline asm, so inside the loop NOTHING
// is loaded into registers from memory: row ids, records and output rows travel by LDS-DMA as well, a window's record comes by a
// scalar load, and what an accumulating launch adds to (the det rows of `out`) is staged as one more chunk of the job.  A wave
// counts the requests (and stores) it issues per iteration; the wait at the top of an iteration lets exactly those of the
// iteration before stay in flight.
static constexpr int SW_CH = 160;                                     // edge rows of a chunk
static constexpr int SW_MAXCH = 24, SW_MAXD = 160, SW_LCAP = 15;      // chunks / dets of a window, steps of a chunk served here
static constexpr int SW_NW = 8, SW_NHG = 128;                         // waves; the plan's lane groups (8 lanes: a 128-byte row half),
                                                                      // two per group of eight lanes here (g and g + 64)
static constexpr int SW_ROWS = SW_CH * 128, SW_RECS = 4096, SW_ACCS = 4 * SW_MAXD * 128, SW_IDS = 768, SW_OROW = 768;
static constexpr int SW_OFF_RECS = 3 * SW_ROWS, SW_OFF_ACCS = SW_OFF_RECS + 3 * SW_RECS, SW_OFF_IDS = SW_OFF_ACCS + SW_ACCS,
                     SW_OFF_OROW = SW_OFF_IDS + 3 * SW_IDS;
static constexpr size_t SW_SHM = SW_OFF_OROW + 2 * SW_OROW;
static_assert(SW_SHM <= 160 * 1024, "k_segsum_win: three chunks and a window's streams must fit the LDS");
static_assert(SW_MAXD <= SW_CH && SW_LCAP * 2 * SW_NHG <= SW_RECS && 4 * SW_CH <= SW_IDS && 4 * SW_MAXD <= SW_OROW &&
                  4 * SW_MAXD <= 5 * SW_NHG && SW_CH / 8 <= 3 * SW_NW && SW_MAXD <= 3 * 8 * SW_NW, "k_segsum_win: capacities");

__device__ __forceinline__ uint32_t sw_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(const char*)p;
}
// 16 bytes (DW = 4) or 4 bytes (DW = 1) per lane from a per-lane global address to (uniform LDS address) + DW * 4 * lane
template <int DW>
__device__ __forceinline__ void sw_glds(const void* gsrc, uint32_t lds_wave_base) {
    unsigned keep;
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);
    if constexpr (DW == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
// everything but this wave's newest n requests / stores has landed (n wave-uniform; more than 8: everything)
__device__ __forceinline__ void sw_wait(int n) {
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// a job = (window, column half); its items: nch chunks of edge rows, then (accumulating launches) the det rows of `out` as one more
// chunk; at least two items per job, so that two buffers of output rows are enough.  nit = 0: no further job
struct SwJob {
    int j, e0, ne, q0, nd, r0, lm2, mb, nit;
    unsigned long long lm01;
    __device__ __forceinline__ int hf() const { return j & 1; }
    __device__ __forceinline__ int nch() const { return (ne + SW_CH - 1) / SW_CH; }
};
struct SwItem { SwJob jb; int c, L, roff; };                            // chunk, its steps, its first step in `recs`
// (field by field: a 60-byte struct assignment is left as a memcpy between stack slots, i.e. scratch traffic inside the loop)
__device__ __forceinline__ void sw_copy(SwItem& d, const SwItem& s) {
    d.jb.j = s.jb.j; d.jb.e0 = s.jb.e0; d.jb.ne = s.jb.ne; d.jb.q0 = s.jb.q0; d.jb.nd = s.jb.nd; d.jb.r0 = s.jb.r0;
    d.jb.lm01 = s.jb.lm01; d.jb.lm2 = s.jb.lm2; d.jb.mb = s.jb.mb; d.jb.nit = s.jb.nit;
    d.c = s.c; d.L = s.L; d.roff = s.roff;
}

// a window's record, by ONE scalar load: the compiler turns plain uniform loads of a kernel that also stores into vector loads
// and waits for them with vmcnt -- which would wait for the chunk requests in flight
__device__ __forceinline__ void sw_window(const int32_t* wrec, int w, SwJob& jb) {
    typedef int sw_v8 __attribute__((ext_vector_type(8)));
    sw_v8 r;
    const int32_t* p = wrec + 8 * (size_t)__builtin_amdgcn_readfirstlane(w);
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    jb.e0 = r[0]; jb.ne = r[1]; jb.q0 = r[2]; jb.nd = r[3]; jb.r0 = r[4]; jb.lm2 = r[7];
    jb.lm01 = ((unsigned long long)(unsigned)r[6] << 32) | (unsigned)r[5];
}

template <bool ACC>
__global__ __launch_bounds__(512) void k_segsum_win(tmpnn_win_plan pl, const float* __restrict__ in, int ld_in,
                                                     float* __restrict__ out, int ld_out, int compact_out) {
    extern __shared__ __attribute__((aligned(16))) char sw_lds[];
    constexpr int NW = SW_NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int njob = 2 * pl.W, G = gridDim.x;
    const int hgl = tid >> 3, c4 = (lane & 7) * 4;                   // this lane's group of eight; its four columns of a row half
    auto rows_of = [&](int b) { return reinterpret_cast<float*>(sw_lds + b * SW_ROWS); };
    auto recs_of = [&](int b) { return reinterpret_cast<uint16_t*>(sw_lds + SW_OFF_RECS + b * SW_RECS); };
    auto ids_of = [&](int b) { return reinterpret_cast<int*>(sw_lds + SW_OFF_IDS + b * SW_IDS); };
    auto orow_of = [&](int b) { return reinterpret_cast<int*>(sw_lds + SW_OFF_OROW + b * SW_OROW); };
    float* const accs = reinterpret_cast<float*>(sw_lds + SW_OFF_ACCS);
    auto steps_of = [&](const SwJob& jb, int c) {
        // (arithmetic on both sides of the select: a select between neighbouring FIELDS becomes an indexed load and the struct
        //  goes to scratch)
        const int lo = (int)(jb.lm01 >> (4 * (c & 15))) & 15, hi = (jb.lm2 >> (4 * (c & 7))) & 15;
        return c < jb.nch() ? (c < 16 ? lo : hi) : 0;
    };
    // the first job at or after j that this kernel serves
    auto job_at = [&](SwJob& jb, int j, int mb) {
        jb.j = j; jb.mb = mb; jb.nit = 0;
        for (; jb.j < njob; jb.j += G) {
            sw_window(pl.wrec, jb.j >> 1, jb);
            if (jb.nd > 0 && jb.nd <= SW_MAXD) {                      // (the plan marks the windows it leaves to the CSR kernel)
                jb.nit = max(2, jb.nch() + (ACC ? 1 : 0));
                break;
            }
        }
    };
    auto next_item = [&](SwItem& it) {
        if (it.jb.nit == 0) return;
        if (it.c + 1 < it.jb.nit) { ++it.c; it.roff += it.L; }
        else { job_at(it.jb, it.jb.j + G, it.jb.mb ^ 1); it.c = 0; it.roff = it.jb.r0; }
        it.L = steps_of(it.jb, it.c);
    };
    // what an item stages: rows of `in` by the window's edge list, or (the last item of an accumulating launch's job) the det
    // rows of `out`; cnt = 0: nothing (a padding item, or the end of the stream)
    auto item_cnt = [&](const SwItem& it, bool& prev) {
        prev = false;
        if (it.jb.nit == 0) return 0;
        if (it.c < it.jb.nch()) return min(SW_CH, it.jb.ne - it.c * SW_CH);
        if (ACC && it.c == it.jb.nit - 1) { prev = true; return it.jb.nd; }
        return 0;
    };
    int fly = 0;                                                      // requests and stores of this wave in this iteration
    // ids of an item into id buffer ib: the edge list is 16-byte aligned (one request of wave 8); the output rows are not
    auto stage_ids = [&](const SwItem& it, int ib) {
        bool prev;
        const int cnt = item_cnt(it, prev);
        if (cnt == 0) return;
        if (!prev) {
            if (wv == 4) {
                const int n4 = (cnt + 3) >> 2;                        // (the list of a window is padded to a multiple of four)
                if (lane < n4) sw_glds<4>(pl.erow + it.jb.e0 + it.c * SW_CH + 4 * lane, sw_lds_addr(ids_of(ib)));   // (<= 640 B)
                ++fly;
            }
        } else if (wv >= 5 && wv < 8 && 64 * (wv - 5) < cnt) {   // (waves 5 .. 7)
            const int32_t* src = (compact_out ? pl.det : pl.drow) + it.jb.q0;
            sw_glds<1>(src + min(64 * (wv - 5) + lane, cnt - 1), sw_lds_addr(ids_of(ib)) + 256u * (wv - 5));
            ++fly;
        }
    };
    auto stage_rows = [&](const SwItem& it, int rb) {
        bool prev;
        const int cnt = item_cnt(it, prev);
        const SwJob& jb = it.jb;
        if (jb.nit == 0) return;
        if (it.c == 0 && wv >= 5 && 64 * (wv - 5) < jb.nd) {              // the job's output rows (for its sums' stores)
            const int32_t* src = (compact_out ? pl.det : pl.drow) + jb.q0;
            sw_glds<1>(src + min(64 * (wv - 5) + lane, jb.nd - 1), sw_lds_addr(orow_of(jb.mb)) + 256u * (wv - 5));
            ++fly;
        }
        if (wv >= 4 && 4 * (7 - wv) < it.L) {                         // the chunk's records: 1 KB (four steps) a request
            const int x = 7 - wv;
            const int nl = min(64, 16 * (it.L - 4 * x));              // (16 lanes a step)
            sw_glds<4>(pl.recs + ((size_t)it.roff + 4 * x) * SW_NHG + 8 * min(lane, nl - 1), sw_lds_addr(recs_of(rb)) + 1024u * x);
            ++fly;
        }
        const float* base = prev ? out : in;
        const int ld = prev ? ld_out : ld_in;
        const int* idl = ids_of(rb);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int piece = wv + NW * k;
            if (8 * piece < cnt) {
                const int id = idl[min(8 * piece + (lane >> 3), cnt - 1)];
                sw_glds<4>(base + (size_t)id * ld + 32 * jb.hf() + c4, sw_lds_addr(rows_of(rb)) + 1024u * piece);
                ++fly;
            }
        }
    };
    SwItem A, B, Cn, D, En;
    A.jb.e0 = A.jb.ne = A.jb.q0 = A.jb.nd = A.jb.r0 = A.jb.lm2 = 0;
    A.jb.lm01 = 0;
    job_at(A.jb, blockIdx.x, 0);
    if (A.jb.nit == 0) return;
    A.c = 0; A.L = steps_of(A.jb, 0); A.roff = A.jb.r0;
    sw_copy(B, A);
    next_item(B);
    sw_copy(Cn, B);
    next_item(Cn);
    sw_copy(D, Cn);
    next_item(D);
    sw_copy(En, D);
    next_item(En);
    // every stream's sum starts at zero (and is put back to zero when its job's sums are written)
    for (int x = tid; x < SW_ACCS / 16; x += 512) reinterpret_cast<float4*>(accs)[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    // prologue: ids of items 0 .. 3, then the rows of items 0 and 1 (item t's rows use id buffer t % 3, as its rows do)
    stage_ids(A, 0);
    stage_ids(B, 1);
    stage_ids(Cn, 2);
    sw_wait(0);
    __syncthreads();
    stage_rows(A, 0);
    sw_wait(0);
    __syncthreads();
    fly = 0;
    stage_rows(B, 1);
    stage_ids(D, 0);
    int fly_prev = fly;
    int W_nd = 0, W_hf = 0, W_mb = 0;                                  // the job whose sums are written at the top of the next iteration
    bool wr = false;
    int rb = 0;                                                        // row / record / id buffer of item t: t % 3
    for (;;) {
        sw_wait(fly_prev);                                            // everything but what the iteration before this one issued
        __syncthreads();
        fly = 0;
        if (wr) {
            // the sums of the job that ended with item t - 1: a det's four streams combined as k_segsum_pipe's xor tree combines
            // its lane groups, (+ the det row of `out`, staged as item t - 1), one store of 8 dets per wave; streams back to zero
            const int* orow = orow_of(W_mb);
            const float* prevr = rows_of(rb == 0 ? 2 : rb - 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (64 * p + 8 * wv >= W_nd) continue;                // (uniform)
                const int q = 64 * p + hgl, qc = min(q, W_nd - 1);
                float* as = accs + qc * 4 * 32 + c4;
                const float4 a0 = *reinterpret_cast<const float4*>(as), a1 = *reinterpret_cast<const float4*>(as + 32);
                const float4 a2 = *reinterpret_cast<const float4*>(as + 64), a3 = *reinterpret_cast<const float4*>(as + 96);
                const int orw = orow[qc];
                float4 t;
                t.x = (a0.x + a1.x) + (a2.x + a3.x);
                t.y = (a0.y + a1.y) + (a2.y + a3.y);
                t.z = (a0.z + a1.z) + (a2.z + a3.z);
                t.w = (a0.w + a1.w) + (a2.w + a3.w);
                if (ACC) {
                    const float4 pv = *reinterpret_cast<const float4*>(prevr + qc * 32 + c4);
                    t.x += pv.x; t.y += pv.y; t.z += pv.z; t.w += pv.w;
                }
                if (q < W_nd) {
                    *reinterpret_cast<float4*>(out + (size_t)orw * ld_out + 32 * W_hf + c4) = t;
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(as) = z;
                    *reinterpret_cast<float4*>(as + 32) = z;
                    *reinterpret_cast<float4*>(as + 64) = z;
                    *reinterpret_cast<float4*>(as + 96) = z;
                }
                ++fly;
            }
            __syncthreads();                                          // (item t + 2 is staged over item t - 1's buffer; streams are zero)
            wr = false;
        }
        if (A.jb.nit == 0) break;
        // rows and records two iterations ahead of their sums, ids two ahead of the rows that use them
        stage_rows(Cn, rb == 0 ? 2 : rb - 1);
        stage_ids(En, rb == 2 ? 0 : rb + 1);
        if (A.L > 0) {
            // (eight lanes serve the plan's lane groups g and g + 64: two records a step, their streams never the same)
            const uint16_t* rc = recs_of(rb) + hgl;
            const float* rows = rows_of(rb) + c4;
            float* ab = accs + hgl * 32 + c4;
            unsigned r0 = rc[0], r1 = rc[64];
            for (int s = 0; s < A.L; ++s) {
                const int sn = min(s + 1, A.L - 1) * SW_NHG;
                const unsigned n0 = rc[sn], n1 = rc[sn + 64];        // (the next records behind these ones' rows)
                const float4 x0 = *reinterpret_cast<const float4*>(rows + (r0 & 255u) * 32);
                const float4 x1 = *reinterpret_cast<const float4*>(rows + (r1 & 255u) * 32);
                float* ap0 = ab + ((r0 >> 9) & 7u) * (SW_NHG * 32);
                float* ap1 = ab + ((r1 >> 9) & 7u) * (SW_NHG * 32) + 64 * 32;
                float4 a0 = *reinterpret_cast<const float4*>(ap0), a1 = *reinterpret_cast<const float4*>(ap1);
                const float w0 = (r0 & 0x100u) ? -1.0f : 1.0f, w1 = (r1 & 0x100u) ? -1.0f : 1.0f;
                a0.x += x0.x * w0; a0.y += x0.y * w0; a0.z += x0.z * w0; a0.w += x0.w * w0;
                a1.x += x1.x * w1; a1.y += x1.y * w1; a1.z += x1.z * w1; a1.w += x1.w * w1;
                if (!(r0 & 0x8000u)) *reinterpret_cast<float4*>(ap0) = a0;
                if (!(r1 & 0x8000u)) *reinterpret_cast<float4*>(ap1) = a1;
                r0 = n0; r1 = n1;
            }
        }
        if (A.c == A.jb.nit - 1) { wr = true; W_nd = A.jb.nd; W_hf = A.jb.hf(); W_mb = A.jb.mb; }
        sw_copy(A, B); sw_copy(B, Cn); sw_copy(Cn, D); sw_copy(D, En);
        next_item(En);
        rb = rb == 2 ? 0 : rb + 1;
        fly_prev = fly;
    }
    sw_wait(0);
}

static int cu_count() {
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus > 0 ? cus : 256;
    }();
    return v;
}

static int segsum_win(const tmpnn_graph* g, const tmpnn_win_plan* pl, const float* in, int ld_in, float* out, int ld_out,
                      int accumulate, int compact_out, hipStream_t st) {
    TM_REQUIRE(pl->W > 0 && pl->wrec && pl->erow && pl->recs && pl->det && pl->drow &&
                   (pl->nbig == 0 || pl->big_order), "segsum: incomplete window plan");
    const int njob = 2 * pl->W, cus = cu_count();
    dim3 grid(njob < cus ? njob : cus), block(512);
    if (accumulate) {
        TM_SHM_ONCE((k_segsum_win<true>), SW_SHM);
        hipLaunchKernelGGL((k_segsum_win<true>), grid, block, SW_SHM, st, *pl, in, ld_in, out, ld_out, compact_out);
    } else {
        TM_SHM_ONCE((k_segsum_win<false>), SW_SHM);
        hipLaunchKernelGGL((k_segsum_win<false>), grid, block, SW_SHM, st, *pl, in, ld_in, out, ld_out, compact_out);
    }
    int rc = check_launch("segsum (window-owned form)");
    if (rc || pl->nbig == 0) return rc;
    // the windows beyond the LDS capacity: the CSR kernel over their dets (its visiting order = the plan's list)
    dim3 g2(grid_for(pl->nbig, 32));
    if (accumulate)
        hipLaunchKernelGGL((k_segsum_pipe<true, 4, 32>), g2, dim3(256), 0, st, pl->nbig, g->det_row, g->rowptr, g->inc, pl->big_order,
                           in, ld_in, out, ld_out, 64, -1.0f, 0, compact_out);
    else
        hipLaunchKernelGGL((k_segsum_pipe<false, 4, 32>), g2, dim3(256), 0, st, pl->nbig, g->det_row, g->rowptr, g->inc, pl->big_order,
                           in, ld_in, out, ld_out, 64, -1.0f, 0, compact_out);
    return check_launch("segsum (windows beyond the LDS capacity)");
}

#endif  // TMPNN_KEEP_VARIANTS

static int segsum(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H, int accumulate,
                  float wneg, int cneg, int compact_out, tmpnn_stream stream, int row_limit = 0x7fffffff) {
    int rc = check_graph(g);
    if (rc) return rc;
    rc = check_rows(in, ld_in, out, ld_out, H, H + cneg, H);
    if (rc) return rc;
    if (g->Dn == 0) return TMPNN_OK;
    hipStream_t st = as_stream(stream);
    if (g->seg_plan && H == 256 && wneg == -1.0f && cneg == 0)        // dense graph with a plan: every edge row read once
        return segsum_dense(g, g->seg_plan, in, ld_in, out, ld_out, accumulate, compact_out, st);
#ifdef TMPNN_KEEP_VARIANTS
    // a batch of small windows with a plan, enough of them to fill the chip: every edge row read once
    if (g->win_plan && H == 64 && wneg == -1.0f && cneg == 0 && 2 * g->win_plan->W >= cu_count() &&
        ld_in % 4 == 0 && ld_out % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0)
        return segsum_win(g, g->win_plan, in, ld_in, out, ld_out, accumulate, compact_out, st);
#endif
    dim3 grid(grid_for(g->Dn, 64)), block(256);
#define LS(K, A) hipLaunchKernelGGL((K<A>), grid, block, 0, st, g->Dn, g->det_row, g->rowptr, g->inc, g->det_order, in, ld_in, out, ld_out, H, wneg, cneg, compact_out)
#define LSP(A, UU, CH)                                                                                       \
    do {                                                                                                     \
        grid = dim3(grid_for(g->Dn, CH));                                                                    \
        hipLaunchKernelGGL((k_segsum_pipe<A, UU, CH>), grid, block, 0, st, g->Dn, g->det_row, g->rowptr, g->inc, \
                           g->det_order, in, ld_in, out, ld_out, H, wneg, cneg, compact_out);                \
    } while (0)
    // rows in flight per lane group and pass: 4 covers a C2-like det (16 incidences at H = 64) in one pass; graphs whose
    // dets average more than 24 incidences (C3: 34) take 8, i.e. two passes instead of three (same box: 0.43 -> 0.45 of
    // 8 TB/s on the C3 shape; on C2 4 is the faster one).  A property of the graph, not of a measurement.
    // A block walks 32-entry chunks of the visiting order (8 dets per wave; 64: 0.44 -> 0.42 ms per 6 M edges).
    const bool deep = (long)2 * g->E > (long)24 * g->Dn;
    if (row_limit < g->N && !accumulate) {         // rows >= row_limit are zero and stay unread (k_segsum_pipe<.., LIM>)
        grid = dim3(grid_for(g->Dn, deep ? 16 : 32));
        if (deep)
            hipLaunchKernelGGL((k_segsum_pipe<false, 8, 16, false, true>), grid, block, 0, st, g->Dn, g->det_row, g->rowptr, g->inc,
                               g->det_order, in, ld_in, out, ld_out, H, wneg, cneg, compact_out, (const float*)nullptr, 0, row_limit);
        else
            hipLaunchKernelGGL((k_segsum_pipe<false, 4, 32, false, true>), grid, block, 0, st, g->Dn, g->det_row, g->rowptr, g->inc,
                               g->det_order, in, ld_in, out, ld_out, H, wneg, cneg, compact_out, (const float*)nullptr, 0, row_limit);
        return check_launch("segsum (live rows)");
    }
    if (agg_variant()) {                 // (high-degree graphs: 16-entry chunks, 4 dets per wave: 0.47 -> 0.49 on the C3 shape)
        if (deep) { if (accumulate) LSP(true, 8, 16); else LSP(false, 8, 16); }
        else      { if (accumulate) LSP(true, 4, 32); else LSP(false, 4, 32); }
    }
#ifdef TMPNN_KEEP_VARIANTS
    else               { if (accumulate) LS(k_segsum, true); else LS(k_segsum, false); }
#endif
#undef LSP
#undef LS
    return check_launch("segsum");
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {

int tmpnn_abi_version(void) { return TMPNN_ABI_VERSION; }
const char* tmpnn_last_error(void) { return tmpnn::g_err; }

// H > 256 (multiples of 128): the row movers run once per slice of <= 256 columns (a row is moved by H / 4 adjacent lanes of
// ONE wave); rows are strided, so a slice is the same call on shifted pointers.  Not for the concat forms (their second half
// sits H columns further on).
#define TM_SLICED(H_, call_)                                                              \
    do {                                                                                  \
        if (supported_H_big(H_)) {                                                        \
            for (int c0 = 0; c0 < (H_); c0 += 256) {                                      \
                const int w = (H_) - c0 < 256 ? (H_) - c0 : 256;                          \
                const int rc_ = call_;                                                    \
                if (rc_) return rc_;                                                      \
            }                                                                             \
            return TMPNN_OK;                                                              \
        }                                                                                 \
    } while (0)

int tmpnn_gather_diff_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H,
                          int accumulate, tmpnn_stream stream) {
    TM_SLICED(H, gather(g, in + c0, ld_in, out + c0, ld_out, w, accumulate, false, stream));
    return gather(g, in, ld_in, out, ld_out, H, accumulate, false, stream);
}
int tmpnn_gather_concat_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H,
                            int accumulate, tmpnn_stream stream) {
    return gather(g, in, ld_in, out, ld_out, H, accumulate, true, stream);
}
int tmpnn_gather_diff_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din, int H,
                          int accumulate, tmpnn_stream stream) {
    TM_SLICED(H, segsum(g, d_out + c0, ld_dout, d_in + c0, ld_din, w, accumulate, -1.0f, 0, 0, stream));
    return segsum(g, d_out, ld_dout, d_in, ld_din, H, accumulate, -1.0f, 0, 0, stream);
}
int tmpnn_gather_concat_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din, int H,
                            int accumulate, tmpnn_stream stream) {
    // (slices of the two halves [h[src] | h[dst]] of a 2H-wide row: the second half stays H columns further on)
    TM_SLICED(H, segsum(g, d_out + c0, ld_dout, d_in + c0, ld_din, w, accumulate, 1.0f, H, 0, stream));
    return segsum(g, d_out, ld_dout, d_in, ld_din, H, accumulate, 1.0f, H, 0, stream);
}
int tmpnn_segsum_fwd(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H, int accumulate,
                     int compact_out, tmpnn_stream stream) {
    TM_SLICED(H, segsum(g, in + c0, ld_in, out + c0, ld_out, w, accumulate, -1.0f, 0, compact_out, stream));
    return segsum(g, in, ld_in, out, ld_out, H, accumulate, -1.0f, 0, compact_out, stream);
}
int tmpnn_segsum_fwd_live(const tmpnn_graph* g, const float* in, int ld_in, float* out, int ld_out, int H, int compact_out,
                          int row_limit, tmpnn_stream stream) {
    TM_REQUIRE(row_limit >= 0, "segsum_fwd_live: row_limit %d", row_limit);
    TM_SLICED(H, segsum(g, in + c0, ld_in, out + c0, ld_out, w, 0, -1.0f, 0, compact_out, stream, row_limit));
    return segsum(g, in, ld_in, out, ld_out, H, 0, -1.0f, 0, compact_out, stream, row_limit);
}
int tmpnn_segsum_bwd(const tmpnn_graph* g, const float* d_out, int ld_dout, float* d_in, int ld_din, int H,
                     int accumulate, tmpnn_stream stream) {
    TM_SLICED(H, gather(g, d_out + c0, ld_dout, d_in + c0, ld_din, w, accumulate, false, stream));
    return gather(g, d_out, ld_dout, d_in, ld_din, H, accumulate, false, stream);
}

int tmpnn_transpose(const float* in, int rows, int cols, float* out, tmpnn_stream stream) {
    TM_REQUIRE(in && out && rows > 0 && cols > 0, "transpose: bad arguments");
    dim3 grid(ceil_div(cols, 32), ceil_div(rows, 32)), block(32, 8);
    hipLaunchKernelGGL(k_transpose, grid, block, 0, as_stream(stream), in, rows, cols, out);
    return check_launch("transpose");
}

}  // extern "C"
