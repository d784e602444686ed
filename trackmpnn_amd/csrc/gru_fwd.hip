// GRU cells of the factor-graph update, FORWARD (SURVEY 8(a) rows E, H', I, J; models/layers.py:90-97,114,116): the generic
// f32-MFMA kernel, the LDS-resident f32 kernel, the bf16x6 forms (per-row gathers, node cell, edge tiles of 32 / 16 rows) and
// the det-row projection GEMM, with their entry points.  Shared helpers: gru_common.h; the backward: gru_bwd.hip.
#include "gru_common.h"

namespace tmpnn {

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
        // (the weights of eight k-steps are requested together: written step by step, every MFMA waited for the one weight
        //  load in front of it -- a full L2 round trip per step, 140 us for the 40 det rows of a KITTI window at IN = 128)
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 8) {
            float bw[8][3 * CT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* __restrict__ b = b0 + (size_t)(s0 + u) * H3;
#pragma unroll
                for (int t = 0; t < CT; ++t) { bw[u][3 * t] = b[t * 32]; bw[u][3 * t + 1] = b[H + t * 32]; bw[u][3 * t + 2] = b[2 * H + t * 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);          // (or the scheduler pairs every load with its MFMA again)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    acc_r[t] = mfma32(av[s0 + u], bw[u][3 * t], acc_r[t]);
                    acc_z[t] = mfma32(av[s0 + u], bw[u][3 * t + 1], acc_z[t]);
                    acc_in[t] = mfma32(av[s0 + u], bw[u][3 * t + 2], acc_in[t]);
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
        for (int s0 = 0; s0 < 16; s0 += 8) {
            float bw[8][3 * CT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* __restrict__ b = b0 + (size_t)(s0 + u) * H3;
#pragma unroll
                for (int t = 0; t < CT; ++t) { bw[u][3 * t] = b[t * 32]; bw[u][3 * t + 1] = b[H + t * 32]; bw[u][3 * t + 2] = b[2 * H + t * 32]; }
            }
            __builtin_amdgcn_sched_barrier(0);          // (or the scheduler pairs every load with its MFMA again)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    acc_r[t] = mfma32(av[s0 + u], bw[u][3 * t], acc_r[t]);
                    acc_z[t] = mfma32(av[s0 + u], bw[u][3 * t + 1], acc_z[t]);
                    acc_hn[t] = mfma32(av[s0 + u], bw[u][3 * t + 2], acc_hn[t]);
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
                const float n = tanhf_(acc_in[t][reg] + bin + r * hn);
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

template <int H, int IN, int XMODE>
__device__ __forceinline__ void a_issue(const GruFwdArgs& a, int kt, const TileIdx& ix, int half, ATile& t) {
    constexpr int NKX = (XMODE == 3) ? 0 : IN / 32;
    if (kt < NKX) {
        const int f0 = kt * 32 + half * 16;
        if (XMODE == 0) load16(a.msg + (size_t)(a.msg_compact ? ix.li : ix.row) * a.ld_msg + f0, t.u);
        else if (XMODE == 1) {
            load16(a.h + (size_t)ix.s * a.ld_h + f0, t.u);
            load16(a.h + (size_t)ix.d * a.ld_h + f0, t.w);
        } else {
            if (f0 < H) load16(a.h + (size_t)ix.s * a.ld_h + f0, t.u);
            else        load16(a.h + (size_t)ix.d * a.ld_h + (f0 - H), t.u);
        }
    } else {
        load16(a.h + (size_t)ix.row * a.ld_h + (kt - NKX) * 32 + half * 16, t.u);
    }
}

// CT = 32-column tiles per wave; (H/32)/CT waves share a 32-row tile (each owning CT column tiles),
// WPB waves per block.  CT = 1 halves the accumulator file of a wave (64 registers), which buys a third
// wave per SIMD and room for the operand prefetch without spilling.
template <int H, int IN, int XMODE, int CT, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gru_fwd_lds(GruFwdArgs a, int ntiles) {
    extern __shared__ float lds[];
    constexpr int H3 = 3 * H;
    constexpr int CW = (H / 32) / CT;          // column waves per row tile
    constexpr int RTB = WPB / CW;              // row tiles per block iteration
    // XMODE 3: the x-part of the gates arrives pre-projected (msg = h[dets] @ W_ih^T, [Dn][3H]) and is
    // gathered in the epilogue -- by linearity (h[src]-h[dst]) W = h[src] W - h[dst] W, so the per-EDGE
    // half of the forward GEMM collapses into one small GEMM over the det rows.
    constexpr int INL = (XMODE == 3) ? 0 : IN;      // x columns that go through the MFMAs
    constexpr int NKX = INL / 32, NK = NKX + H / 32;
    float* sWih = lds;               // [INL][3H]
    float* sWhh = lds + INL * H3;    // [H][3H]
    for (int i = threadIdx.x * 4; i < INL * H3; i += WPB * 64 * 4)
        *reinterpret_cast<float4*>(sWih + i) = *reinterpret_cast<const float4*>(a.wih_t + i);
    for (int i = threadIdx.x * 4; i < H * H3; i += WPB * 64 * 4)
        *reinterpret_cast<float4*>(sWhh + i) = *reinterpret_cast<const float4*>(a.whh_t + i);
    // work distribution: the block owns a contiguous range of (32-row tile, column wave) items and its
    // waves pull them from an LDS counter.  Waves on one SIMD get DIFFERENT static priorities
    // (waves w, w+4, w+8 share a SIMD): identical waves otherwise run in lockstep -- all in their
    // MFMA phase together, then all in their store phase together -- and the matrix pipe idles while
    // the epilogues drain.  With a strict order the top wave runs at full rate and the others fill
    // every gap its loads and stores leave.
    int* next_item = reinterpret_cast<int*>(lds + (INL + H) * H3 + WPB * (32 * STG_LD));
    if (threadIdx.x == 0) *next_item = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    {
        const int grp = __builtin_amdgcn_readfirstlane(wave) >> 2;
        if (grp == 0) __builtin_amdgcn_s_setprio(3);
        else if (grp == 1) __builtin_amdgcn_s_setprio(2);
        else if (grp == 2) __builtin_amdgcn_s_setprio(1);
        // (a fourth group, if any, stays at priority 0)
    }
    const int items_total = ((a.R + 31) / 32) * CW;            // (32-row tile, column wave) pairs
    const int per_block = (items_total + gridDim.x - 1) / gridDim.x;
    const int item_lo = blockIdx.x * per_block;
    const int item_hi = min(items_total, item_lo + per_block);
    (void)ntiles;

    // Items are software pipelined ACROSS tiles: while a tile's MFMAs run, the next item's row ids are
    // already loaded, and its first operand slice is requested before this tile's epilogue -- a global load
    // takes microseconds under load, longer than a tile's matrix work.
    int item = 0;
    if (lane == 0) item = atomicAdd(next_item, 1);
    item = __builtin_amdgcn_readfirstlane(item) + item_lo;
    if (item >= item_hi) return;
    int cw0 = (item % CW) * CT * 32;                           // first output column of this item
    int r0 = (item / CW) * 32;
    TileIdx ix = tile_idx<XMODE>(a, r0, c);
    ATile cur, nxt;
    a_issue<H, IN, XMODE>(a, 0, ix, half, cur);

    for (;;) {
        int nitem = 0;
        if (lane == 0) nitem = atomicAdd(next_item, 1);
        nitem = __builtin_amdgcn_readfirstlane(nitem) + item_lo;
        const bool nvalid = nitem < item_hi;
        const int ncw0 = (nitem % CW) * CT * 32;
        const int nr0 = nvalid ? (nitem / CW) * 32 : r0;
        const TileIdx nix = tile_idx<XMODE>(a, nr0, c);
        const int li = ix.li, row = ix.row;
        f32x16 acc_r[CT], acc_z[CT], acc_in[CT], acc_hn[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc_r[t][i] = 0.f; acc_z[t][i] = 0.f; acc_in[t][i] = 0.f; acc_hn[t][i] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            float av[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) av[i] = (XMODE == 1 && kt < NKX) ? cur.u[i] - cur.w[i] : cur.u[i];
            if (kt + 1 < NK) a_issue<H, IN, XMODE>(a, kt + 1, ix, half, nxt);
            else if (nvalid) a_issue<H, IN, XMODE>(a, 0, nix, half, nxt);      // next item's first slice
            __builtin_amdgcn_sched_barrier(0);
            const bool xpart = kt < NKX;
            const float* b0 = (xpart ? sWih + (kt * 32 + half * 16) * H3 : sWhh + ((kt - NKX) * 32 + half * 16) * H3) + cw0 + c;
            // B operands are read from LDS one k step AHEAD of the MFMAs that consume them, so a step's
            // six matrix instructions never wait on their own ds_read
            float bq[2][3 * CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) { bq[0][3 * t] = b0[t * 32]; bq[0][3 * t + 1] = b0[H + t * 32]; bq[0][3 * t + 2] = b0[2 * H + t * 32]; }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        bq[(s + 1) & 1][3 * t] = b0[(s + 1) * H3 + t * 32];
                        bq[(s + 1) & 1][3 * t + 1] = b0[(s + 1) * H3 + H + t * 32];
                        bq[(s + 1) & 1][3 * t + 2] = b0[(s + 1) * H3 + 2 * H + t * 32];
                    }
                }
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    // weights as the FIRST operand: the accumulator then holds the transposed tile
                    // (lane = state row, registers = 4-wide runs of output features), which makes the
                    // epilogue one row per lane with 16-byte loads and stores
                    acc_r[t] = mfma32(bq[s & 1][3 * t], av[s], acc_r[t]);
                    acc_z[t] = mfma32(bq[s & 1][3 * t + 1], av[s], acc_z[t]);
                    if (xpart) acc_in[t] = mfma32(bq[s & 1][3 * t + 2], av[s], acc_in[t]);
                    else       acc_hn[t] = mfma32(bq[s & 1][3 * t + 2], av[s], acc_hn[t]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 3 * CT, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 3 * CT, 0);
            }
            cur = nxt;
        }
        // epilogue: lane = its own row (li), accumulator register 4q+i <-> feature 8q + 4*half + i of the tile.
        // Outputs leave through a per-wave LDS staging tile so that every global store instruction writes
        // full 128-byte row segments with 16 bytes per lane (8 lanes per row): dword stores are issue-bound
        // and 32-byte runs (what the accumulator layout would give directly) waste the write path.
        float* stg = lds + (INL + H) * H3 + wave * (32 * STG_LD);   // (the item counter sits behind the tiles)
        float4 hp4[CT][4];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                hp4[t][q] = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + cw0 + t * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            // results overwrite the accumulators they were computed from (r -> acc_r, z -> acc_z, n -> acc_in,
            // W_hn h + b -> acc_hn), so the five output tiles cost 16 extra registers, not 80
            f32x16 outv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cw0 + t * 32 + 8 * q + 4 * half;
                const float4 bir = *reinterpret_cast<const float4*>(a.b_ih + col);
                const float4 bhr = *reinterpret_cast<const float4*>(a.b_hh + col);
                const float4 biz = *reinterpret_cast<const float4*>(a.b_ih + H + col);
                const float4 bhz = *reinterpret_cast<const float4*>(a.b_hh + H + col);
                const float4 bin = *reinterpret_cast<const float4*>(a.b_ih + 2 * H + col);
                const float4 bhn = *reinterpret_cast<const float4*>(a.b_hh + 2 * H + col);
                const float br[4] = {bir.x + bhr.x, bir.y + bhr.y, bir.z + bhr.z, bir.w + bhr.w};
                const float bz[4] = {biz.x + bhz.x, biz.y + bhz.y, biz.z + bhz.z, biz.w + bhz.w};
                const float bi[4] = {bin.x, bin.y, bin.z, bin.w};
                const float bh[4] = {bhn.x, bhn.y, bhn.z, bhn.w};
                const float hp[4] = {hp4[t][q].x, hp4[t][q].y, hp4[t][q].z, hp4[t][q].w};
                float xr[4] = {0.f, 0.f, 0.f, 0.f}, xz[4] = {0.f, 0.f, 0.f, 0.f}, xn[4] = {0.f, 0.f, 0.f, 0.f};
                if (XMODE == 3) {
                    const float* ps = a.msg + (size_t)ix.s * a.ld_msg + col;
                    const float* pd = a.msg + (size_t)ix.d * a.ld_msg + col;
                    const float4 sr = *reinterpret_cast<const float4*>(ps), dr = *reinterpret_cast<const float4*>(pd);
                    const float4 sz = *reinterpret_cast<const float4*>(ps + H), dz = *reinterpret_cast<const float4*>(pd + H);
                    const float4 sn = *reinterpret_cast<const float4*>(ps + 2 * H), dn = *reinterpret_cast<const float4*>(pd + 2 * H);
                    xr[0] = sr.x - dr.x; xr[1] = sr.y - dr.y; xr[2] = sr.z - dr.z; xr[3] = sr.w - dr.w;
                    xz[0] = sz.x - dz.x; xz[1] = sz.y - dz.y; xz[2] = sz.z - dz.z; xz[3] = sz.w - dz.w;
                    xn[0] = sn.x - dn.x; xn[1] = sn.y - dn.y; xn[2] = sn.z - dn.z; xn[3] = sn.w - dn.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int reg = 4 * q + i;
                    const float ro = sigmoidf_(acc_r[t][reg] + xr[i] + br[i]);
                    const float zo = sigmoidf_(acc_z[t][reg] + xz[i] + bz[i]);
                    const float ho = acc_hn[t][reg] + bh[i];
                    const float no = tanhf_((XMODE == 3 ? xn[i] : acc_in[t][reg]) + bi[i] + ro * ho);
                    outv[reg] = (1.0f - zo) * no + zo * hp[i];
                    acc_r[t][reg] = ro; acc_z[t][reg] = zo; acc_hn[t][reg] = ho; acc_in[t][reg] = no;
                }
            }
            const int colt = cw0 + t * 32;
            if (a.logit_part) {
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4*>(a.w_head + colt + 8 * q + 4 * half);
                    p += outv[4 * q] * w.x + outv[4 * q + 1] * w.y + outv[4 * q + 2] * w.z + outv[4 * q + 3] * w.w;
                }
                p += __shfl_xor(p, 32);
                if (half == 0 && r0 + c < a.R) a.logit_part[(size_t)(colt / 32) * a.part_stride + row] = p;
            }
            stage_store32(stg, c, half, lane, outv, a.h_out, a.ld_out, colt, row, r0, a.R);
            if (a.gates) {
                stage_store32<true>(stg, c, half, lane, acc_r[t], a.gates, H, colt, row, r0, a.R);
                stage_store32<true>(stg, c, half, lane, acc_z[t], a.gates + a.gate_plane, H, colt, row, r0, a.R);
                stage_store32<true>(stg, c, half, lane, acc_in[t], a.gates + 2 * a.gate_plane, H, colt, row, r0, a.R);
                stage_store32<true>(stg, c, half, lane, acc_hn[t], a.gates + 3 * a.gate_plane, H, colt, row, r0, a.R);
            }
        }
        if (!nvalid) break;
        cw0 = ncw0; r0 = nr0; ix = nix;
    }
}

// XMODE 3 forward on the bf16 pipe (bf16x6, see mfma_x6).  Same persistent structure and the same epilogue as
// k_gru_fwd_lds<H, H, 3, 1, WPB>; what changes is the operand path:
//   * W_hh sits in LDS as three bf16 pieces [piece][3H][H + 8] (k contiguous, rows padded by 16 bytes),
//     read as one ds_read_b128 per piece and 16-deep k block;
//   * a lane's operand is H/2 CONTIGUOUS floats of its state row (k = (H/2)*half + 8*kb + j), split into
//     pieces in registers; the whole next-item operand is requested before this item's matrix phase, so no
//     operand load is ever queued behind this item's gate stores.
template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gru_fwd_split(GruFwdArgs a) {
    extern __shared__ float lds[];
    constexpr int H3 = 3 * H, KP = H + 8, NKB = H / 16, CW = H / 32, NQ4 = H / 8;   // NQ4 float4 per lane operand
    uint16_t* sW = reinterpret_cast<uint16_t*>(lds);                  // [3][3H][KP]
    for (int i = threadIdx.x; i < H * H3 / 4; i += WPB * 64) {
        const int k = i / (H3 / 4), j0 = (i % (H3 / 4)) * 4;
        const float4 w = *reinterpret_cast<const float4*>(a.whh_t + (size_t)k * H3 + j0);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t q1, q2, q3;
            split1(wv[e], q1, q2, q3);
            sW[(0 * H3 + j0 + e) * KP + k] = q1;
            sW[(1 * H3 + j0 + e) * KP + k] = q2;
            sW[(2 * H3 + j0 + e) * KP + k] = q3;
        }
    }
    float* stg_base = reinterpret_cast<float*>(sW + 3 * H3 * KP);
    int* next_item = reinterpret_cast<int*>(stg_base + WPB * (32 * STG_LD));
    // gate biases, pre-added where the cell adds them: [b_ir+b_hr | b_iz+b_hz | b_in | b_hn | w_head], read back with
    // ds_read_b128 in the epilogue (24 fewer vector-memory instructions per item than fetching them from L1)
    float* sBias = reinterpret_cast<float*>(next_item + 4);
    for (int i = threadIdx.x; i < H; i += WPB * 64) {
        sBias[i] = a.b_ih[i] + a.b_hh[i];
        sBias[H + i] = a.b_ih[H + i] + a.b_hh[H + i];
        sBias[2 * H + i] = a.b_ih[2 * H + i];
        sBias[3 * H + i] = a.b_hh[2 * H + i];
        sBias[4 * H + i] = a.logit_part ? a.w_head[i] : 0.f;      // output head slice for the fused partial dot product
    }
    if (threadIdx.x == 0) *next_item = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    {
        const int grp = __builtin_amdgcn_readfirstlane(wave) >> 2;
        if (grp == 0) __builtin_amdgcn_s_setprio(3);
        else if (grp == 1) __builtin_amdgcn_s_setprio(2);
        else if (grp == 2) __builtin_amdgcn_s_setprio(1);
    }
    const int items_total = ((a.R + 31) / 32) * CW;
    const int per_block = (items_total + gridDim.x - 1) / gridDim.x;
    const int item_lo = blockIdx.x * per_block;
    const int item_hi = min(items_total, item_lo + per_block);

    int item = 0;
    if (lane == 0) item = atomicAdd(next_item, 1);
    item = __builtin_amdgcn_readfirstlane(item) + item_lo;
    if (item >= item_hi) return;
    int cw0 = (item % CW) * 32;
    int r0 = (item / CW) * 32;
    TileIdx ix = tile_idx<3>(a, r0, c);
    float4 raw[NQ4];
    {
        const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)ix.row * a.ld_h + (H / 2) * half);
#pragma unroll
        for (int i = 0; i < NQ4; ++i) raw[i] = xr[i];
    }
    float* stg = stg_base + wave * (32 * STG_LD);

    for (;;) {
        int nitem = 0;
        if (lane == 0) nitem = atomicAdd(next_item, 1);
        nitem = __builtin_amdgcn_readfirstlane(nitem) + item_lo;
        const bool nvalid = nitem < item_hi;
        const int ncw0 = (nitem % CW) * 32;
        const int nr0 = nvalid ? (nitem / CW) * 32 : r0;
        const TileIdx nix = tile_idx<3>(a, nr0, c);
        const int row = ix.row;
        Split8 b[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) b[kb] = split8(raw[2 * kb], raw[2 * kb + 1]);
        if (nvalid) {
            const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)nix.row * a.ld_h + (H / 2) * half);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) raw[i] = xr[i];
        }
        f32x16 acc_r, acc_z, acc_hn, acc_in;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_r[i] = 0.f; acc_z[i] = 0.f; acc_hn[i] = 0.f; }
        {
            const uint16_t* wp0 = sW + (cw0 + c) * KP + (H / 2) * half;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint16_t* wp = wp0 + 8 * kb;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const uint4 w1 = *reinterpret_cast<const uint4*>(wp + (g * H) * KP);
                    const uint4 w2 = *reinterpret_cast<const uint4*>(wp + (H3 + g * H) * KP);
                    const uint4 w3 = *reinterpret_cast<const uint4*>(wp + (2 * H3 + g * H) * KP);
                    if (g == 0) acc_r = mfma_x6(w1, w2, w3, b[kb], acc_r);
                    else if (g == 1) acc_z = mfma_x6(w1, w2, w3, b[kb], acc_z);
                    else acc_hn = mfma_x6(w1, w2, w3, b[kb], acc_hn);
                }
            }
        }
        // epilogue: as in k_gru_fwd_lds (lane = its own row, accumulator register 4q+i <-> feature 8q + 4*half + i)
        float4 hp4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            hp4[q] = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + cw0 + 8 * q + 4 * half);
        f32x16 outv;
        {
            // The projected det rows P[src], P[dst] are gathered GATE by gate: the four 16-byte pieces a lane takes from one
            // 128-byte line (its row's 32 columns of one gate) are requested back to back, so the line is fetched from L2
            // once.  Column chunk by chunk (all three gates of one chunk, then the next chunk) the other five line sets of
            // the item -- and the seven other waves' items -- pass through the 32 KiB L1 between two touches of a line.
            const float* ps0 = a.msg + (size_t)ix.s * a.ld_msg + cw0 + 4 * half;
            const float* pd0 = a.msg + (size_t)ix.d * a.ld_msg + cw0 + 4 * half;
            float4 gs[4], gd[4], hs[4], hd[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ps0 + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(pd0 + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) hs[q] = *reinterpret_cast<const float4*>(ps0 + H + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) hd[q] = *reinterpret_cast<const float4*>(pd0 + H + 8 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // r: pre-activation in place
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + cw0 + 8 * q + 4 * half);
                acc_r[4 * q + 0] = acc_r[4 * q + 0] + (gs[q].x - gd[q].x) + b4.x;
                acc_r[4 * q + 1] = acc_r[4 * q + 1] + (gs[q].y - gd[q].y) + b4.y;
                acc_r[4 * q + 2] = acc_r[4 * q + 2] + (gs[q].z - gd[q].z) + b4.z;
                acc_r[4 * q + 3] = acc_r[4 * q + 3] + (gs[q].w - gd[q].w) + b4.w;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ps0 + 2 * H + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(pd0 + 2 * H + 8 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // z
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + H + cw0 + 8 * q + 4 * half);
                acc_z[4 * q + 0] = acc_z[4 * q + 0] + (hs[q].x - hd[q].x) + b4.x;
                acc_z[4 * q + 1] = acc_z[4 * q + 1] + (hs[q].y - hd[q].y) + b4.y;
                acc_z[4 * q + 2] = acc_z[4 * q + 2] + (hs[q].z - hd[q].z) + b4.z;
                acc_z[4 * q + 3] = acc_z[4 * q + 3] + (hs[q].w - hd[q].w) + b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // n: input part
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + 2 * H + cw0 + 8 * q + 4 * half);
                acc_in[4 * q + 0] = (gs[q].x - gd[q].x) + b4.x;
                acc_in[4 * q + 1] = (gs[q].y - gd[q].y) + b4.y;
                acc_in[4 * q + 2] = (gs[q].z - gd[q].z) + b4.z;
                acc_in[4 * q + 3] = (gs[q].w - gd[q].w) + b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4h = *reinterpret_cast<const float4*>(sBias + 3 * H + cw0 + 8 * q + 4 * half);
                const float bh[4] = {b4h.x, b4h.y, b4h.z, b4h.w};
                const float hp[4] = {hp4[q].x, hp4[q].y, hp4[q].z, hp4[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int reg = 4 * q + i;
                    const float ro = sigmoidf_(acc_r[reg]);
                    const float zo = sigmoidf_(acc_z[reg]);
                    const float ho = acc_hn[reg] + bh[i];
                    const float no = tanhf_(acc_in[reg] + ro * ho);
                    outv[reg] = (1.0f - zo) * no + zo * hp[i];
                    acc_r[reg] = ro; acc_z[reg] = zo; acc_hn[reg] = ho; acc_in[reg] = no;
                }
            }
        }
        if (a.logit_part) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(sBias + 4 * H + cw0 + 8 * q + 4 * half);
                p += outv[4 * q] * w.x + outv[4 * q + 1] * w.y + outv[4 * q + 2] * w.z + outv[4 * q + 3] * w.w;
            }
            p += __shfl_xor(p, 32);
            if (half == 0 && r0 + c < a.R) a.logit_part[(size_t)(cw0 / 32) * a.part_stride + row] = p;
        }
        stage_store32(stg, c, half, lane, outv, a.h_out, a.ld_out, cw0, row, r0, a.R);
        if (a.gates) {
            stage_store32<true>(stg, c, half, lane, acc_r, a.gates, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_z, a.gates + a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_in, a.gates + 2 * a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_hn, a.gates + 3 * a.gate_plane, H, cw0, row, r0, a.R);
        }
        if (!nvalid) break;
        cw0 = ncw0; r0 = nr0; ix = nix;
    }
}

// XMODE 0 forward (the NODE cell, models/layers.py:114: x = the compact aggregate es[d], both GEMMs per row) on the bf16 pipe.
// Both weight matrices as three bf16 pieces take 2 x 81 KB at H = 64 -- more than the LDS -- so a block owns ONE 32-column half
// of the outputs: its slices [piece][3 gates x 32 columns][H + 8] of W_ih and W_hh are 2 x 40.5 KB, and the two blocks that
// share a row range sit on the same XCD (block ids b and b + 8), so the second read of a row's operands is an L2 hit.
// Items are 32-row tiles pulled from an LDS counter; per item a lane splits its half row of x, runs the x products
// (r, z, n_in), splits its half row of h and runs the h products (r, z, n_h); the next item's operands are requested right
// behind the split that consumed the registers.  Epilogue, staging stores and the fused head as in k_gru_fwd_split.
template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gru_fwd_split_node(GruFwdArgs a) {
    extern __shared__ float lds[];
    constexpr int KP = H + 8, NKB = H / 16, CW = H / 32, NQ4 = H / 8, NC = 96;
    constexpr int H3 = 3 * H;
    // blocks b and b + 8 (same XCD under the round-robin placement) take the two column halves of row group
    // (b % 8) + 8 * (b / (8 * CW))
    const int cwb = CW == 1 ? 0 : (blockIdx.x >> 3) % CW;
    const int group = CW == 1 ? blockIdx.x : (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * CW));
    const int ngroups = CW == 1 ? gridDim.x : gridDim.x / CW;
    const int cw0 = cwb * 32;
    uint16_t* sWi = reinterpret_cast<uint16_t*>(lds);                 // [3][NC][KP]
    uint16_t* sWh = sWi + 3 * NC * KP;
    for (int i = threadIdx.x; i < H * NC / 4; i += WPB * 64) {
        const int k = i / (NC / 4), j0 = (i % (NC / 4)) * 4;
        const int col = (j0 / 32) * H + cw0 + (j0 % 32);
        const float4 wi = *reinterpret_cast<const float4*>(a.wih_t + (size_t)k * H3 + col);
        const float4 wh = *reinterpret_cast<const float4*>(a.whh_t + (size_t)k * H3 + col);
        const float wiv[4] = {wi.x, wi.y, wi.z, wi.w}, whv[4] = {wh.x, wh.y, wh.z, wh.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t q1, q2, q3;
            split1(wiv[e], q1, q2, q3);
            sWi[(0 * NC + j0 + e) * KP + k] = q1; sWi[(1 * NC + j0 + e) * KP + k] = q2; sWi[(2 * NC + j0 + e) * KP + k] = q3;
            split1(whv[e], q1, q2, q3);
            sWh[(0 * NC + j0 + e) * KP + k] = q1; sWh[(1 * NC + j0 + e) * KP + k] = q2; sWh[(2 * NC + j0 + e) * KP + k] = q3;
        }
    }
    float* stg_base = reinterpret_cast<float*>(sWh + 3 * NC * KP);
    int* next_item = reinterpret_cast<int*>(stg_base + WPB * (32 * STG_LD));
    float* sBias = reinterpret_cast<float*>(next_item + 4);          // [b_ir+b_hr | b_iz+b_hz | b_in | b_hn | w_head] of this half
    for (int i = threadIdx.x; i < 32; i += WPB * 64) {
        const int f = cw0 + i;
        sBias[i] = a.b_ih[f] + a.b_hh[f];
        sBias[32 + i] = a.b_ih[H + f] + a.b_hh[H + f];
        sBias[64 + i] = a.b_ih[2 * H + f];
        sBias[96 + i] = a.b_hh[2 * H + f];
        sBias[128 + i] = a.logit_part ? a.w_head[f] : 0.f;
    }
    if (threadIdx.x == 0) *next_item = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    if ((__builtin_amdgcn_readfirstlane(wave) >> 2) == 0) __builtin_amdgcn_s_setprio(2);   // SIMD partners out of lockstep
    const int tiles_total = (a.R + 31) / 32;
    const int per_group = (tiles_total + ngroups - 1) / ngroups;
    const int item_lo = group * per_group;
    const int item_hi = min(tiles_total, item_lo + per_group);

    int item = 0;
    if (lane == 0) item = atomicAdd(next_item, 1);
    item = __builtin_amdgcn_readfirstlane(item) + item_lo;
    if (item >= item_hi) return;
    int r0 = item * 32;
    int li = min(r0 + c, a.R - 1);
    int row = a.rows[li];
    float4 rawx[NQ4], rawh[NQ4];
    {
        const float4* xr = reinterpret_cast<const float4*>(a.msg + (size_t)(a.msg_compact ? li : row) * a.ld_msg + (H / 2) * half);
        const float4* hr = reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + (H / 2) * half);
#pragma unroll
        for (int i = 0; i < NQ4; ++i) rawx[i] = xr[i];
#pragma unroll
        for (int i = 0; i < NQ4; ++i) rawh[i] = hr[i];
    }
    float* stg = stg_base + wave * (32 * STG_LD);
    const uint16_t* wpi = sWi + c * KP + (H / 2) * half;
    const uint16_t* wph = sWh + c * KP + (H / 2) * half;

    for (;;) {
        int nitem = 0;
        if (lane == 0) nitem = atomicAdd(next_item, 1);
        nitem = __builtin_amdgcn_readfirstlane(nitem) + item_lo;
        const bool nvalid = nitem < item_hi;
        const int nr0 = nvalid ? nitem * 32 : r0;
        const int nli = min(nr0 + c, a.R - 1);
        const int nrow = a.rows[nli];
        f32x16 acc_r, acc_z, acc_hn, acc_in;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_r[i] = 0.f; acc_z[i] = 0.f; acc_hn[i] = 0.f; acc_in[i] = 0.f; }
        Split8 b[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) b[kb] = split8(rawx[2 * kb], rawx[2 * kb + 1]);
        if (nvalid) {
            const float4* xr = reinterpret_cast<const float4*>(a.msg + (size_t)(a.msg_compact ? nli : nrow) * a.ld_msg + (H / 2) * half);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) rawx[i] = xr[i];
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const uint16_t* wp = wpi + (g * 32) * KP + 8 * kb;
                const uint4 w1 = *reinterpret_cast<const uint4*>(wp);
                const uint4 w2 = *reinterpret_cast<const uint4*>(wp + NC * KP);
                const uint4 w3 = *reinterpret_cast<const uint4*>(wp + 2 * NC * KP);
                if (g == 0) acc_r = mfma_x6(w1, w2, w3, b[kb], acc_r);
                else if (g == 1) acc_z = mfma_x6(w1, w2, w3, b[kb], acc_z);
                else acc_in = mfma_x6(w1, w2, w3, b[kb], acc_in);
            }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) b[kb] = split8(rawh[2 * kb], rawh[2 * kb + 1]);
        if (nvalid) {
            const float4* hr = reinterpret_cast<const float4*>(a.h + (size_t)nrow * a.ld_h + (H / 2) * half);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) rawh[i] = hr[i];
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const uint16_t* wp = wph + (g * 32) * KP + 8 * kb;
                const uint4 w1 = *reinterpret_cast<const uint4*>(wp);
                const uint4 w2 = *reinterpret_cast<const uint4*>(wp + NC * KP);
                const uint4 w3 = *reinterpret_cast<const uint4*>(wp + 2 * NC * KP);
                if (g == 0) acc_r = mfma_x6(w1, w2, w3, b[kb], acc_r);
                else if (g == 1) acc_z = mfma_x6(w1, w2, w3, b[kb], acc_z);
                else acc_hn = mfma_x6(w1, w2, w3, b[kb], acc_hn);
            }
        }
        // epilogue: lane = its own row, accumulator register 4q + i <-> feature cw0 + 8q + 4 half + i
        float4 hp4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            hp4[q] = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + cw0 + 8 * q + 4 * half);
        f32x16 outv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 br = *reinterpret_cast<const float4*>(sBias + 8 * q + 4 * half);
            const float4 bz = *reinterpret_cast<const float4*>(sBias + 32 + 8 * q + 4 * half);
            const float4 bi = *reinterpret_cast<const float4*>(sBias + 64 + 8 * q + 4 * half);
            const float4 bh = *reinterpret_cast<const float4*>(sBias + 96 + 8 * q + 4 * half);
            const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w};
            const float biv[4] = {bi.x, bi.y, bi.z, bi.w}, bhv[4] = {bh.x, bh.y, bh.z, bh.w};
            const float hp[4] = {hp4[q].x, hp4[q].y, hp4[q].z, hp4[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int reg = 4 * q + i;
                const float ro = sigmoidf_(acc_r[reg] + brv[i]);
                const float zo = sigmoidf_(acc_z[reg] + bzv[i]);
                const float ho = acc_hn[reg] + bhv[i];
                const float no = tanhf_(acc_in[reg] + biv[i] + ro * ho);
                outv[reg] = (1.0f - zo) * no + zo * hp[i];
                acc_r[reg] = ro; acc_z[reg] = zo; acc_hn[reg] = ho; acc_in[reg] = no;
            }
        }
        if (a.logit_part) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(sBias + 128 + 8 * q + 4 * half);
                p += outv[4 * q] * w.x + outv[4 * q + 1] * w.y + outv[4 * q + 2] * w.z + outv[4 * q + 3] * w.w;
            }
            p += __shfl_xor(p, 32);
            if (half == 0 && r0 + c < a.R) a.logit_part[(size_t)cwb * a.part_stride + row] = p;
        }
        stage_store32(stg, c, half, lane, outv, a.h_out, a.ld_out, cw0, row, r0, a.R);
        if (a.gates) {
            stage_store32<true>(stg, c, half, lane, acc_r, a.gates, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_z, a.gates + a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_in, a.gates + 2 * a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_hn, a.gates + 3 * a.gate_plane, H, cw0, row, r0, a.R);
        }
        if (!nvalid) break;
        r0 = nr0; li = nli; row = nrow;
    }
}

// ---- the same forward over EDGE TILES (struct tmpnn_edge_tiles, rows_per_tile = 32) ----------------------------------
// What holds k_gru_fwd_split is not bytes but a dependent chain inside a wave: after an item's 72 MFMAs the epilogue asks
// for 24 scattered 16-byte pieces of P[src] / P[dst] per lane and waits for them (s_memtime, round 1: 15.5 k of an item's
// 31.5 k cycles), and there is no register left to request them any earlier.  Here the DISTINCT projected det rows of the
// item's tile (16-18 on the KITTI-shaped batches, tile list from trackmpnn_amd.graph.build_edge_tiles) are brought into a
// per-wave LDS area by LDS-DMA -- no register holds them -- a whole item ahead: the DMA for item i + 1 is issued when item
// i's stores have left the staging tile (the area IS the staging tile: P is consumed before the outputs are staged) and
// lands under item i + 1's operand split and matrix phase; the epilogue reads P[src] - P[dst] with ds_read_b128.
// Rows of 3 gates x 32 columns sit 400 bytes apart (16 consecutive det positions -> 16 different 16-byte slots of the bank
// row).  A tile with more than TCAP distinct dets takes the gathers of k_gru_fwd_split.  Same products, same order, same
// epilogue arithmetic: bit-identical results.
#define FT_MARK(i) do { } while (0)
struct FwdTiles { const int32_t* t_row; const int32_t* t_loc; const int32_t* t_dptr; const int32_t* t_dets; int T; };
static constexpr int TCAP = 24, TP_LD = 100;                   // dets staged per item; floats per staged row
static constexpr int TP_AREA = TCAP * TP_LD;                   // floats per wave (>= 32 * STG_LD: the output staging tile)

__device__ __forceinline__ uint32_t lds_addr_g(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(const char*)p;
}
// 16 bytes per lane: per-lane global address -> (wave-uniform LDS byte address) + 16 * lane.  Inline asm on purpose: hipcc
// would wait vmcnt(0) before every later LDS read (the weight operands of the matrix phase) for a builtin DMA it cannot
// disambiguate; this one is waited for by hand where the epilogue needs it.
__device__ __forceinline__ void glds16_g(const void* gsrc, uint32_t lds_wave_base) {
    unsigned keep;
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}

struct TiledIdx { int row, loc, det, nd; };
__device__ __forceinline__ TiledIdx tiled_idx(const FwdTiles& tl, int R, int t, int c, int lane) {
    TiledIdx x;
    const int li = min(32 * t + c, R - 1);                     // (padding slots: the tile's last valid row, never stored)
    x.row = tl.t_row[li];
    x.loc = tl.t_loc[li];
    const int dp0 = tl.t_dptr[t];
    x.nd = tl.t_dptr[t + 1] - dp0;
    x.det = lane < x.nd ? tl.t_dets[dp0 + lane] : 0;
    return x;
}
// request the staged copy of a tile's distinct P rows (columns cw0 .. cw0 + 31 of the three gates) into the wave's area
template <int H>
__device__ __forceinline__ void tiled_stage_p(const GruFwdArgs& a, const TiledIdx& x, int cw0, int lane, uint32_t area) {
    const int nchunk = x.nd * 25;                              // 24 chunks of 16 B per det + one of padding
#pragma unroll 1
    for (int i0 = 0; i0 < nchunk; i0 += 64) {
        const int idx = i0 + lane;
        const int j = idx / 25, rem = idx - 25 * j;
        // the 64 chunks of a pass belong to four det rows at most: their ids by v_readlane (a cross-lane read through the
        // LDS pipe here is a round trip per pass, 2.9 k of an item's 21 k cycles in the s_memtime profile)
        const int j0 = i0 / 25;
        const int d0 = __builtin_amdgcn_readlane(x.det, j0), d1 = __builtin_amdgcn_readlane(x.det, j0 + 1);
        const int d2 = __builtin_amdgcn_readlane(x.det, j0 + 2), d3 = __builtin_amdgcn_readlane(x.det, min(j0 + 3, 63));
        const int dj = j - j0;
        const int det = dj == 0 ? d0 : dj == 1 ? d1 : dj == 2 ? d2 : d3;
        if (idx < nchunk && rem < 24)
            glds16_g(a.msg + (size_t)det * a.ld_msg + (rem >> 3) * H + cw0 + 4 * (rem & 7), area + 16u * i0);
    }
}

// The same requests at a fifth of the vector instructions (the loop above is ~50 per pass -- a division, four cross-lane
// reads, three selects, 64-bit address arithmetic -- seven passes per item: a quarter of the item's vector work, measured
// 8 % of its time with the DMA instructions themselves removed).  TWO det rows per pass: lanes 0-23 take the 24 chunks of
// row 2 k, lanes 25-48 those of row 2 k + 1 (lane 24 is the first row's padding chunk), so a lane's (row half, gate,
// chunk) never changes and the LDS image -- 25-chunk rows, a pass = 50 consecutive slots -- is the one the general loop
// writes.
template <int H>
__device__ __forceinline__ void tiled_stage_p2(const GruFwdArgs& a, const TiledIdx& x, int cw0, int lane, uint32_t area) {
    const int sub = lane >= 25 ? 1 : 0, rem = lane - 25 * sub;
    const int goff = (rem >> 3) * H + cw0 + 4 * (rem & 7);
    const bool lane_on = rem < 24 && lane < 49;
#pragma unroll 1
    for (int k = 0; 2 * k < x.nd; ++k) {
        const int d0 = __builtin_amdgcn_readlane(x.det, 2 * k), d1 = __builtin_amdgcn_readlane(x.det, min(2 * k + 1, 63));
        const int det = sub ? d1 : d0;
        if (lane_on && 2 * k + sub < x.nd) glds16_g(a.msg + (size_t)det * a.ld_msg + goff, area + 800u * k);
    }
}

#define FT_STAGE_P tiled_stage_p2
template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gru_fwd_split_tiled(GruFwdArgs a, FwdTiles tl) {
    extern __shared__ float lds[];
    constexpr int H3 = 3 * H, KP = H + 8, NKB = H / 16, CW = H / 32, NQ4 = H / 8;
    uint16_t* sW = reinterpret_cast<uint16_t*>(lds);                  // [3][3H][KP]
    for (int i = threadIdx.x; i < H * H3 / 4; i += WPB * 64) {
        const int k = i / (H3 / 4), j0 = (i % (H3 / 4)) * 4;
        const float4 w = *reinterpret_cast<const float4*>(a.whh_t + (size_t)k * H3 + j0);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t q1, q2, q3;
            split1(wv[e], q1, q2, q3);
            sW[(0 * H3 + j0 + e) * KP + k] = q1;
            sW[(1 * H3 + j0 + e) * KP + k] = q2;
            sW[(2 * H3 + j0 + e) * KP + k] = q3;
        }
    }
    float* area_base = reinterpret_cast<float*>(sW + 3 * H3 * KP);
    int* next_item = reinterpret_cast<int*>(area_base + WPB * TP_AREA);
    float* sBias = reinterpret_cast<float*>(next_item + 4);
    for (int i = threadIdx.x; i < H; i += WPB * 64) {
        sBias[i] = a.b_ih[i] + a.b_hh[i];
        sBias[H + i] = a.b_ih[H + i] + a.b_hh[H + i];
        sBias[2 * H + i] = a.b_ih[2 * H + i];
        sBias[3 * H + i] = a.b_hh[2 * H + i];
        sBias[4 * H + i] = a.logit_part ? a.w_head[i] : 0.f;
    }
    if (threadIdx.x == 0) *next_item = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    {
        const int grp = __builtin_amdgcn_readfirstlane(wave) >> 2;
        if (grp == 0) __builtin_amdgcn_s_setprio(3);
        else if (grp == 1) __builtin_amdgcn_s_setprio(2);
        else if (grp == 2) __builtin_amdgcn_s_setprio(1);
    }
    const int items_total = tl.T * CW;
    const int per_block = (items_total + gridDim.x - 1) / gridDim.x;
    const int item_lo = blockIdx.x * per_block;
    const int item_hi = min(items_total, item_lo + per_block);

    int item = 0;
    if (lane == 0) item = atomicAdd(next_item, 1);
    item = __builtin_amdgcn_readfirstlane(item) + item_lo;
    if (item >= item_hi) return;
    int cw0 = (item % CW) * 32;
    int t = item / CW;
    TiledIdx ix = tiled_idx(tl, a.R, t, c, lane);
    float* stg = area_base + wave * TP_AREA;                   // staged P rows, then the output staging tile
    const uint32_t area = lds_addr_g(stg);
    if (ix.nd <= TCAP) FT_STAGE_P<H>(a, ix, cw0, lane, area);
    float4 raw[NQ4];
    {
        const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)ix.row * a.ld_h + (H / 2) * half);
#pragma unroll
        for (int i = 0; i < NQ4; ++i) raw[i] = xr[i];
    }

    for (;;) {
        int nitem = 0;
        if (lane == 0) nitem = atomicAdd(next_item, 1);
        nitem = __builtin_amdgcn_readfirstlane(nitem) + item_lo;
        const bool nvalid = nitem < item_hi;
        const int ncw0 = (nitem % CW) * 32;
        const int nt = nvalid ? nitem / CW : t;
        const int r0 = 32 * t;
        const int row = ix.row;
        const bool staged = ix.nd <= TCAP;
        Split8 b[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) b[kb] = split8(raw[2 * kb], raw[2 * kb + 1]);
        const TiledIdx nix = tiled_idx(tl, a.R, nt, c, lane);
        if (nvalid) {
            const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)nix.row * a.ld_h + (H / 2) * half);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) raw[i] = xr[i];
        }
        f32x16 acc_r, acc_z, acc_hn, acc_in;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_r[i] = 0.f; acc_z[i] = 0.f; acc_hn[i] = 0.f; }
        {
            const uint16_t* wp0 = sW + (cw0 + c) * KP + (H / 2) * half;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint16_t* wp = wp0 + 8 * kb;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const uint4 w1 = *reinterpret_cast<const uint4*>(wp + (g * H) * KP);
                    const uint4 w2 = *reinterpret_cast<const uint4*>(wp + (H3 + g * H) * KP);
                    const uint4 w3 = *reinterpret_cast<const uint4*>(wp + (2 * H3 + g * H) * KP);
                    if (g == 0) acc_r = mfma_x6(w1, w2, w3, b[kb], acc_r);
                    else if (g == 1) acc_z = mfma_x6(w1, w2, w3, b[kb], acc_z);
                    else acc_hn = mfma_x6(w1, w2, w3, b[kb], acc_hn);
                }
            }
        }
        // the staged P rows were requested an item ago; only the loads issued since (the next item's operand and index
        // loads, at least NQ4 of them) may still be in flight
        if (staged) {
            if (nvalid) {
                if constexpr (NQ4 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        float4 hp4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            hp4[q] = *reinterpret_cast<const float4*>(a.h + (size_t)row * a.ld_h + cw0 + 8 * q + 4 * half);
        f32x16 outv;
        {
            float4 gs[4], gd[4], hs[4], hd[4];
            const int ls = ix.loc & 0xFFFF, ld_ = ix.loc >> 16;
            // the tile's det list sits one entry per lane: the global det index of a position is a cross-lane read
            const int sdet = __shfl(ix.det, ls, 64), ddet = __shfl(ix.det, ld_, 64);
            const float* ps0 = a.msg + (size_t)sdet * a.ld_msg + cw0 + 4 * half;       // (the gather path of big tiles)
            const float* pd0 = a.msg + (size_t)ddet * a.ld_msg + cw0 + 4 * half;
            if (staged) {
                const float* ls0 = stg + ls * TP_LD + 4 * half;
                const float* ld0 = stg + ld_ * TP_LD + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ls0 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(ld0 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) hs[q] = *reinterpret_cast<const float4*>(ls0 + 32 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) hd[q] = *reinterpret_cast<const float4*>(ld0 + 32 + 8 * q);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ps0 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(pd0 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) hs[q] = *reinterpret_cast<const float4*>(ps0 + H + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) hd[q] = *reinterpret_cast<const float4*>(pd0 + H + 8 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // r: pre-activation in place
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + cw0 + 8 * q + 4 * half);
                acc_r[4 * q + 0] = acc_r[4 * q + 0] + (gs[q].x - gd[q].x) + b4.x;
                acc_r[4 * q + 1] = acc_r[4 * q + 1] + (gs[q].y - gd[q].y) + b4.y;
                acc_r[4 * q + 2] = acc_r[4 * q + 2] + (gs[q].z - gd[q].z) + b4.z;
                acc_r[4 * q + 3] = acc_r[4 * q + 3] + (gs[q].w - gd[q].w) + b4.w;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (staged) {
                const float* ls0 = stg + ls * TP_LD + 4 * half + 64;
                const float* ld0 = stg + ld_ * TP_LD + 4 * half + 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ls0 + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(ld0 + 8 * q);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) gs[q] = *reinterpret_cast<const float4*>(ps0 + 2 * H + 8 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) gd[q] = *reinterpret_cast<const float4*>(pd0 + 2 * H + 8 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // z
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + H + cw0 + 8 * q + 4 * half);
                acc_z[4 * q + 0] = acc_z[4 * q + 0] + (hs[q].x - hd[q].x) + b4.x;
                acc_z[4 * q + 1] = acc_z[4 * q + 1] + (hs[q].y - hd[q].y) + b4.y;
                acc_z[4 * q + 2] = acc_z[4 * q + 2] + (hs[q].z - hd[q].z) + b4.z;
                acc_z[4 * q + 3] = acc_z[4 * q + 3] + (hs[q].w - hd[q].w) + b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {                       // n: input part
                const float4 b4 = *reinterpret_cast<const float4*>(sBias + 2 * H + cw0 + 8 * q + 4 * half);
                acc_in[4 * q + 0] = (gs[q].x - gd[q].x) + b4.x;
                acc_in[4 * q + 1] = (gs[q].y - gd[q].y) + b4.y;
                acc_in[4 * q + 2] = (gs[q].z - gd[q].z) + b4.z;
                acc_in[4 * q + 3] = (gs[q].w - gd[q].w) + b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4h = *reinterpret_cast<const float4*>(sBias + 3 * H + cw0 + 8 * q + 4 * half);
                const float bh[4] = {b4h.x, b4h.y, b4h.z, b4h.w};
                const float hp[4] = {hp4[q].x, hp4[q].y, hp4[q].z, hp4[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int reg = 4 * q + i;
                    const float ro = sigmoidf_(acc_r[reg]);
                    const float zo = sigmoidf_(acc_z[reg]);
                    const float ho = acc_hn[reg] + bh[i];
                    const float no = tanhf_(acc_in[reg] + ro * ho);
                    outv[reg] = (1.0f - zo) * no + zo * hp[i];
                    acc_r[reg] = ro; acc_z[reg] = zo; acc_hn[reg] = ho; acc_in[reg] = no;
                }
            }
        }
        if (a.logit_part) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(sBias + 4 * H + cw0 + 8 * q + 4 * half);
                p += outv[4 * q] * w.x + outv[4 * q + 1] * w.y + outv[4 * q + 2] * w.z + outv[4 * q + 3] * w.w;
            }
            p += __shfl_xor(p, 32);
            if (half == 0 && r0 + c < a.R) a.logit_part[(size_t)(cw0 / 32) * a.part_stride + row] = p;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of the staged P rows is back: the area is free
        stage_store32(stg, c, half, lane, outv, a.h_out, a.ld_out, cw0, row, r0, a.R);
        if (a.gates) {
            stage_store32<true>(stg, c, half, lane, acc_r, a.gates, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_z, a.gates + a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_in, a.gates + 2 * a.gate_plane, H, cw0, row, r0, a.R);
            stage_store32<true>(stg, c, half, lane, acc_hn, a.gates + 3 * a.gate_plane, H, cw0, row, r0, a.R);
        }
        if (!nvalid) break;
        cw0 = ncw0; t = nt; ix = nix;
        if (ix.nd <= TCAP) FT_STAGE_P<H>(a, ix, cw0, lane, area);      // (the staging tile's last reads are back: stage_store32)
    }
}


// ---- the tiled forward at FOUR waves per SIMD (edge tiles of 16 rows) ------------------------------------------------------
// The s_memtime profile of k_gru_fwd_split_tiled (tools/fwd_timeline.py) shows what holds it: an item is one dependent
// chain per wave -- operand wait, split, 72 MFMAs (14 % of the item's time), P reads, gate arithmetic, five planes through
// the staging tile, the next item's DMA requests -- and with 250 registers a SIMD holds two such chains; the vector ALU is
// ~45 % busy, the matrix pipe ~20 %, and every in-order vmcnt wait on a fresh index load also waits for the 20 stores in
// front of it.  Here an item is 16 rows x 32 columns on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand:
//   * lane (n = lane & 15, kg = lane >> 4) holds FOUR CONSECUTIVE COLUMNS of row n of each 16 x 16 result tile, so the five
//     output planes leave as 16-byte stores straight from the accumulators -- no staging tile, no LDS round trips, no
//     cross-lane row ids (a lane stores to its own row);
//   * half the accumulators, operands and epilogue values per wave: <= 128 registers, sixteen waves per CU;
//   * the tile descriptor (row, endpoint positions, det list) is requested TWO items ahead, the operand rows one item
//     ahead from a row id that landed an item ago: no wait of the loop is on a load younger than an item.
// K order inside a product differs from the 32 x 32 x 16 form (32 k per MFMA), so results match k_gru_fwd_split to
// rounding, not bit for bit (both are the fp32-accurate bf16x6 split; tests bound both against fp64).
// MEASURED (C2 stage graph, 6.03 M rows, same box): 2.74 ms with gates + head against 2.55 for the 32-row kernel; 1.51
// against 1.59 without the gate planes.  Twice the waves did not shorten the launch: what a wave waits for is not latency
// it could hide behind its neighbours but its own stores -- vmcnt retires loads and stores in issue order, so the first
// wait on any load issued after an item's ten 1-KB stores is a wait for those stores, for every wave of the CU at once.
// Kept as TMPNN_FWD_TILE_ROWS=16 (tiles of 16 rows); the default stays the 32-row kernel.  DESIGN.md section 4.
static constexpr int T16_CAP = 12;                             // dets staged per item (C2 tiles: 8-10)
static constexpr int T16_AREA = T16_CAP * TP_LD;               // floats per wave

__device__ __forceinline__ f32x4 mfma16_x6(const uint4& a1, const uint4& a2, const uint4& a3, const Split8& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a3), __builtin_bit_cast(bf16x8, b.p1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b.p3), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, b.p2), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, b.p1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b.p2), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b.p1), c, 0, 0, 0);
    return c;
}
// descriptor of tile t for lane (n, .), in two parts so that no load of the loop depends on a load of the same iteration:
// t16_idx (row, endpoint positions, det-list bounds: three loads) two items ahead, t16_det (the list entry of this lane: one
// load, from bounds that landed an item ago) one item ahead.  Indices are clamped, never skipped: the counted waits rely
// on a fixed number of memory operations per iteration.
struct T16Idx { int row, loc, dp0, nd, det; };
__device__ __forceinline__ T16Idx t16_idx(const FwdTiles& tl, int R, int t, int n) {
    T16Idx x;
    const int li = min(16 * t + n, R - 1);
    x.row = tl.t_row[li];
    x.loc = tl.t_loc[li];
    typedef int i32x2u __attribute__((ext_vector_type(2), aligned(4)));
    const i32x2u dp = *reinterpret_cast<const i32x2u*>(tl.t_dptr + t);      // ONE 8-byte load (hipcc merges the pair anyway)
    x.dp0 = dp[0];
    x.nd = dp[1] - dp[0];
    x.det = 0;
    return x;
}
__device__ __forceinline__ int t16_det(const FwdTiles& tl, const T16Idx& x, int lane) {
    return tl.t_dets[x.dp0 + min(lane, max(x.nd - 1, 0))];
}
__device__ __forceinline__ void nt_store4(float* p, const f32x4& v) {
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}

template <int H, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gru_fwd_split_t16(GruFwdArgs a, FwdTiles tl) {
    extern __shared__ float lds[];
    constexpr int H3 = 3 * H, NKS = H / 32, CW = H / 32;
    // weight pieces as the A operand's fragments: sW[piece][k step s][k group kg][column][8 k] (k = 8 NKS kg + 8 s + j), 16 bytes
    // per (column, fragment), no padding.  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ...: sixteen consecutive columns of TWO k groups -- sixteen consecutive 16-byte slots of planes that lie a
    // multiple of 256 bytes apart: conflict-free (a [column][k] image with any row pitch is 2-way on these groups).
    uint16_t* sW = reinterpret_cast<uint16_t*>(lds);
    constexpr int PLANE = H3 * 8;                                     // elements per (piece, s, kg) plane
    for (int i = threadIdx.x; i < H * H3 / 4; i += WPB * 64) {
        const int k = i / (H3 / 4), j0 = (i % (H3 / 4)) * 4;
        const float4 w = *reinterpret_cast<const float4*>(a.whh_t + (size_t)k * H3 + j0);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        const int kgp = k / (8 * NKS), sp = (k / 8) % NKS, jp = k & 7;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t q1, q2, q3;
            split1(wv[e], q1, q2, q3);
            const int o = ((sp * 4 + kgp) * H3 + j0 + e) * 8 + jp;
            sW[o] = q1;
            sW[NKS * 4 * PLANE + o] = q2;
            sW[2 * NKS * 4 * PLANE + o] = q3;
        }
    }
    float* area_base = reinterpret_cast<float*>(sW + 3 * NKS * 4 * PLANE);
    int* next_item = reinterpret_cast<int*>(area_base + WPB * T16_AREA);
    float* sBias = reinterpret_cast<float*>(next_item + 4);
    for (int i = threadIdx.x; i < H; i += WPB * 64) {
        sBias[i] = a.b_ih[i] + a.b_hh[i];
        sBias[H + i] = a.b_ih[H + i] + a.b_hh[H + i];
        sBias[2 * H + i] = a.b_ih[2 * H + i];
        sBias[3 * H + i] = a.b_hh[2 * H + i];
        sBias[4 * H + i] = a.logit_part ? a.w_head[i] : 0.f;
    }
    if (threadIdx.x == 0) *next_item = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, kg = lane >> 4;
    const int items_total = tl.T * CW;
    const int per_block = (items_total + gridDim.x - 1) / gridDim.x;
    const int item_lo = blockIdx.x * per_block;
    const int item_hi = min(items_total, item_lo + per_block);
    auto claim = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(next_item, 1);
        return __builtin_amdgcn_readfirstlane(v) + item_lo;
    };
    int item = claim();
    if (item >= item_hi) return;
    int nitem = claim();
    const int last_t = (item_hi - 1) / CW;                           // (descriptors of items past the range: a valid tile's)
    int t = item / CW, cw0 = (item % CW) * 32;
    int nt = nitem < item_hi ? nitem / CW : last_t;
    T16Idx ix = t16_idx(tl, a.R, t, n);
    T16Idx nix = t16_idx(tl, a.R, nt, n);
    ix.det = t16_det(tl, ix, lane);
    float* stg = area_base + wave * T16_AREA;
    const uint32_t area = lds_addr_g(stg);
    auto stage_p = [&](const T16Idx& x, int c0) {
        if (x.nd <= T16_CAP) tiled_stage_p<H>(a, TiledIdx{x.row, x.loc, x.det, x.nd}, c0, lane, area);
    };
    stage_p(ix, cw0);
    float4 raw[2 * NKS];
    {
        const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)ix.row * a.ld_h + 8 * NKS * kg);
#pragma unroll
        for (int i = 0; i < 2 * NKS; ++i) raw[i] = xr[i];
    }
    // this item's previous state (the merge term): lane (n, kg) takes the columns it will hold after the matrix phase
    f32x4 hp4[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        hp4[ct] = *reinterpret_cast<const f32x4*>(a.h + (size_t)ix.row * a.ld_h + cw0 + 16 * ct + 4 * kg);
    // stores an item issues (h_out, four gate planes, head partial): all in flight when the next item waits for its P rows
    const int nst = __builtin_amdgcn_readfirstlane(2 + (a.gates ? 8 : 0) + (a.logit_part ? 1 : 0));
    bool first = true;
    // the partner lane n ^ 8 of the output exchange below, and what this lane writes in the two store passes
    const bool lo = n < 8;
    const int colw = 16 * (n >> 3) + 4 * kg;                       // pass A: rows n & 7, pass B: rows 8 + (n & 7); 8 lanes = 128 B
    for (;;) {
        const bool nvalid = nitem < item_hi;
        const int ncw0 = (nitem % CW) * 32;
        const int row = ix.row;
        const bool live = 16 * t + n < a.R;
        const bool staged = ix.nd <= T16_CAP;
        // ---- requests first: the next item's operand rows (its row id landed an item ago) and det-list entry, the
        // descriptor of the item after that.  At least 2 NKS + 1 + 3 memory operations (the counted wait below assumes the
        // minimum: more operations in flight only make it stricter).
        Split8 b[NKS];
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) b[s_] = split8(raw[2 * s_], raw[2 * s_ + 1]);
        {
            const float4* xr = reinterpret_cast<const float4*>(a.h + (size_t)nix.row * a.ld_h + 8 * NKS * kg);
#pragma unroll
            for (int i = 0; i < 2 * NKS; ++i) raw[i] = xr[i];
        }
        nix.det = t16_det(tl, nix, lane);
        const int n2item = claim();
        const int n2t = n2item < item_hi ? n2item / CW : last_t;
        const T16Idx n2ix = t16_idx(tl, a.R, n2t, n);
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase: 3 gates x 2 column tiles x NKS steps x 6 products
        f32x4 acc[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[g][ct][i] = 0.f;
        {
            const uint16_t* wp0 = sW + (kg * H3 + cw0 + n) * 8;
#pragma unroll
            for (int s_ = 0; s_ < NKS; ++s_)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const uint16_t* wp = wp0 + s_ * 4 * PLANE + (g * H + 16 * ct) * 8;
                        const uint4 w1 = *reinterpret_cast<const uint4*>(wp);
                        const uint4 w2 = *reinterpret_cast<const uint4*>(wp + NKS * 4 * PLANE);
                        const uint4 w3 = *reinterpret_cast<const uint4*>(wp + 2 * NKS * 4 * PLANE);
                        acc[g][ct] = mfma16_x6(w1, w2, w3, b[s_], acc[g][ct]);
                    }
        }
        // The staged P rows and the previous state of THIS item were requested before the previous item's stores: they
        // have landed once at most those stores and the 2 NKS + 4 requests above are in flight (vmcnt counts loads and stores
        // in issue order).
        if (first) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (NKS == 2) {
            if (nst == 11) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
            else if (nst == 10) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            else if (nst == 3) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            if (nst == 11) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else if (nst == 10) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (nst == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        first = false;
        f32x4 outv[2], o_r[2], o_z[2], o_n[2], o_hn[2];
        {
            const int ls = ix.loc & 0xFFFF, ld_ = ix.loc >> 16;
            const int sdet = __shfl(ix.det, ls, 64), ddet = __shfl(ix.det, ld_, 64);      // (the gather path of big tiles)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int co = 16 * ct + 4 * kg;
                f32x4 gi[3];
                if (staged) {
                    const float* ls0 = stg + ls * TP_LD + co;
                    const float* ld0 = stg + ld_ * TP_LD + co;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        gi[g] = *reinterpret_cast<const f32x4*>(ls0 + 32 * g) - *reinterpret_cast<const f32x4*>(ld0 + 32 * g);
                } else {
                    const float* ps0 = a.msg + (size_t)sdet * a.ld_msg + cw0 + co;
                    const float* pd0 = a.msg + (size_t)ddet * a.ld_msg + cw0 + co;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        gi[g] = *reinterpret_cast<const f32x4*>(ps0 + g * H) - *reinterpret_cast<const f32x4*>(pd0 + g * H);
                }
                const f32x4 b_r = *reinterpret_cast<const f32x4*>(sBias + cw0 + co);
                const f32x4 b_z = *reinterpret_cast<const f32x4*>(sBias + H + cw0 + co);
                const f32x4 b_n = *reinterpret_cast<const f32x4*>(sBias + 2 * H + cw0 + co);
                const f32x4 b_h = *reinterpret_cast<const f32x4*>(sBias + 3 * H + cw0 + co);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ro = sigmoidf_(acc[0][ct][i] + gi[0][i] + b_r[i]);
                    const float zo = sigmoidf_(acc[1][ct][i] + gi[1][i] + b_z[i]);
                    const float ho = acc[2][ct][i] + b_h[i];
                    const float no = tanhf_((gi[2][i] + b_n[i]) + ro * ho);
                    outv[ct][i] = (1.0f - zo) * no + zo * hp4[ct][i];
                    o_r[ct][i] = ro; o_z[ct][i] = zo; o_n[ct][i] = no; o_hn[ct][i] = ho;
                }
            }
        }
        float p_head = 0.f;
        if (a.logit_part) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(sBias + 4 * H + cw0 + 16 * ct + 4 * kg);
                p_head += outv[ct][0] * w[0] + outv[ct][1] * w[1] + outv[ct][2] * w[2] + outv[ct][3] * w[3];
            }
            p_head += __shfl_xor(p_head, 16);
            p_head += __shfl_xor(p_head, 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every read of the staged P rows is back: the area is free
        __builtin_amdgcn_sched_barrier(0);
        // ---- the NEXT item's P rows and previous state are requested BEFORE this item's stores
        const int cw_st = cw0, t_st = t;
        if (nvalid) {
            cw0 = ncw0; t = nt; ix = nix; nix = n2ix; nt = n2t; nitem = n2item;
            stage_p(ix, cw0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                hp4[ct] = *reinterpret_cast<const f32x4*>(a.h + (size_t)ix.row * a.ld_h + cw0 + 16 * ct + 4 * kg);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- stores: lanes n and n ^ 8 exchange one column tile, so that eight lanes cover 128 contiguous bytes of a row
        // and a store instruction writes 8 rows x 128 B (left as the MFMA leaves them it is 16 rows x 64 B: measured 3.4
        // against 2.7 ms per launch)
        {
            constexpr int ROR8 = 0x128;                              // DPP row_ror:8 = lane n ^ 8 of the 16-lane row
            const int prow = __builtin_amdgcn_update_dpp(0, row, ROR8, 0xF, 0xF, false);
            const int plive = __builtin_amdgcn_update_dpp(0, (int)live, ROR8, 0xF, 0xF, false);
            const int rowA = lo ? row : prow, rowB = lo ? prow : row;
            const bool liveA = lo ? live : (plive != 0), liveB = lo ? (plive != 0) : live;
            auto xchg = [&](const f32x4 (&v)[2], f32x4& va, f32x4& vb) {
                f32x4 rcv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float snd = lo ? v[1][i] : v[0][i];
                    rcv[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(snd), ROR8, 0xF, 0xF, false));
                    va[i] = lo ? v[0][i] : rcv[i];
                    vb[i] = lo ? rcv[i] : v[1][i];
                }
            };
            f32x4 va, vb;
            xchg(outv, va, vb);
            if (liveA) *reinterpret_cast<f32x4*>(a.h_out + (size_t)rowA * a.ld_out + cw_st + colw) = va;
            if (liveB) *reinterpret_cast<f32x4*>(a.h_out + (size_t)rowB * a.ld_out + cw_st + colw) = vb;
            if (a.gates) {
                float* gA = a.gates + (size_t)rowA * H + cw_st + colw;
                float* gB = a.gates + (size_t)rowB * H + cw_st + colw;
                xchg(o_r, va, vb);
                if (liveA) nt_store4(gA, va);
                if (liveB) nt_store4(gB, vb);
                xchg(o_z, va, vb);
                if (liveA) nt_store4(gA + a.gate_plane, va);
                if (liveB) nt_store4(gB + a.gate_plane, vb);
                xchg(o_n, va, vb);
                if (liveA) nt_store4(gA + 2 * a.gate_plane, va);
                if (liveB) nt_store4(gB + 2 * a.gate_plane, vb);
                xchg(o_hn, va, vb);
                if (liveA) nt_store4(gA + 3 * a.gate_plane, va);
                if (liveB) nt_store4(gB + 3 * a.gate_plane, vb);
            }
            if (a.logit_part && kg == 0 && live) a.logit_part[(size_t)(cw_st / 32) * a.part_stride + row] = p_head;
        }
        (void)t_st;
        if (!nvalid) break;
    }
}

// out[r][0:NOUT] = in[rows[r]][0:H] @ Wt[H][NOUT]   (no bias; H <= 64; NOUT multiple of 32).  Used to
// project the det rows once per call (P = h[dets] W_ih^T) for the XMODE 3 forward.
template <int H, int NT>     // NT = NOUT / 32 column tiles, all owned by one wave
__global__ __launch_bounds__(512) void k_rows_gemm_lds(const int32_t* __restrict__ rows, int R,
                                                       const float* __restrict__ in, int ld_in,
                                                       const float* __restrict__ wt, float* __restrict__ out,
                                                       int ld_out, int ntiles) {
    extern __shared__ float lds[];
    constexpr int NOUT = NT * 32;
    float* sW = lds;                                   // [H][NOUT]
    for (int i = threadIdx.x * 4; i < H * NOUT; i += 512 * 4)
        *reinterpret_cast<float4*>(sW + i) = *reinterpret_cast<const float4*>(wt + i);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    float* stg = lds + H * NOUT + wave * (32 * STG_LD);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = (tile * 8 + wave) * 32;
        if (r0 >= R) continue;
        const int li = min(r0 + c, R - 1);
        const int row = rows[li];
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int kt = 0; kt < H / 32; ++kt) {
            const int f0 = kt * 32 + half * 16;
            float av[16];
            load16(in + (size_t)row * ld_in + f0, av);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float* b = sW + (f0 + s) * NOUT + c;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma32(b[t * 32], av[s], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            stage_store32(stg, c, half, lane, acc[t], out, ld_out, t * 32, li, r0, R);   // compact rows: list position
        }
    }
}

// The same projection on the bf16 pipe (bf16x6, see mfma_x6): the weight pieces sit in LDS as [piece][out col][k]
// bf16 with k contiguous (rows padded by 16 bytes: conflict-free ds_read_b128), a lane's 8 operand values per
// MFMA are 8 consecutive k of its row.  k is enumerated as k = (H/2)*(lane>>5) + 8*kb + j, so a lane reads H/2
// CONTIGUOUS floats of its row.
// General form (round 4: also the attention projections): out[orow(r)][0:NOUT] (=|+=) in[irow(r)][0:H] @ W, with
//   W[k][n] = wt[k * ld_wt + n]  (wt_trans = 0)  or  wt[n * ld_wt + k]  (wt_trans = 1: a weight used transposed, no copy),
//   irow(r) = rows ? rows[r] : r,  orow(r) = out_rows ? out_rows[r] : r.
template <int H, int NT>
__global__ __launch_bounds__(512) void k_rows_gemm_split(const int32_t* __restrict__ rows, int R,
                                                         const float* __restrict__ in, int ld_in,
                                                         const float* __restrict__ wt, int ld_wt, int wt_trans,
                                                         float* __restrict__ out, int ld_out,
                                                         const int32_t* __restrict__ out_rows, int accumulate, int ntiles) {
    extern __shared__ float lds[];
    constexpr int NOUT = NT * 32, KP = H + 8, NKB = H / 16;
    uint16_t* sW = reinterpret_cast<uint16_t*>(lds);          // [3][NOUT][KP]
    if (!wt_trans) {
        for (int i = threadIdx.x; i < H * NOUT / 4; i += 512) {
            const int k = i / (NOUT / 4), j0 = (i % (NOUT / 4)) * 4;
            const float4 w = *reinterpret_cast<const float4*>(wt + (size_t)k * ld_wt + j0);
            const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint16_t q1, q2, q3;
                split1(wv[e], q1, q2, q3);
                sW[(0 * NOUT + j0 + e) * KP + k] = q1;
                sW[(1 * NOUT + j0 + e) * KP + k] = q2;
                sW[(2 * NOUT + j0 + e) * KP + k] = q3;
            }
        }
    } else {
        for (int i = threadIdx.x; i < H * NOUT; i += 512) {
            const int n = i / H, k = i % H;
            uint16_t q1, q2, q3;
            split1(wt[(size_t)n * ld_wt + k], q1, q2, q3);
            sW[(0 * NOUT + n) * KP + k] = q1;
            sW[(1 * NOUT + n) * KP + k] = q2;
            sW[(2 * NOUT + n) * KP + k] = q3;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, half = lane >> 5;
    float* stg = reinterpret_cast<float*>(sW + 3 * NOUT * KP) + wave * (32 * STG_LD);
    // the wave's rows of the NEXT tile are requested before this tile's matrix phase (round 6: the loop used to load, wait,
    // compute and store tile by tile)
    constexpr int NQ4 = H / 8;
    float4 raw[NQ4];
    int row_n = 0, orow_n = 0;
    auto request = [&](int tile) {
        const int r0 = (tile * 8 + wave) * 32;
        if (tile < ntiles && r0 < R) {
            const int li = min(r0 + c, R - 1);
            row_n = rows ? rows[li] : li;
            orow_n = out_rows ? out_rows[li] : li;
            const float4* xr = reinterpret_cast<const float4*>(in + (size_t)row_n * ld_in + (H / 2) * half);
#pragma unroll
            for (int i = 0; i < NQ4; ++i) raw[i] = xr[i];
        }
    };
    request(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = (tile * 8 + wave) * 32;
        if (r0 >= R) continue;                     // (only the last tile: nothing was requested for it, nothing follows)
        const int orow = orow_n;
        Split8 b[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) b[kb] = split8(raw[2 * kb], raw[2 * kb + 1]);
        request(tile + gridDim.x);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint16_t* wp = sW + (t * 32 + c) * KP + (H / 2) * half + 8 * kb;
                const uint4 w1 = *reinterpret_cast<const uint4*>(wp);
                const uint4 w2 = *reinterpret_cast<const uint4*>(wp + NOUT * KP);
                const uint4 w3 = *reinterpret_cast<const uint4*>(wp + 2 * NOUT * KP);
                acc[t] = mfma_x6(w1, w2, w3, b[kb], acc[t]);
            }
        }
        if (accumulate) {
#pragma unroll
            for (int t = 0; t < NT; ++t) stage_store32<false, true>(stg, c, half, lane, acc[t], out, ld_out, t * 32, orow, r0, R);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) stage_store32(stg, c, half, lane, acc[t], out, ld_out, t * 32, orow, r0, R);
        }
    }
}


bool rows_gemm_supported(int KD, int NOUT) {
    if (!split_enabled() || NOUT <= 0 || NOUT % 32) return false;
    const int nt = NOUT / 32;
    return (KD == 64 && (nt == 1 || nt == 2 || nt == 4 || nt == 6)) || (KD == 32 && nt >= 1 && nt <= 3) || (KD == 128 && (nt == 1 || nt == 2));
}
int launch_rows_gemm(const int32_t* rows, int R, const float* in, int ld_in, int KD, const float* wt, int ld_wt, int wt_trans,
                     int NOUT, float* out, int ld_out, const int32_t* out_rows, int accumulate, hipStream_t st) {
    if (R <= 0) return TMPNN_OK;
    if (!rows_gemm_supported(KD, NOUT)) return set_error(TMPNN_EINVAL, "rows_gemm: unsupported shape KD=%d NOUT=%d", KD, NOUT);
    TM_REQUIRE(in && wt && out && ld_in >= KD && (ld_in & 3) == 0 && aligned16(in) && (ld_out & 3) == 0 && aligned16(out) &&
                   (wt_trans || ((ld_wt & 3) == 0 && aligned16(wt))), "rows_gemm: rows must be 16-byte aligned");
    const int ntiles = ceil_div(R, 256);
    dim3 grid(ntiles < 256 ? ntiles : 256), block(512);
    const size_t shm2 = (size_t)3 * NOUT * (KD + 8) * 2 + sizeof(float) * 8 * 32 * STG_LD;
#define RG(HH, NN)                                                                                                  \
    do {                                                                                                            \
        TM_SHM_ONCE((k_rows_gemm_split<HH, NN>), shm2);                                                             \
        hipLaunchKernelGGL((k_rows_gemm_split<HH, NN>), grid, block, shm2, st, rows, R, in, ld_in, wt, ld_wt, wt_trans, out, \
                           ld_out, out_rows, accumulate, ntiles);                                                   \
    } while (0)
    const int nt = NOUT / 32;
    if (KD == 128)     { if (nt == 1) RG(128, 1); else RG(128, 2); }          // (d_h += d_ha Wcat^T over all heads of an attention call)
    else if (KD == 64) { if (nt == 1) RG(64, 1); else if (nt == 2) RG(64, 2); else if (nt == 4) RG(64, 4); else RG(64, 6); }
    else               { if (nt == 1) RG(32, 1); else if (nt == 2) RG(32, 2); else RG(32, 3); }
#undef RG
    return check_launch("rows_gemm_split");
}

}  // namespace tmpnn

using namespace tmpnn;

extern "C" {


int tmpnn_gru_fwd_head_parts(int H, int IN, int xmode) {
    // column waves per row tile of the LDS-resident kernel; 0 where that kernel cannot run
    if (H != 32 && H != 64) return 0;
    const size_t wpb = (H == 64) ? 12 : 8;
    const size_t shm = sizeof(float) * ((size_t)((xmode == 3 ? 0 : IN) + H) * 3 * H + wpb * 32 * STG_LD + 4);
    return shm > 160 * 1024 ? 0 : H / 32;
}

int tmpnn_gru_fwd(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst, const float* msg,
                  int ld_msg, int msg_compact, int IN, const float* h, int ld_h, int H, const float* wih_t, const float* whh_t,
                  const float* b_ih, const float* b_hh, float* h_out, int ld_out, float* gates, size_t gate_plane,
                  const float* w_head, float* logit_part, size_t part_stride, tmpnn_stream stream) {
    TM_REQUIRE(supported_H_cell(H), "gru_fwd: unsupported H=%d", H);
    TM_REQUIRE(R >= 0, "gru_fwd: R=%d", R);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(xmode >= 0 && xmode <= 3, "gru_fwd: xmode=%d", xmode);
    TM_REQUIRE(IN == (xmode == 2 ? 2 * H : (xmode == 1 || xmode == 3 ? H : IN)) && IN % 32 == 0 && IN > 0,
               "gru_fwd: IN=%d does not match xmode=%d H=%d", IN, xmode, H);
    TM_REQUIRE(rows && h && whh_t && b_ih && b_hh && h_out && (wih_t || xmode == 3), "gru_fwd: null pointer");
    TM_REQUIRE(xmode == 0 ? (msg != nullptr && ld_msg >= IN && (ld_msg & 3) == 0 && aligned16(msg))
                          : (src != nullptr && dst != nullptr),
               "gru_fwd: message source missing/misaligned for xmode=%d", xmode);
    TM_REQUIRE(xmode != 3 || (H <= 64 && msg != nullptr && ld_msg >= 3 * H && (ld_msg & 3) == 0 && aligned16(msg)),
               "gru_fwd: xmode 3 needs H <= 64 and the projected det rows msg [Dn][3H]");
    TM_REQUIRE(ld_h >= H && ld_out >= H && (ld_h & 3) == 0 && aligned16(h), "gru_fwd: bad state layout");
    TM_REQUIRE(gates == nullptr || gate_plane >= (size_t)H, "gru_fwd: gate_plane too small");
    TM_REQUIRE(logit_part == nullptr || (w_head != nullptr && aligned16(w_head) && tmpnn_gru_fwd_head_parts(H, IN, xmode) > 0),
               "gru_fwd: fused head not available for H=%d IN=%d xmode=%d (see tmpnn_gru_fwd_head_parts)", H, IN, xmode);
    GruFwdArgs a{rows, R, src, dst, msg, ld_msg, IN, msg_compact, h, ld_h, H, wih_t, whh_t, b_ih, b_hh, h_out, ld_out, gates,
                 gate_plane, w_head, logit_part, part_stride};
    hipStream_t st = as_stream(stream);
    if (H <= 64 && (xmode == 3 || aligned16(wih_t)) && aligned16(whh_t) && aligned16(h_out) && (ld_out & 3) == 0 && aligned16(b_ih) &&
        aligned16(b_hh) && (gates == nullptr || (aligned16(gates) && (gate_plane & 3) == 0))) {
        // weights resident in LDS, persistent 8-wave blocks (see k_gru_fwd_lds)
        // H = 64: 12 waves per block (3 per SIMD), two per 32-row tile (one per 32-column half) -> 192 rows per pass;
        // H = 32: 8 waves, one per row tile
        const int rows_per_pass = (H == 64) ? 192 : 256;
        const int ntiles = ceil_div(R, rows_per_pass);
        dim3 pgrid(ntiles < 256 ? ntiles : 256), pblock(H == 64 ? 768 : 512);
        const int wpb = (H == 64) ? 12 : 8;
        if (xmode == 3 && split_enabled()) {
            // bf16x6 operand path, 8 waves (two per SIMD: the prefetched next operand needs the registers)
            const size_t shm2 = (size_t)3 * 3 * H * (H + 8) * 2 + sizeof(float) * ((size_t)8 * 32 * STG_LD + 4 + 5 * H);
            if (H == 64) {
                TM_SHM_ONCE((k_gru_fwd_split<64, 8>), shm2);
                hipLaunchKernelGGL((k_gru_fwd_split<64, 8>), pgrid, dim3(512), shm2, st, a);
            } else {
                TM_SHM_ONCE((k_gru_fwd_split<32, 8>), shm2);
                hipLaunchKernelGGL((k_gru_fwd_split<32, 8>), pgrid, dim3(512), shm2, st, a);
            }
            return check_launch("gru_fwd_split");
        }
        if (xmode == 0 && IN == H && split_enabled() && aligned16(msg) && aligned16(wih_t)) {
            // the node cell on the bf16 pipe: one 32-column half of the outputs per block (k_gru_fwd_split_node)
            const size_t shmn = (size_t)2 * 3 * 96 * (H + 8) * 2 + sizeof(float) * ((size_t)8 * 32 * STG_LD + 4 + 5 * 32);
            const int cwn = H / 32;
            const int want = ceil_div(ceil_div(R, 32), 8);                  // >= 8 row tiles per group of blocks
            int groups = want < 256 / cwn ? want : 256 / cwn;
            groups = (groups + 7) & ~7;                                      // whole sets of 8 (block pairs b, b + 8 on one XCD)
            dim3 ngrid(groups * cwn);
            if (H == 64) {
                TM_SHM_ONCE((k_gru_fwd_split_node<64, 8>), shmn);
                hipLaunchKernelGGL((k_gru_fwd_split_node<64, 8>), ngrid, dim3(512), shmn, st, a);
            } else {
                TM_SHM_ONCE((k_gru_fwd_split_node<32, 8>), shmn);
                hipLaunchKernelGGL((k_gru_fwd_split_node<32, 8>), ngrid, dim3(512), shmn, st, a);
            }
            return check_launch("gru_fwd_split_node");
        }
        const size_t shm = sizeof(float) * ((size_t)((xmode == 3 ? 0 : IN) + H) * 3 * H + (size_t)wpb * 32 * STG_LD + 4);
        if (shm > 160 * 1024) goto generic;      // e.g. concat at H = 64: the weights alone take 144 KiB
#define LL(HH, II, X, CC, WW)                                                                                \
    do {                                                                                                     \
        TM_SHM_ONCE((k_gru_fwd_lds<HH, II, X, CC, WW>), shm);                     \
        hipLaunchKernelGGL((k_gru_fwd_lds<HH, II, X, CC, WW>), pgrid, pblock, shm, st, a, ntiles);           \
    } while (0)
        if (H == 64) { if (xmode == 0 && IN == 64) LL(64, 64, 0, 1, 12); else if (xmode == 1) LL(64, 64, 1, 1, 12); else if (xmode == 2) LL(64, 128, 2, 1, 12); else if (xmode == 3) LL(64, 64, 3, 1, 12); else goto generic; }
        else         { if (xmode == 0 && IN == 32) LL(32, 32, 0, 1, 8); else if (xmode == 1) LL(32, 32, 1, 1, 8); else if (xmode == 2) LL(32, 64, 2, 1, 8); else if (xmode == 3) LL(32, 32, 3, 1, 8); else goto generic; }
#undef LL
        return check_launch("gru_fwd_lds");
    }
generic:
    TM_REQUIRE(logit_part == nullptr, "gru_fwd: fused head requested but the LDS path is unavailable (alignment)");
    TM_REQUIRE(xmode != 3, "gru_fwd: xmode 3 is only available on the LDS path (H <= 64, 16-byte aligned buffers)");
    // (a wave walks the whole K range alone: on a batch-1 graph -- a handful of blocks -- narrower column blocks halve each
    //  wave's chain of dependent MFMAs; the sums per output are the same)
    int CT = (H % 64 == 0) ? 2 : 1;
    if (CT == 2 && (long)ceil_div(R, 128) * (H / 64) < 128) CT = 1;
    dim3 grid(ceil_div(R, 128), H / (32 * CT)), block(256);
#define L(C, X) hipLaunchKernelGGL((k_gru_fwd<C, X>), grid, block, 0, st, a)
    if (CT == 2) { if (xmode == 0) L(2, 0); else if (xmode == 1) L(2, 1); else L(2, 2); }
    else         { if (xmode == 0) L(1, 0); else if (xmode == 1) L(1, 1); else L(1, 2); }
#undef L
    return check_launch("gru_fwd");
}

int tmpnn_gru_fwd_tiles(const tmpnn_edge_tiles* tiles, int R, const float* proj, int ld_proj, const float* h, int ld_h, int H,
                        const float* whh_t, const float* b_ih, const float* b_hh, float* h_out, int ld_out, float* gates,
                        size_t gate_plane, const float* w_head, float* logit_part, size_t part_stride, tmpnn_stream stream) {
    TM_REQUIRE(H == 32 || H == 64, "gru_fwd_tiles: H=%d (the tiled forward serves the LDS-resident cells, H in {32, 64})", H);
    TM_REQUIRE(R >= 0, "gru_fwd_tiles: R=%d", R);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(split_enabled(), "gru_fwd_tiles: the tiled forward is the bf16x6 form (TMPNN_SPLIT=0 keeps tmpnn_gru_fwd)");
    TM_REQUIRE(tiles != nullptr, "gru_fwd_tiles: tiles is null");
    const int RPT = tiles->rows_per_tile;
    TM_REQUIRE((RPT == 32 || RPT == 16) && tiles->T > 0 && (long)tiles->T * RPT >= R && (long)(tiles->T - 1) * RPT < R &&
                   tiles->t_row && tiles->t_loc && tiles->t_dptr && tiles->t_dets,
               "gru_fwd_tiles: tile list (T=%d, rows_per_tile=%d) does not cover R=%d rows in 16- or 32-row tiles", tiles->T,
               tiles->rows_per_tile, R);
    TM_REQUIRE(proj && h && whh_t && b_ih && b_hh && h_out, "gru_fwd_tiles: null pointer");
    TM_REQUIRE(ld_proj >= 3 * H && (ld_proj & 3) == 0 && aligned16(proj) && ld_h >= H && ld_out >= H && (ld_h & 3) == 0 &&
                   aligned16(h) && aligned16(whh_t) && aligned16(h_out) && (ld_out & 3) == 0 && aligned16(b_ih) && aligned16(b_hh) &&
                   (gates == nullptr || (aligned16(gates) && (gate_plane & 3) == 0 && gate_plane >= (size_t)H)),
               "gru_fwd_tiles: layout (16-byte alignment, leading dimensions)");
    TM_REQUIRE(logit_part == nullptr || (w_head != nullptr && aligned16(w_head)), "gru_fwd_tiles: fused head needs a 16-byte aligned w_head");
    GruFwdArgs a{nullptr, R, nullptr, nullptr, proj, ld_proj, H, 0, h, ld_h, H, nullptr, whh_t, b_ih, b_hh, h_out, ld_out, gates,
                 gate_plane, w_head, logit_part, part_stride};
    FwdTiles tl{tiles->t_row, tiles->t_loc, tiles->t_dptr, tiles->t_dets, tiles->T};
    hipStream_t st = as_stream(stream);
    const int ntiles = ceil_div(R, (H == 64) ? 192 : 256);
    dim3 pgrid(ntiles < 256 ? ntiles : 256);
    if (RPT == 16) {         // sixteen 128-register waves per CU (k_gru_fwd_split_t16)
        const size_t shm16 = (size_t)3 * 3 * H * H * 2 + sizeof(float) * ((size_t)16 * T16_AREA + 4 + 5 * H);
        if (H == 64) {
            TM_SHM_ONCE((k_gru_fwd_split_t16<64, 16>), shm16);
            hipLaunchKernelGGL((k_gru_fwd_split_t16<64, 16>), pgrid, dim3(1024), shm16, st, a, tl);
        } else {
            TM_SHM_ONCE((k_gru_fwd_split_t16<32, 16>), shm16);
            hipLaunchKernelGGL((k_gru_fwd_split_t16<32, 16>), pgrid, dim3(1024), shm16, st, a, tl);
        }
        return check_launch("gru_fwd_split_t16");
    }
    const size_t shm = (size_t)3 * 3 * H * (H + 8) * 2 + sizeof(float) * ((size_t)8 * TP_AREA + 4 + 5 * H);
    if (H == 64) {
        TM_SHM_ONCE((k_gru_fwd_split_tiled<64, 8>), shm);
        hipLaunchKernelGGL((k_gru_fwd_split_tiled<64, 8>), pgrid, dim3(512), shm, st, a, tl);
    } else {
        TM_SHM_ONCE((k_gru_fwd_split_tiled<32, 8>), shm);
        hipLaunchKernelGGL((k_gru_fwd_split_tiled<32, 8>), pgrid, dim3(512), shm, st, a, tl);
    }
    return check_launch("gru_fwd_split_tiled");
}

int tmpnn_rows_linear(const int32_t* rows, int R, const float* in, int ld_in, int H, const float* wt, int NOUT,
                      float* out, int ld_out, tmpnn_stream stream) {
    TM_REQUIRE((H == 32 || H == 64) && (NOUT == 3 * H), "rows_linear: only [H<=64] x [3H] projections (H=%d NOUT=%d)", H, NOUT);
    if (R == 0) return TMPNN_OK;
    TM_REQUIRE(R > 0 && rows && in && wt && out, "rows_linear: null pointer");
    TM_REQUIRE(ld_in >= H && (ld_in & 3) == 0 && aligned16(in) && aligned16(wt) && ld_out >= NOUT && (ld_out & 3) == 0 &&
                   aligned16(out),
               "rows_linear: rows must be 16-byte aligned");
    const int ntiles = ceil_div(R, 256);
    dim3 grid(ntiles < 256 ? ntiles : 256), block(512);
    hipStream_t st = as_stream(stream);
    if (split_enabled()) return launch_rows_gemm(rows, R, in, ld_in, H, wt, NOUT, 0, NOUT, out, ld_out, nullptr, 0, st);
    const size_t shm = sizeof(float) * ((size_t)H * NOUT + 8 * 32 * STG_LD);
    if (H == 64) {
        TM_SHM_ONCE((k_rows_gemm_lds<64, 6>), shm);
        hipLaunchKernelGGL((k_rows_gemm_lds<64, 6>), grid, block, shm, st, rows, R, in, ld_in, wt, out, ld_out, ntiles);
    } else {
        TM_SHM_ONCE((k_rows_gemm_lds<32, 3>), shm);
        hipLaunchKernelGGL((k_rows_gemm_lds<32, 3>), grid, block, shm, st, rows, R, in, ld_in, wt, out, ld_out, ntiles);
    }
    return check_launch("rows_linear");
}

}  // extern "C"
