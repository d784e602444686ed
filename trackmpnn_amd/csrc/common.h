// Shared helpers for the libtmpnn kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "tmpnn.h"

// The LDS-resident kernels need more than the default 64 KiB of dynamic LDS: raise the limit ONCE per (kernel
// instantiation, device) instead of before every launch (idempotent function-attribute setup, not data state).
#define TM_SHM_ONCE(kernel, bytes)                                                                           \
    do {                                                                                                     \
        static std::atomic<int> done_[16];                                                                   \
        int dev_ = 0;                                                                                        \
        (void)hipGetDevice(&dev_);                                                                           \
        const int b_ = (int)(bytes);                                                                         \
        if (done_[dev_ & 15].load(std::memory_order_relaxed) < b_) {                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, b_);                       \
            done_[dev_ & 15].store(b_, std::memory_order_relaxed);                                           \
        }                                                                                                    \
    } while (0)


// ---- entry-point-shaped helpers that other entry points call (single stages of tmpnn_track_select / _retire, the conversions'
//      small-graph forms): C linkage, NOT exported by the shared library
#define TMPNN_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
#ifndef TMPNN_KEEP_VARIANTS
// (the implementation behind tmpnn_gru_bwd_weights; an entry point of its own only in comparison builds)
TMPNN_INTERNAL int tmpnn_gru_bwd_weights_variant(const int32_t* rows, int R, int xmode, const int32_t* src, const int32_t* dst,
                                  const float* msg, int ld_msg, int msg_compact, int IN, const float* h, int ld_h, int H,
                                  const float* gates, size_t gate_plane, const float* d_hout, int ld_dhout, const float* dy,
                                  const float* w_head, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh, void* ws,
                                  size_t ws_bytes, int variant, tmpnn_stream stream);
#endif
TMPNN_INTERNAL size_t tmpnn_graph_from_coo_ws_ints(int N);
TMPNN_INTERNAL int tmpnn_graph_from_rows(int N, const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst,
                          const tmpnn_dgraph* g, tmpnn_stream stream);
TMPNN_INTERNAL int tmpnn_track_active(int N, const int32_t* ts, const int32_t* assoc, const float* score, int mode, int t,
                       int32_t* active, int32_t* count, tmpnn_stream stream);
TMPNN_INTERNAL int tmpnn_track_associate(const tmpnn_dgraph* g, const int32_t* det_id, const uint8_t* labels, const float* score,
                          int mode, int32_t* assoc, int32_t* status, tmpnn_stream stream);
TMPNN_INTERNAL int tmpnn_track_delete(int N, const int32_t* ts, const int32_t* det_id, const int32_t* assoc, const float* score,
                       const uint8_t* is_edge, const int32_t* row_src, const int32_t* row_dst, const uint8_t* labels,
                       int t_upto, int ret_win, int32_t* keep, int32_t* count, int32_t* o_ts, int32_t* o_det_id,
                       int32_t* o_assoc, uint8_t* o_is_edge, int32_t* o_src, int32_t* o_dst, uint8_t* o_labels,
                       tmpnn_stream stream);
TMPNN_INTERNAL int tmpnn_track_finalize(const tmpnn_dgraph* g, const int32_t* ts, const int32_t* det_id, const int32_t* assoc,
                         const float* score, int t_upto, int32_t* y_track, int ND, int32_t* pos_of_det, void* ws,
                         size_t ws_bytes, tmpnn_stream stream);
TMPNN_INTERNAL int tmpnn_track_gather(const float* in, int ld_in, int W, int max_rows, const int32_t* keep, const int32_t* count,
                       float* out, int ld_out, tmpnn_stream stream);
}

namespace tmpnn {

int set_error(int code, const char* fmt, ...);

#define TM_REQUIRE(cond, ...)                                            \
    do {                                                                 \
        if (!(cond)) return ::tmpnn::set_error(TMPNN_EINVAL, __VA_ARGS__); \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(TMPNN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TMPNN_OK;
}

inline bool supported_H(int H) { return H == 32 || H == 64 || H == 128 || H == 256; }
// widths above 256 (reference: any int, utils/training_options.py:22): multiples of 128 up to 1024 on the H-generic kernels --
// the wide edge cell, the f32-MFMA node cell, the strided GEMMs; the row movers take them in 256-column slices
inline bool supported_H_big(int H) { return H > 256 && H <= 1024 && H % 128 == 0; }
inline bool supported_H_cell(int H) { return supported_H(H) || supported_H_big(H); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline hipStream_t as_stream(tmpnn_stream s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// sum over `nslab` slabs of `n` floats each: dst[i] (+)= sum_s slabs[s*stride + i]   (deterministic)
// More than 64 slabs are first folded 32:1 into `ws2` (reduce_slabs_ws_floats(nslab, n) floats).
size_t reduce_slabs_ws_floats(int nslab, size_t n);
int launch_reduce_slabs(const float* slabs, size_t stride, int nslab, float* dst, size_t n, int accumulate,
                        hipStream_t st, float* ws2 = nullptr);

// Generic small dense product on the vector ALU (tiny operands only: input transform, attention
// projections).  C[crow(m)*ldc + n] (+)= sum_k A[arow(m)*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n])
struct GemmArgs {
    const float* A; long sam, sak; const int32_t* a_rows; const int32_t* a_krows;   // optional row gathers on m / k
    const float* B; long sbk, sbn;
    const float* bias;
    float* C; long ldc; const int32_t* c_rows;
    int M, N, K;
    int accumulate;
};
int launch_gemm(const GemmArgs& g, hipStream_t st);
// split-K variant for K >> M,N (weight gradients): partial products go to `ws`, then reduced into C (+=).
size_t gemm_splitk_ws_floats(int M, int N, int K);
int launch_gemm_splitk(const GemmArgs& g, float* ws, size_t ws_floats, hipStream_t st);
// MFMA row products of the small dense stages (csrc/gru_fwd.hip, bf16x6 = fp32-accurate, see there):
//   out[orow(r)][0:NOUT] (=|+=) in[irow(r)][0:KD] @ W,  W[k][n] = wt[k ld_wt + n] or (wt_trans) wt[n ld_wt + k];
//   KD in {32, 64, 128}; NOUT / 32 in {1, 2, 4, 6} (KD 64), {1, 2, 3} (KD 32) or {1, 2} (KD 128): rows_gemm_supported(); rows / out_rows: optional row lists (NULL = r).
bool rows_gemm_supported(int KD, int NOUT);
int launch_rows_gemm(const int32_t* rows, int R, const float* in, int ld_in, int KD, const float* wt, int ld_wt, int wt_trans,
                     int NOUT, float* out, int ld_out, const int32_t* out_rows, int accumulate, hipStream_t st);
// C[i][n] (+)= sum_r X[xrow(r)][i] * Y[r][n]  (i < M <= 256, n < N; contraction over R rows; f32 MFMA, operands straight from
// global memory: both are row-contiguous for this orientation).  Per-slab partial tiles in `ws`, ordered reduction.
size_t rows_outer_ws_floats(int M, int N, int R);
int launch_rows_outer(const float* X, int ldx, const int32_t* x_rows, const float* Y, int ldy, int R, int M, int N, float* C,
                      int ldc, int accumulate, float* ws, size_t ws_floats, hipStream_t st);
// dst[j] (+)= sum_i src[i*ld + j] * (mul ? mul[i*ldm + j] : 1)     (two level, deterministic)
size_t colsum_ws_floats(int rows, int cols);
int launch_colsum(const float* src, long ld, const float* mul, long ldm, int rows, int cols, float* dst,
                  int accumulate, float* ws, size_t ws_floats, hipStream_t st);

}  // namespace tmpnn
