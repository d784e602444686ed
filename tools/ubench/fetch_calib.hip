// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of this repository's kernels (MI355X_MICROARCH.md:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are
// uncalibrated").  Every kernel reads the same 2 GiB buffer (8 x the 256 MiB Infinity Cache) exactly once; the factor
// bytes_read / FETCH_SIZE per kernel is what a traffic figure of that access pattern has to be multiplied with.
//   k_read16      16 B per lane, a wave reads 1 KiB contiguous                      (the guide's calibrated case)
//   k_read4       4 B per lane, a wave reads 256 B contiguous per instruction        (k_wide_gru_fwd_pp's previous-state reads)
//   k_read4_half  4 B per lane, lanes 0-31 and 32-63 read two 128-B runs 1 KiB apart (its accumulator layout: 32 columns of 2 rows)
//   k_lds16       global_load_lds_dwordx4: 16 B per lane into LDS                    (the A / weight streams of the wide kernels)
//   k_lds4        global_load_lds_dword: 4 B per lane into LDS
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

static constexpr size_t BYTES = (size_t)2 << 30;

__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ p, size_t n16, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_read4(const float* __restrict__ p, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 12345.678f) sink[0] = acc;
}
// a wave's instruction reads 128 B of row r (lanes 0-31) and 128 B of row r + 1 (lanes 32-63); rows are 1 KiB; the 8 column
// blocks of a row pair are read by 8 consecutive instructions of the same wave (every byte once)
__global__ __launch_bounds__(256) void k_read4_half(const float* __restrict__ p, size_t nrows, float* sink) {
    float acc = 0.f;
    const int lane = threadIdx.x & 63, half = lane >> 5, c = lane & 31;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t r = 2 * wave; r + 1 < nrows; r += 2 * nw) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc += p[(r + half) * 256 + cb * 32 + c];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int DW>
__global__ __launch_bounds__(256) void k_lds(const float* __restrict__ p, size_t nchunk, float* sink) {
    __shared__ __attribute__((aligned(16))) float buf[4 * 64 * DW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)(const char*)(buf + wave * 64 * DW);
    const size_t gw = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    float acc = 0.f;
    for (size_t ch = gw; ch < nchunk; ch += nw) {
        const float* src = p + (ch * 64 + lane) * DW;
        unsigned keep;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
        if (DW == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[wave * 64 * DW + lane];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    float *buf, *sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, BYTES);
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const float4*)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, buf, BYTES / 4, sink);
        hipLaunchKernelGGL(k_read4_half, dim3(grid), dim3(256), 0, 0, buf, BYTES / 1024, sink);
        hipLaunchKernelGGL(k_lds<4>, dim3(grid), dim3(256), 0, 0, buf, BYTES / 1024, sink);
        hipLaunchKernelGGL(k_lds<1>, dim3(grid), dim3(256), 0, 0, buf, BYTES / 256, sink);
    }
    hipDeviceSynchronize();
    printf("bytes_per_kernel %zu\n", BYTES);
    return 0;
}
