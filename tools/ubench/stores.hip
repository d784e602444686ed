// Microbenchmark: how do bursts of 16-byte-per-lane stores interact with a wave's compute on gfx950?
// A block is 8 waves; every wave runs ITEMS items of  [matrix phase: NM MFMAs]  [burst: NS stores of 1 KB, streaming]  and,
// depending on the mode, a small load whose result is consumed at a chosen point (vmcnt retires loads and stores in order).
//   mode 0  matrix phases only                      mode 1  store bursts only
//   mode 2  both, no load                           mode 3  both + a load issued AFTER the burst, consumed before the next phase
//   mode 4  both + a load issued BEFORE the burst, consumed after the next matrix phase (nothing younger than an item is waited for)
//   mode 5  as 2, the burst spread over the matrix phase (one store every NM / NS MFMAs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool NT>
__global__ __launch_bounds__(512) void k(int items, float* __restrict__ out, const float* __restrict__ small, float* sink) {
    constexpr int NM = 72, NS = 20;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 8 + wave;
    f32x16 acc[3];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float carry = 0.f, pend = 0.f;
    // wave-private streaming region: items x NS x 1 KB
    float* base = out + gw * (size_t)items * NS * 256;
    for (int it = 0; it < items; ++it) {
        float* p = base + (size_t)it * NS * 256 + lane * 4;
        if (MODE == 4) pend = small[(it * 64 + lane) & 4095];          // requested before the burst of the PREVIOUS... (see below)
        if (MODE != 1 && MODE != 8) {
            if (MODE == 5) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
                    f32x4 v = {acc[0][0] + carry, acc[1][1], acc[2][2], (float)s};
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + s * 256));
                    else *reinterpret_cast<f32x4*>(p + s * 256) = v;
                }
#pragma unroll
                for (int m = 0; m < NM - 3 * NS; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % 3], 0, 0, 0);
            }
        }
        if (MODE == 4) carry += pend;                                   // consumed after the matrix phase: older than the burst below
        if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 4) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f32x4 v = {acc[0][0] + carry, acc[1][1], acc[2][2], (float)s};
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p + s * 256));
                else *reinterpret_cast<f32x4*>(p + s * 256) = v;
            }
        }
        if (MODE == 6 || MODE == 7) {
            // the H = 64 forward's real pattern: five planes, a 32-row x 32-column item = rows of 256 B of which this item
            // writes one 128-byte half (mode 6: the other half comes from another item later; mode 7: both halves now, i.e.
            // 16 full rows per item and plane), four store instructions per plane, each 8 rows x 128 B
            const size_t plane = (size_t)gridDim.x * 8 * items * 1024;              // floats per plane: 5 planes = the whole buffer
            const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int pl = 0; pl < 5; ++pl) {
                float* pb = out + pl * plane;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    f32x4 v = {acc[0][0] + carry, acc[1][1], acc[2][2], (float)kq};
                    size_t row, col;
                    if (MODE == 6) { row = (gw * (size_t)items + it) / 2 * 32 + 8 * kq + rr; col = ((gw * (size_t)items + it) & 1) * 32 + 4 * ch; }
                    else { row = (gw * (size_t)items + it) * 16 + 4 * kq + (lane >> 4); col = 4 * (lane & 15); }
                    float* q = pb + row * 64 + col;
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q));
                    else *reinterpret_cast<f32x4*>(q) = v;
                }
            }
        }
        if (MODE == 3) {                                                // a load behind the burst, needed at once
            const float t = small[(it * 64 + lane) & 4095];
            carry += t;
            asm volatile("" : "+v"(carry));
        }
    }
    float s = carry;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int MODE, bool NT>
static void run(const char* what, int items, float* out, float* small, float* sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(512), 0, 0, items, out, small, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(512), 0, 0, items, out, small, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gb = MODE == 0 ? 0.0 : 256.0 * 8 * items * 20 * 1024 / 1e9;
    printf("%-4s mode %d  %-62s %.3f ms  (%.2f GB stored = %.2f TB/s)\n", NT ? "nt" : "", MODE, what, ms / 3, gb, gb / (ms / 3) / 1e3 * 1e3 / 1e3);
}

int main() {
    const int items = 184;                                   // 256 x 8 x 184 x 20 KB = 7.7 GB, the H = 64 forward's store volume
    float *out, *small, *sink;
    (void)hipMalloc(&out, (size_t)256 * 8 * items * 20 * 1024);
    (void)hipMalloc(&small, 4096 * 4); (void)hipMemset(small, 0, 4096 * 4);
    (void)hipMalloc(&sink, 4096);
    run<0, false>("matrix phases only (72 MFMAs per item)", items, out, small, sink);
    run<1, false>("store bursts only (20 x 1 KB per item)", items, out, small, sink);
    run<1, true>("store bursts only", items, out, small, sink);
    run<2, false>("both, burst after the phase", items, out, small, sink);
    run<2, true>("both, burst after the phase", items, out, small, sink);
    run<3, true>("both + a load BEHIND the burst, consumed at once", items, out, small, sink);
    run<4, true>("both + a load AHEAD of the burst, consumed an item later", items, out, small, sink);
    run<5, true>("both, one store every 3 MFMAs", items, out, small, sink);
    run<5, false>("both, one store every 3 MFMAs", items, out, small, sink);
    run<6, true>("both, real pattern: 128-B halves of 256-B rows, 5 planes", items, out, small, sink);
    run<6, false>("both, real pattern: 128-B halves of 256-B rows, 5 planes", items, out, small, sink);
    run<7, true>("both, full 256-B rows (16 rows x 64 columns per item)", items, out, small, sink);
    run<7, false>("both, full 256-B rows (16 rows x 64 columns per item)", items, out, small, sink);
    return 0;
}
