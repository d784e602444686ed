// Microbenchmark: do MFMA and vector-ALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// A block is 8 waves (2 per SIMD: waves w and w + 4 share a SIMD).  Waves 0-3 run a chain-free MFMA loop, waves 4-7 a
// vector-ALU loop (fp32 FMA, or v_exp_f32); each role alone, then both together.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, int valu_kind, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
    if (do_mfma) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        for (int it = 0; it < n_mfma; it += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0.f;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if (do_valu) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float c0 = 1.0001f, c1 = 1e-4f;
        if (valu_kind == 0) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
            }
        } else if (valu_kind == 1) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
        } else if (valu_kind == 2) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            }
        } else if (valu_kind == 3) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            }
        } else if (valu_kind == 4) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            }
        } else if (valu_kind == 5) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
            }
        } else if (valu_kind == 6) {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&v[i])) : "v"(*reinterpret_cast<const double*>(&v[(i + 2) & 7])));
            }
        } else {
            for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) & 7]));
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int NM = 20000;
    const char* names[8] = {"v_fma_f32", "v_exp_f32", "v_add_f32", "v_and_b32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_pk_fma_f32 (x2 lanes)", "v_mov_b32"};
    for (int kind = 0; kind < 8; ++kind) {
        const int NV = kind == 1 ? 40000 : 160000;
        for (int mode = 1; mode <= 3; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, NM, NV, kind, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, NM, NV, kind, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("valu=%s mode=%s: %.3f ms per launch  (%d MFMA 32x32x16 per MFMA wave, %d %s per VALU wave)\n",
                   names[kind], mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both          ",
                   ms / 5, NM, NV, "ops");
        }
    }
    return 0;
}
