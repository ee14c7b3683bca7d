// Bare MFMA loops on random 16-bit operands held in registers, same wave tile (128 x 64 outputs, K = 32 per step):
// v_mfma_f32_32x32x16 (8 per step) against v_mfma_f32_16x16x32 (32 per step) with the accumulators pinned in
// AGPRs by tied inline-asm operands (the builtin form of the 16x16x32 loop gets accumulator copies from the
// register allocator, which hides the shape's own rate), bf16 and f16, random data and zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int F16>
__global__ void __launch_bounds__(256) k32(const uint4* __restrict__ rnd, float* out, unsigned long long* ts, int iters) {
    const int tid = threadIdx.x;
    uint4 A[4], B[2];
    for (int i = 0; i < 4; ++i) A[i] = rnd[tid * 16 + i];
    for (int j = 0; j < 2; ++j) B[j] = rnd[tid * 16 + 8 + j];
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (F16) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[i]), __builtin_bit_cast(f16x8, B[j]), acc[i * 2 + j], 0, 0, 0);
                    else acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[j]), acc[i * 2 + j], 0, 0, 0);
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 17) { ts[0] = t1 - t0; ts[1] = r1 - r0; }
}

template <int F16>
__global__ void __launch_bounds__(256) k16(const uint4* __restrict__ rnd, float* out, unsigned long long* ts, int iters) {
    const int tid = threadIdx.x;
    u32x4 A[8], B[4];
    for (int i = 0; i < 8; ++i) A[i] = __builtin_bit_cast(u32x4, rnd[tid * 16 + i]);
    for (int j = 0; j < 4; ++j) B[j] = __builtin_bit_cast(u32x4, rnd[tid * 16 + 8 + j]);
    f32x4 acc[32];
    for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(A[i]), "v"(B[j]));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i * 4 + j]) : "v"(A[i]), "v"(B[j]));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // MFMA results -> VALU reads (the hazard the compiler cannot see)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 32; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 17) { ts[0] = t1 - t0; ts[1] = r1 - r0; }
}

int main() {
    const int iters = 20000, blocks = 256;
    uint4* rnd; float* out; unsigned long long* ts;
    CHK(hipMalloc(&rnd, 65536)); CHK(hipMalloc(&out, blocks * 256 * 4)); CHK(hipMalloc(&ts, 64));
    unsigned* h = (unsigned*)malloc(65536);
    for (int f16 = 0; f16 < 2; ++f16)
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 16384; ++i) {
            unsigned a = rand(), b = rand();
            // values in +-[0.25, 1): bf16 0x3E80.. / f16 0x3400..
            unsigned lo, hi;
            if (f16) { lo = ((a & 1) << 15) | (0x3400 + (a >> 1 & 0x7FF)); hi = ((b & 1) << 15) | (0x3400 + (b >> 1 & 0x7FF)); }
            else { lo = ((a & 1) << 15) | (0x3E80 + (a >> 1 & 0xFF)); hi = ((b & 1) << 15) | (0x3E80 + (b >> 1 & 0xFF)); }
            h[i] = zero ? 0u : (lo | (hi << 16));
        }
        CHK(hipMemcpy(rnd, h, 65536, hipMemcpyHostToDevice));
        for (int shape = 0; shape < 2; ++shape)
            for (int rep = 0; rep < 4; ++rep) {
                hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
                CHK(hipEventRecord(e0));
                if (shape == 0) { if (f16) hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, rnd, out, ts, iters); else hipLaunchKernelGGL(k32<0>, dim3(blocks), dim3(256), 0, 0, rnd, out, ts, iters); }
                else { if (f16) hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, rnd, out, ts, iters); else hipLaunchKernelGGL(k16<0>, dim3(blocks), dim3(256), 0, 0, rnd, out, ts, iters); }
                CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long t[2]; CHK(hipMemcpy(t, ts, 16, hipMemcpyDeviceToHost));
                const double flop = (double)blocks * 4 * iters * 128.0 * 64 * 32 * 2;
                if (rep) printf("%s %s data, %s bare: %.3f ms, %.0f TFLOP/s, %.1f cycles per k32-step (512 = MFMA floor), clock %.3f GHz\n",
                       f16 ? "f16 " : "bf16", zero ? "zero" : "random", shape ? "16x16x32" : "32x32x16", ms, flop / (ms * 1e-3) / 1e12,
                       (double)t[0] / iters, (double)t[0] / t[1] * 0.1);
            }
    }
    return 0;
}
