// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 with 1 wave per SIMD (4 per CU),
// 8 independent accumulators, optional ds_read_b128 traffic between MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#ifndef RPG_GAPS
#define RPG_GAPS 6
#endif
template <int READS>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* ts, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 16; i += 256) ((uint4*)lds)[i] = make_uint4(i, i, i, i);
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 A[4], B[2];
    const unsigned char* base = lds + (tid & 63) * 272 % 32768;
    for (int i = 0; i < 4; ++i) A[i] = *(const bf16x8*)(base + i * 16);
    for (int i = 0; i < 2; ++i) B[i] = *(const bf16x8*)(base + 64 + i * 16);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    bf16x8 A1[4], B1[2];
    for (int i = 0; i < 4; ++i) A1[i] = A[i];
    for (int i = 0; i < 2; ++i) B1[i] = B[i];
#define STEP(Ac, Bc, An, Bn, off)                                                               \
    {                                                                                            \
        if (READS) {                                                                             \
            const unsigned char* p = base + (off);                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) Bn[i] = *(const bf16x8*)(p + 32768 + i * 8704); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) An[i] = *(const bf16x8*)(p + i * 8192); \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac[i], Bc[j], acc[i * 2 + j], 0, 0, 0); \
        if (READS) {                                                                             \
            _Pragma("unroll") for (int q = 0; q < RPG_GAPS; ++q) {                               \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
                __builtin_amdgcn_sched_group_barrier(0x100, 6 / RPG_GAPS, 0);                    \
            }                                                                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 8 - RPG_GAPS, 0);                        \
        }                                                                                        \
    }
    for (int it = 0; it < iters; it += 2) {
        const int o = (it & 6) << 5;
        STEP(A, B, A1, B1, o + 32)
        STEP(A1, B1, A, B, o + 64)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { ts[0] = t1 - t0; }
}

int main(int argc, char** argv) {
    const int iters = 2000, blocks = argc > 1 ? atoi(argv[1]) : 256;
    float* out; unsigned long long* ts;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ts, 64);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, ts, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, ts, iters);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; hipMemcpy(&t, ts, 8, hipMemcpyDeviceToHost);
            const double mf = (double)iters * 8;
            printf("gaps=%d reads=%d blocks=%d: %.1f ticks per MFMA, %.3f ms, %.1f TFLOP/s, %.3f ticks/ns\n", RPG_GAPS, mode, blocks, t / mf, ms,
                   blocks * 4.0 * mf * 32768.0 / (ms * 1e-3) / 1e12, t / (ms * 1e6));
        }
    }
    return 0;
}
