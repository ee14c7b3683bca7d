// Micro-benchmark: MFMA issue rate with fragment reads AND a weight stream into LDS
// (one wave per SIMD, 4 per CU), for different ways of issuing the stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gl_void;

// MODE 0: no stream. 1: global_load_lds builtin. 2: global_load_dwordx4 + ds_write_b128 (register staged)
template <int MODE, int PER_STEP>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* ts, const unsigned char* w, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // 64 KB fragments + 64 KB stream
    const int tid = threadIdx.x;
    for (int i = tid; i < 131072 / 16; i += 256) ((uint4*)lds)[i] = make_uint4(i, i, i, i);
    __syncthreads();
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 A[4], B[2], A1[4], B1[2];
    const unsigned char* base = lds + (tid & 63) * 272 % 32768;
    for (int i = 0; i < 4; ++i) A1[i] = A[i] = *(const bf16x8*)(base + i * 16);
    for (int i = 0; i < 2; ++i) B1[i] = B[i] = *(const bf16x8*)(base + 64 + i * 16);
    unsigned char* const st_dst = lds + 65536 + __builtin_amdgcn_readfirstlane(tid >> 6) * 1024;
    const unsigned st_off = tid * 16;
    uint4 stage[PER_STEP > 0 ? PER_STEP : 1];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 162 * 32768 + 65536, 0x00020000);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#define STEP(Ac, Bc, An, Bn, off, piece0, DO)                                                      \
    {                                                                                            \
        const unsigned char* p = base + (off);                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) Bn[i] = *(const bf16x8*)(p + 32768 + i * 8704); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) An[i] = *(const bf16x8*)(p + i * 8192);    \
        if (MODE == 1 && (DO)) {                                                                   \
            _Pragma("unroll") for (int q = 0; q < PER_STEP; ++q)                                 \
                __builtin_amdgcn_global_load_lds((gl_void*)(wchunk + ((piece0) + q) * 4096 + st_off), \
                                                 (lds_void*)(st_dst + (((piece0) + q) & 7) * 4096), 16, 0, 0); \
        }                                                                                        \
        if (MODE == 3 && (DO)) {                                                                 \
            _Pragma("unroll") for (int q = 0; q < PER_STEP; ++q)                                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(st_dst + (((piece0) + q) & 7) * 4096), 16, st_off, \
                                                         wsoff + ((piece0) + q) * 4096, 0, 0);   \
        }                                                                                        \
        if (MODE == 2 && (DO)) {                                                                   \
            _Pragma("unroll") for (int q = 0; q < PER_STEP; ++q) {                               \
                *(uint4*)(st_dst + (((piece0) + q) & 7) * 4096 + (tid & 63) * 16) = stage[q];    \
                stage[q] = *(const uint4*)(wchunk + ((piece0) + q) * 4096 + st_off);             \
            }                                                                                    \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac[i], Bc[j], acc[i * 2 + j], 0, 0, 0); \
        _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
        }                                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        if (DO) __builtin_amdgcn_sched_group_barrier(MODE == 2 ? 0x220 : 0x020, (MODE == 2 ? 2 : 1) * PER_STEP, 0);        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
#ifdef ORDERED
#undef STEP
#define MF(Ac, Bc, i, j) acc[(i) * 2 + (j)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac[i], Bc[j], acc[(i) * 2 + (j)], 0, 0, 0)
#define FENCE __builtin_amdgcn_sched_barrier(0)
#define STEP(Ac, Bc, An, Bn, off, piece0, DO)                                                  \
    {                                                                                            \
        const unsigned char* p = base + (off);                                                   \
        MF(Ac, Bc, 0, 0); Bn[0] = *(const bf16x8*)(p + 32768); FENCE;                            \
        MF(Ac, Bc, 0, 1); Bn[1] = *(const bf16x8*)(p + 32768 + 8704); FENCE;                     \
        MF(Ac, Bc, 1, 0); An[0] = *(const bf16x8*)(p); FENCE;                                    \
        MF(Ac, Bc, 1, 1); An[1] = *(const bf16x8*)(p + 8192); FENCE;                             \
        MF(Ac, Bc, 2, 0); An[2] = *(const bf16x8*)(p + 16384); FENCE;                            \
        MF(Ac, Bc, 2, 1); An[3] = *(const bf16x8*)(p + 24576); FENCE;                            \
        MF(Ac, Bc, 3, 0);                                                                        \
        if (MODE == 1 && (DO)) {                                                                 \
            _Pragma("unroll") for (int q = 0; q < PER_STEP; ++q)                                 \
                __builtin_amdgcn_global_load_lds((gl_void*)(wchunk + ((piece0) + q) * 4096 + st_off), \
                                                 (lds_void*)(st_dst + (((piece0) + q) & 7) * 4096), 16, 0, 0); \
        }                                                                                        \
        if (MODE == 3 && (DO)) {                                                                 \
            _Pragma("unroll") for (int q = 0; q < PER_STEP; ++q)                                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(st_dst + (((piece0) + q) & 7) * 4096), 16, st_off, \
                                                         wsoff + ((piece0) + q) * 4096, 0, 0);   \
        }                                                                                        \
        FENCE;                                                                                   \
        MF(Ac, Bc, 3, 1); FENCE;                                                                 \
    }
#endif
    for (int q = 0; q < (PER_STEP > 0 ? PER_STEP : 1); ++q) stage[q] = make_uint4(0, 0, 0, 0);
    // one "tap" = 8 k-steps, 8 pieces of 4 KB streamed during it (32 KB per tap per workgroup)
    for (int it = 0; it < iters; it += 8) {
        const unsigned char* wchunk = w + (size_t)((it >> 3) % 162) * 32768;
        const int wsoff = ((it >> 3) % 162) * 32768;
        constexpr int NS = PER_STEP > 0 ? 8 / PER_STEP : 0;       // k-steps that carry pieces
        STEP(A, B, A1, B1, 32, 0, NS > 0)
        STEP(A1, B1, A, B, 64, (NS > 1 ? 1 : 0) * PER_STEP, NS > 1)
        STEP(A, B, A1, B1, 96, (NS > 2 ? 2 : 0) * PER_STEP, NS > 2)
        STEP(A1, B1, A, B, 128, (NS > 3 ? 3 : 0) * PER_STEP, NS > 3)
        STEP(A, B, A1, B1, 160, (NS > 4 ? 4 : 0) * PER_STEP, NS > 4)
        STEP(A1, B1, A, B, 192, (NS > 5 ? 5 : 0) * PER_STEP, NS > 5)
        STEP(A, B, A1, B1, 224, (NS > 6 ? 6 : 0) * PER_STEP, NS > 6)
        STEP(A1, B1, A, B, 0, (NS > 7 ? 7 : 0) * PER_STEP, NS > 7)
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int q = 0; q < (PER_STEP > 0 ? PER_STEP : 1); ++q) s += stage[q].x;
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ts[0] = t1 - t0;
}

template <int MODE, int PER_STEP>
void run(const char* name, float* out, unsigned long long* ts, const unsigned char* w, int blocks) {
    const int iters = 1296 * 2;      // 2 x 18 layers x 9 taps x 8 k-steps
    hipFuncSetAttribute((const void*)k<MODE, PER_STEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float best = 1e9f; unsigned long long bt = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PER_STEP>), dim3(blocks), dim3(256), 131072, 0, out, ts, w, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ts, 8, hipMemcpyDeviceToHost);
        if (ms < best) { best = ms; bt = t; }
    }
    const double mf = (double)iters * 8;
    printf("%-34s %.1f ticks/MFMA  %.3f ms  %.0f TFLOP/s\n", name, bt / mf, best, blocks * 4.0 * mf * 32768.0 / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    float* out; unsigned long long* ts; unsigned char* w;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ts, 64); hipMalloc(&w, 162 * 32768 + 65536); hipMemset(w, 1, 162 * 32768 + 65536);
    run<0, 0>("no stream", out, ts, w, blocks);
    run<1, 2>("glds, 2 pieces x 4 k-steps", out, ts, w, blocks);
    run<1, 1>("glds, 1 piece x 8 k-steps", out, ts, w, blocks);
    run<1, 4>("glds, 4 pieces x 2 k-steps", out, ts, w, blocks);
    run<3, 2>("buffer lds, 2 pieces x 4 k-steps", out, ts, w, blocks);
    run<3, 4>("buffer lds, 4 pieces x 2 k-steps", out, ts, w, blocks);
    run<2, 2>("reg-staged, 2 pieces x 4 k-steps", out, ts, w, blocks);
    run<2, 1>("reg-staged, 1 piece x 8 k-steps", out, ts, w, blocks);
    return 0;
}
