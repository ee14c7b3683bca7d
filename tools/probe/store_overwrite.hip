// Does a VALU write into the data registers of buffer_store_dwordx4 (scalar offset in an SGPR), issued right behind
// the store, change what the store writes?  The compiler treats that sequence as safe (its hazard table asks for a
// wait state only when the store has NO scalar offset).  Every lane keeps its four values in v[10:13], queues a few
// stores to a scratch area first (so that the store under test waits its turn in the memory pipeline), then:
//   buffer_store_dwordx4 v[10:13], voff, rsrc, soff offen
//   <writer>                                    ; overwrites v10 (and v11)
// and the host counts the lanes whose stored values are not the original ones.
// writer by variant: 0 = v_pk_add_f32 v[10:11]; 1 = s_nop 3, then the same; 2 = v_add_f32 v10; 3 = v_mov_b32 v10;
// 4 = v_pk_add_f32 with the store's scalar offset replaced by the immediate 0 (the case the compiler does pad:
// here without its padding); 5 = v_pk_mul_f32 v[10:11]; 6 = s_nop 0 in between; 7 / 9 = one other VALU instruction in
// between; 8 = v_accvgpr_read_b32 v10; 10 = v_pk_add_f32 v[12:13]; 11 / 12 = global_store_dwordx4 with one wait state and with
// none; 13 / 14 = buffer_store_dwordx2 with a packed / a plain writer right behind;
// 15 = global_store_dwordx4, one wait state, plain writer; 16 = two wait states (what the compiler puts there), packed writer;
// 17 / 18 = ds_write_b128 with a packed / a plain writer right behind.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/store_overwrite.hip -o tools/probe/store_overwrite && tools/probe/store_overwrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ void __launch_bounds__(256) probe(float* out, float* junk, const float* in, int pre) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    const float4 x = *(const float4*)(in + 4 * (size_t)gid);
    const unsigned long long base = (unsigned long long)out, jbase = (unsigned long long)junk;
    const unsigned bytes = gridDim.x * 256u * 16u;
    u32x4 rs = {(unsigned)base, (unsigned)(base >> 32), bytes, 0x00020000u};
    u32x4 js = {(unsigned)jbase, (unsigned)(jbase >> 32), bytes, 0x00020000u};
    // uniform values in SGPRs
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    js[0] = __builtin_amdgcn_readfirstlane(js[0]); js[1] = __builtin_amdgcn_readfirstlane(js[1]);
    js[2] = __builtin_amdgcn_readfirstlane(js[2]); js[3] = __builtin_amdgcn_readfirstlane(js[3]);
    const unsigned soff = __builtin_amdgcn_readfirstlane(blockIdx.x * 256u * 16u);     // the block's rows, as the scalar offset
    const unsigned voff = threadIdx.x * 16u;
    const float k = 1000.0f;
    const f32x4 xv = {x.x, x.y, x.z, x.w};
    for (int i = 0; i < pre; ++i)       // stores in front of the one under test
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(xv), "v"(voff), "s"(js), "s"(soff) : "memory");
#define HEAD "v_mov_b32 v10, %[a]\n\tv_mov_b32 v11, %[b]\n\tv_mov_b32 v12, %[c]\n\tv_mov_b32 v13, %[d]\n\t" \
             "v_mov_b32 v14, %[k]\n\tv_mov_b32 v15, %[k]\n\ts_nop 7\n\t"
#define STORE "buffer_store_dwordx4 v[10:13], %[voff], %[rs], %[soff] offen\n\t"
#define TAIL "\n\ts_waitcnt vmcnt(0)"
#define OPS : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [voff] "v"(voff), [rs] "s"(rs), [soff] "s"(soff) \
            : "v10", "v11", "v12", "v13", "v14", "v15", "memory"
#define OPS2 : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [voff] "v"(voff), [rs] "s"(rs), [soff] "s"(soff) \
            : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "a0", "memory"
    if constexpr (V == 0) asm volatile(HEAD STORE "v_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS);
    if constexpr (V == 1) asm volatile(HEAD STORE "s_nop 3\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS);
    if constexpr (V == 2) asm volatile(HEAD STORE "v_add_f32 v10, v10, v14" TAIL OPS);
    if constexpr (V == 3) asm volatile(HEAD STORE "v_mov_b32 v10, v14" TAIL OPS);
    if constexpr (V == 5) asm volatile(HEAD STORE "v_pk_mul_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS);
    if constexpr (V == 6) asm volatile(HEAD STORE "s_nop 0\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS);
    if constexpr (V == 7) asm volatile(HEAD STORE "v_mov_b32 v16, v14\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS2);
    if constexpr (V == 8) asm volatile(HEAD STORE "v_accvgpr_read_b32 v10, a0" TAIL OPS2);
    if constexpr (V == 9) asm volatile(HEAD STORE "v_mov_b32 v16, v14\n\tv_mov_b32 v10, v14" TAIL OPS2);
    if constexpr (V == 10) asm volatile(HEAD STORE "v_pk_add_f32 v[12:13], v[12:13], v[14:15]" TAIL OPS);
    if constexpr (V == 11) {                     // global_store_dwordx4 (no scalar offset exists), one wait state in between
        float* ptr = out + 4 * (size_t)gid;
        asm volatile(HEAD "global_store_dwordx4 %[p], v[10:13], off\n\ts_nop 0\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL
                     : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [p] "v"(ptr)
                     : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
    if constexpr (V == 12) {                     // ... and without it
        float* ptr = out + 4 * (size_t)gid;
        asm volatile(HEAD "global_store_dwordx4 %[p], v[10:13], off\n\tv_mov_b32 v10, v14" TAIL
                     : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [p] "v"(ptr)
                     : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
    // 13 / 14: a two-dword store (writes the first two values only; the host compares those), packed and plain writer
    if constexpr (V == 13) asm volatile(HEAD "buffer_store_dwordx2 v[10:11], %[voff], %[rs], %[soff] offen\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL OPS);
    if constexpr (V == 14) asm volatile(HEAD "buffer_store_dwordx2 v[10:11], %[voff], %[rs], %[soff] offen\n\tv_mov_b32 v11, v14" TAIL OPS);
    if constexpr (V == 15) {                     // one wait state, then a plain writer
        float* ptr = out + 4 * (size_t)gid;
        asm volatile(HEAD "global_store_dwordx4 %[p], v[10:13], off\n\ts_nop 0\n\tv_mov_b32 v10, v14" TAIL
                     : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [p] "v"(ptr)
                     : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
    if constexpr (V == 16) {                     // two wait states (the compiler's padding for this store), packed writer
        float* ptr = out + 4 * (size_t)gid;
        asm volatile(HEAD "global_store_dwordx4 %[p], v[10:13], off\n\ts_nop 1\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL
                     : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [p] "v"(ptr)
                     : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
    if constexpr (V == 17 || V == 18) {          // the same question for LDS: ds_write_b128, packed / plain writer right behind
        __shared__ float sh[256 * 4];
        const unsigned laddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)&sh[threadIdx.x * 4];
        if constexpr (V == 17)
            asm volatile(HEAD "ds_write_b128 %[l], v[10:13]\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]\n\ts_waitcnt lgkmcnt(0)"
                         : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [l] "v"(laddr)
                         : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
        else
            asm volatile(HEAD "ds_write_b128 %[l], v[10:13]\n\tv_mov_b32 v10, v14\n\ts_waitcnt lgkmcnt(0)"
                         : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [l] "v"(laddr)
                         : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
        __syncthreads();
        *(float4*)(out + 4 * (size_t)gid) = *(const float4*)&sh[threadIdx.x * 4];
    }
    if constexpr (V == 4) {
        const unsigned voff2 = voff + soff;
        asm volatile(HEAD "buffer_store_dwordx4 v[10:13], %[voff], %[rs], 0 offen\n\tv_pk_add_f32 v[10:11], v[10:11], v[14:15]" TAIL
                     : : [a] "v"(x.x), [b] "v"(x.y), [c] "v"(x.z), [d] "v"(x.w), [k] "v"(k), [voff] "v"(voff2), [rs] "s"(rs)
                     : "v10", "v11", "v12", "v13", "v14", "v15", "memory");
    }
}

template <int V>
static long run(int blocks, int pre, float* d_out, float* d_junk, const float* d_in, const std::vector<float>& h_in) {
    const size_t n = (size_t)blocks * 256 * 4;
    (void)hipMemset(d_out, 0, n * 4);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, d_out, d_junk, d_in, pre);
    if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d: launch failed\n", V); return -1; }
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < n; i += 4)
        if (h[i] != h_in[i] || h[i + 1] != h_in[i + 1] || (V < 13 && (h[i + 2] != h_in[i + 2] || h[i + 3] != h_in[i + 3]))) {
            if (bad < 2 && getenv("PROBE_SHOW"))
                printf("  variant %d lane %zu: stored %g %g %g %g, held %g %g %g %g\n", V, i / 4, h[i], h[i + 1], h[i + 2], h[i + 3],
                       h_in[i], h_in[i + 1], h_in[i + 2], h_in[i + 3]);
            bad++;
        }
    return bad;
}

int main() {
    const int blocks = 4096;
    const size_t n = (size_t)blocks * 256 * 4;
    std::vector<float> h_in(n);
    for (size_t i = 0; i < n; ++i) h_in[i] = (float)(i % 977) + 0.25f;
    float *d_in, *d_out, *d_junk;
    if (hipMalloc(&d_in, n * 4) != hipSuccess || hipMalloc(&d_out, n * 4) != hipSuccess || hipMalloc(&d_junk, n * 4) != hipSuccess) return 1;
    (void)hipMemcpy(d_in, h_in.data(), n * 4, hipMemcpyHostToDevice);
    const char* name[19] = {"v_pk_add_f32 right behind the store", "s_nop 3, then v_pk_add_f32", "v_add_f32 right behind the store",
                            "v_mov_b32 right behind the store", "v_pk_add_f32 behind a store with offset 0 (no SGPR)", "v_pk_mul_f32 right behind the store",
                            "s_nop 0, then v_pk_add_f32", "one other VALU, then v_pk_add_f32", "v_accvgpr_read_b32 right behind the store",
                            "one other VALU, then v_mov_b32", "v_pk_add_f32 into the LAST two data registers", "global_store_dwordx4, s_nop 0, v_pk_add_f32",
                            "global_store_dwordx4, v_mov_b32 right behind",
                            "buffer_store_dwordx2, v_pk_add_f32 right behind", "buffer_store_dwordx2, v_mov_b32 right behind",
                            "global_store_dwordx4, s_nop 0, v_mov_b32", "global_store_dwordx4, s_nop 1, v_pk_add_f32",
                            "ds_write_b128, v_pk_add_f32 right behind", "ds_write_b128, v_mov_b32 right behind"};
    for (int pre : {0, 8, 32}) {
        long bad[19] = {run<0>(blocks, pre, d_out, d_junk, d_in, h_in), run<1>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<2>(blocks, pre, d_out, d_junk, d_in, h_in), run<3>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<4>(blocks, pre, d_out, d_junk, d_in, h_in), run<5>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<6>(blocks, pre, d_out, d_junk, d_in, h_in), run<7>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<8>(blocks, pre, d_out, d_junk, d_in, h_in), run<9>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<10>(blocks, pre, d_out, d_junk, d_in, h_in), run<11>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<12>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<13>(blocks, pre, d_out, d_junk, d_in, h_in), run<14>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<15>(blocks, pre, d_out, d_junk, d_in, h_in), run<16>(blocks, pre, d_out, d_junk, d_in, h_in),
                        run<17>(blocks, pre, d_out, d_junk, d_in, h_in), run<18>(blocks, pre, d_out, d_junk, d_in, h_in)};
        for (int v = 0; v < 19; ++v)
            printf("stores in front: %2d | %-52s | lanes with other values stored: %ld of %d\n", pre, name[v], bad[v], blocks * 256);
    }
    return 0;
}
