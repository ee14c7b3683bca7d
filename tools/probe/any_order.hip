// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?  Kernel A waits (at most a second)
// for a word that only kernel B sets; B is launched behind A on the same stream with the flag.  A's exit code says
// whether it saw the word (1) or gave up (2).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void ka(unsigned* w, unsigned* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 2;
    for (;;) {
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { r = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) break;      // one second at 100 MHz
        __builtin_amdgcn_s_sleep(50);
    }
    out[0] = r;
}
__global__ void kb(unsigned* w) { __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
    unsigned *w, *out, h = 0;
    if (hipMalloc(&w, 4) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 1;
    for (int flags = 0; flags < 2; ++flags) {
        (void)hipMemsetAsync(w, 0, 4, s);
        (void)hipMemsetAsync(out, 0, 4, s);
        hipLaunchKernelGGL(ka, dim3(1), dim3(64), 0, s, w, out);
        void* args[] = {&w};
        const hipError_t e = hipExtLaunchKernel((const void*)kb, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, flags);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
        printf("flags %d: launch rc %d, first kernel %s\n", flags, (int)e, h == 1 ? "saw the second kernel's word (side by side)" : "gave up (in order)");
    }
    return 0;
}
