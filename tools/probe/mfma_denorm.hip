// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  One wave: A = a single subnormal (2^-20) at (row 0, k 0),
// B = 1024 at (k 0, col 0): the product is 2^-10 if the subnormal is kept, 0 if it is flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if (threadIdx.x == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; float h = -1.f;
    hipMalloc(&d, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("product %g (kept: %g, flushed: 0)\n", h, 9.5367431640625e-07 * 1024.0);
    return 0;
}
