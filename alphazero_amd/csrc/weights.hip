// weights.hip -- device-side weight preparation: the flat fp32 blob (module registration order of
// network.hpp:106-114, as torch::save walks it) -> the images the tower kernels read.  Eval-mode
// BatchNorm (eps 1e-5, network.hpp:13,29-30,71) is folded into the preceding convolution in double,
// 3x3 kernels are re-laid per tap and rounded to the MFMA operand format.  Everything runs on the
// context's stream, so a blob that arrived by an RCCL broadcast never visits the host and a swap
// between two evaluator launches needs no synchronisation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "az_weights.h"

__device__ __forceinline__ unsigned short w_to16(float x, int prec) {
    if (prec == 1) {                                          // fp16, saturating, round to nearest even
        x = fminf(fmaxf(x, -65504.f), 65504.f);
        const _Float16 h = (_Float16)x;
        return __builtin_bit_cast(unsigned short, h);
    }
    unsigned u = __float_as_uint(x);                          // bf16, round to nearest even
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// Row of output channel co in the fused kernel's weight images: its 16x16x32 tiles take weights as the B operand
// with lane l of tile a = 4 q + j standing for channel 64 q + 4 l + j, so that the four tiles of a quad give a lane
// four consecutive channels of a position (one 8-byte store).  Image row 16 a + l holds that channel.
__device__ __forceinline__ int fused_row(int co) { return (co & ~63) | ((co & 3) << 4) | ((co >> 2) & 15); }

__device__ __forceinline__ double bn_scale(const float* bn, int C, int ch) {     // bn: weight, bias, mean, var
    return (double)bn[ch] / sqrt((double)bn[3 * C + ch] + 1e-5);
}
__device__ __forceinline__ float bn_bias(const float* bn, int C, int ch, float conv_bias, double s) {
    return (float)(((double)conv_bias - bn[2 * C + ch]) * s + bn[C + ch]);
}

// Every value finite, every running_var + eps positive (a fold through sqrt of a negative number would
// silently kill the channel).  bad: bit 0 non-finite value, bit 1 non-positive variance.
__global__ void az_check_blob_kernel(AzFoldArgs a, int* bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int flag = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_floats; i += stride)
        if (!isfinite(a.blob[i])) flag |= 1;
    const int n_bn = 2 * a.nres + 1;                          // stem bn, then bn1/bn2 of every block
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n_bn * a.C + 1; i += stride) {
        float rv;
        if (i == (size_t)n_bn * a.C) rv = a.blob[a.off_vbn + 3];
        else {
            const int b = (int)(i / a.C), ch = (int)(i % a.C);
            const size_t base = b == 0 ? a.off_stem_bn
                                       : a.off_res0 + (size_t)((b - 1) >> 1) * a.res_stride + 2 * ((size_t)9 * a.C * a.C + a.C) + (size_t)((b - 1) & 1) * 4 * a.C;
            rv = a.blob[base + 3 * (size_t)a.C + ch];
        }
        if (!((double)rv + 1e-5 > 0.0)) flag |= 2;
    }
    if (flag) atomicOr(bad, flag);
}

__device__ __forceinline__ float w_from16(unsigned short h, int prec) {
    if (prec == 1) return (float)__builtin_bit_cast(_Float16, h);
    return __uint_as_float((unsigned)h << 16);
}

// Where the folded 16-bit weights of one output channel sit in fp16's range.  rng[0]: weights beyond +-65504 (the
// image would hold a clamped value), rng[1]: channels whose largest folded weight is nonzero but below 2^-14 (every
// weight of the row is an fp16 subnormal: less than 11 significant bits, down to none), rng[2] / rng[3]: largest
// weight and smallest nonzero row maximum as float bits (positive floats order like integers).  Called by every
// thread of a 256-thread block with its own (over, amax).
__device__ __forceinline__ void range_tally(int over, float amax, int* rng) {
    __shared__ int s_over[4];
    __shared__ float s_max[4];
    for (int o = 32; o > 0; o >>= 1) { over += __shfl_xor(over, o, 64); amax = fmaxf(amax, __shfl_xor(amax, o, 64)); }
    if ((threadIdx.x & 63) == 0) { s_over[threadIdx.x >> 6] = over; s_max[threadIdx.x >> 6] = amax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int ov = s_over[0] + s_over[1] + s_over[2] + s_over[3];
        const float mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (ov) atomicAdd(rng, ov);
        if (mx > 0.f && mx < 7.62939453125e-06f) atomicAdd(rng + 1, 1);
        atomicMax(rng + 2, __float_as_int(mx));
        if (mx > 0.f) atomicMin(rng + 3, __float_as_int(mx));
    }
    __syncthreads();
}

// grid (2 * nres, C): one block per (convolution, output channel); threads stride its C * 9 weights.
// rng != null: a dry run that only tallies the folded weights' range (nothing is written: a blob that is refused
// leaves the images in use untouched).  a.part = 1 writes the compensated mode's image a.conv_w3 instead: three times
// the input channels, [W_hi | W_hi | W_lo] with W_lo = 16-bit(w - W_hi), against activation rows [hi | lo | hi].
__global__ void __launch_bounds__(256) az_fold_conv_kernel(AzFoldArgs a, int* rng) {
    const int L = blockIdx.x, co = blockIdx.y, C = a.C;
    const size_t wsz = (size_t)9 * C * C + C;
    const float* blk = a.blob + a.off_res0 + (size_t)(L >> 1) * a.res_stride;
    const float* w = blk + (size_t)(L & 1) * wsz;
    const float* bn = blk + 2 * wsz + (size_t)(L & 1) * 4 * C;
    const double s = bn_scale(bn, C, co);
    if (threadIdx.x == 0 && !rng && !a.part) a.conv_b[(size_t)L * a.Cp + co] = bn_bias(bn, C, co, w[(size_t)9 * C * C + co], s);
    const int nblk = a.Cp / 128;
    int over = 0;
    float amax = 0.f;
    for (int e = threadIdx.x; e < C * 9; e += 256) {
        const int ci = e / 9, tap = e % 9;
        const float f = (float)(w[(size_t)co * C * 9 + e] * s);
        if (rng) { over += fabsf(f) > 65504.f; amax = fmaxf(amax, fabsf(f)); continue; }
        const unsigned short v = w_to16(f, a.prec);
        if (a.part) {                                         // [layer][nb][sc over 3 Cp][tap][n][k]
            const unsigned short lo = w_to16(f - w_from16(v, a.prec), a.prec);
            const int row = fused_row(co % 128);
            for (int part = 0; part < 3; ++part) {
                const size_t chunk = ((size_t)(co / 128) * 3 * nblk + part * nblk + ci / 128) * 9 + tap;
                a.conv_w3[(size_t)L * 3 * a.conv_layer_elems + chunk * (128 * 128) + (size_t)row * 128 + ci % 128] = part == 2 ? lo : v;
            }
        } else if (a.fused)                                   // one 32 KB chunk per tap: [layer][tap][n][k]
            a.conv_wf[(((size_t)L * 9 + tap) * 128 + fused_row(co)) * 128 + ci] = v;
        else {                                                // [layer][nb][sc][tap][n][k]
            const size_t chunk = ((size_t)(co / 128) * nblk + ci / 128) * 9 + tap;
            const int row = a.layer16 ? fused_row(co % 128) : co % 128;
            a.conv_w[(size_t)L * a.conv_layer_elems + chunk * (128 * 128) + (size_t)row * 128 + ci % 128] = v;
        }
    }
    if (rng) range_tally(over, amax, rng);
}

// The stem's folded weights are 16-bit operands only in the fused kernel: one block, a row (output channel) at a time.
__global__ void __launch_bounds__(256) az_range_stem_kernel(AzFoldArgs a, int* rng) {
    const int C = a.C, KW = a.c_in * 9;
    const float* sw = a.blob + a.off_stem_w;
    const float* sbn = a.blob + a.off_stem_bn;
    for (int co = 0; co < C; ++co) {
        float f = 0.f;
        if (threadIdx.x < KW) f = fabsf((float)(sw[co * KW + threadIdx.x] * bn_scale(sbn, C, co)));
        range_tally(f > 65504.f, f, rng);
    }
}

// Stem and heads: one block.
__global__ void __launch_bounds__(256) az_fold_small_kernel(AzFoldArgs a) {
    const int C = a.C, Cp = a.Cp, A = a.A, KW = a.c_in * 9;
    const float* sw = a.blob + a.off_stem_w;
    const float* sb = sw + (size_t)C * KW;
    const float* sbn = a.blob + a.off_stem_bn;
    for (int e = threadIdx.x; e < C * KW; e += 256) {
        const int co = e / KW, k = e % KW;
        const float f = (float)(sw[e] * bn_scale(sbn, C, co));
        if (a.fused) a.stem_wbf[(size_t)fused_row(co) * 32 + k] = w_to16(f, a.prec);
        else a.stem_w[e] = f;
    }
    for (int co = threadIdx.x; co < C; co += 256) a.stem_b[co] = bn_bias(sbn, C, co, sb[co], bn_scale(sbn, C, co));
    // heads: policy 1x1 as is; value 1x1 with its one-channel BatchNorm folded
    const float* pc = a.blob + a.off_pc;                      // w[c_out][C], b[c_out]
    const float* vc = a.blob + a.off_vc;                      // w[C], b
    const float* vbn = a.blob + a.off_vbn;                    // weight, bias, mean, var
    const double vs = bn_scale(vbn, 1, 0);
    const float* fc1w = a.blob + a.off_fc1;                   // [64][A], then fc1b[64], fc2w[vd][64], fc2b[vd]
    const float bvf = bn_bias(vbn, 1, 0, vc[C], vs);
    const int tail = 64 + 64 * a.vd + a.vd;
    if (a.fused) {                                            // c_out = 1, vd = 2
        float* ht = a.head_pt;                                // {policy, value} weight pairs, biases at [256], [257]
        for (int ch = threadIdx.x; ch < C; ch += 256) { ht[2 * ch] = pc[ch]; ht[2 * ch + 1] = (float)(vc[ch] * vs); }
        if (threadIdx.x == 0) { ht[256] = pc[C]; ht[257] = bvf; }
        for (int e = threadIdx.x; e < 64 * A; e += 256) ht[258 + (size_t)(e % A) * 64 + e / A] = fc1w[e];   // fc1 transposed [A][64]
        for (int e = threadIdx.x; e < tail; e += 256) ht[258 + (size_t)64 * A + e] = fc1w[(size_t)64 * A + e];
    } else {
        float* hp = a.head_p;                                 // wp[c_out][Cp], wv[Cp], bp[c_out], bv, fc1 [A][64], fc1b, fc2w, fc2b
        for (int e = threadIdx.x; e < a.c_out * C; e += 256) hp[(size_t)(e / C) * Cp + e % C] = pc[e];
        float* wv = hp + (size_t)a.c_out * Cp;
        for (int ch = threadIdx.x; ch < C; ch += 256) wv[ch] = (float)(vc[ch] * vs);
        float* bp = wv + Cp;
        for (int co = threadIdx.x; co < a.c_out; co += 256) bp[co] = pc[(size_t)a.c_out * C + co];
        if (threadIdx.x == 0) bp[a.c_out] = bvf;
        float* ft = bp + a.c_out + 1;                             // fc1 transposed [A][64], then fc1b, fc2w, fc2b as they are
        for (int e = threadIdx.x; e < 64 * A; e += 256) ft[(size_t)(e % A) * 64 + e / A] = fc1w[e];
        for (int e = threadIdx.x; e < tail; e += 256) ft[(size_t)64 * A + e] = fc1w[(size_t)64 * A + e];
    }
}

extern "C" void az_launch_check_blob(hipStream_t s, const AzFoldArgs* a, int* bad) {
    hipLaunchKernelGGL(az_check_blob_kernel, dim3(1024), dim3(256), 0, s, *a, bad);
}
// rng: int[4] on the device, {0, 0, 0, 0x7F800000} before the call (see range_tally)
extern "C" void az_launch_fold_range(hipStream_t s, const AzFoldArgs* a, int* rng) {
    hipLaunchKernelGGL(az_fold_conv_kernel, dim3(2 * a->nres, a->C), dim3(256), 0, s, *a, rng);
    if (a->fused) hipLaunchKernelGGL(az_range_stem_kernel, dim3(1), dim3(256), 0, s, *a, rng);
}
extern "C" void az_launch_fold(hipStream_t s, const AzFoldArgs* a) {
    hipLaunchKernelGGL(az_fold_conv_kernel, dim3(2 * a->nres, a->C), dim3(256), 0, s, *a, (int*)nullptr);
    if (a->conv_w3) {
        AzFoldArgs lo = *a;
        lo.part = 1;
        hipLaunchKernelGGL(az_fold_conv_kernel, dim3(2 * a->nres, a->C), dim3(256), 0, s, lo, (int*)nullptr);
    }
    hipLaunchKernelGGL(az_fold_small_kernel, dim3(1), dim3(256), 0, s, *a);
}
