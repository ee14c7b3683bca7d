// tower.hip -- the ResNet policy/value tower (PVNetwork::forward, eval mode,
// network.hpp:117-123) as hand-written gfx950 kernels.
//
//   az_tower_fused_kernel  the whole tower (stem, all ResBlocks, both heads) in ONE
//                     launch, activations resident in LDS (C <= 128, (N+1)^2 <= 256)
//   az_stem_kernel    ConvBlock (network.hpp:15-17): 3x3 conv on the u8 planes,
//                     BN folded, ReLU; fp32 VALU                       } per-layer
//   az_conv3x3_t256s_kernel (default), az_conv3x3_t256_kernel,          } path for
//   az_conv3x3_t128_kernel                                              } wider or
//                     ResBlock convolutions (network.hpp:32-38) as      } larger
//                     implicit GEMM: t256s on v_mfma_f32_16x16x32 with  } shapes
//                     D[pos][ch] = sum_k X[pos][k] W[ch][k], the other  }
//                     two on v_mfma_f32_32x32x16 with D[ch][pos]        }
//   az_heads_kernel   PolicyHead + ValueHead (network.hpp:51-60,75-83)  }
//
// Activation layout in HBM: bf16 rows [guard + M_pad + guard][C_pad]; one row
// per board position in a zero-padded (N+1)x(N+1) raster (column N and row N
// of every board are kept zero), so every 3x3 tap of a real cell is a plain
// row offset dy*(N+1)+dx and padding never needs a mask.  Boards are
// consecutive, 256 rows form one output tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "az_tower.h"
#include "az_dev.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TILE_M 256
#define A_STRIDE 272            // 128 ch * 2 B + 16 B pad: conflict-free ds_read_b128

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// 16-bit operand formats of the MFMA path.  PR = 0: bf16 (8-bit significand,
// fp32 range); PR = 1: fp16 (11-bit significand, saturated at +-65504).  Both
// accumulate in fp32 and run at the same MFMA rate.
template <int PR> struct Prec;
template <> struct Prec<0> {
    typedef bf16x8 vec;
    static __device__ __forceinline__ unsigned short to16(float x) {
        unsigned u = __float_as_uint(x);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
    static __device__ __forceinline__ float from16(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // relu + round-to-nearest-even pack of two values: v_max x2 + one v_cvt_pk_bf16_f32
    static __device__ __forceinline__ unsigned pack2_relu(float a, float b) {
        typedef __bf16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {fmaxf(a, 0.f), fmaxf(b, 0.f)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
    }
    // largest storable activation; pack2 takes values already clamped to [0, cap]
    static __device__ __forceinline__ float cap() { return __builtin_inff(); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef __bf16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
    }
};
template <> struct Prec<1> {
    typedef f16x8 vec;
    static __device__ __forceinline__ unsigned short to16(float x) {
        x = fminf(fmaxf(x, -65504.f), 65504.f);
        const _Float16 h = (_Float16)x;
        return __builtin_bit_cast(unsigned short, h);
    }
    static __device__ __forceinline__ float from16(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
    static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // relu + saturation in one v_med3_f32 each, then one packed convert
    static __device__ __forceinline__ unsigned pack2_relu(float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {__builtin_amdgcn_fmed3f(a, 0.f, 65504.f), __builtin_amdgcn_fmed3f(b, 0.f, 65504.f)};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
    }
    static __device__ __forceinline__ float cap() { return 65504.f; }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 f = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2));
    }
};

// The residual stream as a pair of 16-bit values: x = hi + lo / AZ_LO_SCALE, hi = x rounded to the operand format
// (the copy the next convolution reads), lo = the remainder x - hi (exact in fp32), scaled by 2^11 before it is
// rounded: unscaled, the remainder of an activation below 0.125 is a subnormal fp16 number and keeps few or no bits.
#define AZ_LO_SCALE 2048.0f
#define AZ_LO_INV (1.0f / 2048.0f)

// ------------------------------------------------------------------- stem
// One block per board.  w: [C_pad][27] fp32 (BN folded), bias [C_pad].
template <int PR>
__global__ void __launch_bounds__(256)
az_stem_kernel(const uint8_t* __restrict__ planes, unsigned short* __restrict__ out, float* __restrict__ out32,
               unsigned short* __restrict__ out_lo /* non-null: the stream as out + out_lo (16-bit pair) instead of out32 */,
               const float* __restrict__ w, const float* __restrict__ bias,
               int N, int C_pad, int Gd, int c_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int A = N * N, W = N + 1, P = W * W, KW = c_in * 9;
    float* s_w = (float*)smem;                       // [C_pad][c_in * 9]
    float* s_b = s_w + C_pad * KW;                   // [C_pad]
    uint8_t* s_in = (uint8_t*)(s_b + C_pad);         // [c_in][A]
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < C_pad * KW; i += 256) s_w[i] = w[i];
    for (int i = threadIdx.x; i < C_pad; i += 256) s_b[i] = bias[i];
    for (int i = threadIdx.x; i < c_in * A; i += 256) s_in[i] = planes[(size_t)b * c_in * A + i];
    __syncthreads();
    const int cpairs = C_pad / 2;
    for (int idx = threadIdx.x; idx < P * cpairs; idx += 256) {
        const int p = idx / cpairs, c = (idx % cpairs) * 2;
        const int y = p / W, x = p % W;
        float a0 = 0.f, a1 = 0.f;
        if (y < N && x < N) {
            a0 = s_b[c]; a1 = s_b[c + 1];
            for (int ci = 0; ci < c_in; ++ci)
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= N) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= N) continue;
                        const float v = (float)s_in[ci * A + yy * N + xx];
                        a0 += v * s_w[c * KW + ci * 9 + ky * 3 + kx];
                        a1 += v * s_w[(c + 1) * KW + ci * 9 + ky * 3 + kx];
                    }
                }
            a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f);
        }
        const unsigned pk = (unsigned)Prec<PR>::to16(a0) | ((unsigned)Prec<PR>::to16(a1) << 16);
        *(unsigned*)(out + ((size_t)(Gd + b * P + p)) * C_pad + c) = pk;
        if (out_lo)                                                                           // the stream = this 16-bit copy + a 16-bit remainder
            *(unsigned*)(out_lo + ((size_t)(Gd + b * P + p)) * C_pad + c) =
                (unsigned)Prec<PR>::to16((a0 - Prec<PR>::from16((unsigned short)pk)) * AZ_LO_SCALE) |
                ((unsigned)Prec<PR>::to16((a1 - Prec<PR>::from16((unsigned short)(pk >> 16))) * AZ_LO_SCALE) << 16);
        else
            *(float2*)(out32 + ((size_t)(Gd + b * P + p)) * C_pad + c) = make_float2(a0, a1); // the residual stream stays fp32
    }
}

// The same stem as an fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products of fp32 weights and the u8 planes):
// rows = 32 output channels, columns = 32 raster positions, K = CIN * 9 taken two at a time.  A wave takes every
// fourth block of 32 positions and four row blocks (128 channels) at once; lane l carries position l & 31 and the
// k half kh = l >> 5, i.e. im2col entry k = 2 t + kh of MFMA step t.  A lane's weights for a 128-channel slab stay
// in registers across its position blocks; its im2col entries are single bytes of the board's planes in LDS.
template <int PR, int CIN>
__global__ void __launch_bounds__(256)
az_stem_mfma_kernel(const uint8_t* __restrict__ planes, unsigned short* __restrict__ out, float* __restrict__ out32,
                    unsigned short* __restrict__ out_lo /* non-null: the stream as out + out_lo instead of out32 */,
                    const float* __restrict__ w, const float* __restrict__ bias, int N, int C_pad, int Gd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16s;
    constexpr int KW = CIN * 9, T = (KW + 1) / 2;
    const int A = N * N, W = N + 1, P = W * W;
    float* s_bs = (float*)smem;                      // [C_pad]: the biases (read per position block; a trip to memory each otherwise)
    uint8_t* s_in = smem + C_pad * 4;                // [CIN][A]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < C_pad; i += 256) s_bs[i] = bias[i];
    for (int i = tid; i < CIN * A; i += 256) s_in[i] = planes[(size_t)b * CIN * A + i];
    __syncthreads();
    const int lane = tid & 63, wvi = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    for (int c0 = 0; c0 < C_pad; c0 += 128) {
        float wr[4][T];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int k = 2 * t + kh;
                wr[m][t] = k < KW ? w[(size_t)(c0 + 32 * m + l31) * KW + k] : 0.f;
            }
        for (int pb = wvi; pb * 32 < P; pb += 4) {
            const int p = pb * 32 + l31;
            const int y = p / W, x = p % W;
            const bool cell = p < P && y < N && x < N;
            unsigned vm = 0;                         // bit tap: the tap's source cell is on the board
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (cell && yy >= 0 && yy < N && xx >= 0 && xx < N) vm |= 1u << tap;
            }
            const int base = y * N + x;
            f32x16s acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int k0 = 2 * t, k1 = 2 * t + 1;
                const int tap0 = k0 % 9, tap1 = k1 % 9;
                const int off0 = (k0 / 9) * A + (tap0 / 3 - 1) * N + (tap0 % 3 - 1);
                const int off1 = (k1 / 9) * A + (tap1 / 3 - 1) * N + (tap1 % 3 - 1);
                const bool ok = kh ? (k1 < KW && (vm >> tap1 & 1u)) : (vm >> tap0 & 1u) != 0;
                const int src = ok ? base + (kh ? off1 : off0) : 0;
                const float v = ok ? (float)s_in[src] : 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[m][t], v, acc[m], 0, 0, 0);
            }
            if (p < P) {
                const size_t orow = ((size_t)(Gd + b * P + p)) * C_pad + c0 + 4 * kh;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *(const float4*)(s_bs + c0 + 32 * m + 8 * q + 4 * kh);
                        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (cell) o = make_float4(fmaxf(acc[m][4 * q] + bq.x, 0.f), fmaxf(acc[m][4 * q + 1] + bq.y, 0.f),
                                                  fmaxf(acc[m][4 * q + 2] + bq.z, 0.f), fmaxf(acc[m][4 * q + 3] + bq.w, 0.f));
                        uint2 pk;
                        pk.x = (unsigned)Prec<PR>::to16(o.x) | ((unsigned)Prec<PR>::to16(o.y) << 16);
                        pk.y = (unsigned)Prec<PR>::to16(o.z) | ((unsigned)Prec<PR>::to16(o.w) << 16);
                        *(uint2*)(out + orow + 32 * m + 8 * q) = pk;
                        if (out_lo) {                                       // the stream = this 16-bit copy + a 16-bit remainder
                            uint2 lo;
                            lo.x = (unsigned)Prec<PR>::to16((o.x - Prec<PR>::from16((unsigned short)pk.x)) * AZ_LO_SCALE) |
                                   ((unsigned)Prec<PR>::to16((o.y - Prec<PR>::from16((unsigned short)(pk.x >> 16))) * AZ_LO_SCALE) << 16);
                            lo.y = (unsigned)Prec<PR>::to16((o.z - Prec<PR>::from16((unsigned short)pk.y)) * AZ_LO_SCALE) |
                                   ((unsigned)Prec<PR>::to16((o.w - Prec<PR>::from16((unsigned short)(pk.y >> 16))) * AZ_LO_SCALE) << 16);
                            *(uint2*)(out_lo + orow + 32 * m + 8 * q) = lo;
                        } else
                            *(float4*)(out32 + orow + 32 * m + 8 * q) = o;  // the residual stream stays fp32
                    }
            }
        }
    }
}

// The stem once more, on the 16-bit MFMAs without giving up the fp32 weights: every folded weight goes in as a pair
// of fp16 numbers, w = hi + lo / 2048 (hi = w rounded to fp16, or 0 where that would be a subnormal; lo = the remainder,
// exact in fp32, scaled so that it keeps its bits), and the planes as the pair (v, v / 2048), both exact for a byte.
// v_mfma_f32_32x32x16_f16 forms exact products and adds in fp32, so acc += hi * v and acc += lo * (v / 2048) give the
// fp32 convolution to 2^-22 of each weight, at a sixteenth of the fp32 MFMA's cycles per product.  Rows = 32 output
// channels, columns = 32 raster positions, K = CIN * 9 in steps of 16 (lane l: position l & 31, k = 8 (l >> 5) + j).
// The outputs of a wave's 32 positions x 128 channels go through a 32 x 256-byte image in LDS (the wave's own, no
// block-wide barrier) so that every store instruction writes four whole 256-byte rows: straight from the
// accumulators a store covers 16 bytes of each of 32 rows.
#define STEM_SROW 272            // staged row: 256 bytes + 16 (bank spread of the 8- and 16-byte writes)
template <int PR, int CIN>
__global__ void __launch_bounds__(256, 2)
az_stem_split_kernel(const uint8_t* __restrict__ planes, unsigned short* __restrict__ out, float* __restrict__ out32,
                     unsigned short* __restrict__ out_lo /* non-null: the stream as out + out_lo instead of out32 */,
                     const float* __restrict__ w, const float* __restrict__ bias, int N, int C_pad, int Gd,
                     int B, int bpb /* boards per block: the blocks of 32 positions of all of them are dealt out to the four
                                       waves in turn (14 blocks of a 20x20 board alone leave two waves a block short), and
                                       the weight image is built once for all of them */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int KW = CIN * 9, KS = (KW + 15) / 16;
    const int A = N * N, W = N + 1, P = W * W;
    const int SI = (CIN * A + 15) & ~15;             // bytes of one board's planes in LDS
    const int b0 = blockIdx.x * bpb, nb = min(bpb, B - b0), tid = threadIdx.x;
    const int lane = tid & 63, wvi = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    float* s_bs = (float*)smem;                      // [C_pad]
    uint8_t* s_in0 = smem + C_pad * 4;               // [bpb][CIN][A]
    unsigned char* stage = smem + C_pad * 4 + bpb * SI + wvi * 32 * STEM_SROW;
    for (int i = tid; i < C_pad; i += 256) s_bs[i] = bias[i];
    for (int bi = 0; bi < nb; ++bi)
        for (int i = tid; i < CIN * A; i += 256) s_in0[bi * SI + i] = planes[(size_t)(b0 + bi) * CIN * A + i];
    __syncthreads();
    h8* s_w = (h8*)(smem + C_pad * 4 + bpb * SI + 4 * 32 * STEM_SROW);   // [hi, lo][KS][4 tiles][64 lanes]
    const int PB = (P + 31) / 32;
    for (int c0 = 0; c0 < C_pad; c0 += 128) {
        // the weight pairs of a 128-channel slab, as the operand registers of lane `lane` for tile m: wave m builds tile m
        if (c0) __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 wh, wl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * kh + j;
                const float wv = k < KW ? w[(size_t)(c0 + 32 * wvi + l31) * KW + k] : 0.f;
                float hf = (float)(_Float16)fminf(fmaxf(wv, -65504.f), 65504.f);
                if (fabsf(hf) < 6.103515625e-05f) hf = 0.f;
                wh[j] = (_Float16)hf;
                wl[j] = (_Float16)((wv - hf) * 2048.f);
            }
            s_w[(ks * 4 + wvi) * 64 + lane] = wh;
            s_w[((KS + ks) * 4 + wvi) * 64 + lane] = wl;
        }
        __syncthreads();
        for (int f = wvi; f < nb * PB; f += 4) {
            const int bi = f / PB, pb = f - bi * PB, b = b0 + bi;
            const uint8_t* s_in = s_in0 + bi * SI;
            const int p = pb * 32 + l31;
            const int y = p / W, x = p % W;
            const bool cell = p < P && y < N && x < N;
            unsigned vm = 0;                         // bit tap: the tap's source cell is on the board
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (cell && yy >= 0 && yy < N && xx >= 0 && xx < N) vm |= 1u << tap;
            }
            const int base = y * N + x;
            f32x16 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                h8 bh;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ka = 16 * ks + j, kb = ka + 8;                      // this entry for k half 0 / 1
                    const int ta = ka % 9, tb = kb % 9;
                    const int offa = (ka / 9) * A + (ta / 3 - 1) * N + (ta % 3 - 1);
                    const int offb = (kb / 9) * A + (tb / 3 - 1) * N + (tb % 3 - 1);
                    const bool ok = kh ? (kb < KW && (vm >> tb & 1u)) : (ka < KW && (vm >> ta & 1u));
                    const int src = ok ? base + (kh ? offb : offa) : 0;
                    bh[j] = ok ? (_Float16)(float)s_in[src] : (_Float16)0.f;
                }
                const h8 bl = bh * (_Float16)0.00048828125f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[((KS + ks) * 4 + m) * 64 + lane], bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(ks * 4 + m) * 64 + lane], bh, acc[m], 0, 0, 0);
                }
            }
            // bias + ReLU in place; register 4 q + i of tile m is channel c0 + 32 m + 8 q + 4 kh + i of position l31
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *(const float4*)(s_bs + c0 + 32 * m + 8 * q + 4 * kh);
                    acc[m][4 * q] = cell ? fmaxf(acc[m][4 * q] + bq.x, 0.f) : 0.f;
                    acc[m][4 * q + 1] = cell ? fmaxf(acc[m][4 * q + 1] + bq.y, 0.f) : 0.f;
                    acc[m][4 * q + 2] = cell ? fmaxf(acc[m][4 * q + 2] + bq.z, 0.f) : 0.f;
                    acc[m][4 * q + 3] = cell ? fmaxf(acc[m][4 * q + 3] + bq.w, 0.f) : 0.f;
                }
            const size_t row0 = (size_t)(Gd + b * P + pb * 32);
            // the staged image -> rows of `pitch` bytes at g0: lane i moves bytes [16 (i & 15), +16) of row 4 it + (i >> 4)
            auto flush = [&](unsigned char* g0, size_t pitch) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                uint4 v[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) v[it] = *(const uint4*)(stage + (4 * it + (lane >> 4)) * STEM_SROW + 16 * (lane & 15));
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int r = 4 * it + (lane >> 4);
                    if (pb * 32 + r < P) *(uint4*)(g0 + (size_t)r * pitch + 16 * (lane & 15)) = v[it];
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            };
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 pk;
                    pk.x = (unsigned)Prec<PR>::to16(acc[m][4 * q]) | ((unsigned)Prec<PR>::to16(acc[m][4 * q + 1]) << 16);
                    pk.y = (unsigned)Prec<PR>::to16(acc[m][4 * q + 2]) | ((unsigned)Prec<PR>::to16(acc[m][4 * q + 3]) << 16);
                    *(uint2*)(stage + l31 * STEM_SROW + (32 * m + 8 * q + 4 * kh) * 2) = pk;
                }
            flush((unsigned char*)(out + row0 * C_pad + c0), (size_t)C_pad * 2);
            if (out_lo) {                                                   // the stream = the 16-bit copy + a 16-bit remainder
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float r4[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float o = acc[m][4 * q + i];
                            r4[i] = (o - Prec<PR>::from16(Prec<PR>::to16(o))) * AZ_LO_SCALE;
                        }
                        uint2 lo;
                        lo.x = (unsigned)Prec<PR>::to16(r4[0]) | ((unsigned)Prec<PR>::to16(r4[1]) << 16);
                        lo.y = (unsigned)Prec<PR>::to16(r4[2]) | ((unsigned)Prec<PR>::to16(r4[3]) << 16);
                        *(uint2*)(stage + l31 * STEM_SROW + (32 * m + 8 * q + 4 * kh) * 2) = lo;
                    }
                flush((unsigned char*)(out_lo + row0 * C_pad + c0), (size_t)C_pad * 2);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {                               // the fp32 stream, 64 channels (256 bytes of a row) at a time
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int m = 2 * h + mm;
                            *(float4*)(stage + l31 * STEM_SROW + (32 * mm + 8 * q + 4 * kh) * 4) =
                                make_float4(acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]);
                        }
                    flush((unsigned char*)(out32 + row0 * C_pad + c0 + 64 * h), (size_t)C_pad * 4);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- conv 3x3
// Per-layer implicit GEMMs (any C multiple of 128, any N <= 21): used where the fused kernel does not
// apply (C = 256, or boards larger than one tile).  Shared pieces first.
#define W_BUF_BYTES (128 * 256)  // one tap of weights: 128 rows x 128 k, 16-bit
typedef __attribute__((address_space(3))) void lds_void;
// Every LDS-DMA piece this wave has issued has landed (s_waitcnt vmcnt(0); gfx9 encoding: vmcnt in
// bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima).  Must precede the barrier
// that hands a weight buffer to the other waves: the compiler orders a wave's own LDS reads after
// its own DMA, but a barrier alone says nothing about the pieces the OTHER waves are waiting for.
#define AZ_DMA_LANDED() __builtin_amdgcn_s_waitcnt(0x0F70)

// Behind a buffer_store_dwordx4 whose offset comes in a scalar register, before anything may write the registers the
// store takes its data from.  The MI355X reads those registers a quarter wave at a time after the store has issued,
// and an instruction right behind it that writes one of them can get there first: v_pk_add_f32 changed what a quarter
// of the lanes stored (the compensated mode's fp32 stream came out different from run to run), a plain v_mov_b32 /
// v_add_f32 / v_accvgpr_read_b32 what one lane in two thousand stored; one instruction in between and the stored
// values were the right ones every time (tools/probe/store_overwrite.hip).  The compiler pads that sequence (with
// two wait states) only for stores WITHOUT a scalar offset.  The data registers are the asm's operand, so whatever overwrites them is
// ordered behind it, and the two wait states are the distance.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void wide_store_guard(const u32x4& d) { asm volatile("s_nop 1" : : "v"(d)); }

// ReLU, the 16-bit format's saturation and the padding mask in ONE v_med3_f32 per value:
// cap is Prec::cap() for a real board cell and 0 for a padding row of the raster (whose
// LDS copy must stay zero: neighbours read it as the convolution's zero padding).
__device__ __forceinline__ float relu_cap(float x, float cap) { return __builtin_amdgcn_fmed3f(x, 0.f, cap); }

// ------------------------------------------------- conv 3x3, two workgroups per CU
// Implicit GEMM on 128-row tiles: 80 KB of LDS and <= 256 registers per wave, so TWO workgroups
// share a CU and one computes while the other waits for HBM (activation slab, residual tile,
// output rows).
// grid (M_pad/128, C_pad/128), 256 threads = 4 waves; wave w owns positions [32w, 32w+32) and
// all 128 output channels: 4 accumulator tiles of 32(ch) x 32(pos).  The activation tile with its
// halo is staged once per 128-channel slab and reused by all 9 taps.  Weights are pre-packed as
// [nb][sc][tap][n=128][k=128] images (BN scale folded) and stream as half-tap chunks [n=128][k=64]
// (16 KB) through a double buffer by buffer-addressed LDS DMA, 16-byte piece c of row n at piece
// c ^ ((n>>1)&7) (128-byte rows: two rows per bank sweep), applied through the source address.
// The output goes back through LDS so that HBM sees whole 256-byte rows.
#define T2_M 128
#define W2_BYTES (128 * 128)     // half a tap of weights: 128 rows x 64 k, 16-bit

template <int PR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))   // LDS allows two workgroups: 256 registers each
az_conv3x3_t128_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                       float* resid /* fp32 residual stream, updated in place; null: none */, const unsigned short* __restrict__ wpk,
                       const float* __restrict__ bias, int C_pad, int M, int N, int Gd, int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = N + 1, P = W * W;
    const int a_rows = T2_M + 2 * Gd;
    unsigned char* sA = smem;
    unsigned char* sB = smem + (size_t)a_rows * A_STRIDE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * T2_M;
    const int nb = blockIdx.y;
    const int n_sc = C_pad / 128;
    const int total = n_sc * 18;                             // half-tap chunks of this column block

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // weight stream: chunk t = (sc*9 + tap)*2 + half lives in the packed image at
    // ((nb*n_sc + sc)*9 + tap) * 32 KB, columns [64*half, 64*half + 64) of every row
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wpk + (size_t)nb * n_sc * 9 * (128 * 128)), 0, n_sc * 9 * (128 * 128 * 2), 0x00020000);
    // piece q = it*256 + tid of a chunk: LDS row (q >> 3) = it*32 + (tid >> 3), physical piece tid & 7,
    // logical piece = physical ^ ((row >> 1) & 7) = (tid & 7) ^ ((tid >> 4) & 7) for every it
    const unsigned st_off = (unsigned)((tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2u;
    unsigned char* const st_dst = sB + __builtin_amdgcn_readfirstlane(wv) * 1024;
#define STAGE_W2(t_, it_) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
        wrsrc, (lds_void*)(st_dst + ((t_) & 1) * W2_BYTES + (it_) * 4096), 16, st_off, \
        ((t_) >> 1) * (128 * 128 * 2) + ((t_) & 1) * 128 + (it_) * 8192, 0, 0)
    // activation slab with its halo: all of a thread's loads (<= 11 rows) are in flight together
    // (a macro, not a lambda: the staging array must stay in registers)
#define STAGE_A2(sc_) do { \
        uint4 sv_[11]; \
        _Pragma("unroll") for (int u = 0; u < 11; ++u) { \
            const int r_ = min((tid >> 4) + 16 * u, a_rows - 1); \
            sv_[u] = *(const uint4*)(in + (size_t)(m0 + r_) * C_pad + (sc_) * 128 + (tid & 15) * 8); \
        } \
        _Pragma("unroll") for (int u = 0; u < 11; ++u) { \
            const int r_ = (tid >> 4) + 16 * u; \
            if (r_ < a_rows) *(uint4*)(sA + r_ * A_STRIDE + (tid & 15) * 16) = sv_[u]; \
        } \
    } while (0)

#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W2(0, it);
#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W2(1, it);          // total >= 18: chunk 1 always exists
    STAGE_A2(0);
    AZ_DMA_LANDED();
    __syncthreads();

    const int posrow0 = Gd + 32 * wv + l31;                  // LDS row of this lane's position
    const int swz = (l31 >> 1) & 7;
    typename Prec<PR>::vec fx[2], fw[2][4];
    {
        const unsigned char* bA = sA + (posrow0 - W - 1) * A_STRIDE + h * 16;
        const unsigned char* bB = sB + l31 * 128;
        fx[0] = *(const typename Prec<PR>::vec*)(bA);
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[0][i] = *(const typename Prec<PR>::vec*)(bB + i * 32 * 128 + ((h ^ swz) << 4));
    }
    for (int t = 0; t < total; ++t) {
        const int tap = (t >> 1) % 9, hf = t & 1;
        const int tn_ = t + 1 < total ? t + 1 : t;           // the chunk after this one (the last re-reads itself)
        const int ntap = (tn_ >> 1) % 9, nhf = tn_ & 1;
        const bool new_slab = t + 1 < total && (t + 1) % 18 == 0;
        const unsigned char* bA = sA + (posrow0 + (tap / 3 - 1) * W + (tap % 3 - 1)) * A_STRIDE + hf * 128 + h * 16;
        const unsigned char* bB = sB + (t & 1) * W2_BYTES + l31 * 128;
        const unsigned char* nA = sA + (posrow0 + (ntap / 3 - 1) * W + (ntap % 3 - 1)) * A_STRIDE + nhf * 128 + h * 16;
        const unsigned char* nB = sB + ((t + 1) & 1) * W2_BYTES + l31 * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            const unsigned char* rA = ks < 3 ? bA + (ks + 1) * 32 : nA;
            const unsigned char* rB = ks < 3 ? bB + ((((ks + 1) * 2 + h) ^ swz) << 4) : nB + ((h ^ swz) << 4);
            if (ks == 3) {
                // hand-over: every wave holds its last fragments of this chunk, the next chunk has landed
                AZ_DMA_LANDED();
                __syncthreads();
                // this chunk's buffer is free from here on: start the chunk after next right away, so it
                // has five k-steps (plus whatever the co-resident workgroup takes) to arrive from L2
                if (t + 2 < total) { STAGE_W2(t + 2, 0); STAGE_W2(t + 2, 1); STAGE_W2(t + 2, 2); STAGE_W2(t + 2, 3); }
                if (new_slab) {                              // next 128 input channels: restage the activations
                    STAGE_A2((t + 1) / 18);
                    __syncthreads();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#define MF2_(i_) acc[i_] = Prec<PR>::mfma(fw[c][i_], fx[c], acc[i_])
#define FENCE2_ __builtin_amdgcn_sched_barrier(0)
            MF2_(0); fx[n] = *(const typename Prec<PR>::vec*)(rA); FENCE2_;
            MF2_(1); fw[n][0] = *(const typename Prec<PR>::vec*)(rB); fw[n][1] = *(const typename Prec<PR>::vec*)(rB + 32 * 128); FENCE2_;
            MF2_(2); fw[n][2] = *(const typename Prec<PR>::vec*)(rB + 2 * 32 * 128); fw[n][3] = *(const typename Prec<PR>::vec*)(rB + 3 * 32 * 128); FENCE2_;
            MF2_(3); FENCE2_;
#undef MF2_
#undef FENCE2_
        }
    }
#undef STAGE_W2
#undef STAGE_A2
    __syncthreads();                                         // all waves done with sA: reuse rows 0..127 for the epilogue

    // epilogue (this column block's 128 channels of the tile's 128 rows): + bias (+ the fp32 residual stream, read
    // and rewritten in place straight from the accumulator layout: this kernel is the cross-check, not the fast
    // path), ReLU, zero padding rows -> 16-bit rows through LDS -> coalesced store
    {
        const int rr = 32 * wv + l31;
        const int m = m0 + rr;
        const int p = m % P;
        const bool real = m < M && (p / W) < N && (p % W) < N;
        const unsigned keep = real ? 0xFFFFFFFFu : 0u;
        const float cap = real ? Prec<PR>::cap() : 0.f;
        unsigned char* rowp = sA + rr * A_STRIDE + 8 * h;
        float* const rrow = resid ? resid + (size_t)(Gd + m) * C_pad + nb * 128 + 4 * h : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cch = 32 * i + 8 * q;
                const float4 bv = *(const float4*)(bias + nb * 128 + cch + 4 * h);
                float v0 = acc[i][4 * q + 0] + bv.x, v1 = acc[i][4 * q + 1] + bv.y;
                float v2 = acc[i][4 * q + 2] + bv.z, v3 = acc[i][4 * q + 3] + bv.w;
                if (resid) {
                    const float4 rv = *(const float4*)(rrow + cch);
                    v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
                }
                uint2 o;
                if (relu) {                                  // ReLU + saturation + padding mask: one v_med3 per value
                    v0 = relu_cap(v0, cap); v1 = relu_cap(v1, cap); v2 = relu_cap(v2, cap); v3 = relu_cap(v3, cap);
                    o.x = Prec<PR>::pack2(v0, v1);
                    o.y = Prec<PR>::pack2(v2, v3);
                } else {
                    if (!real) { v0 = 0.f; v1 = 0.f; v2 = 0.f; v3 = 0.f; }
                    o.x = ((unsigned)Prec<PR>::to16(v0) | ((unsigned)Prec<PR>::to16(v1) << 16)) & keep;
                    o.y = ((unsigned)Prec<PR>::to16(v2) | ((unsigned)Prec<PR>::to16(v3) << 16)) & keep;
                }
                if (resid) *(float4*)(rrow + cch) = make_float4(v0, v1, v2, v3);
                *(uint2*)(rowp + cch * 2) = o;
            }
    }
    __syncthreads();
    for (int i = tid; i < T2_M * 16; i += 256) {
        const int r = i >> 4, c = i & 15;
        *(uint4*)(out + (size_t)(Gd + m0 + r) * C_pad + nb * 128 + c * 8) = *(const uint4*)(sA + r * A_STRIDE + c * 16);
    }
}

// ------------------------------------- conv 3x3, 256 rows x 128 channels, two workgroups per CU
// The default per-layer kernel (boards larger than one LDS tile, N > 15, or C > 128).  The 128-row
// kernel above asks L2 for a weight byte per 128 rows (32 B/clk per CU with two workgroups resident:
// at the limit of what a CU gets from L2); this one keeps the two-workgroups-per-CU overlap (72 KB of
// LDS, 256 registers) but covers 256 rows per weight byte: wave w owns positions [64w, 64w+64) and all
// 128 output channels of the column block (4 x 2 accumulator tiles, 8 MFMAs per k-step on 4 weight + 2
// activation fragments).  The reduction runs over 64-channel input slices x 9 taps: per (slice, tap) one
// 16 KB weight chunk [n=128][k=64] through a double buffer (same packed image and source-address swizzle
// as the 128-row kernel), per slice one activation slab [256 + 2 Gd rows][64 ch] in a SINGLE buffer,
// by buffer-addressed LDS DMA with piece c of row r at c ^ ((r >> 1) & 7).  Re-loading the slab between
// slices stalls this workgroup; the co-resident one computes meanwhile.  Epilogue through a swizzled
// 256-byte-pitch image (piece c of row r at c ^ (r & 15)): residual rows in by LDS DMA, arithmetic in
// place, whole rows out.
#define T4_M 256
static inline __host__ __device__ int t4_slab_rows(int Gd) { return (T4_M + 2 * Gd + 7) / 8 * 8; }

template <int PR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
az_conv3x3_t256_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                       float* resid /* fp32 residual stream, updated in place; null: none */, const unsigned short* __restrict__ wpk,
                       const float* __restrict__ bias, int C_pad, int M, int n_tiles, int N, int Gd, int relu, int xcd_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Prec<PR>::vec vec;
    const int W = N + 1, P = W * W;
    const int rows_total = n_tiles * T4_M + 2 * Gd;          // rows of the activation buffers
    const int slab_rows = t4_slab_rows(Gd);
    unsigned char* sW = smem;                                // 2 x W2_BYTES
    unsigned char* sA = smem + 2 * W2_BYTES;                 // slab_rows x 128 B; the epilogue image (64 KB) starts at smem

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int l31 = lane & 31, h = lane >> 5;
    // Workgroups go to the eight XCDs round robin by their linear index, each XCD with an L2 of its own.  Index
    // -> (tile, output-channel block) so that an XCD works through one contiguous eighth of the tiles with the
    // channel blocks of a tile back to back: the second block finds the tile's input rows in that L2, and so does
    // the next tile for the halo rows it shares.  AZ_T256_PLAIN_ORDER (host) keeps the plain (x = tile, y = block) order.
    int tile_i = blockIdx.x, nb = blockIdx.y;
    if (gridDim.y == 1 && xcd_tiles > 0) {
        const int n_nb = C_pad / 128;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        nb = slot % n_nb;
        tile_i = xcd * xcd_tiles + slot / n_nb;
        if (slot / n_nb >= xcd_tiles || tile_i >= n_tiles) return;
    }
    const int m0 = tile_i * T4_M;
    const int n_sc = C_pad / 128;
    const int n_kc = C_pad / 64;
    const int total = n_kc * 9;                              // (slice, tap) chunks

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // weights: chunk t = kc*9 + tap lives in the packed image at ((nb*n_sc + kc/2)*9 + tap) * 32 KB, columns
    // [64*(kc&1), +64) of every 256-byte row; piece q = it*256 + tid of a chunk: LDS row it*32 + (tid >> 3),
    // physical piece tid & 7 = logical piece ^ ((row >> 1) & 7)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wpk + (size_t)nb * n_sc * 9 * (128 * 128)), 0, n_sc * 9 * (128 * 128 * 2), 0x00020000);
    const unsigned w_voff = (unsigned)((tid >> 3) * 256 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4));
    unsigned char* const w_dst = sW + wvu * 1024;
#define CHUNK_SRC4(t_) ((((t_) / 9 >> 1) * 9 + (t_) % 9) * (128 * 128 * 2) + (((t_) / 9) & 1) * 128)
#define STAGE_W4(t_, it_) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
        wrsrc, (lds_void*)(w_dst + ((t_) & 1) * W2_BYTES + (it_) * 4096), 16, w_voff, CHUNK_SRC4(t_) + (it_) * 8192, 0, 0)
    // slab: DMA q covers rows 8q .. 8q+7 (lane: row 8q + (lane >> 3), physical piece lane & 7); wave w issues
    // q = w, w+4, ..., so (row >> 1) & 7 = 4 (w & 1) + (lane >> 4) is a per-lane constant
    const size_t a_left = (size_t)(rows_total - m0) * C_pad * 2;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in + (size_t)m0 * C_pad), 0, (int)(a_left < 0x7FFFFFFFull ? a_left : 0x7FFFFFFFull), 0x00020000);
    const unsigned a_voff = (unsigned)((lane >> 3) * C_pad * 2 + (((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7)) << 4));
    const int n_q = slab_rows / 8;
#define STAGE_A4(kc_) do { \
        for (int q_ = wvu; q_ < n_q; q_ += 4) \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(sA + q_ * 1024), 16, \
                                                     a_voff, q_ * 8 * C_pad * 2 + (kc_) * 128, 0, 0); \
    } while (0)

#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W4(0, it);
#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W4(1, it);          // total >= 18: chunk 1 always exists
    STAGE_A4(0);
    AZ_DMA_LANDED();
    __syncthreads();

    const int s_base = Gd + 64 * wv + l31;                   // slab row of this lane's first position
    const int swzW = (l31 >> 1) & 7;
    const unsigned char* const wrow = sW + l31 * 128;
    vec fx[2][2], fw[2][4];
    {
        const int s0 = s_base - W - 1;                       // tap 0
        const int sz = (s0 >> 1) & 7;
        const unsigned char* bX = sA + s0 * 128 + ((h ^ sz) << 4);
        const unsigned char* bW = wrow + ((h ^ swzW) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[0][j] = *(const vec*)(bX + j * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[0][i] = *(const vec*)(bW + i * 4096);
    }
    for (int t = 0; t < total; ++t) {
        const int kc = t / 9, tap = t - 9 * kc;
        const int tn_ = t + 1 < total ? t + 1 : t;
        const int ntap = tn_ - 9 * (tn_ / 9);
        const int s0 = s_base + (tap / 3 - 1) * W + (tap % 3 - 1);
        const int ns0 = s_base + (ntap / 3 - 1) * W + (ntap % 3 - 1);
        const int sz = (s0 >> 1) & 7, nsz = (ns0 >> 1) & 7;
        const unsigned char* bX = sA + s0 * 128;
        const unsigned char* nX = sA + ns0 * 128;
        const unsigned char* bW = wrow + (t & 1) * W2_BYTES;
        const unsigned char* nW = wrow + ((t + 1) & 1) * W2_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            const unsigned char* rX = ks < 3 ? bX + ((((ks + 1) * 2 + h) ^ sz) << 4) : nX + ((h ^ nsz) << 4);
            const unsigned char* rW = ks < 3 ? bW + ((((ks + 1) * 2 + h) ^ swzW) << 4) : nW + ((h ^ swzW) << 4);
            if (ks == 3) {
                // hand-over: every wave holds its last fragments of this chunk, the next chunk has landed
                AZ_DMA_LANDED();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#define MF4_(i_, j_) acc[i_][j_] = Prec<PR>::mfma(fw[c][i_], fx[c][j_], acc[i_][j_])
#define FENCE4_ __builtin_amdgcn_sched_barrier(0)
#define DMA4_(it_) do { if (ks == 3 && t + 2 < total) STAGE_W4(t + 2, it_); } while (0)
            MF4_(0, 0); fx[n][0] = *(const vec*)(rX); FENCE4_;
            MF4_(0, 1); fx[n][1] = *(const vec*)(rX + 4096); DMA4_(0); FENCE4_;
            MF4_(1, 0); fw[n][0] = *(const vec*)(rW); FENCE4_;
            MF4_(1, 1); fw[n][1] = *(const vec*)(rW + 4096); DMA4_(1); FENCE4_;
            MF4_(2, 0); fw[n][2] = *(const vec*)(rW + 2 * 4096); FENCE4_;
            MF4_(2, 1); fw[n][3] = *(const vec*)(rW + 3 * 4096); DMA4_(2); FENCE4_;
            MF4_(3, 0); FENCE4_;
            MF4_(3, 1); DMA4_(3); FENCE4_;
#undef MF4_
#undef FENCE4_
#undef DMA4_
        }
        if (tap == 8 && t + 1 < total) {
            // slice boundary: the slab is single-buffered.  The activation fragments just prefetched came from
            // the old slab (harmless, dropped); reload the slab -- the co-resident workgroup has the MFMA
            // pipes meanwhile -- and read them again.  The weight fragments are already those of chunk t + 1.
            __syncthreads();                                 // every wave is done with the slab
            STAGE_A4(kc + 1);
            AZ_DMA_LANDED();
            __syncthreads();
            const unsigned char* rX = nX + ((h ^ nsz) << 4);
            fx[0][0] = *(const vec*)(rX); fx[0][1] = *(const vec*)(rX + 4096);
        }
    }
#undef STAGE_W4
#undef STAGE_A4
#undef CHUNK_SRC4
    __syncthreads();                                         // all waves done with the buffers: LDS becomes the output image

    if (resid) {
        // Second convolution of a ResBlock: x = relu(conv + bias + residual) with the residual stream in fp32
        // (network.hpp:36 in the reference's own precision; only the copy that feeds the next convolution is
        // rounded to 16 bits).  Two passes of 128 tile rows; a wave handles its own 32 rows per pass end to end
        // (rows 64 wv + 32 p + [0, 32) of the tile = rows 32 wv + [0, 32) of the LDS image), so the passes need
        // no hand-over between waves.  Image: fp32 [128 rows][128 ch], 512-byte rows, 16-byte piece c of row r
        // at c ^ (r & 31); in by LDS DMA, arithmetic in place, fp32 rows and their 16-bit copies out whole.
        unsigned char* const img = smem + (size_t)wvu * 32 * 512;            // this wave's 32 image rows
        const size_t rpitch = (size_t)C_pad * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int trow0 = 64 * wv + 32 * p;                              // first tile row of this wave's pass
            float* const gbase = resid + (size_t)(Gd + m0 + trow0) * C_pad + nb * 128;
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)gbase, 0, (int)(32 * rpitch), 0x00020000);
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {                                   // DMA q: image rows 2q, 2q + 1
                const int r = 2 * q + (lane >> 5);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, (lds_void*)(img + q * 1024), 16,
                                                         (unsigned)(r * rpitch) + (unsigned)(((lane & 31) ^ r) << 4), 0, 0, 0);
            }
            AZ_DMA_LANDED();
            __builtin_amdgcn_sched_barrier(0);
            {
                const int m = m0 + trow0 + l31;
                const int pp = m % P;
                const bool real = m < M && (pp / W) < N && (pp % W) < N;
                const float cap = real ? Prec<PR>::cap() : 0.f;
                unsigned char* const rowp = img + l31 * 512;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bv = *(const float4*)(bias + nb * 128 + 32 * i + 8 * q + 4 * h);
                        float4* const slot = (float4*)(rowp + (((8 * i + 2 * q + h) ^ l31) << 4));   // channels 32 i + 8 q + 4 h .. +3
                        const float4 rv = *slot;
                        float v0 = acc[i][p][4 * q + 0] + bv.x, v1 = acc[i][p][4 * q + 1] + bv.y;
                        float v2 = acc[i][p][4 * q + 2] + bv.z, v3 = acc[i][p][4 * q + 3] + bv.w;
                        v0 += rv.x; v1 += rv.y; v2 += rv.z; v3 += rv.w;
                        if (relu) { v0 = relu_cap(v0, cap); v1 = relu_cap(v1, cap); v2 = relu_cap(v2, cap); v3 = relu_cap(v3, cap); }
                        else if (!real) { v0 = 0.f; v1 = 0.f; v2 = 0.f; v3 = 0.f; }
                        *slot = make_float4(v0, v1, v2, v3);
                    }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                              // this wave's LDS writes are done (lgkmcnt 0)
            __builtin_amdgcn_sched_barrier(0);
            // fp32 rows back to the residual stream: 2 rows per trip
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int r = 2 * it + (lane >> 5), ph = lane & 31;
                const float4 v = *(const float4*)(img + r * 512 + (ph << 4));
                *(float4*)(gbase + (size_t)r * C_pad + ((ph ^ r) << 2)) = v;
            }
            // and their 16-bit copies, the next convolution's input: 4 rows per trip, 8 channels per lane
            unsigned short* const obase = out + (size_t)(Gd + m0 + trow0) * C_pad + nb * 128;
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int r = 4 * it + (lane >> 4), c = lane & 15;
                const float4 a = *(const float4*)(img + r * 512 + (((2 * c) ^ r) << 4));
                const float4 bq = *(const float4*)(img + r * 512 + (((2 * c + 1) ^ r) << 4));
                uint4 o;
                if (relu) {                                                  // already in [0, cap]
                    o.x = Prec<PR>::pack2(a.x, a.y); o.y = Prec<PR>::pack2(a.z, a.w);
                    o.z = Prec<PR>::pack2(bq.x, bq.y); o.w = Prec<PR>::pack2(bq.z, bq.w);
                } else {
                    o.x = (unsigned)Prec<PR>::to16(a.x) | ((unsigned)Prec<PR>::to16(a.y) << 16);
                    o.y = (unsigned)Prec<PR>::to16(a.z) | ((unsigned)Prec<PR>::to16(a.w) << 16);
                    o.z = (unsigned)Prec<PR>::to16(bq.x) | ((unsigned)Prec<PR>::to16(bq.y) << 16);
                    o.w = (unsigned)Prec<PR>::to16(bq.z) | ((unsigned)Prec<PR>::to16(bq.w) << 16);
                }
                *(uint4*)(obase + (size_t)r * C_pad + c * 8) = o;
            }
            __syncthreads();                                                 // reads of this pass are done before the next pass's DMA lands
        }
        return;
    }
    // first convolution of a ResBlock: 256 rows x 128 channels of 16-bit output, 256-byte rows, piece c of row r at c ^ (r & 15)
    unsigned char* const sO = smem;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = 64 * wv + 32 * j + l31;
        const int m = m0 + rr;
        const int pp = m % P;
        const bool real = m < M && (pp / W) < N && (pp % W) < N;
        const unsigned keep = real ? 0xFFFFFFFFu : 0u;
        const float cap = real ? Prec<PR>::cap() : 0.f;
        unsigned char* const rowp = sO + rr * 256 + 8 * h;
        const int sw = rr & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4*)(bias + nb * 128 + 32 * i + 8 * q + 4 * h);
                const float v0 = acc[i][j][4 * q + 0] + bv.x, v1 = acc[i][j][4 * q + 1] + bv.y;
                const float v2 = acc[i][j][4 * q + 2] + bv.z, v3 = acc[i][j][4 * q + 3] + bv.w;
                uint2* const slot = (uint2*)(rowp + (((4 * i + q) ^ sw) << 4));   // channels 32 i + 8 q + 4 h .. +3
                uint2 o;
                if (relu) {                                  // ReLU + saturation + padding mask: one v_med3 per value
                    o.x = Prec<PR>::pack2(relu_cap(v0, cap), relu_cap(v1, cap));
                    o.y = Prec<PR>::pack2(relu_cap(v2, cap), relu_cap(v3, cap));
                } else {
                    o.x = ((unsigned)Prec<PR>::to16(v0) | ((unsigned)Prec<PR>::to16(v1) << 16)) & keep;
                    o.y = ((unsigned)Prec<PR>::to16(v2) | ((unsigned)Prec<PR>::to16(v3) << 16)) & keep;
                }
                *slot = o;
            }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < T4_M * 16; i += 256) {
        const int r = i >> 4, c = i & 15;
        *(uint4*)(out + (size_t)(Gd + m0 + r) * C_pad + nb * 128 + c * 8) = *(const uint4*)(sO + r * 256 + ((c ^ (r & 15)) << 4));
    }
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int PR>
__device__ __forceinline__ void mfma16(f32x4& acc, const u32x4& a, const u32x4& b);
// The 256-row kernel above on v_mfma_f32_16x16x32 with the activations as the A operand: the chip holds a higher
// clock on this shape and with the zeros (ReLU outputs, padding) in A (tools/microbench/mfma_operand_power.hip).
// Same tiles, LDS images, weight stream and slab handling; what changes is the k-step (two of 32 per chunk, 32 MFMAs
// on 4 activation + 8 weight fragments), the accumulator layout, and the order of the weight image's rows
// (AzFoldArgs::layer16: row 16 a + l of a 128-channel block holds channel 64 (a >> 2) + 4 l + (a & 3)).  In that
// layout a lane holds four consecutive channels of a position and a 16-lane MFMA row 64 of them, so the outputs go
// from the accumulators to HBM in whole cache lines; LDS is used in the epilogue only to bring the residual in.
// RM: 0 = first convolution of a ResBlock; otherwise its second (x = relu(conv + bias + x), network.hpp:36) and how the
// residual stream x is kept: 1 = one fp32 image (resid), updated in place; 2 = a pair of 16-bit images hi + lo, hi =
// the 16-bit image the next convolution reads anyway (`out`, read here before it is overwritten), lo = the rounding
// remainder x - hi, scaled (`lo`, updated in place; AZ_LO_SCALE): 22 significant bits in fp16, and 8 bytes per element through HBM
// instead of 10; 3 = hi + lo in, fp32 out (the last block: the heads read resid).  4 and 5 (the compensated mode): 0 and 1
// with `out` an image of 3 C_pad values per row that takes the output as [hi | lo | hi] (az_cat3_kernel's split, here
// in the epilogue: the next convolution's input without another pass over HBM).  RELU: network.hpp:33,36.
// Template parameters so that the epilogue is straight-line code (as run-time flags they cost three branches and
// both conversions per four values).
template <int PR, int RM, bool RELU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
az_conv3x3_t256s_kernel(const unsigned short* __restrict__ in, unsigned short* out,
                       float* resid /* RM 1, 5: the fp32 stream, in and out; RM 3: out */, unsigned short* lo /* RM 2, 3 */,
                       const unsigned short* __restrict__ wpk,
                       const float* __restrict__ bias, int C_pad, int M, int n_tiles, int N, int Gd, int xcd_tiles,
                       int Cin /* channels (and row pitch) of `in` = the reduction's extent; C_pad except in the compensated mode */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = N + 1, P = W * W;
    const int rows_total = n_tiles * T4_M + 2 * Gd;          // rows of the activation buffers
    const int slab_rows = t4_slab_rows(Gd);
    unsigned char* sW = smem;                                // 2 x W2_BYTES
    unsigned char* sA = smem + 2 * W2_BYTES;                 // slab_rows x 128 B; the epilogue image (64 KB) starts at smem

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int l15 = lane & 15, gq = lane >> 4;               // MFMA row / column of the lane, k slice
    const int lrc = l15 < 8 ? l15 ^ 4 : l15;                 // the position / weight-image row it stands for (see AccTile16)
    const int gc = gq < 2 ? gq ^ 1 : gq;                     // its accumulators: positions 4 gc .. 4 gc + 3 of a block
    const int kpc = ((gq & 1) << 2) | (gq >> 1);             // 16-byte piece of its k slice in k-step 0 (0, 4, 1, 5); + 2 in k-step 1
    // Workgroups go to the eight XCDs round robin by their linear index, each XCD with an L2 of its own.  Index
    // -> (tile, output-channel block) so that an XCD works through one contiguous eighth of the tiles with the
    // channel blocks of a tile back to back: the second block finds the tile's input rows in that L2, and so does
    // the next tile for the halo rows it shares.  AZ_T256_PLAIN_ORDER (host) keeps the plain (x = tile, y = block) order.
    int tile_i = blockIdx.x, nb = blockIdx.y;
    if (gridDim.y == 1 && xcd_tiles > 0) {
        const int n_nb = C_pad / 128;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        nb = slot % n_nb;
        tile_i = xcd * xcd_tiles + slot / n_nb;
        if (slot / n_nb >= xcd_tiles || tile_i >= n_tiles) return;
    }
    const int m0 = tile_i * T4_M;
    const int n_sc = Cin / 128;
    const int n_kc = Cin / 64;
    const int total = n_kc * 9;                              // (slice, tap) chunks

    // acc[a][b][r]: position 64 wv + 16 b + 4 gc + r, channel 64 (a >> 2) + 4 lrc + (a & 3) of the block
    f32x4 acc[8][4];

    // weights: chunk t = kc*9 + tap lives in the packed image at ((nb*n_sc + kc/2)*9 + tap) * 32 KB, columns
    // [64*(kc&1), +64) of every 256-byte row; piece q = it*256 + tid of a chunk: LDS row it*32 + (tid >> 3),
    // physical piece tid & 7 = logical piece ^ ((row >> 1) & 7)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wpk + (size_t)nb * n_sc * 9 * (128 * 128)), 0, n_sc * 9 * (128 * 128 * 2), 0x00020000);
    const unsigned w_voff = (unsigned)((tid >> 3) * 256 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4));
    unsigned char* const w_dst = sW + wvu * 1024;
#define CHUNK_SRC4(t_) ((((t_) / 9 >> 1) * 9 + (t_) % 9) * (128 * 128 * 2) + (((t_) / 9) & 1) * 128)
#define STAGE_W4(t_, it_) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
        wrsrc, (lds_void*)(w_dst + ((t_) & 1) * W2_BYTES + (it_) * 4096), 16, w_voff, CHUNK_SRC4(t_) + (it_) * 8192, 0, 0)
#define STAGE_W4S(buf_, src_, it_) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
        wrsrc, (lds_void*)(w_dst + (buf_) * W2_BYTES + (it_) * 4096), 16, w_voff, (src_) + (it_) * 8192, 0, 0)
    // slab: DMA q covers rows 8q .. 8q+7 (lane: row 8q + (lane >> 3), physical piece lane & 7); wave w issues
    // q = w, w+4, ..., so (row >> 1) & 7 = 4 (w & 1) + (lane >> 4) is a per-lane constant
    const size_t a_left = (size_t)(rows_total - m0) * Cin * 2;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(in + (size_t)m0 * Cin), 0, (int)(a_left < 0x7FFFFFFFull ? a_left : 0x7FFFFFFFull), 0x00020000);
    const unsigned a_voff = (unsigned)((lane >> 3) * Cin * 2 + (((lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7)) << 4));
    const int n_q = slab_rows / 8;
#define STAGE_A4(kc_) do { \
        for (int q_ = wvu; q_ < n_q; q_ += 4) \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(sA + q_ * 1024), 16, \
                                                     a_voff, q_ * 8 * Cin * 2 + (kc_) * 128, 0, 0); \
    } while (0)

#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W4(0, it);
#pragma unroll
    for (int it = 0; it < 4; ++it) STAGE_W4(1, it);          // total >= 18: chunk 1 always exists
    STAGE_A4(0);
    // The accumulators start from the bias.  Buffer addressing for the epilogue: scalar resource over the tile's 256
    // rows of this channel block, the lane's first value as a 32-bit offset, (row, channel half) as a scalar offset
    // -- no address registers per access.
    const size_t tile0 = (size_t)(Gd + m0) * C_pad + nb * 128;
    const unsigned lane0 = (unsigned)((64 * wv + 4 * gc) * C_pad + 4 * lrc);                 // in elements
    constexpr bool RESID = RM == 1 || RM == 5, HILO = RM == 2 || RM == 3, CAT = RM >= 4, R32 = RM == 1 || RM == 3 || RM == 5;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(R32 ? resid + tile0 : (float*)nullptr), 0, R32 ? T4_M * C_pad * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(CAT ? (unsigned short*)nullptr : out + tile0), 0, CAT ? 0 : T4_M * C_pad * 2, 0x00020000);
    // CAT: `out` is an image with rows of 3 C_pad values, [hi | lo | hi] of the layer's output (see az_cat3_kernel)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(CAT ? out + (size_t)(Gd + m0) * (3 * C_pad) + nb * 128 : (unsigned short*)nullptr), 0, CAT ? T4_M * 3 * C_pad * 2 : 0, 0x00020000);
    const unsigned lane0c = (unsigned)((64 * wv + 4 * gc) * (3 * C_pad) + 4 * lrc);
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(HILO ? lo + tile0 : (unsigned short*)nullptr), 0, HILO ? T4_M * C_pad * 2 : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 bv = *(const float4*)(bias + nb * 128 + 64 * q + 4 * lrc);              // channels 64 q + 4 lrc .. + 3
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[4 * q][b][r] = bv.x; acc[4 * q + 1][b][r] = bv.y; acc[4 * q + 2][b][r] = bv.z; acc[4 * q + 3][b][r] = bv.w;
            }
    }
    // Which of the wave's 64 rows are cells of a board (the others are the zero column and row of the padded raster
    // and the rows past the last board): lane l works it out for row 64 wv + l, the ballot keeps it in two scalars.
    // floor((v + 0.5) / d) through the reciprocal is exact here: v < 1024, d <= 512.
    unsigned long long realmask;
    {
        const int pp0 = m0 % P;                              // scalar
        int pp = pp0 + 64 * wv + lane;
        pp -= (int)(((float)pp + 0.5f) * (1.0f / (float)P)) * P;
        const int y = (int)(((float)pp + 0.5f) * (1.0f / (float)W));
        const int x = pp - y * W;
        realmask = __ballot(m0 + 64 * wv + lane < M && y < N && x < N);
    }
    AZ_DMA_LANDED();
    __syncthreads();

    // Fragments: activations = the MFMA's A operand, weights = B (see AccTile16 for why and for the lane labels).
    // Slab and weight images have 128-byte rows (64 k = 8 pieces, piece c of row r at c ^ ((r >> 1) & 7)); the two
    // k slices that share a ds_read_b128 lane group read pieces 4 apart and rows that are complementary sets closed
    // under + 8, so every group covers 16 distinct slots.  A chunk is two k-steps of 32.  The four activation
    // fragments are double-buffered; a weight fragment is refilled in place behind its tile's last MFMA (tile-major
    // order: its next use is at least 21 MFMAs away), which keeps the kernel inside 256 registers.
    // A wave must keep the MFMA pipe fed on its own while its neighbour on the SIMD is in a memory phase, so the
    // loop has no branches and no wait it could avoid: the hand-over barrier sits behind MFMA 7 of the second
    // k-step, when the reads of the first have long completed (a barrier waits for all of a wave's LDS reads), and
    // the second k-step's own reads follow it; the weight DMA of chunk t + 2 is issued unconditionally, past the
    // end as the last chunk once more, into a buffer that is dead by then.
    const int s_base = Gd + 64 * wv + lrc;                   // slab row of this lane's position in block 0
    const int swzW = (lrc >> 1) & 7;
    const unsigned char* const wrow = sW + lrc * 128;
    u32x4 fx[2][4], fw[8];
    {
        const int s0 = s_base - W - 1;                       // tap 0
        const int sz = (s0 >> 1) & 7;
        const unsigned char* bX = sA + s0 * 128 + ((kpc ^ sz) << 4);
        const unsigned char* bW = wrow + ((kpc ^ swzW) << 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) fx[0][b] = *(const u32x4*)(bX + b * 2048);
#pragma unroll
        for (int a = 0; a < 8; ++a) fw[a] = *(const u32x4*)(bW + a * 2048);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+a"(acc[a][b]));
    asm volatile("s_nop 7");
    for (int t = 0; t < total; ++t) {
        const int kc = t / 9, tap = t - 9 * kc;
        const int tn_ = t + 1 < total ? t + 1 : t;
        const int ntap = tn_ - 9 * (tn_ / 9);
        const int s0 = s_base + (tap / 3 - 1) * W + (tap % 3 - 1);
        const int ns0 = s_base + (ntap / 3 - 1) * W + (ntap % 3 - 1);
        const int sz = (s0 >> 1) & 7, nsz = (ns0 >> 1) & 7;
        const unsigned char* bX = sA + s0 * 128;
        const unsigned char* nX = sA + ns0 * 128;
        const unsigned char* bW = wrow + (t & 1) * W2_BYTES;
        const unsigned char* nW = wrow + ((t + 1) & 1) * W2_BYTES;
        const int t2 = t + 2 < total ? t + 2 : total - 1;
        const int w_src = CHUNK_SRC4(t2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks, n = ks ^ 1;
            const unsigned char* rX = ks == 0 ? bX + (((kpc + 2) ^ sz) << 4) : nX + ((kpc ^ nsz) << 4);
            const unsigned char* rW = ks == 0 ? bW + (((kpc + 2) ^ swzW) << 4) : nW + ((kpc ^ swzW) << 4);
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int m = 4 * a + b;
                    mfma16<PR>(acc[a][b], fx[c][b], fw[a]);
                    if (ks == 0) {
                        if (b == 3) fw[a] = *(const u32x4*)(rW + a * 2048);
                        if (m == 0 || m == 1 || m == 2 || m == 5) fx[n][m < 3 ? m : 3] = *(const u32x4*)(rX + (m < 3 ? m : 3) * 2048);
                    } else {
                        if (m == 7) {
                            // hand-over: every wave has its last fragments of this chunk, the next chunk has landed
                            AZ_DMA_LANDED();
                            __syncthreads();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (m == 8 || m == 9 || m == 10 || m == 12) fx[n][m < 11 ? m - 8 : 3] = *(const u32x4*)(rX + (m < 11 ? m - 8 : 3) * 2048);
                        if (m == 11) fw[0] = *(const u32x4*)(rW);
                        if (m == 13) fw[1] = *(const u32x4*)(rW + 2048);
                        if (m == 14) fw[2] = *(const u32x4*)(rW + 2 * 2048);
                        if (m >= 15 && b == 3) fw[a] = *(const u32x4*)(rW + a * 2048);
                        if (m >= 16 && (m & 3) == 0) STAGE_W4S(t & 1, w_src, (m - 16) >> 2);
                    }
                }
        }
        if (tap == 8 && t + 1 < total) {
            // slice boundary: the slab is single-buffered.  The activation fragments just prefetched came from
            // the old slab (harmless, dropped); reload the slab -- the co-resident workgroup has the MFMA
            // pipes meanwhile -- and read them again.  The weight fragments are already those of chunk t + 1.
            __syncthreads();                                 // every wave is done with the slab
            STAGE_A4(kc + 1);
            AZ_DMA_LANDED();
            __syncthreads();
            const unsigned char* rX = nX + ((kpc ^ nsz) << 4);
#pragma unroll
            for (int b = 0; b < 4; ++b) fx[0][b] = *(const u32x4*)(rX + b * 2048);
            __builtin_amdgcn_s_waitcnt(0xC07F);              // nothing pending where this path rejoins the loop
        }
    }
    AZ_DMA_LANDED();                                         // the weight pieces requested past the end
#undef STAGE_W4
#undef STAGE_W4S
#undef STAGE_A4
#undef CHUNK_SRC4
    asm volatile("s_nop 15\n\ts_nop 7");                     // last MFMAs -> first read of their results
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+a"(acc[a][b]));
    // Epilogue.  A lane holds four consecutive channels of a position and a 16-lane MFMA row 64 of them, so results
    // go from the accumulators to HBM in whole cache lines: one store instruction = four rows x 64 channels, 256
    // contiguous bytes per row of the fp32 residual stream, 128 of the 16-bit image the next convolution reads.  One
    // value = one v_med3 (ReLU + 16-bit saturation + padding-row mask through a cap of 0) and half a packed convert.
    //
    // Second convolution of a ResBlock: x = relu(conv + bias + x) with the residual stream x in fp32 (network.hpp:36
    // in the reference's own precision; only the copy that feeds the next convolution is rounded to 16 bits).  The
    // stream comes in through LDS, by DMA, in two passes of 128 tile rows: a wave's own 32 rows per pass (tile rows
    // 64 wv + 32 p + [0, 32)) as a plain fp32 [32][128] image -- in the accumulator layout a 16-lane group reads 16
    // distinct 16-byte slots whatever the row, no swizzle.  Pass 1 is requested as soon as pass 0 sits in registers
    // and lands under pass 0's arithmetic and stores.  (Straight into registers instead, in the prologue or here,
    // the same loads cost 225 us more per launch at batch 8 192: 32 per lane in flight at once, all exposed.)
    unsigned char* const img = smem + (size_t)wvu * 32 * 512;                // this wave's 32 image rows
    const unsigned dma_voff = (unsigned)((lane >> 5) * C_pad * 4 + (lane & 31) * 16);
#define RES_DMA(p_) do { _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_)      /* DMA q: image rows 2q, 2q + 1 */ \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, (lds_void*)(img + q_ * 1024), 16, dma_voff, \
                                                 (64 * wvu + 32 * (p_) + 2 * q_) * C_pad * 4, 0, 0); } while (0)
    // hi + lo stream: the same two passes, a wave's 32 rows of both 16-bit images ([32][128] x 2 B each, hi then lo)
    // in the 16 KB the fp32 image takes; one DMA = four image rows
    const unsigned dma16_voff = (unsigned)((lane >> 4) * C_pad * 2 + (lane & 15) * 16);
#define RES_DMA16(p_) do { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) { \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ors, (lds_void*)(img + q_ * 1024), 16, dma16_voff, \
                                                 (64 * wvu + 32 * (p_) + 4 * q_) * C_pad * 2, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (lds_void*)(img + 8192 + q_ * 1024), 16, dma16_voff, \
                                                 (64 * wvu + 32 * (p_) + 4 * q_) * C_pad * 2, 0, 0); } } while (0)
    if constexpr (RESID) {
        __syncthreads();                                                     // all waves done with the buffers
        RES_DMA(0);
    }
    if constexpr (HILO) {
        __syncthreads();
        RES_DMA16(0);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const unsigned rmask = (unsigned)(realmask >> (32 * p));                             // rows 32 p .. 32 p + 31 of the wave
        float4 rv[2][2][4];
        if constexpr (HILO) {
            AZ_DMA_LANDED();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned char* at = img + (16 * bb + 4 * gc + r) * 256 + 128 * q + 8 * lrc;
                        const uint2 h = *(const uint2*)at, l = *(const uint2*)(at + 8192);
                        rv[q][bb][r] = make_float4(
                            fmaf(Prec<PR>::from16((unsigned short)l.x), AZ_LO_INV, Prec<PR>::from16((unsigned short)h.x)),
                            fmaf(Prec<PR>::from16((unsigned short)(l.x >> 16)), AZ_LO_INV, Prec<PR>::from16((unsigned short)(h.x >> 16))),
                            fmaf(Prec<PR>::from16((unsigned short)l.y), AZ_LO_INV, Prec<PR>::from16((unsigned short)h.y)),
                            fmaf(Prec<PR>::from16((unsigned short)(l.y >> 16)), AZ_LO_INV, Prec<PR>::from16((unsigned short)(h.y >> 16))));
                    }
            if (p == 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F);                          // the images are in registers (lgkmcnt 0)
                __builtin_amdgcn_sched_barrier(0);
                RES_DMA16(1);
            }
        }
        if constexpr (RESID) {
            AZ_DMA_LANDED();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[q][bb][r] = *(const float4*)(img + (16 * bb + 4 * gc + r) * 512 + ((16 * q + lrc) << 4));
            if (p == 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F);                          // the image is in registers (lgkmcnt 0)
                __builtin_amdgcn_sched_barrier(0);
                RES_DMA(1);
            }
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 2 * p + bb;
                const bool real = rmask >> (16 * bb + 4 * gc + r) & 1u;
                const float cap = real ? Prec<PR>::cap() : 0.f;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int so = (16 * b + r) * C_pad + 64 * q;                            // scalar: row and channel half
                    float v0 = acc[4 * q][b][r], v1 = acc[4 * q + 1][b][r], v2 = acc[4 * q + 2][b][r], v3 = acc[4 * q + 3][b][r];
                    if constexpr (RESID || HILO) { v0 += rv[q][bb][r].x; v1 += rv[q][bb][r].y; v2 += rv[q][bb][r].z; v3 += rv[q][bb][r].w; }
                    uint2 o;
                    if constexpr (RELU) {
                        v0 = relu_cap(v0, cap); v1 = relu_cap(v1, cap); v2 = relu_cap(v2, cap); v3 = relu_cap(v3, cap);
                        o.x = Prec<PR>::pack2(v0, v1); o.y = Prec<PR>::pack2(v2, v3);        // already in [0, cap]
                    } else {
                        if (!real) { v0 = 0.f; v1 = 0.f; v2 = 0.f; v3 = 0.f; }
                        o.x = (unsigned)Prec<PR>::to16(v0) | ((unsigned)Prec<PR>::to16(v1) << 16);
                        o.y = (unsigned)Prec<PR>::to16(v2) | ((unsigned)Prec<PR>::to16(v3) << 16);
                    }
                    if constexpr (R32) {
                        const u32x4 d4 = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
                        __builtin_amdgcn_raw_buffer_store_b128(d4, rrs, lane0 * 4, so * 4, 0);
                        wide_store_guard(d4);
                    }
                    if constexpr (RM == 2) {                                 // what the 16-bit copy drops, as a second 16-bit value
                        const unsigned l0 = Prec<PR>::pack2((v0 - Prec<PR>::from16((unsigned short)o.x)) * AZ_LO_SCALE,
                                                            (v1 - Prec<PR>::from16((unsigned short)(o.x >> 16))) * AZ_LO_SCALE);
                        const unsigned l1 = Prec<PR>::pack2((v2 - Prec<PR>::from16((unsigned short)o.y)) * AZ_LO_SCALE,
                                                            (v3 - Prec<PR>::from16((unsigned short)(o.y >> 16))) * AZ_LO_SCALE);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){l0, l1}, lrs, lane0 * 2, so * 2, 0);
                    }
                    if constexpr (CAT) {                                     // hi, the remainder as it is (22 bits together), hi again
                        // (the remainders are computed in the registers the four-dword store above has just been given:
                        // see wide_store_guard)
                        const unsigned l0 = Prec<PR>::pack2(v0 - Prec<PR>::from16((unsigned short)o.x), v1 - Prec<PR>::from16((unsigned short)(o.x >> 16)));
                        const unsigned l1 = Prec<PR>::pack2(v2 - Prec<PR>::from16((unsigned short)o.y), v3 - Prec<PR>::from16((unsigned short)(o.y >> 16)));
                        const int soc = (16 * b + r) * (3 * C_pad) + 64 * q;
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){o.x, o.y}, crs, lane0c * 2, soc * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){l0, l1}, crs, lane0c * 2, (soc + C_pad) * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){o.x, o.y}, crs, lane0c * 2, (soc + 2 * C_pad) * 2, 0);
                    }
                    if constexpr (RM != 3 && !CAT)                           // (the last block's output feeds the heads only)
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){o.x, o.y}, ors, lane0 * 2, so * 2, 0);
                }
            }
    }
#undef RES_DMA
#undef RES_DMA16
}

// ------------------------------------------------------------ fused tower
// The whole tower for one tile of boards in ONE launch (C <= 128, (N+1)^2 <= 256):
// a workgroup keeps its boards' activations resident in LDS (bf16, zero-padded
// raster with zero guard rows), streams only weights (16 KB chunks, L2-resident,
// shared by every workgroup) and carries the residual stream in fp32 registers
// in the MFMA C/D layout, so between the stem and the heads no activation byte
// touches HBM.  A tile holds bpt = 256 / (N+1)^2 whole boards; no halo exchange.
//   stem  : im2col of the u8 planes (K = 27 -> 32) + 2 MFMA k-steps
//   block : conv1 -> ReLU -> LDS ; conv2 + fp32 residual -> ReLU -> LDS
//   heads : policy softmax and value MLP straight from the LDS-resident activations
#define I_STRIDE 80             // 32 k * 2 B + 16 B pad

// Optional phase timing of the fused kernel (build with -DAZ_TOWER_TIMING): wave 0 of
// two workgroups stamps the shader clock at phase boundaries; read with az_tower_timing_read.
#ifdef AZ_TOWER_TIMING
__device__ unsigned long long az_tower_ts[2][64];
#define TSTAMP(slot_) do { if (ts_blk >= 0 && tid == 0) az_tower_ts[ts_blk][slot_] = __builtin_readcyclecounter(); } while (0)
extern "C" int az_tower_timing_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(az_tower_ts), sizeof(unsigned long long) * 128);
}
#else
#define TSTAMP(slot_) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------
// The convolutions run on v_mfma_f32_16x16x32_{f16,bf16}.  A wave owns 32 J positions (J = 2: 256-row workgroup
// tiles; J = 1: 128-row tiles for small boards, so that a small batch still fills the chip) x 128 channels =
// 2 J position blocks x 8 channel tiles of 16 x 16; one k-step covers K = 32 input channels, a tap four k-steps.
//
// Operand roles.  The ACTIVATIONS are the MFMA's A operand (rows of D = positions), the WEIGHTS its B operand
// (columns of D = output channels).  The kernel is power-bound and what the chip draws depends on the data and on
// the side it enters: about half of a ReLU output is zeros, and zeros in A let a bare loop hold 2.29 GHz where the
// same zeros in B hold 2.18 (dense weights on the other side; tools/microbench/mfma_operand_power.hip).  With the
// roles the other way round this kernel took 4.6 % longer.
//
// What a lane stands for.  The hardware fixes which lane feeds which row / column / k slice of the MFMA (lane =
// 16 g + l15: A row l15, B column l15, k slice g; D rows 4 g .. 4 g + 3 of column l15), not which position,
// channel or 8-wide slice of K that is.  With lrc = l15 ^ 4 below 8 and l15 from 8 up, gc = g ^ 1 below 2 and g
// from 2 up, koff = 0, 8, 4, 12 for g = 0..3:
//   A fragment (activations), block b: LDS row of position 16 b + lrc (+ the tap's offset), 16-byte piece ks + koff
//   B fragment (weights), tile a: row 16 a + lrc of the tap's [128][128] image, piece ks + koff, and that image
//       row holds output channel 64 (a >> 2) + 4 lrc + (a & 3) (the pack kernel orders the rows so)
//   T[a][b][r] = position 16 b + 4 gc + r, channel 64 (a >> 2) + 4 lrc + (a & 3)
// so the four tiles a = 4 q .. 4 q + 3 give a lane FOUR CONSECUTIVE CHANNELS of each of its positions: the
// epilogue packs them into one 8-byte LDS store, and the 16 lanes of a row fill 128 contiguous bytes.
// lrc and koff are what makes both fragment reads conflict-free: a ds_read_b128 is served in four groups of 16
// lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32, i.e. half of one k slice's lanes with the
// OTHER half of the next slice's.  On the rotated activation image (17 slots of 16 B per row) the two halves land
// on disjoint slots only if their rows are complementary sets closed under + 8 and their pieces differ by 8; on
// the XOR-ed weight image the same holds with ^ 8.  lrc sends lanes {0-3, 12-15} to rows {4-7, 12-15} and the rest
// to {0-3, 8-11}, and slices g, g + 1 of a group read pieces p, p + 8: every group of both reads covers all 16
// slots (with lrc = l15 and piece 4 ks + g, SQ_LDS_BANK_CONFLICT was 27 % of the LDS cycles).
// The MFMAs are inline asm with the accumulator tied in an AGPR quad: with the builtin the register allocator
// copies accumulators around inside the loop.  What the compiler cannot see behind the asm is handled here:
// wait states between the last MFMAs and the first read of their results, and between the epilogue's writes of
// the accumulators and the first MFMA.
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int J> using AccTile16 = f32x4[8][2 * J];

template <int PR>
__device__ __forceinline__ void mfma16(f32x4& acc, const u32x4& a, const u32x4& b) {
    if constexpr (PR == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// every accumulator of the tile is in its AGPR quad from here on (an empty asm the values pass through)
template <int J>
__device__ __forceinline__ void acc_fence16(AccTile16<J>& T) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 2 * J; ++b) asm volatile("" : "+a"(T[a][b]));
}

// T = bias[channel]; bl = bias + 4 lrc: tiles 4q .. 4q+3 of a lane hold channels 64 q + 4 lrc + (0..3)
template <int J>
__device__ __forceinline__ void fused_bias_init16(AccTile16<J>& T, const float* bl) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 bv = *(const float4*)(bl + 64 * q);
#pragma unroll
        for (int b = 0; b < 2 * J; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) { T[4 * q][b][r] = bv.x; T[4 * q + 1][b][r] = bv.y; T[4 * q + 2][b][r] = bv.z; T[4 * q + 3][b][r] = bv.w; }
    }
}

// fused_epi_hidden / fused_epi_residual in this layout.  row0: the lane's first position (4 gc of block 0) + 8 lrc bytes;
// caps[b][r]: the cap of position 16 b + 4 gc + r; bl = next bias + 4 lrc.  One 8-byte store = the four channels
// 64 q + 4 lrc .. + 3 of one position, taken from four tiles; a 16-lane group fills 128 contiguous bytes of a row.
template <int PR, int J, bool RESID>
__device__ __forceinline__ void fused_epi16(AccTile16<J>& T, const float* bl, unsigned char* row0, const float (&caps)[2 * J][4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 bv = *(const float4*)(bl + 64 * q);
#pragma unroll
        for (int b = 0; b < 2 * J; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cap = caps[b][r];
                const float r0 = relu_cap(T[4 * q][b][r], cap), r1 = relu_cap(T[4 * q + 1][b][r], cap);
                const float r2 = relu_cap(T[4 * q + 2][b][r], cap), r3 = relu_cap(T[4 * q + 3][b][r], cap);
                uint2 o;
                o.x = Prec<PR>::pack2(r0, r1);
                o.y = Prec<PR>::pack2(r2, r3);
                *(uint2*)(row0 + (16 * b + r) * A_STRIDE + 128 * q) = o;
                if (RESID) { T[4 * q][b][r] = r0 + bv.x; T[4 * q + 1][b][r] = r1 + bv.y; T[4 * q + 2][b][r] = r2 + bv.z; T[4 * q + 3][b][r] = r3 + bv.w; }
                else { T[4 * q][b][r] = bv.x; T[4 * q + 1][b][r] = bv.y; T[4 * q + 2][b][r] = bv.z; T[4 * q + 3][b][r] = bv.w; }
            }
            __builtin_amdgcn_sched_barrier(0);             // 16 values at a time: short live ranges
        }
    }
}

// One 3x3 convolution (nine taps) accumulated onto `acc`.  Weight chunk g + tap sits in buffer (g + tap) & 1 when
// the tap starts; chunk g + tap + 1 streams in behind it.  One chunk = one tap = [n=128][k=128] 16-bit = 32 KB,
// copied global -> LDS by the buffer-addressed LDS DMA (scalar resource + scalar chunk offset + per-lane 32-bit
// offset, 16 B per lane, no staging registers; the compiler keeps counted s_waitcnt lgkmcnt(N) for the fragment
// reads while pieces are in flight).  The LDS image is lane-linear, so the bank-conflict fix is an XOR swizzle
// applied to the SOURCE address and again on the read: 16-byte piece c of row n sits at piece c ^ (n & 15).
// Four k-steps of 32 per tap; the pipeline runs across tap boundaries: the barrier that hands over the weight
// buffers sits before the LAST k-step of a tap, whose MFMAs then cover the first fragment reads of the next tap.  A k-step is 8 x 2 J MFMAs, channel block major; the
// 8 + 2 J fragment reads of the NEXT k-step go one per gap behind its first MFMAs, in the order the next
// k-step consumes them (activations, then weights by channel block), which leaves the last of them more than
// a dozen MFMAs to return; the eight 4 KB pieces of the next tap's weights go out in k-steps 0 and 1, in the
// second half of the step, away from the fragment reads.
template <int PR, int J>
__device__ __forceinline__ void fused_conv_taps16(AccTile16<J>& acc, int g, int total, const unsigned char* sA, unsigned char* sB,
                                                  __amdgpu_buffer_rsrc_t wrsrc, unsigned st_off, unsigned char* st_dst,
                                                  int posrow16, int W, int koff, int lp) {
    constexpr int NB = 2 * J;
#define STAGE_W(g_, buf_, p_) __builtin_amdgcn_raw_ptr_buffer_load_lds( \
        wrsrc, (lds_void*)(st_dst + (buf_) * W_BUF_BYTES + (p_) * 4096), 16, st_off, (g_) * (128 * 128 * 2) + (p_) * 4096, 0, 0)
    u32x4 fx[2][NB], fw[2][8];
    const unsigned char* bA = sA + (posrow16 - W - 1) * A_STRIDE + koff * 16;      // tap 0: (-1,-1)
    const unsigned char* bB = sB + (g & 1) * W_BUF_BYTES + lp * 256;
#pragma unroll
    for (int b = 0; b < NB; ++b) fx[0][b] = *(const u32x4*)(bA + b * 16 * A_STRIDE);
#pragma unroll
    for (int a = 0; a < 8; ++a) fw[0][a] = *(const u32x4*)(bB + a * 16 * 256 + ((koff ^ lp) << 4));
    acc_fence16<J>(acc);
    asm volatile("s_nop 7");                                 // accumulator writes of the epilogue -> first MFMA
    for (int tap = 0; tap < 9; ++tap, ++g) {
        const int gn = g + 1 < total ? g + 1 : g;            // after the very last chunk: re-fetch it into the idle buffer
        const int tn = tap < 8 ? tap + 1 : 8;
        const unsigned char* nA = sA + (posrow16 + (tn / 3 - 1) * W + (tn % 3 - 1)) * A_STRIDE + koff * 16;
        const unsigned char* nB = sB + ((g + 1) & 1) * W_BUF_BYTES + lp * 256;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            const unsigned char* rA = ks < 3 ? bA + (ks + 1) * 16 : nA;
            const unsigned char* rB = ks < 3 ? bB + (((ks + 1 + koff) ^ lp) << 4) : nB + ((koff ^ lp) << 4);
            if (ks == 3) {
                // every wave holds its last fragments of this tap in registers and the next tap's weights
                // have landed: hand the buffers over
                AZ_DMA_LANDED();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int m = a * NB + b;                // MFMA number inside the k-step
                    mfma16<PR>(acc[a][b], fx[c][b], fw[c][a]);       // activations are the A operand (see AccTile16)
                    if (m < NB) fx[n][m] = *(const u32x4*)(rA + m * 16 * A_STRIDE);
                    else if (m < NB + 8) fw[n][m - NB] = *(const u32x4*)(rB + (m - NB) * 16 * 256);
                    if (ks < 2 && m >= 4 * NB && (m - 4 * NB) % NB == 0) {        // 4 pieces per k-step, NB MFMAs apart
                        const int piece = 4 * ks + (m - 4 * NB) / NB;
                        STAGE_W(gn, (g + 1) & 1, piece);
                    }
                }
        }
        bA = nA; bB = nB;
    }
#undef STAGE_W
    asm volatile("s_nop 15\n\ts_nop 7");                     // last MFMAs -> first read of their results
    acc_fence16<J>(acc);
}

// Register budget: 400 of the SIMD's 512 per lane (amdgpu_num_vgpr counts half of the unified
// file) -- nothing spills, and a 64-register wave of the tree kernel can live on every SIMD
// beside the tower's wave (its block needs 7.5 KB of LDS, the tower leaves 8.8 KB), so when a
// process runs two contexts on two streams one context's tree round executes under the
// other's tower instead of waiting for it.
#ifndef AZ_TOWER_REG_CAP
#define AZ_TOWER_REG_CAP __attribute__((amdgpu_num_vgpr(224)))
#endif
// WT: the policy and value go out write-through (agent-scope stores), for a reader on another CU inside this launch
template <bool WT>
__device__ __forceinline__ void fused_out(float* p, float v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// One tile (boards tile * bpt ...) through the whole tower; every thread of the workgroup calls it.
// lds_warm: an earlier tile of this workgroup has left the bias table and the zeroed guard rows in LDS (nothing
// after the set-up writes either).  next_seq / next_want / next_nb (WT only): the hand-over words of the boards
// this workgroup takes next; wave 0 looks at them while the heads run and, if they are all there, issues the
// acquire for them right away.  Returns that (wave 0 only): the next tile can then start without waiting.
template <int PR, int J, bool WT>
__device__ __forceinline__ bool
fused_tile(unsigned char* smem, int tile, const uint8_t* planes, const unsigned short* __restrict__ stem_w,
           const float* __restrict__ stem_b, const unsigned short* __restrict__ wpk,
           const float* __restrict__ bias, const float* __restrict__ hp,
           float* policy, float* value,
           int B, int N, int Gd, int n_layers, int bpt, unsigned long long* __restrict__ sat, int plane_stride,
           bool lds_warm = false, const unsigned* next_seq = nullptr, unsigned next_want = 0u, int next_nb = 0) {
    constexpr int TM = 128 * J;                              // rows of the workgroup's tile
    constexpr int PW = 32 * J;                               // positions per wave
    const int W = N + 1, P = W * W, A = N * N;
    const int a_rows = TM + 2 * Gd;
    unsigned char* sA = smem;
    unsigned char* sB = smem + (size_t)a_rows * A_STRIDE;
    unsigned char* sI = sB;                                  // stem im2col [TM][I_STRIDE]
    unsigned char* sW = sB + TM * I_STRIDE;                  // stem weights [128][I_STRIDE]
    float* sBias = (float*)(sB + 2 * W_BUF_BYTES);           // [n_layers + 2][128]: stem bias, every conv's, a row of zeros
    uint8_t* sP = (uint8_t*)(sBias + (n_layers + 2) * 128);  // planes of this tile's boards

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, gq = lane >> 4;               // 16x16x32 shape: MFMA row / column of the lane, k slice
    const int lrc = l15 < 8 ? l15 ^ 4 : l15;                 // ... the position / channel row it stands for (see AccTile16)
    const int gc = gq < 2 ? gq ^ 1 : gq;                     // ... the group of four channels its accumulators hold
    const int koff = ((gq & 1) << 3) | ((gq & 2) << 1);      // ... the 16-byte piece its k slice starts from
    constexpr int NPB = 2 * J;                               // 16-position blocks of a wave
    const int b0 = tile * bpt;                               // first board of the tile
    const int nb_here = min(bpt, B - b0);
    const int rows_here = nb_here * P;                       // tile rows that belong to boards
#ifdef AZ_TOWER_TIMING
    const int ts_blk = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 1 : -1);
    if (ts_blk >= 0 && tid == 0) az_tower_ts[ts_blk][63] = __builtin_amdgcn_s_memrealtime();
#endif
    TSTAMP(0);

    // Set-up: every global load is issued before the first LDS store that depends on one, so
    // the latencies overlap.  Only the guard rows need zeroing: the stem epilogue writes all
    // 256 tile rows (padding and idle rows as zeros).
    const int np = nb_here * 3 * A;                          // <= 768 plane bytes per tile
    uint8_t pl[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {                            // board k of the tile starts at byte k * plane_stride (3 A when packed)
        const int i = tid + 256 * t, bi = i / (3 * A);
        pl[t] = i < np ? planes[(size_t)(b0 + bi) * plane_stride + (i - bi * 3 * A)] : (uint8_t)0;
    }
    uint4 sw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) sw[t] = *(const uint4*)(stem_w + (size_t)(tid + 256 * t) * 8);
    // all biases into LDS once: the per-layer epilogue then never waits on HBM/L2
    const int nb4 = (n_layers + 1) * 32;                     // float4s: stem bias, then every conv's
    if (!lds_warm) {
#pragma unroll 3
        for (int i = tid; i < nb4 + 32; i += 256)
            ((float4*)sBias)[i] = i < 32 ? ((const float4*)stem_b)[i] : (i < nb4 ? ((const float4*)bias)[i - 32] : make_float4(0.f, 0.f, 0.f, 0.f));
        for (int i = tid; i < 2 * Gd * (A_STRIDE / 16); i += 256) {
            const int r = i / (A_STRIDE / 16), c = i % (A_STRIDE / 16);
            *(uint4*)(sA + (size_t)(r < Gd ? r : r + TM) * A_STRIDE + c * 16) = make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) if (tid + 256 * t < np) sP[tid + 256 * t] = pl[t];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = tid + 256 * t;
        *(uint4*)(sW + (i >> 2) * I_STRIDE + (i & 3) * 16) = sw[t];
    }
    __syncthreads();
    TSTAMP(40);
    if (tid < TM) {   // im2col row of tile position `tid`: k = ci*9 + ky*3 + kx, bf16(u8) is exact
        unsigned short v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = 0;
        const int r = tid;
        if (r < rows_here) {
            const int bi = r / P, p = r % P, y = p / W, x = p % W;
            if (y < N && x < N) {
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yy = y + ky - 1, xx = x + kx - 1;
                            if (yy >= 0 && yy < N && xx >= 0 && xx < N)
                                v[ci * 9 + ky * 3 + kx] = Prec<PR>::to16((float)sP[(bi * 3 + ci) * A + yy * N + xx]);
                        }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 pk;
            pk.x = v[8 * q + 0] | ((unsigned)v[8 * q + 1] << 16); pk.y = v[8 * q + 2] | ((unsigned)v[8 * q + 3] << 16);
            pk.z = v[8 * q + 4] | ((unsigned)v[8 * q + 5] << 16); pk.w = v[8 * q + 6] | ((unsigned)v[8 * q + 7] << 16);
            *(uint4*)(sI + r * I_STRIDE + q * 16) = pk;
        }
    }
    __syncthreads();
    TSTAMP(41);

    AccTile16<J> X, Y;                   // X: fp32 residual stream = accumulator of every second convolution; Y: of every first
    fused_bias_init16<J>(X, sBias + 4 * lrc);                // stem bias

    // which of this lane's positions (16 b + 4 gc + r of the wave's rows) are real board cells
    float cap16[NPB][4];
#pragma unroll
    for (int b = 0; b < NPB; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = PW * wv + 16 * b + 4 * gc + r4;
            const int p = r % P;
            cap16[b][r4] = (r < rows_here && (p / W) < N && (p % W) < N) ? Prec<PR>::cap() : 0.f;
        }

    {
        // stem: the im2col tile is exactly one k-step of 32
        u32x4 xb[NPB], wa[8];
#pragma unroll
        for (int b = 0; b < NPB; ++b) xb[b] = *(const u32x4*)(sI + (PW * wv + 16 * b + lrc) * I_STRIDE + gq * 16);
#pragma unroll
        for (int a = 0; a < 8; ++a) wa[a] = *(const u32x4*)(sW + (16 * a + lrc) * I_STRIDE + gq * 16);
        acc_fence16<J>(X);
        asm volatile("s_nop 7");
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < NPB; ++b) mfma16<PR>(X[a][b], xb[b], wa[a]);
        asm volatile("s_nop 15\n\ts_nop 7");
        acc_fence16<J>(X);
    }
    __syncthreads();                                         // every wave is done with sI / sW
    TSTAMP(42);
    // bias table: [0] stem, [1 + L] convolution L.  Block b = convolutions 2b (onto Y) and 2b+1 (onto X).
    unsigned char* const row0 = sA + (size_t)(Gd + PW * wv + 4 * gc) * A_STRIDE + 8 * lrc;
    fused_epi16<PR, J, true>(X, sBias + 2 * 128 + 4 * lrc, row0, cap16);
    fused_bias_init16<J>(Y, sBias + 1 * 128 + 4 * lrc);

    // weight stream set-up (see fused_conv_taps): piece p (0..7) of a chunk is 256 lanes x 16 B;
    // lane's piece index q = p*256 + tid -> LDS row q>>4 = p*16 + (tid>>4), physical piece tid&15,
    // logical piece ^ (row&15)
    const int total = n_layers * 9;
    const int st_row = tid >> 4;
    // per-lane byte offset inside a piece (32-bit, added to a scalar base by the load itself)
    const unsigned st_off = (unsigned)(st_row * 128 + (((tid & 15) ^ (st_row & 15)) << 3)) * 2u;   // row&15 == st_row
    // the LDS side is wave-uniform: keep it in scalar registers so that setting M0 costs no VALU work
    unsigned char* const st_dst = sB + __builtin_amdgcn_readfirstlane(wv) * 1024;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, total * (128 * 128 * 2), 0x00020000);
#pragma unroll
    for (int p = 0; p < 8; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(st_dst + p * 4096), 16, st_off, p * 4096, 0, 0);
    AZ_DMA_LANDED();
    __syncthreads();
    TSTAMP(1);

    const int posrow0 = Gd + PW * wv + lrc;
    const int n_blocks = n_layers >> 1;
    for (int b = 0; b < n_blocks; ++b) {
        // next block's biases; after the last block X restarts from the zero row: the heads read relu(x) itself
        const int nb1 = min(3 + 2 * b, n_layers), nb2 = b + 1 < n_blocks ? 4 + 2 * b : n_layers + 1;
        fused_conv_taps16<PR, J>(Y, 18 * b, total, sA, sB, wrsrc, st_off, st_dst, posrow0, W, koff, lrc);
        // (the barrier before the last k-step: all waves have finished reading this layer's input)
        TSTAMP(2 + 4 * b);
        fused_epi16<PR, J, false>(Y, sBias + nb1 * 128 + 4 * lrc, row0, cap16);
        __syncthreads();
        TSTAMP(3 + 4 * b);
        fused_conv_taps16<PR, J>(X, 18 * b + 9, total, sA, sB, wrsrc, st_off, st_dst, posrow0, W, koff, lrc);
        TSTAMP(4 + 4 * b);
        fused_epi16<PR, J, true>(X, sBias + nb2 * 128 + 4 * lrc, row0, cap16);
        __syncthreads();
        TSTAMP(5 + 4 * b);
    }
    bool next_ready = false;
    // ---- heads, on the activations still resident in LDS --------------------
    // PolicyHead (network.hpp:51-60): 1x1 conv -> softmax over each board's N*N cells.
    // ValueHead (network.hpp:75-83): 1x1 conv -> BN(1) -> ReLU -> fc1 -> ReLU -> fc2 -> softmax*2-1.
    // hp (fp32): {wp[c], wv[c]} pairs for c < 128 (BN folded), bp, bv, fc1 weights TRANSPOSED [A][64]
    // (so the 64 outputs of one cell are one coalesced 256-byte read), fc1b[64], fc2w[2][64], fc2b[2].
    {
        float* s_logit = (float*)sB;                 // [256] the weight buffers are free now
        float* s_u = s_logit + 256;                  // [256]
        float* s_part = s_u + 256;                   // [4][64]
        float* s_red = s_part + 256;                 // [8]
        int* s_map = (int*)(s_red + 8);              // [A] cell -> padded-raster row
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // the planes of the next tile were published most of a tile ago: one look now, under the heads' own latencies
        unsigned next_v = 0u;
        const bool next_look = WT && next_seq != nullptr && wv == 0 && lane < next_nb;
        if (next_look) next_v = az_pair_peek(next_seq + lane);
        const float bp = hp[256], bvv = hp[257];
        const float* fc1t = hp + 258;
        const float* fc1b = fc1t + 64 * A;
        const float* fc2w = fc1b + 64;
        const float* fc2b = fc2w + 128;
        const int r = tid;
        const int pp = r % P, py = pp / W, px = pp % W;
        const bool cell = r < rows_here && py < N && px < N;
        // Both 1x1 convolutions straight from the fp32 residual tile X (= relu(x), the last epilogue added a
        // zero bias).  Nothing is rounded to 16 bits between the last ResBlock and the heads.
        {
            // a lane holds 8 of the 128 channels (64 q + 4 lrc + 0..3) of each of its 4 x 2 J positions; the other
            // fifteen lanes of its 16-lane row hold the rest
            f32x2 sp16[NPB][4];
#pragma unroll
            for (int b = 0; b < NPB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) sp16[b][r] = (f32x2){0.f, 0.f};
            // fp16 operands: a residual stream that reached the format's largest value was clamped on the way
            // (relu_cap) -- count it, the answers of those boards are not the network's (az_stats.saturated)
            if (PR == 1) {
                float xm = 0.f;
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int b = 0; b < NPB; ++b) xm = fmaxf(fmaxf(xm, fmaxf(X[a][b][0], X[a][b][1])), fmaxf(X[a][b][2], X[a][b][3]));
                if (__ballot(xm >= 65504.f) != 0ull && lane == 0) atomicAdd(sat, 1ull);   // one per wave = 32 J positions
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 w01 = *(const float4*)(hp + 2 * (64 * q + 4 * lrc));          // {wp, wv} of channels c, c + 1
                const float4 w23 = *(const float4*)(hp + 2 * (64 * q + 4 * lrc) + 4);      // c + 2, c + 3
#pragma unroll
                for (int b = 0; b < NPB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x0 = X[4 * q][b][r], x1 = X[4 * q + 1][b][r], x2 = X[4 * q + 2][b][r], x3 = X[4 * q + 3][b][r];
                        sp16[b][r] += (f32x2){x0, x0} * (f32x2){w01.x, w01.y};
                        sp16[b][r] += (f32x2){x1, x1} * (f32x2){w01.z, w01.w};
                        sp16[b][r] += (f32x2){x2, x2} * (f32x2){w23.x, w23.y};
                        sp16[b][r] += (f32x2){x3, x3} * (f32x2){w23.z, w23.w};
                    }
            }
            TSTAMP(43);
#pragma unroll
            for (int b = 0; b < NPB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sp_ = sp16[b][r].x, sv_ = sp16[b][r].y;
#pragma unroll
                    for (int o2 = 1; o2 < 16; o2 <<= 1) { sp_ += __shfl_xor(sp_, o2, 64); sv_ += __shfl_xor(sv_, o2, 64); }
                    if (l15 == 0) {
                        const int row = PW * wv + 16 * b + 4 * gc + r;
                        const bool real = cap16[b][r] != 0.f;
                        s_logit[row] = real ? sp_ + bp : -3.4e38f;
                        s_u[row] = real ? fmaxf(sv_ + bvv, 0.f) : 0.f;
                    }
                }
        }
        if (TM < 256 && tid < 256 - TM) { s_logit[TM + tid] = -3.4e38f; s_u[TM + tid] = 0.f; }
        for (int a = tid; a < A; a += 256) s_map[a] = (a / N) * W + (a % N);
        if (WT && next_seq != nullptr && wv == 0) {                  // wave-uniform
            next_ready = __ballot(next_look && (int)(next_v - next_want) < 0) == 0ull;
            // the invalidate completes on its own; the wait for every store of the tile, at its end, covers it
            if (next_ready) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // softmax over each board's cells, reductions in a fixed order (deterministic bits)
        const int base = (r / P) * P;
        float mx, sum;
        if (bpt == 1) {                              // one board per tile: block-wide reduce
            float m = s_logit[r];
            for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
            if (lane == 0) s_red[wv] = m;
            __syncthreads();
            mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
            const float e = cell ? expf(s_logit[r] - mx) : 0.f;
            float t = e;
            for (int o2 = 32; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2, 64);
            if (lane == 0) s_red[4 + wv] = t;
            __syncthreads();
            sum = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
            if (cell) fused_out<WT>(&policy[(size_t)b0 * A + py * N + px], e / sum);
        } else {                                     // several small boards per tile
            mx = -3.4e38f;
            for (int k = 0; k < P; ++k) mx = fmaxf(mx, s_logit[base + k]);
            const float e = cell ? expf(s_logit[r] - mx) : 0.f;
            __syncthreads();
            s_logit[r] = e;
            __syncthreads();
            sum = 0.f;
            for (int k = 0; k < P; ++k) sum += s_logit[base + k];
            if (cell) fused_out<WT>(&policy[(size_t)(b0 + r / P) * A + py * N + px], e / sum);
        }
        TSTAMP(44);
        for (int bi = 0; bi < nb_here; ++bi) {       // value head, one board at a time
            const int o = tid & 63, part = tid >> 6;
            float acc1 = 0.f;
            int a = part;
            for (; a + 124 < A; a += 128) {          // 32 independent L2 loads in flight per lane
                float wv32[32], uv32[32];
#pragma unroll
                for (int t = 0; t < 32; ++t) wv32[t] = fc1t[(a + 4 * t) * 64 + o];
#pragma unroll
                for (int t = 0; t < 32; ++t) uv32[t] = s_u[bi * P + s_map[a + 4 * t]];
#pragma unroll
                for (int t = 0; t < 32; ++t) acc1 += wv32[t] * uv32[t];
            }
            for (; a + 60 < A; a += 64) {
                float wv16[16], uv16[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) wv16[t] = fc1t[(a + 4 * t) * 64 + o];
#pragma unroll
                for (int t = 0; t < 16; ++t) uv16[t] = s_u[bi * P + s_map[a + 4 * t]];
#pragma unroll
                for (int t = 0; t < 16; ++t) acc1 += wv16[t] * uv16[t];
            }
            for (; a + 28 < A; a += 32) {
                float wv8[8], uv8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) wv8[t] = fc1t[(a + 4 * t) * 64 + o];
#pragma unroll
                for (int t = 0; t < 8; ++t) uv8[t] = s_u[bi * P + s_map[a + 4 * t]];
#pragma unroll
                for (int t = 0; t < 8; ++t) acc1 += wv8[t] * uv8[t];
            }
            for (; a < A; a += 4) acc1 += fc1t[a * 64 + o] * s_u[bi * P + s_map[a]];
            s_part[part * 64 + o] = acc1;
            __syncthreads();
            if (tid < 64) {
                const float f = fmaxf(s_part[tid] + s_part[64 + tid] + s_part[128 + tid] + s_part[192 + tid] + fc1b[tid], 0.f);
                float z0 = fc2w[tid] * f, z1 = fc2w[64 + tid] * f;
                for (int o2 = 32; o2 > 0; o2 >>= 1) { z0 += __shfl_xor(z0, o2, 64); z1 += __shfl_xor(z1, o2, 64); }
                if (tid == 0) {
                    z0 += fc2b[0]; z1 += fc2b[1];
                    const float m2 = fmaxf(z0, z1);
                    const float e0 = expf(z0 - m2), e1 = expf(z1 - m2);
                    fused_out<WT>(&value[2 * (size_t)(b0 + bi)], e0 / (e0 + e1) * 2.f - 1.f);
                    fused_out<WT>(&value[2 * (size_t)(b0 + bi) + 1], e1 / (e0 + e1) * 2.f - 1.f);
                }
            }
            __syncthreads();
        }
    }
    TSTAMP(60);
#ifdef AZ_TOWER_TIMING
    if (ts_blk >= 0 && tid == 0) az_tower_ts[ts_blk][62] = __builtin_amdgcn_s_memrealtime();
#endif
    return next_ready;
}

template <int PR, int J>
__global__ void __launch_bounds__(256) AZ_TOWER_REG_CAP
az_tower_fused_kernel(const uint8_t* __restrict__ planes, const unsigned short* __restrict__ stem_w,
                      const float* __restrict__ stem_b, const unsigned short* __restrict__ wpk,
                      const float* __restrict__ bias, const float* __restrict__ hp,
                      float* __restrict__ policy, float* __restrict__ value,
                      int B, int N, int Gd, int n_layers, int bpt, unsigned long long* __restrict__ sat) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fused_tile<PR, J, false>(smem, blockIdx.x, planes, stem_w, stem_b, wpk, bias, hp, policy, value, B, N, Gd, n_layers, bpt, sat, 3 * N * N);
}

// The tower half of the persistent pair (AzPairDev, az_dev.h): one workgroup per CU walks its tiles round after
// round; a tile starts when the tree waves have published the planes of its boards and ends by publishing their
// policy and value.  Nothing else couples the workgroups: no grid-wide step between rounds.
// The arguments travel as one struct and are read from the kernel-argument segment again for every tile (scalar
// loads behind an opaque pointer): held in registers across the tile loop they cost the tile body 40 scalar
// registers it does not have.
struct AzTowerPairArgs {
    const uint8_t* planes; const unsigned short* stem_w; const float* stem_b; const unsigned short* wpk;
    const float* bias; const float* hp; float* policy; float* value;
    int B, N, Gd, n_layers, bpt, tiles;
    unsigned long long* sat;
    AzPairDev pr;
};
template <int PR, int J>
__global__ void __launch_bounds__(256) AZ_TOWER_REG_CAP
az_tower_pair_kernel(AzTowerPairArgs) {
    typedef const __attribute__((address_space(4))) AzTowerPairArgs* KArgs;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_go;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int rounds = ka->pr.rounds, tiles = ka->tiles;
    __builtin_amdgcn_s_setprio(3);                                       // ahead of the tree wave that shares this SIMD
    if (threadIdx.x == 0)                                                // this CU has its tower workgroup (the tree kernel's gate counts)
        __hip_atomic_fetch_add(ka->pr.start + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) s_go = 0;                                      // 0: wait for the tile's planes, 1: already acquired, -1: leave
    __syncthreads();
    bool warm = false;
    for (int i = 1; i <= rounds; ++i) {
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
            asm volatile("" : "+s"(ka));                                 // forget what was loaded through it
            int go = s_go;                                               // written in front of the last barrier
            if (go == 0) {
                __syncthreads();                                         // every wave has read s_go: wave 0 may rewrite it
                const unsigned want = ka->pr.base + (unsigned)i;
                const int b = t * ka->bpt + (int)threadIdx.x;
                if (threadIdx.x < 64) {                                  // lane k of wave 0 stands for board k of the tile
                    bool ok = true;
                    // round 1 evaluates the planes that were there before the launch
                    if ((int)threadIdx.x < ka->bpt && b < ka->B && i > 1 && !(ka->pr.flags & 1u))
                        ok = az_pair_wait<4>(ka->pr.plane_seq + b, want, ka->pr.abort, ka->pr.timeout_ticks, (unsigned)b);
                    const bool all_ok = __ballot(!ok) == 0ull;
                    // the planes were written by other CUs: drop this CU's L1 copies before any wave loads them
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (threadIdx.x == 0) s_go = all_ok ? 1 : -1;
                }
                __syncthreads();
                go = s_go;
            }
            if (go < 0) return;                                          // a wait ran out somewhere: leave
            // what this workgroup takes next: its next tile of this round, else its first of the next round
            int tn = t + (int)gridDim.x, in = i;
            if (tn >= tiles) { tn = (int)blockIdx.x; in = i + 1; }
            const bool look = in <= rounds && !(ka->pr.flags & 3u);
            const bool pre = fused_tile<PR, J, true>(smem, t, ka->planes, ka->stem_w, ka->stem_b, ka->wpk, ka->bias, ka->hp, ka->policy,
                                                     ka->value, ka->B, ka->N, ka->Gd, ka->n_layers, ka->bpt, ka->sat, ka->pr.plane_stride,
                                                     warm, look ? ka->pr.plane_seq + tn * ka->bpt : nullptr, ka->pr.base + (unsigned)in,
                                                     min(ka->bpt, ka->B - tn * ka->bpt));
            warm = true;
            // every wave's write-through stores have left (and wave 0's early acquire has completed) before the flags say so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) s_go = pre ? 1 : 0;
            __syncthreads();
            asm volatile("" : "+s"(ka));
            {
                const int b = t * ka->bpt + (int)threadIdx.x;
                if ((int)threadIdx.x < ka->bpt && b < ka->B)
                    __hip_atomic_store(ka->pr.eval_seq + b, ka->pr.base + (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------- heads
#ifdef AZ_TOWER_TIMING
__device__ unsigned long long az_heads_ts[32];       // 100 MHz stamps of thread 0 of block 0 at the phase boundaries
#define HSTAMP(slot_) do { if (blockIdx.x == 0 && threadIdx.x == 0) az_heads_ts[slot_] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int az_heads_timing_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(az_heads_ts), sizeof(unsigned long long) * 32);
}
#else
#define HSTAMP(slot_) do {} while (0)
#endif
// One block per board.  PolicyHead (network.hpp:51-60): 1x1 conv to c_out channels -> softmax over the
// c_out*N*N entries (layout [c_out][N][N] = get_action_shape()).  ValueHead (network.hpp:75-83): 1x1 conv ->
// BN(1) -> ReLU -> fc1 (N*N -> 64) -> ReLU -> fc2 (64 -> vd) -> softmax*2-1 (vd = 2 in the reference; a game
// with more players asks for one entry per player).
// hp layout (fp32): wp[c_out][C_pad], wv[C_pad], bp[c_out], bv (BN folded into wv/bv), then
// fc1 transposed [A][64], fc1b [64], fc2w [vd][64], fc2b [vd].
//
// Both 1x1 convolutions are one fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation):
// rows = output channels (policy, then value) in blocks of 32, columns = 32 cells of the board, K = C_pad.  The
// work splits into units of (32 cells) x (32 output channels); a wave takes every fourth unit, consecutive units
// being the row blocks of one cell block (the four waves read the same activations at about the same time).
// Lane l carries cell l & 31 and the k half kh = l >> 5; MFMA step t of a 128-channel slab pairs kh with channel
// 64 kh + t, so a lane's 64 activations of a slab are one contiguous 256-byte run of its cell's fp32 row
// (16 x 16-byte loads) and its weights four consecutive floats of one LDS row per read.  The activations of the
// next (unit, slab) are loaded into a second register set before the MFMAs of this one issue.  The accumulators
// start at the biases.  The softmax's maximum and sum are kept as this lane's running (max, sum of 2^((logit -
// max) log2 e)) over the logits it produces; the logits cross the output buffer once (written here, rescaled in
// place below with the exact exponential).
// SPLIT (fp16 operands only): the same GEMM on v_mfma_f32_32x32x16_f16 with both operands as fp16 pairs.  A weight is
// wh + wl / 2048 (wh = w rounded to fp16, wl = the remainder scaled so that it keeps its bits), an activation
// xh + xl (xl = the remainder as it is: the MFMA keeps fp16 subnormals, and what a subnormal xl loses is below
// 2^-25 times a weight).  acc += wh xh + wh xl, corr += wl xh, logit = acc + corr / 2048: exact products, fp32 sums,
// every product within 2^-22 of the fp32 one (wl xl is left out), at 3/16 of the fp32 MFMA's cycles.
template <int PR, bool SPLIT>
__global__ void __launch_bounds__(256, 2)
az_heads_kernel(const float* __restrict__ x /* fp32 residual stream */, const float* __restrict__ hp,
                float* __restrict__ policy, float* __restrict__ value, int N, int C_pad, int Gd, int c_out, int vd,
                unsigned long long* __restrict__ sat,
                const unsigned* __restrict__ mask /* null, or per board [c_out * N] row masks: bit x of word co * N + y
                                                     set = policy entry (co, y, x) is wanted */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int A = N * N, W = N + 1, P = W * W;
    const int MR = (c_out + 1 + 31) / 32 * 32;       // policy channels, then the value channel, padded to MFMA row blocks
    const int WS = C_pad + (SPLIT ? 8 : 4);          // floats of the weight image per row: fp32 + 16 bytes (ds_read_b128 of 32 rows
                                                     // without bank conflicts), or an fp16 row and its remainder row, each + 16 bytes
    const int RS = 2 * C_pad + 16;                   // SPLIT: bytes of one fp16 row; the remainder image follows the MR main rows
    float* s_wp = (float*)smem;                      // [MR][WS]
    float* s_bias = s_wp + MR * WS;                  // [MR]: policy biases, the value bias, zeros
    unsigned* s_mask = (unsigned*)(s_bias + MR);     // [N][MR]: the board's masks with the channel index innermost, zeros past c_out
    float* s_u = (float*)(s_mask + N * MR);          // [A]
    float* s_red = s_u + A;                          // [264]
    float* s_f1 = s_red + 264;                       // [64]
    const int b = blockIdx.x, tid = threadIdx.x;
    // Masked form: the search reads a leaf's policy at its legal actions only (_append_children, mcts.hpp:425-455) --
    // a few hundred of the c_out * N * N entries when c_out is large -- so only those are written; the softmax's
    // maximum and sum still run over every logit, and the entries written have the bits of the dense form.
    const float* wvv = hp + (size_t)c_out * C_pad;
    const float* bp = wvv + C_pad;
    const float* fc1t = bp + c_out + 1;
    const float* fc1b = fc1t + 64 * A;
    const float* fc2w = fc1b + 64;
    const float* fc2b = fc2w + 64 * vd;
    float* pol = policy + (size_t)b * c_out * A;
    typedef float f32x4h __attribute__((ext_vector_type(4)));
    typedef __attribute__((ext_vector_type(16))) float f32x16h;
    constexpr float LOG2E = 1.44269504088896340736f;
    const int lane = tid & 63, wvi = tid >> 6, l31h = lane & 31, kh = lane >> 5;
    const int n_mb = MR / 32, n_units = (A + 31) / 32 * n_mb, n_slab = C_pad / 128;
    // the 256-byte run of (unit u, slab sl) this lane multiplies
    // SPLIT: MFMA step s8 takes channels 16 s8 + 8 kh + j of the slab, so that the two k halves of a cell read adjacent
    // 32-byte runs: a load instruction touches one 128-byte line per cell (32 per instruction) instead of two -- the
    // kernel's time is the address unit's, which handles a line per cycle, more than the MFMAs' or HBM's.
#define HEADS_XROW(u_, sl_) (x + ((size_t)(Gd + b * P + (min((u_) / n_mb * 32 + l31h, A - 1) / N) * W + min((u_) / n_mb * 32 + l31h, A - 1) % N)) * C_pad + (sl_) * 128 + kh * (SPLIT ? 8 : 64))
#define HEADS_XLOAD(dst_, u_, sl_) do { const float* row_ = HEADS_XROW(u_, sl_); _Pragma("unroll") for (int q = 0; q < 16; ++q) dst_[q] = *(const f32x4h*)(row_ + (SPLIT ? 16 * (q >> 1) + 4 * (q & 1) : 4 * q)); } while (0)
    // (an empty statement that reads the sixteen registers: the wait for their loads lands here, not in front of the MFMAs)
#define HEADS_XWAIT(r_) asm volatile("" :: "v"(r_[0]), "v"(r_[1]), "v"(r_[2]), "v"(r_[3]), "v"(r_[4]), "v"(r_[5]), "v"(r_[6]), "v"(r_[7]), \
                                     "v"(r_[8]), "v"(r_[9]), "v"(r_[10]), "v"(r_[11]), "v"(r_[12]), "v"(r_[13]), "v"(r_[14]), "v"(r_[15]))
    f32x4h xa[16], xb[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xa[q] = xb[q] = (f32x4h){0.f, 0.f, 0.f, 0.f};
    HSTAMP(0);
    if (wvi < n_units) HEADS_XLOAD(xa, wvi, 0);                  // the first unit's activations travel under the staging below
    // staging, every load of a group issued before the first write to LDS (one trip to memory per group, not one per row)
    if (mask) {
        for (int i = tid; i < (MR - c_out) * N; i += 256) s_mask[(i % N) * MR + c_out + i / N] = 0u;
        for (int i0 = tid; i0 < c_out * N; i0 += 8 * 256) {
            unsigned mw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) mw[k] = i0 + 256 * k < c_out * N ? mask[(size_t)b * c_out * N + i0 + 256 * k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + 256 * k; if (i < c_out * N) s_mask[(i % N) * MR + i / N] = mw[k]; }
        }
    }
    for (int i = tid; i < MR; i += 256) s_bias[i] = i <= c_out ? bp[i] : 0.f;          // bp[c_out] is the value bias
    if (tid == 0) s_f1[0] = 0.f;                                 // saturation flag until the value head needs the array
    for (int i0 = tid; i0 < MR * (C_pad / 4); i0 += 12 * 256) {  // wp rows and the wv row are contiguous in hp
        float4 wr[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int i = i0 + 256 * k, r = i / (C_pad / 4), c4 = i % (C_pad / 4);
            wr[k] = r <= c_out ? *(const float4*)(hp + (size_t)r * C_pad + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int i = i0 + 256 * k, r = i / (C_pad / 4), c4 = i % (C_pad / 4);
            if (i < MR * (C_pad / 4)) {
                if (SPLIT) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    const float wv[4] = {wr[k].x, wr[k].y, wr[k].z, wr[k].w};
                    h4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (_Float16)fminf(fmaxf(wv[e], -65504.f), 65504.f);
                        lo[e] = (_Float16)((wv[e] - (float)hi[e]) * 2048.f);
                    }
                    *(h4*)((unsigned char*)s_wp + (size_t)r * RS + 8 * c4) = hi;
                    *(h4*)((unsigned char*)s_wp + (size_t)(MR + r) * RS + 8 * c4) = lo;
                } else
                    *(float4*)(s_wp + r * WS + 4 * c4) = wr[k];
            }
        }
    }
    HEADS_XWAIT(xa);                                             // (unconditional: a wait behind a branch is invisible to the next branch)
    __syncthreads();
    HSTAMP(1);

    float run_m = -3.4e38f, run_s = 0.f;
    float xmax = 0.f;                                // largest value of the stream this lane read (fp16: saturation check)
    {
        int u = wvi, sl = 0;
        f32x16h acc, corr;
        unsigned want = 0u;
        // one (unit, slab) step: xc holds its activations (arrived), xn receives the next step's
        auto step = [&](f32x4h (&xc)[16], f32x4h (&xn)[16]) {
            int un = u, sn = sl + 1;
            if (sn == n_slab) { sn = 0; un = u + 4; }
            const bool more = un < n_units;
            if (more) HEADS_XLOAD(xn, un, sn);
            const int cb = u / n_mb, mb = u - cb * n_mb;
            const int a = cb * 32 + l31h, ac = min(a, A - 1);
            if (sl == 0) {
                HSTAMP(2 + 3 * min(u >> 2, 5));
                // accumulators start at the biases; this lane's outputs are channels 32 mb + 8 j + 4 kh + i
                // (C/D layout of the 32x32 MFMAs), j, i = 0..3: register 4 j + i
                want = mask ? 0u : 0xffffu;
                if (SPLIT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) corr[r] = 0.f;
                }
                const int ay = ac / N, ax = ac - ay * N;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 b4 = *(const float4*)(s_bias + 32 * mb + 8 * j + 4 * kh);
                    acc[4 * j] = b4.x; acc[4 * j + 1] = b4.y; acc[4 * j + 2] = b4.z; acc[4 * j + 3] = b4.w;
                    if (mask) {
                        const uint4 m4 = *(const uint4*)(s_mask + ay * MR + 32 * mb + 8 * j + 4 * kh);
                        want |= ((m4.x >> ax & 1u) | (m4.y >> ax & 1u) << 1 | (m4.z >> ax & 1u) << 2 | (m4.w >> ax & 1u) << 3) << (4 * j);
                    }
                }
            }
            if (PR == 1 && mb == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) xmax = fmaxf(fmaxf(xmax, fmaxf(xc[q].x, xc[q].y)), fmaxf(xc[q].z, xc[q].w));
            }
            if (SPLIT) {
                // MFMA step s8 takes this lane's channels 16 s8 + 8 kh .. + 7 of the slab (k = 8 kh + j of the 32x32x16 form)
                typedef _Float16 h8s __attribute__((ext_vector_type(8)));
                const unsigned char* whr = (const unsigned char*)s_wp + (size_t)(32 * mb + l31h) * RS + (sl * 128 + kh * 8) * 2;
                const unsigned char* wlr = whr + (size_t)MR * RS;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const h8s wh = *(const h8s*)(whr + 32 * s8), wl = *(const h8s*)(wlr + 32 * s8);
                    h8s xh, xl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = xc[2 * s8 + (j >> 2)][j & 3];
                        const _Float16 h = (_Float16)v;
                        xh[j] = h;
                        xl[j] = (_Float16)(v - (float)h);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
                    corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, corr, 0, 0, 0);
                }
            } else {
                const float* wrow = s_wp + (32 * mb + l31h) * WS + sl * 128 + kh * 64;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 w4 = *(const float4*)(wrow + 4 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, xc[q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, xc[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, xc[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, xc[q].w, acc, 0, 0, 0);
                }
            }
            // the next step's activations have had the MFMAs' time to arrive; waiting here keeps the wait in front of
            // this unit's stores (loads and stores retire through one in-order counter)
            HEADS_XWAIT(xn);
            if (sl == n_slab - 1) {
                HSTAMP(3 + 3 * min(u >> 2, 5));
                if (SPLIT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = fmaf(corr[r], 1.f / 2048.f, acc[r]);
                }
                if (a < A) {
                    const int co0 = 32 * mb + 4 * kh;                 // register r holds channel co0 + 8 (r >> 2) + (r & 3)
                    float bm = -3.4e38f, bs = 0.f;
                    if (32 * mb + 32 <= c_out) {                      // a row block of policy channels only (the same for every lane)
#pragma unroll
                        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, acc[r]);
                        if (want) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (want >> r & 1u) pol[(co0 + 8 * (r >> 2) + (r & 3)) * A + a] = acc[r];
                        }
                        const float nm = fmaxf(run_m, bm);
#pragma unroll
                        for (int r = 0; r < 16; ++r) bs += __builtin_amdgcn_exp2f((acc[r] - nm) * LOG2E);
                        run_s = run_s * __builtin_amdgcn_exp2f((run_m - nm) * LOG2E) + bs;
                        run_m = nm;
                    } else {
                        // the last row block: policy channels, then the value channel, then padding; which of a lane's
                        // registers are which differs between the two k halves, so selects, not branches
                        const int rem = c_out - co0;                  // register r is a policy channel iff 8 (r >> 2) + (r & 3) < rem
                        float uval = 0.f;
                        unsigned pmask = 0u;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ci = 8 * (r >> 2) + (r & 3);
                            pmask |= (ci < rem ? 1u : 0u) << r;
                            bm = ci < rem ? fmaxf(bm, acc[r]) : bm;
                            uval = ci == rem ? acc[r] : uval;
                        }
                        if (rem >= 0 && rem < 32 && !(rem & 4)) s_u[a] = fmaxf(uval, 0.f);
                        const unsigned wsel = want & pmask;
                        if (wsel) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (wsel >> r & 1u) pol[(co0 + 8 * (r >> 2) + (r & 3)) * A + a] = acc[r];
                        }
                        const float nm = fmaxf(run_m, bm);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float e = __builtin_amdgcn_exp2f((acc[r] - nm) * LOG2E);
                            bs += 8 * (r >> 2) + (r & 3) < rem ? e : 0.f;
                        }
                        run_s = run_s * __builtin_amdgcn_exp2f((run_m - nm) * LOG2E) + bs;
                        run_m = nm;
                    }
                }
                HSTAMP(4 + 3 * min(u >> 2, 5));
            }
            u = un; sl = sn;
        };
        while (u < n_units) {
            step(xa, xb);
            if (u >= n_units) break;
            step(xb, xa);
        }
    }
#undef HEADS_XROW
#undef HEADS_XLOAD
#undef HEADS_XWAIT
    HSTAMP(25);
    // fp16 operands: a residual stream that reached the format's largest value was clamped on the way (relu_cap in
    // the convolutions' epilogues): count the board, its answer is not the network's (az_stats.saturated)
    if (PR == 1 && xmax >= 65504.f) s_f1[0] = 1.f;   // every writer stores the same value
    // softmax over the c_out * A entries: combine the lanes' running pairs, then one pass over the logits
    const int n_pol = c_out * A;
    float wm = run_m;
    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
    if (lane == 0) s_red[wvi] = wm;
    __syncthreads();
    HSTAMP(26);
    if (PR == 1 && tid == 0 && s_f1[0] != 0.f) atomicAdd(sat, 1ull);
    const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float ws = run_s * expf(run_m - mx);             // a lane without logits: 0 * exp(-huge) = 0
    for (int o = 32; o > 0; o >>= 1) ws += __shfl_xor(ws, o, 64);
    if (lane == 0) s_red[4 + wvi] = ws;
    __syncthreads();
    const float inv = 1.f / ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    HSTAMP(27);
    // value: fc1 with the weights transposed (a wave reads whole rows of 64 outputs), the cells split four ways;
    // issued before the softmax pass so that its loads are in flight under it
    {
        const int o = tid & 63, part = tid >> 6;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int a = part;
        // sixteen rows in flight per thread: the loop is a chain of trips to L2 (the 100 KB of weights are read by every
        // block), four at a time it took a sixth of a block's life
        for (; a + 60 < A; a += 64) {
            float w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = fc1t[(a + 4 * k) * 64 + o];
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                s0 += w[k] * s_u[a + 4 * k];
                s1 += w[k + 1] * s_u[a + 4 * k + 4];
                s2 += w[k + 2] * s_u[a + 4 * k + 8];
                s3 += w[k + 3] * s_u[a + 4 * k + 12];
            }
        }
        for (; a + 12 < A; a += 16) {
            s0 += fc1t[a * 64 + o] * s_u[a];
            s1 += fc1t[(a + 4) * 64 + o] * s_u[a + 4];
            s2 += fc1t[(a + 8) * 64 + o] * s_u[a + 8];
            s3 += fc1t[(a + 12) * 64 + o] * s_u[a + 12];
        }
        for (; a < A; a += 4) s0 += fc1t[a * 64 + o] * s_u[a];
        s_red[8 + tid] = (s0 + s1) + (s2 + s3);      // (s_red holds 256 + 8 floats)
    }
    const float f1b = fc1b[lane];
    float f2w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f2w[k] = k < vd ? fc2w[k * 64 + lane] : 0.f;
    HSTAMP(28);
    if (mask) {
        // each thread's words eight at a time, one wanted entry of every word per round: the rounds' loads are
        // independent of each other
        for (int i0 = tid; i0 < N * MR; i0 += 8 * 256) {
            unsigned wd[8];
            float* prow[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + 256 * k;
                wd[k] = i < N * MR ? s_mask[i] : 0u;
                const int y = i / MR, co = i - y * MR;
                prow[k] = wd[k] ? pol + co * A + y * N : pol;            // (a word without bits is read at entry 0, and not written)
            }
            unsigned any = wd[0] | wd[1] | wd[2] | wd[3] | wd[4] | wd[5] | wd[6] | wd[7];
            while (any) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = prow[k][wd[k] ? __ffs((int)wd[k]) - 1 : 0];
                any = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (wd[k]) {
                        prow[k][__ffs((int)wd[k]) - 1] = expf(v[k] - mx) * inv;
                        wd[k] &= wd[k] - 1u;
                        any |= wd[k];
                    }
            }
        }
    } else
        for (int i = tid; i < n_pol; i += 256) pol[i] = expf(pol[i] - mx) * inv;
    __syncthreads();
    HSTAMP(29);
    if (tid < 64) {
        const float f1 = fmaxf(s_red[8 + tid] + s_red[8 + 64 + tid] + s_red[8 + 128 + tid] + s_red[8 + 192 + tid] + f1b, 0.f);
        float z[8];
        float m2 = -3.4e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= vd) break;
            float zk = f2w[k] * f1;
            for (int o = 32; o > 0; o >>= 1) zk += __shfl_xor(zk, o, 64);
            z[k] = zk + fc2b[k];
            m2 = fmaxf(m2, z[k]);
        }
        if (tid == 0) {
            float den = 0.f;
            for (int k = 0; k < vd; ++k) { z[k] = expf(z[k] - m2); den += z[k]; }
            for (int k = 0; k < vd; ++k) value[(size_t)vd * b + k] = z[k] / den * 2.f - 1.f;
        }
    }
    HSTAMP(30);
}

// ------------------------------------------------------------ host launchers

// dynamic LDS of the fused kernel: activations + two weight buffers + bias table + planes
extern "C" size_t az_tower_fused_lds_bytes(int N, int n_layers, int rows) {
    const int P = (N + 1) * (N + 1), Gd = N + 2;
    return (size_t)(rows + 2 * Gd) * A_STRIDE + 2 * W_BUF_BYTES + (size_t)(n_layers + 2) * 128 * 4 +
           (size_t)(rows / P) * 3 * N * N + 16;
}

extern "C" size_t az_tower_heads_lds_bytes(int N, int C_pad, int c_out) {
    return ((size_t)((c_out + 1 + 31) / 32 * 32) * (C_pad + 9 + N) + (size_t)N * N + 264 + 64) * 4;
}
extern "C" size_t az_tower_stem_lds_bytes(int N, int C_pad, int c_in) {
    return (size_t)C_pad * (c_in * 9 + 1) * 4 + (size_t)c_in * N * N + 16;
}

extern "C" int az_tower_configure(void) {
    hipError_t e = hipSuccess;
#define SETA(k, bytes) do { if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); } while (0)
    SETA(az_conv3x3_t128_kernel<0>, 80 * 1024 + 512); SETA(az_conv3x3_t128_kernel<1>, 80 * 1024 + 512);
    SETA(az_conv3x3_t256_kernel<0>, 80 * 1024); SETA(az_conv3x3_t256_kernel<1>, 80 * 1024);
#define SETA4(pr) SETA((az_conv3x3_t256s_kernel<pr, 0, true>), 80 * 1024); SETA((az_conv3x3_t256s_kernel<pr, 1, true>), 80 * 1024); \
                  SETA((az_conv3x3_t256s_kernel<pr, 2, true>), 80 * 1024); SETA((az_conv3x3_t256s_kernel<pr, 3, true>), 80 * 1024); \
                  SETA((az_conv3x3_t256s_kernel<pr, 4, true>), 80 * 1024); SETA((az_conv3x3_t256s_kernel<pr, 5, true>), 80 * 1024)
    SETA4(0); SETA4(1);
#undef SETA4
    SETA(az_stem_kernel<0>, 64 * 1024); SETA(az_stem_kernel<1>, 64 * 1024);
    SETA((az_stem_split_kernel<1, 3>), 80 * 1024); SETA((az_stem_split_kernel<1, 5>), 80 * 1024);
    SETA((az_heads_kernel<0, false>), 128 * 1024); SETA((az_heads_kernel<1, false>), 128 * 1024); SETA((az_heads_kernel<1, true>), 128 * 1024);
    SETA((az_tower_fused_kernel<0, 2>), 160 * 1024); SETA((az_tower_fused_kernel<1, 2>), 160 * 1024);
    SETA((az_tower_fused_kernel<0, 1>), 160 * 1024); SETA((az_tower_fused_kernel<1, 1>), 160 * 1024);
    SETA((az_tower_pair_kernel<0, 2>), 160 * 1024 - 64); SETA((az_tower_pair_kernel<1, 2>), 160 * 1024 - 64);
    SETA((az_tower_pair_kernel<0, 1>), 160 * 1024 - 64); SETA((az_tower_pair_kernel<1, 1>), 160 * 1024 - 64);
#undef SETA
    return (int)e;
}

// prec: 0 = bf16 operands, 1 = fp16 operands
#define BY_PREC(kern, ...) do { if (prec == 1) hipLaunchKernelGGL(kern<1>, __VA_ARGS__); else hipLaunchKernelGGL(kern<0>, __VA_ARGS__); } while (0)

extern "C" void az_launch_stem(hipStream_t s, int prec, const uint8_t* planes, unsigned short* out, float* out32,
                               unsigned short* out_lo, const float* w, const float* bias, int B, int N, int C_pad, int Gd, int c_in) {
    // The planes the two games here use run on MFMAs, any other count on the generic kernel.  fp16 operands and 128
    // channels (where the stem is 4 % of a Blokus evaluation): the weights as fp16 pairs on the 16-bit MFMAs;
    // AZ_STEM_FP32 in the environment: the fp32 MFMA form, which the bf16 mode always takes (a bf16 tower's folded
    // weights may lie outside fp16's range) and the 256-channel towers too -- there the stem is 0.05 % of the time,
    // and the 41-convolution fixtures sit so close to the 1e-3 bound in fp16 that any change of the last bit of the
    // stem's fp32 sums, which re-rolls the roundings of everything behind it, moves some of them across (the
    // compensated mode's errors are the same to 1 % with either stem).
    if (c_in == 3 || c_in == 5) {
        static const bool stem32 = getenv("AZ_STEM_FP32") != nullptr;
        if (prec == 1 && !stem32 && C_pad == 128) {
            const int bpb = B >= 2048 ? 4 : 1;                           // (a small batch keeps one board per block: more blocks than CUs first)
            const size_t sms = (size_t)C_pad * 4 + (size_t)bpb * ((c_in * N * N + 15) & ~15) + 4 * 32 * STEM_SROW + (size_t)2 * ((c_in * 9 + 15) / 16) * 4 * 64 * 16;
            const dim3 grid((B + bpb - 1) / bpb);
            if (c_in == 3) hipLaunchKernelGGL((az_stem_split_kernel<1, 3>), grid, dim3(256), sms, s, planes, out, out32, out_lo, w, bias, N, C_pad, Gd, B, bpb);
            else hipLaunchKernelGGL((az_stem_split_kernel<1, 5>), grid, dim3(256), sms, s, planes, out, out32, out_lo, w, bias, N, C_pad, Gd, B, bpb);
            return;
        }
        const size_t smm = (size_t)c_in * N * N + 16 + (size_t)C_pad * 4;
#define STEM_MFMA(PR_, CIN_) hipLaunchKernelGGL((az_stem_mfma_kernel<PR_, CIN_>), dim3(B), dim3(256), smm, s, planes, out, out32, out_lo, w, bias, N, C_pad, Gd)
        if (prec == 1) { if (c_in == 3) STEM_MFMA(1, 3); else STEM_MFMA(1, 5); }
        else { if (c_in == 3) STEM_MFMA(0, 3); else STEM_MFMA(0, 5); }
#undef STEM_MFMA
        return;
    }
    const size_t sm = (size_t)C_pad * (c_in * 9 + 1) * 4 + (size_t)c_in * N * N + 16;
    BY_PREC(az_stem_kernel, dim3(B), dim3(256), sm, s, planes, out, out32, out_lo, w, bias, N, C_pad, Gd, c_in);
}

// Which per-layer convolution kernel a tower uses; read once when the tower is created, because the order of the
// weight image's rows goes with it.  0: 256 rows on 16x16x32 MFMAs, activations as A (rows in the order of
// AzFoldArgs::layer16); 1: 256 rows on 32x32x16, weights as A (AZ_T256_32); 2: 128 rows (AZ_CONV_T128).
extern "C" int az_conv_kind(void) {
    if (getenv("AZ_CONV_T128")) return 2;
    if (getenv("AZ_T256_32")) return 1;
    return 0;
}

extern "C" void az_launch_conv3x3(hipStream_t s, int prec, int kind, const unsigned short* in, unsigned short* out,
                                  float* resid, unsigned short* lo, const unsigned short* wpk, const float* bias,
                                  int C_pad, int M, int N, int Gd, int relu, int Cin) {
    if (Cin <= 0) Cin = C_pad;                                           // the kernels of kinds 1 and 2 take square layers only
    if (kind != 2) {                                                      // 256 rows x 128 channels, two workgroups per CU
        const int tiles = (M + T4_M - 1) / T4_M;
        size_t sm = 2 * (size_t)W2_BYTES + (size_t)t4_slab_rows(Gd) * 128;
        if (sm < (size_t)T4_M * 256) sm = (size_t)T4_M * 256;               // the epilogue image
        const bool plain = getenv("AZ_T256_PLAIN_ORDER") != nullptr;
        const int xcd_tiles = plain ? 0 : (tiles + 7) / 8;
        const dim3 grid = plain ? dim3(tiles, C_pad / 128) : dim3(8 * xcd_tiles * (C_pad / 128));
        if (kind == 0) {
#define T4S_GO(pr, rs, rl) hipLaunchKernelGGL((az_conv3x3_t256s_kernel<pr, rs, rl>), grid, dim3(256), sm, s, \
                                              in, out, resid, lo, wpk, bias, C_pad, M, tiles, N, Gd, xcd_tiles, Cin)
            // every 3x3 convolution of the tower is followed by a ReLU (network.hpp:33,36): the forms without one are
            // not instantiated (relu is ignored here; the other two kernels take it at run time)
            // residual stream: none (a block's first convolution), fp32 in place, hi + lo in place, hi + lo in and fp32 out
#define T4S_BY_FLAGS(pr) do { if (lo && resid) T4S_GO(pr, 3, true); else if (lo) T4S_GO(pr, 2, true); \
                              else if (resid) T4S_GO(pr, 1, true); else T4S_GO(pr, 0, true); } while (0)
            if (prec == 1) T4S_BY_FLAGS(1); else T4S_BY_FLAGS(0);
#undef T4S_BY_FLAGS
#undef T4S_GO
        }
        else
            BY_PREC(az_conv3x3_t256_kernel, grid, dim3(256), sm, s, in, out, resid, wpk, bias, C_pad, M, tiles, N, Gd, relu, xcd_tiles);
        return;
    }
    const int tiles = (M + T2_M - 1) / T2_M;
    const size_t sm = (size_t)(T2_M + 2 * Gd) * A_STRIDE + 2 * W2_BYTES;
    BY_PREC(az_conv3x3_t128_kernel, dim3(tiles, C_pad / 128), dim3(256), sm, s,
            in, out, resid, wpk, bias, C_pad, M, N, Gd, relu);
}

// The same convolution with its output written as the [hi | lo | hi] image the compensated mode's next convolution
// reads (rows of 3 C_pad values; az_cat3_kernel's arithmetic in the epilogue).  resid: the fp32 residual stream,
// updated in place (a block's second convolution), or null (its first).
extern "C" void az_launch_conv3x3_cat(hipStream_t s, int prec, const unsigned short* in, unsigned short* cat_out,
                                      float* resid, const unsigned short* wpk, const float* bias,
                                      int C_pad, int M, int N, int Gd, int Cin) {
    const int tiles = (M + T4_M - 1) / T4_M;
    size_t sm = 2 * (size_t)W2_BYTES + (size_t)t4_slab_rows(Gd) * 128;
    if (sm < (size_t)T4_M * 256) sm = (size_t)T4_M * 256;
    const int xcd_tiles = (tiles + 7) / 8;
    const dim3 grid(8 * xcd_tiles * (C_pad / 128));
#define T4C_GO(pr, rm) hipLaunchKernelGGL((az_conv3x3_t256s_kernel<pr, rm, true>), grid, dim3(256), sm, s, \
                                          in, cat_out, resid, (unsigned short*)nullptr, wpk, bias, C_pad, M, tiles, N, Gd, xcd_tiles, Cin)
    if (prec == 1) { if (resid) T4C_GO(1, 5); else T4C_GO(1, 4); }
    else { if (resid) T4C_GO(0, 5); else T4C_GO(0, 4); }
#undef T4C_GO
}

// Tile height of the fused kernel for a batch
static int fused_rows(int B, int P) {
    int rows = TILE_M;
    const char* force = getenv("AZ_FUSED_ROWS");
    if (P <= 128) {
        const int t256 = (B + TILE_M / P - 1) / (TILE_M / P), t128 = (B + 128 / P - 1) / (128 / P);
        const int r256 = (t256 + 255) / 256, r128 = (t128 + 255) / 256;
        if (3 * r128 < 5 * r256) rows = 128;
        if (force) rows = atoi(force) == 128 ? 128 : TILE_M;
    }
    return rows;
}

extern "C" void az_launch_tower_fused(hipStream_t s, int prec, const uint8_t* planes, const unsigned short* stem_w,
                                      const float* stem_b, const unsigned short* wpk, const float* bias,
                                      const float* hp, float* policy, float* value, int B, int N, int Gd, int n_layers,
                                      unsigned long long* sat) {
    const int P = (N + 1) * (N + 1);
    // Tile height: 256 rows (four 16-position blocks per wave), or 128 rows where a whole board fits in 128
    // and the batch is too small to fill the chip with 256-row tiles.  A 128-row tile does half the MFMA
    // work per weight byte, so it costs ~0.6 of a 256-row tile, not 0.5: take it when it saves tile rounds.
    const int rows = fused_rows(B, P);                       // AZ_FUSED_ROWS = "128" / "256" forces one: A/B and tests
    const int bpt = rows / P;
    const int tiles = (B + bpt - 1) / bpt;
    const size_t sm = az_tower_fused_lds_bytes(N, n_layers, rows);
#define FUSED_ARGS dim3(tiles), dim3(256), sm, s, planes, stem_w, stem_b, wpk, bias, hp, policy, value, B, N, Gd, n_layers, bpt, sat
#define FUSED_LAUNCH(J_) do { if (prec == 1) hipLaunchKernelGGL((az_tower_fused_kernel<1, J_>), FUSED_ARGS); \
                              else hipLaunchKernelGGL((az_tower_fused_kernel<0, J_>), FUSED_ARGS); } while (0)
    if (rows == 128) FUSED_LAUNCH(1); else FUSED_LAUNCH(2);
#undef FUSED_LAUNCH
#undef FUSED_ARGS
}

// The persistent form: at most `max_blocks` workgroups (one per CU), each walks tiles b, b + blocks, ... every
// round.  Returns the number of workgroups launched, 0 when the kernel's LDS does not fit.
extern "C" int az_launch_tower_pair(hipStream_t s, int prec, const uint8_t* planes, const unsigned short* stem_w,
                                    const float* stem_b, const unsigned short* wpk, const float* bias,
                                    const float* hp, float* policy, float* value, int B, int N, int Gd, int n_layers,
                                    unsigned long long* sat, const AzPairDev* pr, int max_blocks) {
    const int P = (N + 1) * (N + 1);
    const int rows = fused_rows(B, P);
    const int bpt = rows / P;
    const int tiles = (B + bpt - 1) / bpt;
    const size_t sm = az_tower_fused_lds_bytes(N, n_layers, rows);
    if (sm > 160 * 1024 - 64 || bpt > 64) return 0;
    const int blocks = tiles < max_blocks ? tiles : max_blocks;
    AzTowerPairArgs ka;
    ka.planes = planes; ka.stem_w = stem_w; ka.stem_b = stem_b; ka.wpk = wpk; ka.bias = bias; ka.hp = hp;
    ka.policy = policy; ka.value = value; ka.B = B; ka.N = N; ka.Gd = Gd; ka.n_layers = n_layers; ka.bpt = bpt;
    ka.tiles = tiles; ka.sat = sat; ka.pr = *pr;
#define PAIR_LAUNCH(J_) do { if (prec == 1) hipLaunchKernelGGL((az_tower_pair_kernel<1, J_>), dim3(blocks), dim3(256), sm, s, ka); \
                             else hipLaunchKernelGGL((az_tower_pair_kernel<0, J_>), dim3(blocks), dim3(256), sm, s, ka); } while (0)
    if (rows == 128) PAIR_LAUNCH(1); else PAIR_LAUNCH(2);
#undef PAIR_LAUNCH
    return blocks;
}

extern "C" void az_launch_heads(hipStream_t s, int prec, const float* x, const float* hp, float* policy,
                                float* value, int B, int N, int C_pad, int Gd, int c_out, int vd, unsigned long long* sat,
                                const unsigned* mask) {
    const size_t sm = az_tower_heads_lds_bytes(N, C_pad, c_out);
    // fp16 operands: the 1x1 convolutions on the 16-bit MFMAs with fp16-pair operands (AZ_HEADS_FP32: on the fp32 MFMAs,
    // which the bf16 mode always takes: its stream is not capped at fp16's largest value)
    static const bool heads32 = getenv("AZ_HEADS_FP32") != nullptr;
    if (prec == 1 && !heads32) hipLaunchKernelGGL((az_heads_kernel<1, true>), dim3(B), dim3(256), sm, s, x, hp, policy, value, N, C_pad, Gd, c_out, vd, sat, mask);
    else if (prec == 1) hipLaunchKernelGGL((az_heads_kernel<1, false>), dim3(B), dim3(256), sm, s, x, hp, policy, value, N, C_pad, Gd, c_out, vd, sat, mask);
    else hipLaunchKernelGGL((az_heads_kernel<0, false>), dim3(B), dim3(256), sm, s, x, hp, policy, value, N, C_pad, Gd, c_out, vd, sat, mask);
}

// Compensated mode of the per-layer path (AZ_PREC_FP16X3).  An activation x (fp32) enters a convolution as the pair
// hi = 16-bit(x), lo = 16-bit(x - hi), 22 significant bits together, and conv(W, x) = W_hi * x_hi + W_hi * x_lo +
// W_lo * x_hi (the lo * lo term is 2^-22 of the result) is ONE reduction over three times the channels: the input
// image's rows are [hi | lo | hi] (3 C_pad channels), the weight image's [W_hi | W_hi | W_lo], and the convolution
// kernel runs unchanged with Cin = 3 C_pad -- a single fp32 accumulation, the usual epilogue.  This kernel writes
// that image from the fp32 one: rows x C_pad values in, rows x 3 C_pad out, eight values per thread.
template <int PR>
__global__ void __launch_bounds__(256) az_cat3_kernel(const float* __restrict__ x, unsigned short* __restrict__ cat, size_t rows, int C_pad) {
    const size_t n8 = rows * (size_t)(C_pad / 8), stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const size_t r = i / (size_t)(C_pad / 8);
        const int c8 = (int)(i - r * (size_t)(C_pad / 8)) * 8;
        const float4 a = *(const float4*)(x + 8 * i), b = *(const float4*)(x + 8 * i + 4);
        const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned short h0 = Prec<PR>::to16(xv[2 * k]), h1 = Prec<PR>::to16(xv[2 * k + 1]);
            const unsigned short l0 = Prec<PR>::to16(xv[2 * k] - Prec<PR>::from16(h0)), l1 = Prec<PR>::to16(xv[2 * k + 1] - Prec<PR>::from16(h1));
            h[k] = (unsigned)h0 | ((unsigned)h1 << 16);
            l[k] = (unsigned)l0 | ((unsigned)l1 << 16);
        }
        unsigned short* row = cat + r * (size_t)(3 * C_pad) + c8;
        *(uint4*)(row) = make_uint4(h[0], h[1], h[2], h[3]);
        *(uint4*)(row + C_pad) = make_uint4(l[0], l[1], l[2], l[3]);
        *(uint4*)(row + 2 * C_pad) = make_uint4(h[0], h[1], h[2], h[3]);
    }
}
extern "C" void az_launch_cat3(hipStream_t s, int prec, const float* x, unsigned short* cat, size_t rows, int C_pad) {
    const size_t n8 = rows * (size_t)(C_pad / 8);
    const int blocks = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    BY_PREC(az_cat3_kernel, dim3(blocks), dim3(256), 0, s, x, cat, rows, C_pad);
}
