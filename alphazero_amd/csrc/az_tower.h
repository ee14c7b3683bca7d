// Host-side launchers of the tower kernels (tower.hip).  prec: 0 bf16, 1 fp16 operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "az_dev.h"

extern "C" {
int  az_tower_configure(void);
size_t az_tower_fused_lds_bytes(int N, int n_layers, int rows);   // dynamic LDS of az_tower_fused_kernel

// out_lo non-null: the residual stream leaves as the pair out + out_lo of 16-bit images instead of the fp32 image out32
void az_launch_stem(hipStream_t s, int prec, const uint8_t* planes, unsigned short* out, float* out32, unsigned short* out_lo,
                    const float* w, const float* bias, int B, int N, int C_pad, int Gd, int c_in);
int  az_conv_kind(void);                                          // per-layer convolution kernel (0, 1, 2), from the environment
// resid / lo (a ResBlock's second convolution): resid alone = the fp32 residual stream, updated in place; lo alone
// (kind 0 only) = the stream as the pair out + lo, both updated in place; both = the pair in, resid (fp32) out
void az_launch_conv3x3(hipStream_t s, int prec, int kind, const unsigned short* in, unsigned short* out,
                       float* resid, unsigned short* lo, const unsigned short* wpk, const float* bias,
                       int C_pad, int M, int N, int Gd, int relu, int Cin /* channels of `in`, 0 = C_pad (kind 0 only) */);
// kind 0 with the output written as the compensated mode's [hi | lo | hi] image (rows of 3 C_pad values); resid: the
// fp32 residual stream, updated in place, or null for a block's first convolution
void az_launch_conv3x3_cat(hipStream_t s, int prec, const unsigned short* in, unsigned short* cat_out,
                           float* resid, const unsigned short* wpk, const float* bias,
                           int C_pad, int M, int N, int Gd, int Cin);
void az_launch_tower_fused(hipStream_t s, int prec, const uint8_t* planes, const unsigned short* stem_w,
                           const float* stem_b, const unsigned short* wpk, const float* bias,
                           const float* hp, float* policy, float* value, int B, int N, int Gd, int n_layers,
                           unsigned long long* sat);
// the tower half of the persistent pair (AzPairDev): returns the workgroups launched, 0 if it does not fit
int  az_launch_tower_pair(hipStream_t s, int prec, const uint8_t* planes, const unsigned short* stem_w,
                          const float* stem_b, const unsigned short* wpk, const float* bias,
                          const float* hp, float* policy, float* value, int B, int N, int Gd, int n_layers,
                          unsigned long long* sat, const AzPairDev* pr, int max_blocks);
// sat: device counter of work items whose residual stream reached fp16's largest value (az_stats.saturated)
void az_launch_heads(hipStream_t s, int prec, const float* x, const float* hp, float* policy,
                     float* value, int B, int N, int C_pad, int Gd, int c_out, int vd, unsigned long long* sat,
                     const unsigned* mask /* null = every policy entry; else [B][c_out * N] row masks of the entries wanted */);
// compensated mode: rows [hi | lo | hi] (3 C_pad 16-bit channels) from an fp32 image of `rows` x C_pad values
void az_launch_cat3(hipStream_t s, int prec, const float* x, unsigned short* cat, size_t rows, int C_pad);
size_t az_tower_heads_lds_bytes(int N, int C_pad, int c_out);     // dynamic LDS of az_heads_kernel / az_stem_kernel (<= 64 KB)
size_t az_tower_stem_lds_bytes(int N, int C_pad, int c_in);
}
