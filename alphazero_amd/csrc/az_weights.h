// Device-side weight preparation (weights.hip): where each field of the flat fp32 blob sits, and the
// images the fold kernels fill.  Offsets are in floats from the start of the blob.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct AzFoldArgs {
    const float* blob;          // device pointer, n_floats values
    size_t n_floats;
    int N, A, C, Cp, nres;
    int c_in, c_out, vd;        // input planes, policy channels, value entries
    int fused;                  // 1: images of az_tower_fused_kernel, 0: of the per-layer kernels
    int prec;                   // 0 bf16, 1 fp16 operands
    int layer16;                // per-layer path: rows of each 128-channel block in the order of the 16x16x32 kernel
    int part;                   // az_fold_conv_kernel: 0 the 16-bit weights, 1 the compensated mode's image conv_w3
    size_t off_stem_w;          // CBlock.conv.weight [C][c_in][3][3], then bias [C]
    size_t off_stem_bn;         // CBlock.bn weight, bias, running_mean, running_var [C] each
    size_t off_pc, off_vc;      // PHead.conv: weight [c_out][C], bias [c_out]; VHead.conv: weight [C], bias
    size_t off_vbn;             // VHead.bn: weight, bias, running_mean, running_var
    size_t off_fc1;             // VHead.fc1.weight [64][A], fc1.bias [64], fc2.weight [vd][64], fc2.bias [vd]
    size_t off_res0, res_stride;// RBlock_i at off_res0 + i * res_stride: conv1 w,b; conv2 w,b; bn1; bn2
    // outputs (device)
    float* stem_w;              // per-layer path: [Cp][27]
    float* stem_b;              // [Cp]
    unsigned short* stem_wbf;   // fused path: [128][32]
    unsigned short* conv_wf;    // fused path: [layer][tap][128][128]
    unsigned short* conv_w;     // per-layer path: [layer][nb][sc][tap][128][128]
    unsigned short* conv_w3;    // per-layer path, compensated mode: [layer][nb][sc over 3 Cp][tap][128][128] = [W_hi | W_hi | W_lo] (else null)
    size_t conv_layer_elems;
    float* conv_b;              // [2 nres][Cp]
    float* head_p;              // per-layer heads image
    float* head_pt;             // fused heads image
};

extern "C" {
void az_launch_check_blob(hipStream_t s, const AzFoldArgs* a, int* bad);
void az_launch_fold_range(hipStream_t s, const AzFoldArgs* a, int* rng);
void az_launch_fold(hipStream_t s, const AzFoldArgs* a);
}
