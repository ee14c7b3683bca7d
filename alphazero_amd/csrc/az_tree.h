// Host-side launchers of the environment / tree kernels (tree.hip).
#pragma once
#include "az_dev.h"

// dynamic LDS of a tree block (board bytes + per-descent records): at most, and for a board of A cells
#define AZ_TREE_SMEM_BYTES (512 + 4 * 448 * 4)
static inline int az_tree_smem_bytes(int A) { return ((A + 15) & ~15) + 4 * ((A + 1 + 7) & ~7) * 4; }

extern "C" {
int  az_tree_configure(const AzTreeDev* d);
// one lockstep round over games [g0, g0 + count)
void az_launch_round(hipStream_t s, const AzTreeDev* d, int phase_mask, int mode, int g0, int count);
// the tree half of the persistent pair: `blocks` one-wave workgroups, each takes games b, b + blocks, ... every round
void az_launch_round_pair(hipStream_t s, const AzTreeDev* d, const AzPairDev* pr, int mode, int blocks);
void az_launch_begin(hipStream_t s, const AzTreeDev* d, int budget);
void az_launch_selfplay_begin(hipStream_t s, const AzTreeDev* d);
void az_launch_stats(hipStream_t s, const AzTreeDev* d, unsigned long long* out);   // out: 32 words, see az_stats_kernel
void az_launch_env_step(hipStream_t s, const AzTreeDev* d, const int* actions, float* reward,
                        uint8_t* done, int* illegal);
void az_launch_env_reset(hipStream_t s, const AzTreeDev* d);
void az_launch_env_export(hipStream_t s, const AzTreeDev* d, uint8_t* planes, uint8_t* mask);
void az_launch_eval_hash(hipStream_t s, const uint8_t* planes, float* policy, float* value, int B, int A);
}
