// Shared device-side definitions of libazhip (tree / env state in HBM).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One child stub = one reference Node before add() (node.hpp:21-22,63-67):
// prior p, visits n, total value v; q is v/n (recomputed, identical bits).
// child: >=0 the expanded child node = index of the first edge of ITS block of child edges
// (a node is nothing but that block: one edge per empty cell of its position, so an expanded
// node l plies below the root owns A - root_history - l consecutive edges and the root's block
// starts at edge 0), -1 stub, -2/-3/-4 terminal leaf whose
// stored reward is {+1,-1} / {-1,+1} / {0,0} (Node::terminal_add, node.hpp:36-42),
// -5 stub whose evaluation was issued in the current round (the reference holds
// the node's lock over that span, mcts.hpp:224-231).
struct __attribute__((aligned(16))) AzEdge {
    float p;
    int   n;
    float v;
    int   child;
};

enum { AZ_CHILD_WAITING = -5 };
enum { AZ_PEND_NONE = 0, AZ_PEND_ROOT = 1, AZ_PEND_LEAF = 2, AZ_PEND_TERMINAL = 3 };
enum { AZ_ST_IDLE = 0, AZ_ST_RUN = 1 };
enum { AZ_MODE_SEARCH = 0, AZ_MODE_SELFPLAY = 1 };

// All per-game state, structure-of-arrays over the G games of one GPU.
enum { AZ_GC_EVALS = 0, AZ_GC_DESCENTS = 1, AZ_GC_LEVELS = 2, AZ_GC_SCANNED = 3, AZ_GC_EXPANDED = 4, AZ_GC_RESTARTS = 5,
       // phase clocks (100 MHz ticks, only while AzTreeDev::timing is set): the buckets of the reference's verbosity-3 report
       AZ_GC_T_SELECT = 8, AZ_GC_T_STEP = 9, AZ_GC_T_APPEND = 10, AZ_GC_T_BACKUP = 11, AZ_GC_T_ROOT = 12, AZ_GC_N_STEP = 13,
       AZ_GC_STRIDE = 16 };

struct AzTreeDev {
    int G, N, A, K;
    int L;                      // leaves (descents) per tree per round; evaluation slot = g*L + k
    int node_cap, edge_cap;     // per game, per arena
    int path_cap;               // A + 1
    float vl;
    int c_puct_i;               // Node::ucb takes int (node.hpp:47)
    int noise_mode;
    unsigned long long seed;
    int slot_base;
    int budget;                 // selfplay: iter_budget of every search
    int chain;                  // simulations ending in a stored terminal leaf (no evaluation) a round may run back to back
    int timing;                 // 1: the tree kernels clock their phases into g_ctr (az_tree_timing)

    // real game (= root position)
    uint8_t*  cells;            // [G][A]
    uint8_t*  to_move;          // [G]
    int*      hist_len;         // [G]
    uint16_t* hist;             // [G][A]
    uint8_t*  game_over;        // [G]
    float*    final_reward;     // [G][2]

    // tree arenas (ping-pong for re-root compaction)
    int*      arena;            // [G] 0/1
    AzEdge*   edges[2];         // [G][edge_cap]
    uint16_t* eact[2];          // [G][edge_cap] action (cell index) of each edge
    int*      n_nodes;          // [G] expanded nodes (= edge blocks) in the arena
    int*      n_edges;          // [G]
    int*      root_n;           // [G]
    int*      root_hist_len;    // [G] -1 = no tree
    uint16_t* root_hist;        // [G][A]
    int*      queue;            // [G][node_cap] BFS queue of the re-root copy

    // search state
    int*      status;           // [G]
    int*      sims_done;        // [G]
    int*      sims_target;      // [G]
    int*      pending;          // [G*L]
    int*      path_len;         // [G*L]
    int*      path;             // [G*L][path_cap] edge indices root->leaf
    float*    pend_result;      // [G*L][2] reward of a terminal leaf
    int*      err;              // [G] sticky error flag (2: no selectable child, mcts.hpp:374-375)

    // evaluator I/O
    uint8_t*  leaf_planes;      // [G*L][3][A]
    float*    policy;           // [G*L][A]
    float*    value;            // [G*L][2]
    float*    noise;            // [G][A]
    float*    out_policy;       // [G][A]

    // self-play
    int*      move_idx;         // [G] moves played in the current game, -1 during its warm-up search
    float*    avg_len;          // [G] running mean game length (generate.cpp:69)
    long long* game_count;      // [G]
    unsigned* sample_ctr;       // [G]
    unsigned* noise_ctr;        // [G]
    uint8_t*  row_cells;        // [G][A][A]
    uint8_t*  row_tomove;       // [G][A]
    float*    row_policy;       // [G][A][A]
    // replay ring: the one of the context's two that launches currently append to (az_replay_flip swaps them)
    int       ring_cap;
    unsigned long long* ring_ctr;       // [0] rows in this ring, [1] rows dropped because it was full (= counters + 4 or + 6)
    uint8_t*  ring_cells;       // [ring_cap][A]
    uint8_t*  ring_tomove;      // [ring_cap]
    float*    ring_policy;      // [ring_cap][A]
    float*    ring_reward;      // [ring_cap][2]
    // counters touched at most once per move: [0] sims of completed searches [2] moves [3] games
    // [4] rows [5] dropped of ring 0, [6] [7] of ring 1.  Everything counted per round lives per game (no hot shared word):
    unsigned long long* counters;
    unsigned long long* g_ctr;          // [G][AZ_GC_STRIDE] per-game tallies (one 128-byte line per game, no shared hot word):
                                        //   AZ_GC_EVALS boards submitted to the evaluator, AZ_GC_DESCENTS descents that stood,
                                        //   AZ_GC_LEVELS tree levels they walked, AZ_GC_SCANNED child edges they read,
                                        //   AZ_GC_EXPANDED child edges written by expansions, AZ_GC_RESTARTS searches
                                        //   restarted from a fresh root because the arena was full
    int*      g_hw_nodes;       // [G] arena high-water marks, noted when a tree shrinks
    int*      g_hw_edges;       // [G]
};

// Hand-over words of the persistent pair: az_tower_pair_kernel (one workgroup per CU, its tiles round after round)
// and az_round_pair_kernel (one tree wave per CU beside it) run a whole batch of rounds in ONE launch each and pass
// every game's leaf planes / policy + value to each other through memory, game by game, with no grid-wide step:
//   tree wave, game g, round i : waits eval_seq[g] >= base + i, applies + selects, copies the leaf's planes to
//                                planes_wt with write-through stores, publishes plane_seq[g] = base + i + 1
//   tower workgroup, tile t    : waits plane_seq[g] >= base + i for its boards (i = 1: ready before the launch),
//                                evaluates, publishes eval_seq[g] = base + i
// The tree kernel's stream does not wait for the tower kernel's stream.  Every tower workgroup counts itself in
// start[1] when it begins (everything queued ahead of the tower kernel on the context's stream has finished by then),
// and a one-wave gate kernel in front of the tree kernel, on the tree kernel's stream, waits until all of them have:
// the tree waves are handed out only when every CU already holds its tower workgroup, which leaves room for exactly
// one tree wave per CU.  (Handed out first, onto a chip that a copy kernel is still leaving, two tree waves may land
// on one CU; the tower workgroup of that CU then never fits, and both kernels wait until the time-out.)
// A wait that lasts longer than timeout_ticks (the two kernels are not resident together) raises abort[0]; every
// wave of both kernels then leaves at its next wait, so the grids always drain.
struct AzPairDev {
    unsigned* plane_seq;            // [G]
    unsigned* eval_seq;             // [G]
    unsigned* planes_wt;            // [G][plane_stride / 4] the planes once more, as whole words stored write-through
    int       plane_stride;         // bytes per game in planes_wt: 3 A rounded up to 16
    unsigned* abort;                // [8] [0] raised by a wait that ran out, [1..3] that wait's notes, [4] clock steps seen by waits
    unsigned* start;                // [4] [1] tower workgroups that have begun, over all launches so far
    unsigned  started_want;         // ... once every workgroup of this launch has
    unsigned long long* busy;       // [G] 100 MHz ticks the tree wave worked on the game (waits excluded)
    unsigned  base;                 // evaluations of every game published by earlier launches
    int       rounds;
    unsigned  timeout_ticks;        // 100 MHz
    unsigned  flags;                // timing experiments: bit 0 the tower never waits for planes (wrong games), bit 1 it
                                    // does not look at its next tile's words ahead of time
};

#ifdef __HIPCC__
__device__ __forceinline__ unsigned az_pair_peek(const unsigned* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // bypasses this CU's L1
}
// One lane's wait until *w has reached `want` (sequence numbers: compared as a signed difference).  The lane whose
// wait runs out notes who it is in abort[1..3] (tag, the value it wanted, the value it last saw) before it raises
// abort[0]; with several such lanes the notes are of any one of them.
template <int SLEEP>
__device__ inline bool az_pair_wait(const unsigned* w, unsigned want, unsigned* abort, unsigned long long limit, unsigned tag) {
    if ((int)(az_pair_peek(w) - want) >= 0) return true;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), tp = t0;
    for (;;) {
        __builtin_amdgcn_s_sleep(SLEEP);
        const unsigned seen = az_pair_peek(w);
        if ((int)(seen - want) >= 0) return true;
        if (az_pair_peek(abort) != 0u) return false;
        // The clock between two polls (microseconds apart) does not count when it stepped back or forward by more than
        // 0.1 s: the wave was taken off the GPU and put back in between (the driver does that to every queue of the
        // process whenever the process makes or drops a hardware queue -- a new stream, the first copy on a DMA
        // engine), during which its partner did not run either, and it may be reading another die's counter now.
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        const long long step = (long long)(now - tp);
        if (step < 0 || step > 10000000ll) {
            t0 += (unsigned long long)step;
            __hip_atomic_fetch_add(abort + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tallied for the error text
        }
        tp = now;
        if ((long long)(now - t0) > (long long)limit) {
            __hip_atomic_store(abort + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort + 2, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort + 3, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}
#endif

__host__ __device__ inline unsigned long long az_splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Philox4x32-10, ctr = {counter, stream, 0, 0}, key = {seed lo, seed hi}.
__host__ __device__ inline void az_philox4(unsigned long long seed, unsigned stream, unsigned counter, unsigned out[4]) {
    unsigned c0 = counter, c1 = stream, c2 = 0u, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = 0xD2511F53ull * c0;
        unsigned long long p1 = 0xCD9E8D57ull * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One gamma(alpha) draw for the Dirichlet noise of util.h:42-58: Marsaglia-Tsang with the alpha+1 boost (the
// algorithm std::gamma_distribution uses) on counter-based uniforms, counters base .. base + 16 of `stream`.
// Only the distribution is defined by the reference (it seeds from random_device).
#ifdef __HIPCC__
__device__ inline float az_gamma_draw(unsigned long long seed, unsigned stream, unsigned base, float alpha) {
    const float a1 = alpha + 1.0f;
    const float dd = a1 - 1.0f / 3.0f;
    const float cc = 1.0f / sqrtf(9.0f * dd);
    float gv = dd;
    unsigned r[4];
    for (int it = 0; it < 16; ++it) {
        az_philox4(seed, stream, base + (unsigned)it, r);
        const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u3 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float x = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
        float v = 1.0f + cc * x;
        if (v <= 0.0f) continue;
        v = v * v * v;
        if (logf(u3) < 0.5f * x * x + dd - dd * v + dd * logf(v)) { gv = dd * v; break; }
    }
    az_philox4(seed, stream, base + 16u, r);
    const float u4 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return gv * powf(u4, 1.0f / alpha);
}
#endif
